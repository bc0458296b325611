// How long does the in-order sum of n doubles parked in LDS take for ONE lane - the tail of every pair-distance kernel
// (vft_pair_block / vft_pair_wave: lane 0 adds the terms, lane 1 the weights, in column order)?  Variants: which lanes are
// active, how many loads are in flight, other wavefronts of the workgroup waiting at a barrier or not.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/sumprobe/sumprobe tools/sumprobe/sumprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long *out, int n, double seed) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) lds[i] = seed + 1e-9 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long c0 = 0, c1 = 0;
    double acc = 0;
    if (wave == 0) {
        c0 = clock64();
        if (MODE == 0) {   // all 64 lanes add the same array, 8 reads in flight (vft_pair_block's loop)
            const double *src = lds;
            for (int p = 0; p + 8 <= n; p += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = src[p + u];
#pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u];
            }
        } else if (MODE == 1) {   // lanes 0 and 1 only, each its own array
            if (lane < 2) {
                const double *src = lds + lane * n;
                for (int p = 0; p + 8 <= n; p += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = src[p + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) acc += v[u];
                }
            }
        } else if (MODE == 2) {   // every lane loads ITS element of a 64-block (one ds_read per 64 terms), lane 0 adds them via readlane
            for (int p = 0; p < n; p += 64) {
                const double mine = lds[p + lane];
#pragma unroll
                for (int u = 0; u < 64; u++) {
                    const double t = __shfl(mine, u, 64);
                    acc += t;
                }
            }
        } else if (MODE == 3) {   // like 2 with scalar broadcasts (readlane into SGPRs, uniform adds)
            for (int p = 0; p < n; p += 64) {
                const double mine = lds[p + lane];
                const int lo = (int) (__double_as_longlong(mine) & 0xFFFFFFFFll), hi = (int) (__double_as_longlong(mine) >> 32);
#pragma unroll
                for (int u = 0; u < 64; u++) {
                    const int l2 = __builtin_amdgcn_readlane(lo, u), h2 = __builtin_amdgcn_readlane(hi, u);
                    acc += __longlong_as_double(((long long) h2 << 32) | (unsigned int) l2);
                }
            }
        }
        else if (MODE == 4) {   // a rotating window of eight 16-byte reads: every pair of adds waits for the OLDEST read only
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 *src = (const d2 *) lds;
            d2 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6], v7 = src[7];
            const int n2 = n / 2;
            int p = 0;
            for (; p + 16 <= n2; p += 8) {
#define STEP(V, K) acc += V.x; acc += V.y; V = src[p + 8 + K]; __builtin_amdgcn_sched_barrier(0);
                STEP(v0, 0) STEP(v1, 1) STEP(v2, 2) STEP(v3, 3) STEP(v4, 4) STEP(v5, 5) STEP(v6, 6) STEP(v7, 7)
#undef STEP
            }
            acc += v0.x; acc += v0.y; acc += v1.x; acc += v1.y; acc += v2.x; acc += v2.y; acc += v3.x; acc += v3.y;
            acc += v4.x; acc += v4.y; acc += v5.x; acc += v5.y; acc += v6.x; acc += v6.y; acc += v7.x; acc += v7.y;
            for (p = (p + 8) * 2; p < n; p++) acc += lds[p];
        }
        c1 = clock64();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = (unsigned long long) (acc * 1e6);
    }
    if (acc == 12345.678) out[2] = 1;
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 64);
    const int ns[3] = {208, 304, 1008};
    for (int n: ns)
        for (int block: {64, 512}) {
            printf("n = %4d, %3d threads:", n, block);
#define RUN(M)                                                                                             \
    hipLaunchKernelGGL(probe<M>, dim3(1), dim3(block), 2 * n * sizeof(double), 0, d, n, 0.5);              \
    hipLaunchKernelGGL(probe<M>, dim3(1), dim3(block), 2 * n * sizeof(double), 0, d, n, 0.5);              \
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);                                                            \
    printf("  mode %d: %6llu cycles (%.1f per term)", M, h[0], (double) h[0] / n);
            RUN(0) RUN(1) RUN(3) RUN(4)
            printf("\n");
        }
    return 0;
}
