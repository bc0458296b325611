// Does the shader clock depend on how much of the chip is busy?  One dependent FMA chain per wavefront, timed with HIP events for
// grids of 1, 3, 256 and 4096 workgroups, with and without a second stream keeping the rest of the chip busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain(float *out, int n, float a, float b) {
    float x = (float) threadIdx.x;
    for (int i = 0; i < n; i++) x = __builtin_fmaf(x, a, b);
    if (x == 12345.678f) out[0] = x;
}
// shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) across a chain of n x 64 dependent FMAs
__global__ void clocks(unsigned long long *out, int n, float a, float b) {
    float x = (float) threadIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) x = __builtin_fmaf(x, a, b);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
    if (x == 12345.678f) out[2] = (unsigned long long) x;
}
// dependent double-precision chains, 64 operations per trip: plain multiply, multiply written as fma(x, a, 0), add
template <int OP>
__global__ void clocks64(unsigned long long *out, int n, double a) {
    double x = 1.0 + (double) threadIdx.x * 1e-9;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            if (OP == 0) x = x * a;
            else if (OP == 1) x = __builtin_fma(x, a, 0.0);
            else x = x + a;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
    if (x == 12345.678) out[2] = (unsigned long long) x;
}
__global__ void chain64(double *out, int n, double a, double b) {
    double x = (double) threadIdx.x;
    for (int i = 0; i < n; i++) x = x * a + b;
    if (x == 12345.678) out[0] = x;
}
__global__ void spin(float *out, volatile int *stop, float a, float b) {
    float x = (float) threadIdx.x;
    for (int round = 0; round < 400000 && !*stop; round++)   // (bounded: a few seconds at most)
        for (int i = 0; i < 4096; i++) x = __builtin_fmaf(x, a, b);
    if (x == 12345.678f) out[0] = x;
}
static double timeit(int grid, int block, int n, hipStream_t s, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain, dim3(grid), dim3(block), 0, s, out, 1000, 1.0001f, 0.5f);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(chain, dim3(grid), dim3(block), 0, s, out, n, 1.0001f, 0.5f);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float *out;
    hipMalloc(&out, 64);
    int *stopH, *stopD;
    hipHostMalloc(&stopH, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void **) &stopD, stopH, 0);
    hipStream_t s, s2;
    hipStreamCreate(&s);
    hipStreamCreate(&s2);
    const int n = 4000000;
    for (int rep = 0; rep < 2; rep++)
        for (int grid : {1, 3, 256, 4096}) {
            const double ms = timeit(grid, 64, n, s, out);
            printf("grid %5d x 64 threads: %8.3f ms  -> %.2f ns per dependent FMA\n", grid, ms, ms * 1e6 / n);
        }
    // the same single workgroup while 1000 other workgroups spin on a second stream
    *stopH = 0;
    hipLaunchKernelGGL(spin, dim3(1000), dim3(256), 0, s2, out, (volatile int *) stopD, 1.0001f, 0.5f);
    for (int rep = 0; rep < 3; rep++) {
        const double ms = timeit(1, 64, n, s, out);
        printf("grid     1 x 64 threads, chip kept busy: %8.3f ms  -> %.2f ns per dependent FMA\n", ms, ms * 1e6 / n);
    }
    *stopH = 1;
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        const double ms = timeit(1, 64, n, s, out);
        printf("grid     1 x 64 threads, idle again:     %8.3f ms  -> %.2f ns per dependent FMA\n", ms, ms * 1e6 / n);
    }
    unsigned long long *ck;
    hipHostMalloc(&ck, 64, hipHostMallocMapped);
    for (int grid : {1, 4096}) {
        const int m = 100000;
        hipLaunchKernelGGL(clocks, dim3(grid), dim3(64), 0, s, ck, m, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        printf("grid %5d: %llu shader-clock ticks, %llu ticks of 100 MHz for %d x 64 dependent FMAs: s_memtime runs at %.1f MHz; %.2f ns, %.2f s_memtime ticks per FMA\n",
               grid, ck[0], ck[1], m, (double) ck[0] / ((double) ck[1] / 100.0), (double) ck[1] * 10.0 / (64.0 * m), (double) ck[0] / (64.0 * m));
    }
    const char *names[3] = {"v_mul_f64", "v_fma_f64 (x * a + 0)", "v_add_f64"};
    for (int op = 0; op < 3; op++) {
        const int m = 50000;
        if (op == 0) hipLaunchKernelGGL(clocks64<0>, dim3(1), dim3(64), 0, s, ck, m, 1.0000001);
        else if (op == 1) hipLaunchKernelGGL(clocks64<1>, dim3(1), dim3(64), 0, s, ck, m, 1.0000001);
        else hipLaunchKernelGGL(clocks64<2>, dim3(1), dim3(64), 0, s, ck, m, 1.0000001);
        hipDeviceSynchronize();
        printf("dependent %-24s %.2f ns = %.2f shader clocks per operation\n", names[op], (double) ck[1] * 10.0 / (64.0 * m), (double) ck[0] / (64.0 * m));
    }
    return 0;
}
