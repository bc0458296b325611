// Latency probe for the walk server (csrc/vft_kernels_walk.h): what does it cost to hand a command to a RESIDENT kernel and get a
// number back, without a launch?  Build: hipcc --offload-arch=gfx950 -O3 -o mailbox_probe mailbox_probe.hip
//   A  host mailbox:   the command is an 8-byte {data, tag} granule in pinned host memory, the kernel polls it over PCIe (one or three
//                      polls in flight), answers with one 8-byte granule in pinned host memory; the host spins on that.
//   B  device mailbox: the same with the command word in device memory written by the CPU through the PCIe aperture (tried in a
//                      child process: a box without a CPU-visible aperture faults there).
//   C  six workgroups: every workgroup polls the command and answers by itself; the host waits for all six.
//   D  hop:            workgroup 0 takes the command, publishes a 4 KB payload (16-byte sc1 stores) + flag in device memory, workgroups
//                      1..5 wait for the flag, read the payload (sc1 loads), answer; same / different XCD by block placement.
// Every spin in the kernel is bounded (2 s of the 100 MHz clock); the host's spins are bounded too.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csetjmp>
#include <csignal>
#include <unistd.h>

#define CHK(x)                                                                          \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(2);                                                                    \
        }                                                                               \
    } while (0)

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 ld_sys(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_dev(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define LIMIT_TICKS 200000000ull   // 2 s

// A / B / C: `stride` blocks apart answer (the others exit at once: placement tests); every answering block polls `cmd`
template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_pingpong(const u64 *cmd, u64 *res, int rounds, int stride) {
    if (blockIdx.x % stride) return;
    const int w = blockIdx.x / stride;
    if (threadIdx.x >= 64) return;
    const u64 t0 = wall_clock64();
    for (u64 seq = 1; seq <= (u64) rounds; seq++) {
        u64 v;
        if (INFLIGHT == 1) {
            do {
                v = ld_sys(cmd);
                if (wall_clock64() - t0 > LIMIT_TICKS) return;
            } while ((v >> 32) != seq);
        } else {
            u64 a = ld_sys(cmd), b = ld_sys(cmd), c2 = ld_sys(cmd);
            for (;;) {
                if ((a >> 32) == seq) {
                    v = a;
                    break;
                }
                a = b;
                b = c2;
                c2 = ld_sys(cmd);
                if (wall_clock64() - t0 > LIMIT_TICKS) return;
            }
        }
        if (threadIdx.x == 0) st_sys(&res[8 * w], (seq << 32) | ((v + 1) & 0xFFFFFFFFu));
    }
}

// W: the walk server's own poll - 64 granules of 8 bytes (one per lane) per slot, a command of `cnt` granules is there when the first
// cnt tags match; the host writes cnt granules per round
__global__ __launch_bounds__(256) void k_widepoll(const u64 *cmd, u64 *res, int rounds, int stride, int cnt) {
    if (blockIdx.x % stride) return;
    const int w = blockIdx.x / stride;
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const u64 t0 = wall_clock64();
    const u64 need = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
    for (u64 seq = 1; seq <= (u64) rounds; seq++) {
        for (;;) {
            const u64 g = ld_sys(cmd + lane);
            const u64 ok = __ballot((g >> 32) == seq);
            if ((ok & need) == need) break;
            if (wall_clock64() - t0 > LIMIT_TICKS) return;
        }
        if (threadIdx.x == 0) st_sys(&res[8 * w], (seq << 32) | 1u);
    }
}

// D: block 0 = producer, blocks `stride`, 2*stride ... 5*stride = consumers (the rest exit)
__global__ __launch_bounds__(256) void k_hop(const u64 *cmd, u64 *res, uint4 *payload, u64 *flag, int rounds, int stride, int payloadVec) {
    if (blockIdx.x % stride) return;
    const int w = blockIdx.x / stride;
    if (w > 5) return;
    const u64 t0 = wall_clock64();
    __shared__ u64 sv;
    for (u64 seq = 1; seq <= (u64) rounds; seq++) {
        if (w == 0) {
            if (threadIdx.x == 0) {
                u64 v;
                do {
                    v = ld_sys(cmd);
                    if (wall_clock64() - t0 > LIMIT_TICKS) break;
                } while ((v >> 32) != seq);
                sv = v;
            }
            __syncthreads();
            if ((sv >> 32) != seq) return;
            // the payload: 16-byte write-through stores, drained, then the flag
            for (int t = threadIdx.x; t < payloadVec; t += 256) {
                u32x4 x;
                x.x = (unsigned) seq;
                x.y = x.z = x.w = (unsigned) t;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&payload[t]), "v"(x) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                st_dev(flag, seq);
                st_sys(&res[0], (seq << 32) | 1u);
            }
        } else {
            if (threadIdx.x == 0) {
                u64 v;
                do {
                    v = ld_dev(flag);
                    if (wall_clock64() - t0 > LIMIT_TICKS) break;
                } while (v != seq);
                sv = v;
            }
            __syncthreads();
            if (sv != seq) return;
            unsigned bad = 0;
            for (int t = threadIdx.x; t < payloadVec; t += 256) {
                u32x4 x;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(x) : "v"(&payload[t]) : "memory");
                bad += x.x != (unsigned) seq;
            }
            bad = __syncthreads_or(bad);
            if (threadIdx.x == 0) st_sys(&res[8 * w], (seq << 32) | (bad ? 0xBADu : 1u));
        }
    }
}

static sigjmp_buf faultJmp;
static void on_fault(int) { siglongjmp(faultJmp, 1); }

static double drive_wide(volatile u64 *cmd, volatile u64 *res, int nRes, int rounds, int cnt, bool wc) {
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    for (u64 seq = 1; seq <= (u64) rounds; seq++) {
        for (int g = cnt - 1; g >= 0; g--) cmd[g] = (seq << 32) | (u64) g;
        if (wc) __builtin_ia32_sfence();
        for (int w = 0; w < nRes; w++) {
            long spins = 0;
            while ((res[8 * w] >> 32) != seq)
                if (++spins > 400000000L) return -1.0;
        }
    }
    return (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0) * 1e6 / rounds;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// host side of a ping-pong over nRes answering blocks; returns microseconds per round, < 0 on timeout
static double drive(volatile u64 *cmd, volatile u64 *res, int nRes, int rounds, bool wc) {
    const double t0 = now();
    for (u64 seq = 1; seq <= (u64) rounds; seq++) {
        *cmd = (seq << 32) | 7u;
        if (wc) __builtin_ia32_sfence();
        for (int w = 0; w < nRes; w++) {
            long spins = 0;
            while ((res[8 * w] >> 32) != seq)
                if (++spins > 400000000L) return -1.0;
        }
    }
    return (now() - t0) * 1e6 / rounds;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    CHK(hipSetDevice(0));
    {
        int largeBar = -1;
        (void) hipDeviceGetAttribute(&largeBar, hipDeviceAttributeIsLargeBar, 0);
        printf("hipDeviceAttributeIsLargeBar = %d\n", largeBar);
    }
    u64 *hCmd, *hRes, *dCmd, *dRes;
    CHK(hipHostMalloc((void **) &hCmd, 4096, hipHostMallocMapped));
    CHK(hipHostMalloc((void **) &hRes, 4096, hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void **) &dCmd, hCmd, 0));
    CHK(hipHostGetDevicePointer((void **) &dRes, hRes, 0));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto reset = [&]() {
        memset(hCmd, 0, 4096);
        memset(hRes, 0, 4096);
    };
    // A: host mailbox, one block
    reset();
    hipLaunchKernelGGL(k_pingpong<1>, dim3(1), dim3(256), 0, st, dCmd, dRes, rounds, 1);
    printf("A1 host mailbox, 1 workgroup, 1 poll in flight : %.2f us per round trip\n", drive(hCmd, hRes, 1, rounds, false));
    CHK(hipStreamSynchronize(st));
    reset();
    hipLaunchKernelGGL(k_pingpong<3>, dim3(1), dim3(256), 0, st, dCmd, dRes, rounds, 1);
    printf("A3 host mailbox, 1 workgroup, 3 polls in flight: %.2f us per round trip\n", drive(hCmd, hRes, 1, rounds, false));
    CHK(hipStreamSynchronize(st));
    // C: six blocks
    reset();
    hipLaunchKernelGGL(k_pingpong<1>, dim3(6), dim3(256), 0, st, dCmd, dRes, rounds, 1);
    printf("C1 host mailbox, 6 workgroups (6 XCDs), 1 poll  : %.2f us per round trip\n", drive(hCmd, hRes, 6, rounds, false));
    CHK(hipStreamSynchronize(st));
    reset();
    hipLaunchKernelGGL(k_pingpong<3>, dim3(6), dim3(256), 0, st, dCmd, dRes, rounds, 1);
    printf("C3 host mailbox, 6 workgroups (6 XCDs), 3 polls : %.2f us per round trip\n", drive(hCmd, hRes, 6, rounds, false));
    CHK(hipStreamSynchronize(st));
    reset();
    hipLaunchKernelGGL(k_pingpong<3>, dim3(41), dim3(256), 0, st, dCmd, dRes, rounds, 8);
    printf("C3 host mailbox, 6 workgroups (one XCD), 3 polls: %.2f us per round trip\n", drive(hCmd, hRes, 6, rounds, false));
    CHK(hipStreamSynchronize(st));
    for (int cnt : {1, 14, 22, 62}) {
        reset();
        hipLaunchKernelGGL(k_widepoll, dim3(41), dim3(256), 0, st, dCmd, dRes, rounds, 8, cnt);
        printf("W  host mailbox, 6 workgroups (one XCD), 64-lane poll, command of %2d granules: %.2f us per round trip\n", cnt, drive_wide(hCmd, hRes, 6, rounds, cnt, false));
        CHK(hipStreamSynchronize(st));
    }
    // for comparison: an empty launch + wait
    {
        reset();
        const double t0 = now();
        for (int r = 0; r < 2000; r++) {
            hipLaunchKernelGGL(k_pingpong<1>, dim3(6), dim3(256), 0, st, dCmd, dRes, 0, 1);
            CHK(hipStreamSynchronize(st));
        }
        printf("-- empty 6-workgroup launch + hipStreamSynchronize: %.2f us\n", (now() - t0) * 1e6 / 2000);
    }
    // D: hop through device memory
    uint4 *payload;
    u64 *flag;
    CHK(hipMalloc((void **) &payload, 1 << 20));
    CHK(hipMalloc((void **) &flag, 4096));
    for (int stride : {1, 8}) {
        for (int pv : {0, 256, 2048}) {
            reset();
            CHK(hipMemset(flag, 0, 4096));
            CHK(hipDeviceSynchronize());
            hipLaunchKernelGGL(k_hop, dim3(5 * stride + 1), dim3(256), 0, st, dCmd, dRes, payload, flag, rounds, stride, pv);
            const double us = drive(hCmd, hRes, 6, rounds, false);
            unsigned bad = 0;
            for (int w = 1; w < 6; w++) bad += (hRes[8 * w] & 0xFFFFFFFFu) == 0xBADu;
            printf("D  hop, %s, payload %5d B: %.2f us per round trip%s\n", stride == 1 ? "six XCDs" : "one XCD ", pv * 16, us, bad ? "  (STALE PAYLOAD SEEN)" : "");
            CHK(hipStreamSynchronize(st));
        }
    }
    // B: the command word in device memory, written by the CPU (the first touch under a SIGSEGV / SIGBUS handler)
    fflush(stdout);
    int status = 0;
    {
        u64 *dMail = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **) &dMail, 4096, hipDeviceMallocFinegrained);
        if (e != hipSuccess) {
            printf("B  device mailbox: hipExtMallocWithFlags(finegrained) failed: %s\n", hipGetErrorString(e));
        } else {
            CHK(hipMemset(dMail, 0, 4096));
            CHK(hipDeviceSynchronize());
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof(at));
            (void) hipPointerGetAttributes(&at, dMail);
            printf("B  device mailbox at %p (host pointer %p)\n", (void *) dMail, at.hostPointer);
            fflush(stdout);
            signal(SIGSEGV, on_fault);
            signal(SIGBUS, on_fault);
            if (sigsetjmp(faultJmp, 1) == 0) {   // can the CPU touch it at all?
                volatile u64 x = *(volatile u64 *) dMail;
                (void) x;
                status = 0;
            } else {
                status = 1;
            }
            signal(SIGSEGV, SIG_DFL);
            signal(SIGBUS, SIG_DFL);
            if (status == 0) {
                reset();
                hipLaunchKernelGGL(k_pingpong<1>, dim3(1), dim3(256), 0, st, dMail, dRes, rounds, 1);
                printf("B1 device mailbox (CPU writes through the aperture), 1 workgroup: %.2f us per round trip\n", drive(dMail, hRes, 1, rounds, true));
                CHK(hipStreamSynchronize(st));
                CHK(hipMemset(dMail, 0, 4096));
                reset();
                hipLaunchKernelGGL(k_pingpong<1>, dim3(6), dim3(256), 0, st, dMail, dRes, rounds, 1);
                printf("B6 device mailbox, 6 workgroups: %.2f us per round trip\n", drive(dMail, hRes, 6, rounds, true));
                CHK(hipStreamSynchronize(st));
                for (int cnt : {1, 14, 22, 62}) {
                    CHK(hipMemset(dMail, 0, 4096));
                    reset();
                    hipLaunchKernelGGL(k_widepoll, dim3(41), dim3(256), 0, st, dMail, dRes, rounds, 8, cnt);
                    printf("BW device mailbox, 6 workgroups (one XCD), 64-lane poll, command of %2d granules: %.2f us per round trip\n", cnt, drive_wide(dMail, hRes, 6, rounds, cnt, true));
                    CHK(hipStreamSynchronize(st));
                }
            } else {
                printf("B  device mailbox: the CPU cannot touch device memory on this box (child status %d)\n", status);
            }
        }
    }
    return 0;
}
