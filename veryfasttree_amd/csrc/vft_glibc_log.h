// log() as this image's glibc (2.35) computes it on an FMA-capable x86-64, bit for bit, for positive normal doubles:
// the table-driven algorithm of ARM's optimized-routines (x = 2^k z, z in [0x1.6p-1, 0x1.6p0), 128 subintervals with
// tabulated 1/c and log c, a degree-5 polynomial in r = z/c - 1; a degree-11 polynomial with a double-double head for
// x within [1 - 2^-4, 1 + 0x1.09p-4)) with the fused multiply-adds exactly where the library's FMA build has them
// (libm's __log_fma, selected by its ifunc on every CPU with FMA + AVX2).  Everything outside the fma() calls is
// compiled without contraction (-ffp-contract=off is part of the build contract).
//
// Why the ML kernels need it: pairLogLk ends with std::log (NJ.tcc:1444) and Brent's line searches use the values
// arithmetically; with the likelihood of a float-precision matrix model being rough, a one-ulp difference in one
// logarithm is enough for a search to end on another point.  Usable from host code (the CPU
// test compares it with libm on 10^8 arguments) and from device code.
#ifndef VFT_GLIBC_LOG_H
#define VFT_GLIBC_LOG_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#include "vft_glibc_log_data.h"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define VFT_GLOG_FN __host__ __device__ __forceinline__
#else
#define VFT_GLOG_FN static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
static __constant__ double vft_glog_tab_dev[256] = {VFT_GLOG_TABLE};
#define VFT_GLOG_TAB vft_glog_tab_dev
#else
static const double vft_glog_tab_host[256] = {VFT_GLOG_TABLE};
#define VFT_GLOG_TAB vft_glog_tab_host
#endif

VFT_GLOG_FN double vft_glibc_log(double x) {
    uint64_t ix;
    memcpy(&ix, &x, 8);
    const uint64_t lo1 = 0x3fee000000000000ull;               // 1 - 2^-4
    if (ix - lo1 < 0x0003090000000000ull) {                   // [1 - 2^-4, 1 + 0x1.09p-4)
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        double q1 = fma(r, VFT_GLOG_B2, VFT_GLOG_B1);
        double q2 = fma(r, VFT_GLOG_B5, VFT_GLOG_B4);
        double q3 = fma(r, VFT_GLOG_B8, VFT_GLOG_B7);
        q1 = fma(r2, VFT_GLOG_B3, q1);
        q2 = fma(r2, VFT_GLOG_B6, q2);
        q3 = fma(r2, VFT_GLOG_B9, q3);
        q3 = fma(r3, VFT_GLOG_B10, q3);
        q2 = fma(q3, r3, q2);
        const double poly = fma(q2, r3, q1);
        // r split into rhi + rlo so that rhi * rhi is exact
        const double t = fma(r, 0x1p27, r);
        const double rhi = fma(-0x1p27, r, t);
        const double rlo = r - rhi;
        const double sq = rhi * rhi;
        const double hi = fma(sq, VFT_GLOG_B0, r);
        double lo = fma(sq, VFT_GLOG_B0, r - hi);
        lo = fma(VFT_GLOG_B0 * rlo, rhi + r, lo);
        const double y = fma(poly, r3, lo);
        return y + hi;
    }
    // x = 2^k z, z in [0x1.6p-1, 0x1.6p0)
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int) ((tmp >> 45) & 127);
    const int64_t k = (int64_t) tmp >> 52;
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    double z;
    memcpy(&z, &iz, 8);
    const double invc = VFT_GLOG_TAB[2 * i], logc = VFT_GLOG_TAB[2 * i + 1];
    const double r = fma(z, invc, -1.0);
    const double kd = (double) k;
    const double w = fma(kd, VFT_GLOG_LN2HI, logc);
    const double hi = w + r;
    const double lo = fma(kd, VFT_GLOG_LN2LO, (w - hi) + r);
    const double r2 = r * r;
    double p = fma(r, VFT_GLOG_A2, VFT_GLOG_A1);
    const double q = fma(r, VFT_GLOG_A4, VFT_GLOG_A3);
    const double t = fma(r2, VFT_GLOG_A0, lo);
    p = fma(q, r2, p);
    const double y = fma(r * r2, p, t);
    return y + hi;
}

#endif
