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
static __constant__ uint64_t vft_gexp_tab_dev[256] = {VFT_GEXP_TABLE};
#define VFT_GLOG_TAB vft_glog_tab_dev
#define VFT_GEXP_TAB vft_gexp_tab_dev
#else
static const double vft_glog_tab_host[256] = {VFT_GLOG_TABLE};
static const uint64_t vft_gexp_tab_host[256] = {VFT_GEXP_TABLE};
#define VFT_GLOG_TAB vft_glog_tab_host
#define VFT_GEXP_TAB vft_gexp_tab_host
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

// exp() of the same library (its FMA build, __exp_fma): x = k ln2 / 128 + r, 2^(k/128) from a 128-entry table carrying
// a correction term, a degree-5 polynomial in r.  The matrix models' P(t) tables are exp(eigenvalue * rate * length)
// narrowed to numeric_t (expEigenRates, NJ.tcc:2020-2038 with fastexp level 0): in double precision the device's own exp
// differs from this in the last place of some arguments, enough to move a few optimised lengths in the ninth decimal and
// a handful of SH-like supports; with this one the tables are the reference's bit for bit.
VFT_GLOG_FN double vft_glibc_exp(double x) {
    uint64_t ix;
    memcpy(&ix, &x, 8);
    uint32_t abstop = (uint32_t) (ix >> 52) & 0x7ff;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;              // |x| < 2^-54
        if (abstop >= 0x409u) {                                          // |x| >= 1024, inf, nan
            if (ix == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (ix >> 63) ? 0.0 : INFINITY;
        }
        abstop = 0;                                                      // 512 <= |x| < 1024: scaled carefully below
    }
    const double kd0 = fma(x, VFT_GEXP_INVLN2N, VFT_GEXP_SHIFT);
    uint64_t ki;
    memcpy(&ki, &kd0, 8);
    const double kd = kd0 - VFT_GEXP_SHIFT;
    double r = fma(kd, VFT_GEXP_NEGLN2HIN, x);
    r = fma(kd, VFT_GEXP_NEGLN2LON, r);
    const unsigned idx = 2u * (unsigned) (ki & 127u);
    const uint64_t top = ki << 45;
    const uint64_t tbits = VFT_GEXP_TAB[idx];
    double tail;
    memcpy(&tail, &tbits, 8);
    uint64_t sbits = VFT_GEXP_TAB[idx + 1] + top;
    const double r2 = r * r;
    double p = fma(r, VFT_GEXP_C3, VFT_GEXP_C2);
    const double q = fma(r, VFT_GEXP_C5, VFT_GEXP_C4);
    p = fma(p, r2, r + tail);
    const double tmp = fma(r2 * r2, q, p);
    double scale;
    if (abstop == 0) {
        if ((ki & 0x80000000ull) == 0) {   // k > 0: the result may overflow
            sbits -= 1009ull << 52;
            memcpy(&scale, &sbits, 8);
            return 0x1p1009 * fma(scale, tmp, scale);
        }
        sbits += 1022ull << 52;            // k < 0: the result may be subnormal
        memcpy(&scale, &sbits, 8);
        const double st = scale * tmp;
        double y = scale + st;
        if (y < 1.0) {
            double lo = (scale - y) + st;
            const double hi = 1.0 + y;
            lo = ((1.0 - hi) + y) + lo;
            y = (hi + lo) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        return 0x1p-1022 * y;
    }
    memcpy(&scale, &sbits, 8);
    return fma(scale, tmp, scale);
}

#endif
