// Device-side view of the arena and the per-column primitives shared by every kernel.
// All arithmetic follows the reference's types literally (see oracle/vft_oracle_impl.h for the same statements
// on the CPU and the NeighbourJoining.tcc lines they mirror); the file is compiled with -ffp-contract=off so
// no multiply-add is ever fused.
#pragma once
#include <hip/hip_runtime.h>
#include "vft_layout.h"

template <typename REAL>
struct Arena {
    VftDims d;
    const uint4 *leafT;
    REAL *profW;
    REAL *profF;
    uint4 *profC;
    ColMask *colMask;
    ColOff *colOff;
    int32_t *parent;
    REAL *diameter, *selfweight, *selfdist, *outDist;
    int32_t *nOutActive;
    // host-mapped mirrors of outDist / nOutActive (zero-copy): every refresh also lands in host memory, so the
    // host driver evaluates criteria (NJ.tcc:1099-1107) without fetching anything
    REAL *mOutDist;
    int32_t *mNOut;
    // out-profile, row-major: outW[nPos], outF[nPos][nCodes], outCD[nPos][nCodes] (only with a distance matrix)
    REAL *outW, *outF, *outCD;
    // distance matrix (NULL when %-different distances are used)
    const REAL *dmDist, *dmCodeFreq, *dmEigenval, *dmEigentot;
    // transition matrix (NULL for Jukes-Cantor): codefreq has nCodes+1 rows
    const REAL *tmStat, *tmStatInv, *tmEigenval, *tmCodeFreq, *tmEigenInv, *tmEigenInvT;
    const REAL *rates;
    const int32_t *ratecat;
    int32_t nRates;
    int32_t jcExact;   // Jukes-Cantor likelihoods the reference's way to the last bit: glibc's exp in P(t), the ordered total (vft_kernels_ml.h)
    // ML-phase profiles (vft_layout.h, "dense ML rows"): a node whose mlIs byte is set lives in mlW/mlC/mlF instead of
    // the tile streams.  Indexed by (node - nSeqs) * nPos + p; NULL until the ML phase allocates them.
    uint8_t *mlIs;
    REAL *mlW;
    uint8_t *mlC;
    REAL *mlF;
};

__device__ __forceinline__ uint32_t vft_byte(const uint4 &v, int b) {
    const uint32_t w = b < 8 ? (b < 4 ? v.x : v.y) : (b < 12 ? v.z : v.w);
    return (w >> ((b & 3) * 8)) & 0xFFu;
}

// reference code (0..nCodes-1 or 127) of a stored leaf byte
template <int NC>
__device__ __forceinline__ int vft_decode(uint32_t enc) {
    if (NC != 4) return (int) enc;
    return (enc & 0x10u) ? (__ffs((int) (enc & 0xFu)) - 1) : VFT_NOCODE_;
}

// weight of a column that stores none (vft_layout.h): 1 under a code or a vector, 0 for an empty gap
template <typename REAL>
__device__ __forceinline__ REAL vft_implicit_weight(int code, bool hasVector) {
    return (hasVector || code != VFT_NOCODE_) ? (REAL) 1 : (REAL) 0;
}

// One alignment column of one node, in the reference's terms.
template <typename REAL, int NC>
struct Col {
    REAL w;
    int code;
    bool vec;      // the reference holds a frequency vector here (code == NOCODE && w > 0)
    REAL f[NC];    // valid only when vec
};

// Element k of the vector with rank `rank` among the nvec vectors a tile holds at one column, whose block starts at
// vector slot offVec of the tile's stream (vft_layout.h).  4-state alphabets: the vectors back to back (16 / 32 bytes
// each - one load per lane).  20-state alphabets: the block is transposed in 16-byte pieces - piece e of all nvec
// vectors, then piece e + 1 ... - so that the lanes of a wavefront, which own consecutive ranks, read consecutive
// 16 bytes: a 160-byte vector per lane would otherwise touch 64 different cache lines per load instruction.
template <typename REAL, int NC>
__device__ __forceinline__ int64_t vft_fidx(uint32_t offVec, int nvec, int rank, int k) {
    if (NC == 4) return ((int64_t) offVec + rank) * NC + k;
    constexpr int E = 16 / (int) sizeof(REAL);   // elements per piece
    return (int64_t) offVec * NC + ((int64_t) (k / E) * nvec + rank) * E + (k % E);
}

template <typename REAL, int NC>
__device__ __forceinline__ void vft_load_col(const Arena<REAL> &A, int64_t node, int64_t p, Col<REAL, NC> &c) {
    const int lane = (int) (node & (VFT_TILE - 1));
    const int64_t tile = node >> 6;
    if (node < A.d.nSeqs) {
        const uint4 t = A.leafT[vft_leaf_idx(A.d, tile, (int) (p >> 4), lane)];
        c.code = vft_decode<NC>(vft_byte(t, (int) (p & 15)));
        c.w = c.code != VFT_NOCODE_ ? (REAL) 1 : (REAL) 0;
        c.vec = false;
    } else {
        const int64_t pt = tile - A.d.firstProfTile;
        const int64_t mi = vft_meta_idx(A.d, pt, p);
        const ColMask m = A.colMask[mi];
        const ColOff o = A.colOff[mi];
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint4 t = A.profC[vft_c_idx(A.d, pt, (int) (p >> 4), lane)];
        c.code = (int) vft_byte(t, (int) (p & 15));
        const bool hv = (m.vec >> lane) & 1ull;
        if ((m.w >> lane) & 1ull) c.w = A.profW[vft_wstream_base(A.d, pt) + o.w + __popcll(m.w & below)];
        else c.w = vft_implicit_weight<REAL>(c.code, hv);
        c.vec = c.w > 0 && c.code == VFT_NOCODE_;   // == hv (k_tile_commit)
        if (c.vec) {
            const REAL *src = A.profF + vft_fstream_base(A.d, pt);
            const int nvec = __popcll(m.vec), rank = __popcll(m.vec & below);
#pragma unroll
            for (int k = 0; k < NC; k++) c.f[k] = src[vft_fidx<REAL, NC>(o.vec, nvec, rank, k)];
        }
    }
}

// one column of a node's dense row (vft_layout.h)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_load_row(const Arena<REAL> &A, int64_t node, int64_t p, Col<REAL, NC> &c) {
    const int64_t idx = (node - A.d.nSeqs) * A.d.nPos + p;
    c.w = A.mlW[idx];
    c.code = (int) A.mlC[idx];
    c.vec = c.w > 0 && c.code == VFT_NOCODE_;
    if (c.vec) {
        const REAL *src = A.mlF + idx * NC;
#pragma unroll
        for (int k = 0; k < NC; k++) c.f[k] = src[k];
    }
}

// ML-phase read: dense row if the node has one, the tile streams otherwise (leaves, NJ-phase averages)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_load_col_ml(const Arena<REAL> &A, int64_t node, int64_t p, Col<REAL, NC> &c) {
    if (node >= A.d.nSeqs && A.mlIs != nullptr && A.mlIs[node - A.d.nSeqs]) {
        vft_load_row<REAL, NC>(A, node, p, c);
        return;
    }
    vft_load_col<REAL, NC>(A, node, p, c);
}

// ML-phase write of one column into the node's dense row (the caller sets mlIs[node - nSeqs] once per node)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_store_col_ml(const Arena<REAL> &A, int64_t node, int64_t p, REAL w, int code, const REAL *f) {
    const int64_t idx = (node - A.d.nSeqs) * A.d.nPos + p;
    A.mlW[idx] = w;
    A.mlC[idx] = (uint8_t) code;
    if (w > 0 && code == VFT_NOCODE_) {
        REAL *dst = A.mlF + idx * NC;
#pragma unroll
        for (int k = 0; k < NC; k++) dst[k] = f[k];
    }
}

template <typename REAL, int NC>
__device__ __forceinline__ REAL vft_pick(const REAL (&f)[NC], int code) {
    REAL v = 0;
#pragma unroll
    for (int k = 0; k < NC; k++) v = (k == code) ? f[k] : v;
    return v;
}

// SSE128 / AVX256 horizontal order: four strided accumulators, then (s0+s1)+(s2+s3)
template <typename REAL, int NC>
__device__ __forceinline__ REAL vft_red4_mul3(const REAL *a, const REAL *b, const REAL *c) {
    REAL s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NC; i += 4)
#pragma unroll
        for (int l = 0; l < 4; l++) {
            REAL p = a[i + l] * b[i + l];
            p = p * c[i + l];
            s[l] = p + s[l];
        }
    const REAL lo = s[0] + s[1], hi = s[2] + s[3];
    return lo + hi;
}
template <typename REAL, int NC>
__device__ __forceinline__ REAL vft_red4_mul(const REAL *a, const REAL *b) {
    REAL s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NC; i += 4)
#pragma unroll
        for (int l = 0; l < 4; l++) {
            REAL p = a[i + l] * b[i + l];
            s[l] = p + s[l];
        }
    const REAL lo = s[0] + s[1], hi = s[2] + s[3];
    return lo + hi;
}
template <typename REAL, int NC>
__device__ __forceinline__ REAL vft_red4_sum(const REAL *a) {
    REAL s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NC; i += 4)
#pragma unroll
        for (int l = 0; l < 4; l++) s[l] = a[i + l] + s[l];
    const REAL lo = s[0] + s[1], hi = s[2] + s[3];
    return lo + hi;
}

// Where a kernel reads the distance-matrix tables from: the arena (global memory, the default of every function below), or a copy
// a latency-bound kernel made in LDS (k_walk_server: a table read from global memory inside a loop that also stores is a wait for
// the stores).  The tables: distances [NC][NC], codeFreq [NC][NC], eigenval [NC], eigentot [NC].
template <typename REAL>
struct DmGlobal {
    const REAL *dist, *codeFreq, *eigenval, *eigentot;
    __device__ __forceinline__ explicit DmGlobal(const Arena<REAL> &A) : dist(A.dmDist), codeFreq(A.dmCodeFreq), eigenval(A.dmEigenval), eigentot(A.dmEigentot) {}
};
template <typename REAL>
struct DmLds {
    const __attribute__((address_space(3))) REAL *dist, *codeFreq, *eigenval, *eigentot;
};

// profileDistPiece (NJ.tcc:900-941).  cd2 = codeDist row of profile 2 for this column, or nullptr.
template <typename REAL, int NC, typename DM>
__device__ __forceinline__ double vft_piece(const Arena<REAL> &A, const Col<REAL, NC> &c1, const Col<REAL, NC> &c2,
                                            const REAL *cd2, const DM &T) {
    if (A.dmDist) {
        if (c1.code != VFT_NOCODE_ && c2.code != VFT_NOCODE_) return (double) T.dist[c1.code * NC + c2.code];
        if (cd2 != nullptr && c1.code != VFT_NOCODE_) return (double) cd2[c1.code];
        REAL f1[NC], f2[NC], ev[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            f1[k] = c1.vec ? c1.f[k] : T.codeFreq[(c1.code == VFT_NOCODE_ ? 0 : c1.code) * NC + k];
            f2[k] = c2.vec ? c2.f[k] : T.codeFreq[(c2.code == VFT_NOCODE_ ? 0 : c2.code) * NC + k];
            ev[k] = T.eigenval[k];
        }
        if ((!c1.vec && c1.code == VFT_NOCODE_) || (!c2.vec && c2.code == VFT_NOCODE_)) return 10.0;
        return (double) vft_red4_mul3<REAL, NC>(f1, f2, ev);
    }
    if (c1.code != VFT_NOCODE_) {
        if (c2.code != VFT_NOCODE_) return c1.code == c2.code ? 0.0 : 1.0;
        if (!c2.vec) return 10.0;
        return 1.0 - (double) vft_pick<REAL, NC>(c2.f, c1.code);
    }
    if (c2.code != VFT_NOCODE_) {
        if (!c1.vec) return 10.0;
        return 1.0 - (double) vft_pick<REAL, NC>(c1.f, c2.code);
    }
    if (!c1.vec || !c2.vec) return 10.0;
    double piece = 1.0;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const REAL p = c1.f[k] * c2.f[k];
        piece -= (double) p;
    }
    return piece;
}

template <typename REAL, int NC>
__device__ __forceinline__ double vft_piece(const Arena<REAL> &A, const Col<REAL, NC> &c1, const Col<REAL, NC> &c2, const REAL *cd2) {
    return vft_piece<REAL, NC, DmGlobal<REAL>>(A, c1, c2, cd2, DmGlobal<REAL>(A));
}

// One column's addends to (denom, top) of a profile distance (profileDist NJ.tcc:1176-1182, seqDist :1614-1620), parked in LDS for the
// in-order sum of the pair kernels (vft_kernels_nj.h vft_pair_wave / vft_pair_block, vft_kernels_walk.h)
template <typename REAL, int NC, typename DM>
__device__ __forceinline__ void vft_pair_addends(const Arena<REAL> &A, bool leaves, bool jIsOut, int64_t p,
                                                 const Col<REAL, NC> &c1, const Col<REAL, NC> &c2, double *sW, double *sT, const DM &T) {
    double wgt = 0.0, term = 0.0;
    if (leaves) {   // seqDist with a distance matrix (NJ.tcc:1614-1620): top += distances[c1][c2], in order
        if (c1.code != VFT_NOCODE_ && c2.code != VFT_NOCODE_) {
            wgt = 1.0;
            term = A.dmDist ? (double) T.dist[c1.code * NC + c2.code] : (c1.code != c2.code ? 1.0 : 0.0);
        }
    } else if (c1.w > 0 && c2.w > 0) {
        const REAL ww = c1.w * c2.w;
        wgt = (double) ww;
        term = wgt * vft_piece<REAL, NC, DM>(A, c1, c2, (jIsOut && A.outCD) ? A.outCD + p * NC : nullptr, T);
    }
    sW[p] = wgt;
    sT[p] = term;
}
template <typename REAL, int NC>
__device__ __forceinline__ void vft_pair_addends(const Arena<REAL> &A, bool leaves, bool jIsOut, int64_t p,
                                                 const Col<REAL, NC> &c1, const Col<REAL, NC> &c2, double *sW, double *sT) {
    vft_pair_addends<REAL, NC, DmGlobal<REAL>>(A, leaves, jIsOut, p, c1, c2, sW, sT, DmGlobal<REAL>(A));
}

// setOutDistance's closed form (NJ.tcc:1046-1053): numeric_t products, one double division.
template <typename REAL>
__device__ __forceinline__ REAL vft_out_distance(REAL dist, REAL weight, int64_t nActive, REAL selfweight,
                                                 REAL selfdist, REAL diameter, double totdiam) {
    const REAL t1 = dist * weight;
    const REAL t2 = t1 * (REAL) nActive;
    const REAL t3 = selfweight * selfdist;
    const REAL t4 = t2 - t3;
    const REAL topr = (REAL) (nActive - 1) * t4;
    const REAL b1 = weight * (REAL) nActive;
    const REAL botr = b1 - selfweight;
    const double top = topr, bottom = botr;
    const double pd = top / bottom;
    const REAL dn = diameter * (REAL) (nActive - 1);
    const double r = bottom > 0.01 ? pd - (double) dn - (totdiam - (double) diameter) : 3.0;
    return (REAL) r;
}

// setCriterion's formula with the stale out-distance rescale (NJ.tcc:1099-1107)
template <typename REAL>
__device__ __forceinline__ REAL vft_criterion(REAL dist, REAL outI, int64_t nOutI, REAL outJ, int64_t nOutJ,
                                              int64_t nActive) {
    double oi = outI, oj = outJ;
    if (nOutI != nActive) oi *= (double) (nActive - 1) / (double) (nOutI - 1);
    if (nOutJ != nActive) oj *= (double) (nActive - 1) / (double) (nOutJ - 1);
    return (REAL) ((double) dist - (oi + oj) / (double) (nActive - 2));
}

// Wave-uniform, read-only data (the staged query) goes through the scalar cache: a 16/32-byte s_load per column
// instead of per-lane vector loads.  The constant address space tells the compiler the data cannot change under
// the kernel, a vector type keeps it from splitting the load into per-lane selects of addresses.
template <typename REAL> struct UVec4;
template <> struct UVec4<float> {
    typedef float __attribute__((ext_vector_type(4))) type;
};
template <> struct UVec4<double> {
    typedef double __attribute__((ext_vector_type(4))) type;
};
template <typename REAL>
__device__ __forceinline__ typename UVec4<REAL>::type vft_uniform_load4(const REAL *p) {
    typedef const __attribute__((address_space(4))) typename UVec4<REAL>::type *cp_t;
    return *(cp_t) p;
}
template <typename REAL>
__device__ __forceinline__ REAL vft_uniform_load(const REAL *p) {
    typedef const __attribute__((address_space(4))) REAL *cp_t;
    return *(cp_t) p;
}

// Sort key of a hit: ascending criterion, ties by DESCENDING node id.  Smaller key = earlier.
__device__ __forceinline__ uint32_t vft_order_f32(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint64_t vft_order_f64(double x) {
    const uint64_t u = (uint64_t) __double_as_longlong(x);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
