// The walk server: the steps of a host-driven refinement walk (SPR chains, NJ.tcc:1805-1927 / :6185-6312; the minimum-evolution NNIs of
// the one-thread order, NJ.tcc:5797-5990) WITHOUT a launch per step.
//
// A step = the unweighted averages queued since the last step (recomputeProfile, up-profiles down a path: a chain, later ones read
// earlier outputs) + the six raw profile distances AB AC AD BC BD CD of the quartet they lead up to (chooseNNI, NJ.tcc:4836-4846).
// Round 4 did that in one launch per step (k_walk_step_args, since removed): ~11 us of launch + completion wait around ~10 us of work on
// six workgroups that each ran the whole chain, and a second launch for the 28 % of steps that rewrite a node.  Here six workgroups stay
// RESIDENT for a whole round.  The host keeps what a CPU is good at - the tree, the cache flags, the pointer chasing of traverseSPR (a wavefront needs
// ~12 us per step for that) - and hands every step over as a command in a mailbox the workgroups poll:
//
//   host      writes the command as self-tagged 8-byte granules {data, seq} into slot seq % RING of the mailbox (pinned host memory,
//             or device memory written through the PCIe aperture); no ordering between the granules is needed, a granule is
//             valid when its tag is the expected sequence number                        (round trip measured: 2.3-2.7 us, against 11.4)
//   chain     the alignment's columns are owned in slices of 64 by (workgroup, wavefront) pairs, the SAME owner for the whole life of
//             the server: a thread takes its column through the whole chain, reading earlier outputs back from where it stored
//             them itself.  No node is ever written by two workgroups, so steps that rewrite a node need nothing special, and a
//             long alignment's chain spreads over six CUs with one wavefront per SIMD (at 1 000 columns one workgroup's SIMDs were
//             issue-bound with four).  Rows are stored write-through (sc1), drained, then the workgroup raises chainDone[w] = seq.
//   pairs     workgroup w waits for all six chainDone flags, reads its pair's two rows (sc1 loads: another workgroup wrote them),
//             parks the addends of all columns in LDS, raises readsDone[w] = seq (the next chain may overwrite the rows now), adds the
//             addends in column order (two lanes: `top` and `denom`, the reference's sequence of double additions, exactly as
//             vft_pair_block) and sends the distance to the host as one or two self-tagged granules - no fence, no flag.
// Same operations on the same values in the same order as k_average_chain + k_pairs_fused: the trees stay
// byte-identical (tests/test_gpu_walk_server.py and every SPR / NNI fixture, which run through the server by default).
// Every spin is bounded: a workgroup that sees no command for idleTicks, or no flag for flagTicks, reports a status and exits.
#ifndef VFT_KERNELS_WALK_H
#define VFT_KERNELS_WALK_H
#include "vft_glibc_log.h"

#define VFT_WS_WG_OF(NC) ((NC) == 4 ? 512 : 256)   // threads per workgroup: 8 wavefronts for 4-state columns, 4 for 20-state ones (a pair of
                                                  // columns of four 20-state profiles does not fit 256 registers), each owning column slices
#define VFT_WS_NWG 6         // one workgroup per pair of the quartet
#define VFT_WS_GRAN 64       // granules per command slot (one per lane of the polling wavefront)
#define VFT_WS_RING 16       // command slots
#define VFT_WS_MAXOPS 19     // averages per command: 5 + 3 * 19 = 62 granules
#define VFT_WS_RESG 16       // result granules per slot: workgroup w writes [2 w] (and [2 w + 1]: the high half of a double)
#define VFT_WS_CMD_WORK 1u
#define VFT_WS_CMD_STOP 2u
// A DUAL command (round 6) carries BOTH continuations of an SPR chain: which NNI follows step k is one comparison of step k's own
// distances (findSPRSteps, NJ.tcc:1805-1859: "criteria[1] < criteria[2]"), so the host - which cannot know the outcome yet - builds
// the next step for either outcome and hands both over while step k is still running; the workgroups exchange step k's six distances
// through device memory, evaluate the comparison themselves (logCorrect with glibc's log, bit for bit: vft_glibc_log.h) and take
// their alternative without a host round trip.  Header: bit 18 = dual, bits 8-15 = averages of alternative 0 (B and C swapped:
// criteria[1] < criteria[2]), bits 24-31 = averages of alternative 1, bit 19 / 20 = alternative 0 / 1 is NOT a device step (the chain
// ends there, the quartet is answered from the host's memo table, too many averages): the command then counts as an empty one;
// bit 21 = logCorrect's scoredist flavour (NJ.tcc:322-330).
// Granules: [0] header, [1..4] quartet 0, [5 .. 5 + 3 n0) averages 0, then quartet 1 (4) and averages 1 (3 n1): n0 + n1 <= 18.
#define VFT_WS_DUAL_BIT 18
#define VFT_WS_DUAL_MAXOPS 18
#define VFT_WS_RES_CHOICE 12   // result granule of a dual command: {seq, 1 | alternative << 1 | not-a-device-step << 2} (workgroup 0)
#define VFT_WS_FLAG_DIST 16    // flags[16 + 16 * (seq & 1) + 2 w (+ 1)]: workgroup w's LOG-CORRECTED distance of command seq (a double), tagged low / high half
#define VFT_WS_SC1 16        // aux bits of the buffer intrinsics on gfx950: write-through stores / L1-bypassing loads

struct WalkServerArgs {
    const unsigned long long *mail;   // [RING][GRAN] command granules {data (low 32 bits), tag (high 32 bits)}
    unsigned long long *res;          // host-mapped: [RING][RESG] result granules
    unsigned long long *flags;        // device: [0..5] chainDone, [8..13] readsDone
    unsigned long long *status;       // host-mapped: [w] = how workgroup w ended (1 stop command, 2 idle limit, 3 flag limit)
    unsigned long long *ticks;        // device, tools builds only: per-phase clock ticks of workgroup 0
    double tol;
    uint32_t firstSeq;                // sequence number of the first command this launch will see
    int32_t stride;                   // block b works iff b % stride == 0 (stride 8: all six on one XCD - for speed only)
    long long idleTicks, flagTicks;   // give-up limits (100 MHz ticks)
};

#ifdef VFT_WALK_TIMING   // tools-only variant build (tools/walk_ticks.py): thread 0 of workgroup 0 adds up the ticks of its phases
#define VFT_WS_TICK(k)                                                   \
    do {                                                                 \
        if (w == 0 && threadIdx.x == 0) {                                \
            const unsigned long long now_ = wall_clock64();              \
            atomicAdd(&S.ticks[k], now_ - wsTick_);                      \
            wsTick_ = now_;                                              \
        }                                                                \
    } while (0)
#else
#define VFT_WS_TICK(k) do { } while (0)
#endif

typedef unsigned int vft_ws_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int vft_ws_u32x2 __attribute__((ext_vector_type(2)));

// One row of the refinement phase as three buffer resources (weights, codes, vectors).  The node id is wave-uniform, so the resources
// live in scalar registers and a column costs one 32-bit offset per array.  Reads beyond the alignment return zeros and writes there
// are dropped by the bounds check: whole wavefronts run without a tail test.  A LEAF looks the same to the loads: its weight and
// vector resources are empty (every read returns zero, no memory traffic) and its code resource is the leaf's byte lane of its tile,
// so that a column is always the same three loads with no branch around them - the compiler's s_waitcnt bookkeeping stays exact and
// loads requested an average ahead really stay in flight (a load inside a branch makes every later wait a wait for everything).
template <typename REAL, int NC>
struct WsRow {
    __amdgpu_buffer_rsrc_t w, c, f;
    bool leaf;
};
template <typename REAL, int NC>
__device__ __forceinline__ WsRow<REAL, NC> vft_ws_row(const Arena<REAL> &A, int32_t node) {
    WsRow<REAL, NC> R;
    const int32_t nPos = (int32_t) A.d.nPos;
    R.leaf = node < (int32_t) A.d.nSeqs;
    const int64_t idx = (int64_t) (R.leaf ? 0 : node - (int32_t) A.d.nSeqs) * nPos;
    const char *leafBase = (const char *) A.leafT + (((int64_t) ((R.leaf ? node : 0) >> 6) * A.d.nChunk) * VFT_TILE + (node & (VFT_TILE - 1))) * 16;
    R.w = __builtin_amdgcn_make_buffer_rsrc((void *) (A.mlW + idx), 0, R.leaf ? 0 : nPos * (int) sizeof(REAL), 0x00020000);
    R.c = __builtin_amdgcn_make_buffer_rsrc(R.leaf ? (void *) leafBase : (void *) (A.mlC + idx), 0, R.leaf ? A.d.nChunk * (VFT_TILE * 16) : nPos, 0x00020000);
    R.f = __builtin_amdgcn_make_buffer_rsrc((void *) (A.mlF + idx * NC), 0, R.leaf ? 0 : nPos * NC * (int) sizeof(REAL), 0x00020000);
    return R;
}
// AUX: the cache policy of the loads - VFT_WS_SC1 (past the CU's L1: rows another workgroup has written) or 0 (the chain's own columns)
template <typename REAL, int AUX>
__device__ __forceinline__ REAL vft_ws_ld_real(__amdgpu_buffer_rsrc_t r, int32_t off) {
    if constexpr (sizeof(REAL) == 4) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX));
    } else {
        const vft_ws_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX);
        return __hiloint2double((int) v.y, (int) v.x);
    }
}
// the NC numbers of a vector as 16-byte loads
template <typename REAL, int NC, int AUX>
__device__ __forceinline__ void vft_ws_ld_vec(__amdgpu_buffer_rsrc_t r, int32_t off, REAL *f) {
    constexpr int PER = 16 / (int) sizeof(REAL), NLD = NC / PER;
#pragma unroll
    for (int t = 0; t < NLD; t++) {
        const vft_ws_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16 * t, 0, AUX);
        if constexpr (sizeof(REAL) == 4) {
            f[4 * t] = __uint_as_float(v.x);
            f[4 * t + 1] = __uint_as_float(v.y);
            f[4 * t + 2] = __uint_as_float(v.z);
            f[4 * t + 3] = __uint_as_float(v.w);
        } else {
            f[2 * t] = __hiloint2double((int) v.y, (int) v.x);
            f[2 * t + 1] = __hiloint2double((int) v.w, (int) v.z);
        }
    }
}
template <typename REAL, int NC>
__device__ __forceinline__ void vft_ws_st_vec(__amdgpu_buffer_rsrc_t r, int32_t p, const REAL *f) {
    constexpr int PER = 16 / (int) sizeof(REAL), NST = NC / PER;
#pragma unroll
    for (int t = 0; t < NST; t++) {
        vft_ws_u32x4 v;
        if constexpr (sizeof(REAL) == 4) {
            v.x = __float_as_uint(f[4 * t]);
            v.y = __float_as_uint(f[4 * t + 1]);
            v.z = __float_as_uint(f[4 * t + 2]);
            v.w = __float_as_uint(f[4 * t + 3]);
        } else {
            v.x = (unsigned) __double2loint(f[2 * t]);
            v.y = (unsigned) __double2hiint(f[2 * t]);
            v.z = (unsigned) __double2loint(f[2 * t + 1]);
            v.w = (unsigned) __double2hiint(f[2 * t + 1]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, r, p * (NC * (int) sizeof(REAL)) + 16 * t, 0, VFT_WS_SC1);
    }
}

// One column in two steps, so that loads can be in flight long before they are looked at (both inputs of an average, and the NEXT
// average's inputs while the current one is computed): vft_ws_request issues weight, code and (EAGER) the vector - always the same
// loads, leaf or row; vft_ws_finish interprets them.  !EAGER (the pair phase of 20-state alphabets, whose vectors are 160 bytes a
// column behind loads that cannot use the L1): the vector is fetched in vft_ws_finish once the column is known to hold one.
template <typename REAL, int NC>
struct WsRaw {
    REAL w;
    uint32_t code;
    REAL f[NC];
};
// the thread's offsets into the arrays of a row / into a leaf's byte lane: fixed for a column
struct WsOff {
    int32_t w, c, f, leaf;
};
template <typename REAL, int NC>
__device__ __forceinline__ WsOff vft_ws_off(int32_t p) {
    WsOff o;
    o.w = p * (int) sizeof(REAL);
    o.c = p;
    o.f = p * (NC * (int) sizeof(REAL));
    o.leaf = (p >> 4) * (VFT_TILE * 16) + (p & 15);
    return o;
}
template <typename REAL, int NC, int AUX, bool EAGER>
__device__ __forceinline__ void vft_ws_request(const WsRow<REAL, NC> &R, const WsOff &o, WsRaw<REAL, NC> &r) {
    r.w = vft_ws_ld_real<REAL, AUX>(R.w, o.w);
    r.code = __builtin_amdgcn_raw_buffer_load_b8(R.c, R.leaf ? o.leaf : o.c, 0, AUX);
    if constexpr (EAGER) vft_ws_ld_vec<REAL, NC, AUX>(R.f, o.f, r.f);
}
template <typename REAL, int NC, int AUX, bool EAGER>
__device__ __forceinline__ void vft_ws_finish(const WsRow<REAL, NC> &R, const WsOff &o, const WsRaw<REAL, NC> &r, Col<REAL, NC> &c) {
    const int lcode = vft_decode<NC>(r.code);
    c.code = R.leaf ? lcode : (int) r.code;
    c.w = R.leaf ? (lcode != VFT_NOCODE_ ? (REAL) 1 : (REAL) 0) : r.w;
    c.vec = c.w > 0 && c.code == VFT_NOCODE_ && !R.leaf;
    if constexpr (EAGER) {
#pragma unroll
        for (int k = 0; k < NC; k++) c.f[k] = r.f[k];   // (zeros for a leaf: its vector resource is empty)
    } else {
#pragma unroll
        for (int k = 0; k < NC; k++) c.f[k] = 0;
        if (c.vec) vft_ws_ld_vec<REAL, NC, AUX>(R.f, o.f, c.f);
    }
}
// the row's vector slot is written whatever the column holds (zeros under a code or a gap): a store inside a branch would spoil the
// wait bookkeeping like a load does, and readers only look at the slot of a vector column
template <typename REAL, int NC>
__device__ __forceinline__ void vft_ws_store(const WsRow<REAL, NC> &R, int32_t p, REAL w, int code, const REAL *f) {
    if constexpr (sizeof(REAL) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w), R.w, p * 4, 0, VFT_WS_SC1);
    } else {
        vft_ws_u32x2 v;
        v.x = (unsigned) __double2loint(w);
        v.y = (unsigned) __double2hiint(w);
        __builtin_amdgcn_raw_buffer_store_b64(v, R.w, p * 8, 0, VFT_WS_SC1);
    }
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char) code, R.c, p, 0, VFT_WS_SC1);
    vft_ws_st_vec<REAL, NC>(R.f, p, f);
}

__device__ __forceinline__ unsigned long long vft_ws_ld_sys(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long vft_ws_ld_dev(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0: wait until the six words at f have all reached seq; false when the limit passed first
__device__ __forceinline__ bool vft_ws_wait6(const unsigned long long *f, unsigned long long seq, long long limit) {
    const int lane = threadIdx.x & 63;
    const long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long v = lane < VFT_WS_NWG ? vft_ws_ld_dev(f + lane) : ~0ull;
        if (__all(v >= seq)) return true;
        if ((long long) wall_clock64() - t0 > limit) return false;
    }
}

// the column-ordered sum of n doubles in LDS (n a multiple of 32; the tail beyond the alignment holds +0.0): the adds are one
// dependent chain (~13 cycles each on gfx950, tools/sumprobe), so the reads of the sixteen after next are issued while sixteen
// are added - two fixed register blocks, each refilled right after it has been consumed
__device__ __forceinline__ double vft_ws_ordered_sum(const double *src, int32_t n) {
    double acc = 0, b0[16], b1[16];
#pragma unroll
    for (int u = 0; u < 16; u++) b0[u] = src[u];
#pragma unroll
    for (int u = 0; u < 16; u++) b1[u] = src[16 + u];
    for (int32_t p = 0;; p += 32) {
        const bool more = p + 32 < n;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; u++) acc += b0[u];
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int u = 0; u < 16; u++) b0[u] = src[p + 32 + u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; u++) acc += b1[u];
        __builtin_amdgcn_sched_barrier(0);
        if (!more) return acc;
#pragma unroll
        for (int u = 0; u < 16; u++) b1[u] = src[p + 48 + u];
    }
}

template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WS_WG_OF(NC)) void k_walk_server(Arena<REAL> A, WalkServerArgs S) {
    constexpr int WG = VFT_WS_WG_OF(NC);
    extern __shared__ __attribute__((aligned(16))) double pwLds[];   // addends: sW, sT, (nPos rounded up to 32) doubles each
    if (blockIdx.x % S.stride) return;
    const int w = (int) (blockIdx.x / S.stride);
    if (w >= VFT_WS_NWG) return;
    __shared__ uint32_t sCmd[VFT_WS_GRAN];
    __shared__ unsigned long long sPre[VFT_WS_GRAN];   // the next slot as the last wavefront saw it during this step's pair phase (dual commands arrive that early)
    __shared__ double sSum[2];
    __shared__ int sStop;
    // amino acids: the distance-matrix tables (distances, codeFreq, eigenval, eigentot) in LDS: a table read from global memory between the chain's stores is a wait for the stores
    constexpr int NT = NC == 20 ? 2 * NC * NC + 2 * NC : 1;
    __shared__ REAL sDm[NT];
    typedef const __attribute__((address_space(3))) REAL *lds_t;
    DmLds<REAL> T;
    T.dist = (lds_t) sDm;
    T.codeFreq = (lds_t) sDm + (NC == 20 ? NC * NC : 0);
    T.eigenval = (lds_t) sDm + (NC == 20 ? 2 * NC * NC : 0);
    T.eigentot = (lds_t) sDm + (NC == 20 ? 2 * NC * NC + NC : 0);
    // (the host starts the server only for 4-state alphabets without a distance matrix and 20-state ones with one)
    if (NC == 20)
        for (int t = threadIdx.x; t < NT; t += WG)
            sDm[t] = t < NC * NC ? A.dmDist[t] : t < 2 * NC * NC ? A.dmCodeFreq[t - NC * NC] : t < 2 * NC * NC + NC ? A.dmEigenval[t - 2 * NC * NC] : A.dmEigentot[t - 2 * NC * NC - NC];
    const int32_t nPos = (int32_t) A.d.nPos, nPosPad = (nPos + 31) & ~31;   // (the ordered sums add whole thirty-twos)
    double *sW = pwLds, *sT = pwLds + nPosPad;
    for (int32_t t = nPos + (int32_t) threadIdx.x; t < nPosPad; t += WG) sW[t] = sT[t] = 0.0;   // +0.0 beyond the alignment, for good
    if (threadIdx.x == 0) sStop = 0;
    if (threadIdx.x < VFT_WS_GRAN) sPre[threadIdx.x] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t nSlices = (nPos + 63) >> 6;
    const int pi = w < 3 ? 0 : w < 5 ? 1 : 2, pj = w == 0 ? 1 : (w == 1 || w == 3) ? 2 : 3;   // AB AC AD BC BD CD
#ifdef VFT_WALK_TIMING
    unsigned long long wsTick_ = wall_clock64();
#endif
    for (uint32_t seq = S.firstSeq;; seq++) {
        // ---- the next command (wave 0 polls the slot: one 8-byte granule per lane, valid when its tag is seq)
        if (wave == 0) {
            const unsigned long long *slot = S.mail + (size_t) (seq % VFT_WS_RING) * VFT_WS_GRAN;
            const long long t0 = wall_clock64();
            int end = 0;
            // A poll is a read over PCIe (~1.2 us).  A dual command is in the mailbox long before the step in front of it ends: the last
            // wavefront read this slot while that step's pairs were summed (sPre) - complete there, no poll at all.
            bool first = true;
            for (;;) {
                const unsigned long long g = first ? sPre[lane] : vft_ws_ld_sys(slot + lane);
                first = false;
                const unsigned long long ok = __ballot((uint32_t) (g >> 32) == seq);
                if (ok & 1ull) {
                    const uint32_t hdr = (uint32_t) __shfl(g, 0, 64);
                    const bool dual = (hdr >> VFT_WS_DUAL_BIT) & 1u;
                    const int cnt = (hdr & 0xFFu) == VFT_WS_CMD_STOP ? 1 : dual ? 9 + 3 * (int) (((hdr >> 8) & 0xFFu) + ((hdr >> 24) & 0xFFu)) : 5 + 3 * (int) ((hdr >> 8) & 0xFFu);
                    const unsigned long long need = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
                    if ((ok & need) == need) {
                        sCmd[lane] = (uint32_t) g;
                        if ((hdr & 0xFFu) == VFT_WS_CMD_STOP) end = 1;
                        break;
                    }
                }
                if ((long long) wall_clock64() - t0 > S.idleTicks) {
                    end = 2;
                    break;
                }
            }
            // ---- a dual command: the comparison of the command before (its six distances, from the workgroups' device-memory words),
            //      then this command becomes the plain command of the alternative taken
            if (!end && ((sCmd[0] >> VFT_WS_DUAL_BIT) & 1u)) {
                const unsigned long long *dw = S.flags + VFT_WS_FLAG_DIST + 16 * ((seq - 1u) & 1u);
                const long long t1 = wall_clock64();
                unsigned long long g = 0;
                for (;;) {   // words 2 w and 2 w + 1: the halves of workgroup w's log-corrected distance of the command before
                    g = lane < 2 * VFT_WS_NWG ? vft_ws_ld_dev(dw + lane) : ((unsigned long long) (seq - 1u) << 32);
                    if (__all((uint32_t) (g >> 32) == seq - 1u)) break;
                    if ((long long) wall_clock64() - t1 > S.flagTicks) {
                        end = 3;
                        break;
                    }
                }
                if (!end) {
                    double c[6];
#pragma unroll
                    for (int t = 0; t < 6; t++) {
                        const uint32_t lo = (uint32_t) __shfl(g, 2 * t, 64), hi = (uint32_t) __shfl(g, 2 * t + 1, 64);
                        c[t] = __hiloint2double((int) hi, (int) lo);
                    }
                    // criteria AB+CD, AC+BD, AD+BC over the distances AB AC AD BC BD CD (meCollect)
                    const double c1 = c[1] + c[4], c2 = c[2] + c[3];
                    const int alt = c1 < c2 ? 0 : 1;
                    const uint32_t hdr = sCmd[0];
                    const int n0 = (int) ((hdr >> 8) & 0xFFu), n1 = (int) ((hdr >> 24) & 0xFFu);
                    const bool skip = (hdr >> (19 + alt)) & 1u;
                    const int nAlt = skip ? 0 : (alt ? n1 : n0);
                    // alternative 1's granules move down to where a plain command has them (alternative 0's are there already)
                    const int src = alt ? 5 + 3 * n0 + lane : 1 + lane;
                    const uint32_t v = src < VFT_WS_GRAN ? sCmd[src] : 0u;
                    if (lane < 4 + 3 * nAlt) sCmd[1 + lane] = v;   // (one wavefront: every lane has read before any lane writes)
                    if (lane == 0) {
                        sCmd[0] = (hdr & 0xFFu) | ((uint32_t) nAlt << 8) | (skip ? 0u : 1u << 16) | (hdr & (1u << 17)) | (hdr & (1u << 21));
                        if (w == 0)
                            __hip_atomic_store(S.res + (size_t) (seq % VFT_WS_RING) * VFT_WS_RESG + VFT_WS_RES_CHOICE,
                                               ((unsigned long long) seq << 32) | 1ull | ((unsigned long long) alt << 1) | ((unsigned long long) (skip ? 1 : 0) << 2),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            // the rows the chain is about to overwrite may still be read by a slower workgroup's pair phase of the command before
            // (only when the host sent this command without having seen that one's six answers: bit 17 clear)
            if (!end && !((sCmd[0] >> 17) & 1u) && seq != S.firstSeq && !vft_ws_wait6(S.flags + 8, (unsigned long long) seq - 1ull, S.flagTicks)) end = 3;
            if (end && lane == 0) sStop = end;
        }
        __syncthreads();
        if (sStop) break;
        VFT_WS_TICK(0);
        const uint32_t hdr = sCmd[0];
        const int n = (int) ((hdr >> 8) & 0xFFu);
        const bool hasDist = (hdr >> 16) & 1u;
        // ---- the averages: this wavefront's slices, a column per lane through the whole chain
        if (n > 0) {
            for (int32_t s = w + VFT_WS_NWG * wave; s < nSlices; s += VFT_WS_NWG * (WG / 64)) {
                const int32_t p = s * 64 + lane;
                const WsOff off = vft_ws_off<REAL, NC>(p);
                int32_t prevOut = -1;
                Col<REAL, NC> prev;
                prev.w = 0;
                prev.code = VFT_NOCODE_;
                prev.vec = false;
#pragma unroll
                for (int k = 0; k < NC; k++) prev.f[k] = 0;
                // The inputs of average k + 1 are requested BEFORE average k is computed - always, branch-free (the last average
                // requests its own inputs again; an input that is k's output row is requested too and never looked at: that one
                // is handed over in registers): rows written by earlier averages of the command were stored by this very thread
                // earlier in program order, so the only row a request can miss is k's output.  An average then costs its
                // arithmetic, not arithmetic + a memory round.  Plain (L1) loads: these are the thread's own columns.
                int32_t o = __builtin_amdgcn_readfirstlane((int32_t) sCmd[5]), a = __builtin_amdgcn_readfirstlane((int32_t) sCmd[6]),
                        b = __builtin_amdgcn_readfirstlane((int32_t) sCmd[7]);
                WsRow<REAL, NC> Ra = vft_ws_row<REAL, NC>(A, a), Rb = vft_ws_row<REAL, NC>(A, b);
                WsRaw<REAL, NC> na, nb;
                vft_ws_request<REAL, NC, 0, true>(Ra, off, na);
                vft_ws_request<REAL, NC, 0, true>(Rb, off, nb);
                for (int k = 0; k < n; k++) {
                    const WsRaw<REAL, NC> ra = na, rb = nb;
                    const bool leafA = Ra.leaf, leafB = Rb.leaf;
                    const int k1 = k + 1 < n ? k + 1 : k;
                    const int32_t o1 = __builtin_amdgcn_readfirstlane((int32_t) sCmd[5 + 3 * k1]), a1 = __builtin_amdgcn_readfirstlane((int32_t) sCmd[6 + 3 * k1]),
                                  b1 = __builtin_amdgcn_readfirstlane((int32_t) sCmd[7 + 3 * k1]);
                    Ra = vft_ws_row<REAL, NC>(A, a1);
                    Rb = vft_ws_row<REAL, NC>(A, b1);
                    vft_ws_request<REAL, NC, 0, true>(Ra, off, na);
                    vft_ws_request<REAL, NC, 0, true>(Rb, off, nb);
                    Col<REAL, NC> c1, c2;
                    {
                        WsRow<REAL, NC> Rc;   // (only the leaf flag matters to an eager finish)
                        Rc.leaf = leafA;
                        vft_ws_finish<REAL, NC, 0, true>(Rc, off, ra, c1);
                        Rc.leaf = leafB;
                        vft_ws_finish<REAL, NC, 0, true>(Rc, off, rb, c2);
                    }
                    if (a == prevOut) c1 = prev;
                    if (b == prevOut) c2 = prev;
                    REAL wo, f[NC];
                    int co;
                    if constexpr (NC == 4) vft_average_col_nt_select<REAL>(c1, c2, S.tol, wo, co, f);
                    else vft_average_col<REAL, NC, DmLds<REAL>>(A, c1, c2, 0.5, S.tol, wo, co, f, T);
                    vft_ws_store<REAL, NC>(vft_ws_row<REAL, NC>(A, o), p, wo, co, f);
                    prevOut = o;
                    prev.w = wo;
                    prev.code = co;
                    prev.vec = wo > 0 && co == VFT_NOCODE_;
#pragma unroll
                    for (int k2 = 0; k2 < NC; k2++) prev.f[k2] = f[k2];
                    o = o1;
                    a = a1;
                    b = b1;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every row byte of this wavefront has left the chip's caches
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(S.flags + w, (unsigned long long) seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        VFT_WS_TICK(1);
        unsigned long long *out = S.res + (size_t) (seq % VFT_WS_RING) * VFT_WS_RESG + 2 * w;
        if (!hasDist) {   // averages only: acknowledge (the host's flow control and its final wait)
            if (threadIdx.x == 0) {
                __hip_atomic_store(S.flags + 8 + w, (unsigned long long) seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out, (unsigned long long) seq << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            continue;
        }
        // ---- the pair of this workgroup, once every workgroup's slices of the chain are out
        if (wave == 0 && !vft_ws_wait6(S.flags, (unsigned long long) seq, S.flagTicks) && lane == 0) sStop = 3;
        __syncthreads();
        if (sStop) break;
        VFT_WS_TICK(2);
        {
            const int32_t i = (int32_t) sCmd[1 + pi], j = (int32_t) sCmd[1 + pj];
            const WsRow<REAL, NC> Ri = vft_ws_row<REAL, NC>(A, __builtin_amdgcn_readfirstlane(i)), Rj = vft_ws_row<REAL, NC>(A, __builtin_amdgcn_readfirstlane(j));
            const bool leaves = Ri.leaf && Rj.leaf;
            // 4-state columns: two per thread and trip, the loads of both issued before the first is consumed (as in vft_pair_block);
            // 20-state columns one at a time (two pairs of 20-state profiles do not fit the registers)
            constexpr int PER = NC == 4 ? 2 : 1;
            for (int32_t p = threadIdx.x; p < nPos; p += PER * WG) {
                const int32_t pb = p + WG;
                const bool hasB = PER == 2 && pb < nPos;
                constexpr bool EAGER = NC == 4;
                const WsOff oa = vft_ws_off<REAL, NC>(p), ob = vft_ws_off<REAL, NC>(pb);
                WsRaw<REAL, NC> r1, r2, r3, r4;
                vft_ws_request<REAL, NC, VFT_WS_SC1, EAGER>(Ri, oa, r1);
                vft_ws_request<REAL, NC, VFT_WS_SC1, EAGER>(Rj, oa, r2);
                if (hasB) {
                    vft_ws_request<REAL, NC, VFT_WS_SC1, EAGER>(Ri, ob, r3);
                    vft_ws_request<REAL, NC, VFT_WS_SC1, EAGER>(Rj, ob, r4);
                }
                Col<REAL, NC> a1, a2;
                vft_ws_finish<REAL, NC, VFT_WS_SC1, EAGER>(Ri, oa, r1, a1);
                vft_ws_finish<REAL, NC, VFT_WS_SC1, EAGER>(Rj, oa, r2, a2);
                if constexpr (NC == 20) vft_pair_addends<REAL, NC, DmLds<REAL>>(A, leaves, false, p, a1, a2, sW, sT, T);
                else vft_pair_addends<REAL, NC>(A, leaves, false, p, a1, a2, sW, sT);
                if (hasB) {
                    Col<REAL, NC> b1, b2;
                    vft_ws_finish<REAL, NC, VFT_WS_SC1, EAGER>(Ri, ob, r3, b1);
                    vft_ws_finish<REAL, NC, VFT_WS_SC1, EAGER>(Rj, ob, r4, b2);
                    if constexpr (NC == 20) vft_pair_addends<REAL, NC, DmLds<REAL>>(A, leaves, false, pb, b1, b2, sW, sT, T);
                    else vft_pair_addends<REAL, NC>(A, leaves, false, pb, b1, b2, sW, sT);
                }
            }
        }
        __syncthreads();
        VFT_WS_TICK(3);
        if (threadIdx.x == 0) __hip_atomic_store(S.flags + 8 + w, (unsigned long long) seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the last wavefront, idle while two lanes add the columns up, reads the next slot of the mailbox: a PCIe read, ~1.2 us, under
        //  the ~1.6 us of the sums - a dual command is there by now and the next step starts without a poll)
        if (wave == WG / 64 - 1) sPre[lane] = vft_ws_ld_sys(S.mail + (size_t) ((seq + 1u) % VFT_WS_RING) * VFT_WS_GRAN + lane);
        if (threadIdx.x < 2) sSum[threadIdx.x] = vft_ws_ordered_sum(threadIdx.x == 0 ? sT : sW, nPosPad);   // `top`, `denom`: each in column order
        __syncthreads();
        VFT_WS_TICK(4);
        if (threadIdx.x == 0) {
            const double top = sSum[0], denom = sSum[1];
            const REAL d = (REAL) (denom > 0 ? top / denom : 1.0);   // profileDist / seqDist (NJ.tcc:1183-1189, :1621-1623)
            if constexpr (sizeof(REAL) == 4) {
                __hip_atomic_store(out, ((unsigned long long) seq << 32) | (unsigned long long) __float_as_uint(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                __hip_atomic_store(out, ((unsigned long long) seq << 32) | (unsigned long long) (unsigned) __double2loint(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(out + 1, ((unsigned long long) seq << 32) | (unsigned long long) (unsigned) __double2hiint(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (threadIdx.x == 0) {
            // ... and, log-corrected (logCorrect, NJ.tcc:322-330; host/MLLengths.h: Jukes-Cantor or scoredist-like by bit 21 of the command,
            // capped at 3; glibc's log bit for bit, vft_glibc_log.h), to the other workgroups: a dual command that follows compares the six
            // numbers on the device.  After the answer has gone out - the host does not wait for this - and one logarithm per workgroup,
            // side by side, instead of six in front of the next step's chain.
            const double top = sSum[0], denom = sSum[1];
            const double dd = (double) (REAL) (denom > 0 ? top / denom : 1.0);
            const double maxscore = 3.0;
            double x;
            if (!((hdr >> 21) & 1u)) x = dd < 0.74 ? -0.75 * vft_glibc_log(1.0 - dd * 4.0 / 3.0) : maxscore;
            else x = dd < 0.99 ? -1.3 * vft_glibc_log(1.0 - dd) : maxscore;
            x = x < maxscore ? x : maxscore;
            unsigned long long *dw = S.flags + VFT_WS_FLAG_DIST + 16 * (seq & 1u) + 2 * w;
            __hip_atomic_store(dw, ((unsigned long long) seq << 32) | (unsigned long long) (unsigned) __double2loint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dw + 1, ((unsigned long long) seq << 32) | (unsigned long long) (unsigned) __double2hiint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        VFT_WS_TICK(5);
#ifdef VFT_WALK_TIMING
        if (w == 0 && threadIdx.x == 0) {
            atomicAdd(&S.ticks[6], 1ull);
            atomicAdd(&S.ticks[7], (unsigned long long) n);
        }
#endif
    }
    if (threadIdx.x == 0) __hip_atomic_store(S.status + w, (unsigned long long) sStop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define VFT_WALK_SERVER_INSTANCES(PFX)                                                          \
    PFX template __global__ void k_walk_server<float, 4>(Arena<float>, WalkServerArgs);         \
    PFX template __global__ void k_walk_server<float, 20>(Arena<float>, WalkServerArgs);        \
    PFX template __global__ void k_walk_server<double, 4>(Arena<double>, WalkServerArgs);       \
    PFX template __global__ void k_walk_server<double, 20>(Arena<double>, WalkServerArgs);

#endif
