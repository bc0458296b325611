// C-ABI implementation (include/vft_hip.h): context, arena allocation, launches.
// Built with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>
#include <csignal>
#include <unistd.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <utility>
#include <vector>
#include <chrono>

#include "../../include/vft_hip.h"
#define VFT_PAIR_STAGE_CAP 2048   // = the longest list the workgroup-per-pair kernels take
#define VFT_SMALL_BYTES_ (256u << 10)   // = VFT_SMALL_BYTES below: what goes through the host-mapped ring
#include "vft_kernels_ml.h"

VFT_ML_HEAVY_INSTANCES(extern)   // compiled in vft_ml_kernels_*.hip
#include "vft_kernels_ml_long.h"
VFT_ML_LONG_INSTANCES(extern)    // compiled in vft_ml_kernels_long.hip
#include "vft_kernels_nj.h"
#include "vft_kernels_aa.h"
#include "vft_kernels_profile.h"
#include "vft_kernels_tophits.h"
#include "vft_kernels_njengine.h"
#include "vft_kernels_walk.h"

VFT_WALK_SERVER_INSTANCES(extern)   // compiled in vft_walk_kernels.hip

struct vft_ctx {
    vft_config cfg;
    VftDims d;
    hipStream_t stream = nullptr, ownStream = nullptr;
    char err[512] = {0};
    size_t rs = 4;   // sizeof(real)
    int64_t maxnode = 0, shardLo = 0, shardHi = 0;
    bool refreshAll = false;   // vft_set_shard_mode: lazy refreshes ignore the shard
    // vft_join_fused: joined nodes whose tile streams have not been rebuilt yet (their rows and stash slots are valid)
    std::vector<int64_t> staleIds;             // pair lists: the distinct stale nodes of the current list
    std::vector<uint32_t> staleMark;
    std::vector<int32_t> staleIdx;
    unsigned int *refDone = nullptr;           // k_pairs_refresh_fused: completion tags of the refresh workgroups
    uint32_t staleEpoch = 0;
    std::vector<int64_t> pend;
    char *pendBase = nullptr;        // [stash of 64 nodes | meta | commit scratch], commit_plan(c, 64)
    int64_t *pendIdsDev = nullptr;   // device copy of pend[], written by the kernel
    int64_t nProfTiles = 0, nLeafTiles = 0;
    bool leavesUp = false;

    // arena
    uint4 *leafT = nullptr, *profC = nullptr;
    void *profW = nullptr, *profF = nullptr;
    ColMask *colMask = nullptr;
    ColOff *colOff = nullptr;
    std::vector<int32_t> hParent;          // host copy of parent[], to recognise "all active nodes, ascending" lists
    unsigned long long *tileMask = nullptr;
    // test / tool hooks, set through vft_debug_option only (no environment variable changes which kernels run)
    bool noFusedRefresh = false;   // VFT_DEBUG_NO_FUSED_REFRESH: pair lists with refreshes as two launches
    int pairWG = 0;                // VFT_DEBUG_PAIR_THREADS: threads per pair of the short-list kernels
    bool genericOutProfile = false;   // VFT_DEBUG_GENERIC_OUTPROFILE: the one-thread-per-column out-profile kernel
    int fusedLimit = 2048;         // workgroups of k_pairs_refresh_fused that are resident at once (set by vft_create)
    bool faultNoFlag = false;      // VFT_DEBUG_FAULT_NO_FLAG: the next wait for a completion flag waits for one that never comes
    double waitLimitS = 120.0;     // how long a wait for a completion flag may last while the stream is busy
    bool wideGlue = false;         // test hook: the 1 024-thread instance of k_nj_glue_scan at any size
    bool jcExact = true;           // vft_set_jc_exact: Jukes-Cantor likelihoods bit for bit the reference's (glibc exp, ordered totals) - the default
    bool mlLong = false;           // test hook: the line searches through the workspace kernels (vft_kernels_ml_long.h) at any length
    char *mlLongWs = nullptr;      // their workspaces (one per workgroup of a launch)
    size_t mlLongWsBytes = 0;
    size_t shLdsSet = 0;           // the largest dynamic LDS k_sh_support has been configured for
    unsigned int *opHist = nullptr;            // k_leaf_hist: per-(column, code) counts of the active leaves
    int32_t *parent = nullptr, *nOutActive = nullptr;
    void *diameter = nullptr, *selfweight = nullptr, *selfdist = nullptr, *outDist = nullptr;
    void *outW = nullptr, *outF = nullptr, *outCD = nullptr;
    // query staging (node query and out-profile-as-query)
    void *qW[2] = {nullptr, nullptr}, *qF[2] = {nullptr, nullptr};
    uint8_t *qC[2] = {nullptr, nullptr};
    uint4 *qEnc[2] = {nullptr, nullptr};
    double2 *qTab[2] = {nullptr, nullptr};
    void *qPT[2] = {nullptr, nullptr};   // amino acids: per-(column, target code) piece table of the query (vft_kernels_aa.h)
    size_t aaLds = 0;                    // dynamic LDS of k_sweep_aa, 0 = alignment too long for it (generic kernels)
    // sweep outputs
    void *swDist = nullptr, *swWeight = nullptr, *swCrit = nullptr;
    void *partMin = nullptr, *partMax = nullptr;
    int nPart = 0;
    // select scratch
    SelectState *sel = nullptr;
    unsigned int *slices = nullptr;
    uint64_t *candKey = nullptr;
    int32_t *candId = nullptr;
    char *dRes = nullptr;    // SelectHeader followed by the k hit records
    char *hRes = nullptr;    // pinned, device-mapped host mirror of dRes, written by the last workgroup of k_select_rank (zero-copy)
    char *hResDev = nullptr; // device address of hRes
    // One set of sweep-result + selection buffers per seed of a batch (vft_sweep_batch); slot 0 aliases the members
    // above, further slots are allocated on first use.
    struct SweepSlotHost {
        void *swDist = nullptr, *swWeight = nullptr, *swCrit = nullptr, *partMin = nullptr, *partMax = nullptr;
        SelectState *sel = nullptr;
        unsigned int *slices = nullptr;
        uint64_t *candKey = nullptr;
        int32_t *candId = nullptr;
        char *dRes = nullptr, *hRes = nullptr, *hResDev = nullptr;
        int nPart = 0;
        // the seed's staged query (nt without a distance matrix; slot 0 uses the context's qW[0] ...)
        void *qW = nullptr, *qF = nullptr;
        uint8_t *qC = nullptr;
        uint4 *qEnc = nullptr;
        double2 *qTab = nullptr;
    };
    std::vector<SweepSlotHost> slots;
    void *mqBuf = nullptr;        // interleaved queries of the profile-seed groups of a batch (QuerySlot::mq): VFT_MQ_GROUPS groups
    size_t mqGroupBytes = 0;
    char *dMerge = nullptr, *hMerge = nullptr, *hMergeDev = nullptr;   // result blocks of vft_merge_hits_batch
    size_t mergeBytes = 0;
    int32_t hitsCap = 0;
    // models
    void *dm[4] = {nullptr, nullptr, nullptr, nullptr};
    bool hasDm = false;
    void *tm[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool hasTm = false;
    void *rates = nullptr;
    int32_t *ratecat = nullptr;
    int32_t nRates = 0;
    double minLen = 5e-4, minRel = 2.5e-4, fpostTol = 1e-10;
    bool rowMode = false;                      // vft_set_profile_rows: averages write dense rows as well
    bool allRows = false;                      // every internal node is flagged as a row (no tile-path write since the
                                               // switch): the chain calls then need not raise flags
    uint8_t *mlIs = nullptr, *mlC = nullptr;   // dense ML rows (vft_layout.h), allocated by the first ML-phase write
    void *mlW = nullptr, *mlF = nullptr;
    void *blen = nullptr;              // branchlength[] (numeric_t) for the ML length optimiser
    unsigned int *mlEvals = nullptr;   // likelihood evaluations made by k_ml_node_lengths
    // generic device scratch (index lists, staging)
    void *scratch = nullptr;
    size_t scratchBytes = 0;
    // host-mapped command/response ring: small id lists go to the device and small results come back through
    // zero-copy pinned memory instead of hipMemcpy (whose fixed latency dominates a join's worth of tiny calls)
    char *hIO = nullptr, *dIO = nullptr;
    size_t ioCap = 8u << 20, ioHead = 0;
    unsigned long long *hFlag = nullptr, *dFlag = nullptr, signalSeq = 0;
    unsigned int *doneCtr = nullptr;   // completion counter of k_pairs_fused
    void *pairIn = nullptr;            // device copy of the ids of a short pair list (k_copy16x2)
    bool noPairStaging = false;        // VFT_DEBUG_NO_PAIR_STAGING
    void *pairStage = nullptr;         // device staging of the results of lists of up to VFT_PAIR_STAGE_CAP pairs
    // Upper bound of nOutActive over the nodes of the shard (host bookkeeping; VFT_STAMP_UNKNOWN = no bound): lets
    // vft_sweep skip its lazy out-distance pre-pass when provably no target can be stale.
    int64_t maxStamp = (int64_t) 1 << 62;
    int pwWaves = 4;                   // items (waves) per workgroup of the wave-per-item kernels: 4, or fewer when the
                                       // LDS staging of a long alignment would not fit (raise_pair_kernel_lds)
    // host-mapped mirrors of outDist / nOutActive, written by the kernels that refresh them
    void *hOutDist = nullptr, *dOutDistM = nullptr;
    int32_t *hNOut = nullptr, *dNOutM = nullptr;
    // top-hit lists on the device (vft_kernels_tophits.h)
    void *thHits = nullptr, *thStD = nullptr, *thStC = nullptr;
    int32_t *thLen = nullptr, *thStJ = nullptr;
    unsigned int *thMark = nullptr, *thDone = nullptr;
    int32_t *thSorted = nullptr;
    int32_t thM = 0, thCap = 0;
    int64_t thLists = 0;
    unsigned int thTag = 0;
    size_t thLds = 0;                  // dynamic LDS of k_th_best / k_th_join
    size_t thRefreshLds = 0;           // the largest dynamic LDS k_th_refresh has been configured for
    size_t pbLdsSet = 0;               // ... and k_pairs_block_tiled
    // the walk server (vft_kernels_walk.h): six resident workgroups that take the steps of a refinement walk from a mailbox
    struct WalkServerHost {
        bool up = false;
        hipStream_t stream = nullptr;
        unsigned long long *hMail = nullptr, *dMail = nullptr;   // the mailbox as the CPU writes it / as the kernel polls it
        bool mailOnDevice = false;                                // ... in device memory behind the PCIe aperture (sfence after writing)
        bool wantDeviceMail = false;                              // VFT_DEBUG_WALK_DEVICE_MAILBOX
        unsigned long long *hRes = nullptr, *dRes = nullptr;      // answers, host-mapped
        unsigned long long *hStatus = nullptr, *dStatus = nullptr;
        unsigned long long *dFlags = nullptr;
        uint32_t seq = 0;       // the last sequence number handed out
        uint32_t acked = 0;     // every command up to this one has been answered by all six workgroups
        int stride = 8;         // VFT_DEBUG_WALK_SERVER_STRIDE: 8 = the six workgroups on one XCD, 1 = on six
        bool disabled = false;  // VFT_DEBUG_NO_WALK_SERVER
        bool allocated = false; // every buffer of the server exists (set behind the last allocation of the first start)
        bool scoredist = false; // logCorrect's flavour of the walk (bit 21 of every command: the workgroups log-correct their distance for a dual command that may follow)
    } ws;
    // the join loop on the device (vft_kernels_njengine.h)
    void *njState = nullptr, *njVisD = nullptr;
    int32_t *njVisJ = nullptr, *njTop = nullptr, *njAge = nullptr;
    NjJoinRec *njLogDev = nullptr, *njLogHost = nullptr, *njLogHostDev = nullptr;
    long long *njStatusHost = nullptr, *njStatusDev = nullptr;
    vft_nj_engine_config njCfg{};
    size_t njScanLds = 0, njTailLds = 0;
    int njP = 0, njTailThreads = 0;
    unsigned int *njClaim = nullptr, njClaimTag = 0;   // speculative double walks: one writer per refreshed node
    int32_t *njLogNode = nullptr, *njLogStamp = nullptr;
    void *njLogOut = nullptr;
    int32_t *njSlotI = nullptr, *njCandI = nullptr;
    void *njSlotR = nullptr, *njCandR = nullptr;
    int njTopPad = 0, njCapPad = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;
    size_t kevUsed = 0;
    int64_t kevSweeps = 0;       // sweeps beyond one per timed launch (a pass of k_sweep_nt_leafq_multi covers several)
    bool noMultiSweep = false;   // VFT_DEBUG_NO_MULTI_SWEEP
    int multiMax = 0;            // seeds per shared pass: 0 = by the shard's size (vft_sweep_batch), 2 / 4 = fixed (VFT_DEBUG_NO_MULTI_SWEEP values 2 / 4)
    bool timeKernels = false;
};

static int fail(vft_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, call)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return fail(c, VFT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
    } while (0)

#define LAUNCHCHK(c)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = hipGetLastError();                                                                   \
        if (e_ != hipSuccess) return fail(c, VFT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

template <typename T>
static hipError_t dalloc(T **p, size_t n) {
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void **) p, n * sizeof(T));
    return e;
}
static hipError_t dallocb(void **p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 1); }

static int ensure_scratch(vft_ctx *c, size_t bytes) {
    if (bytes <= c->scratchBytes) return VFT_OK;
    if (c->scratch) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->scratch));
        c->scratch = nullptr;
        c->scratchBytes = 0;
    }
    size_t want = bytes + bytes / 2 + 4096;
    HIPCHK(c, hipMalloc(&c->scratch, want));
    c->scratchBytes = want;
    return VFT_OK;
}

template <typename REAL>
static Arena<REAL> arena(const vft_ctx *c) {
    Arena<REAL> A;
    A.d = c->d;
    A.leafT = c->leafT;
    A.profW = (REAL *) c->profW;
    A.profF = (REAL *) c->profF;
    A.profC = c->profC;
    A.colMask = c->colMask;
    A.colOff = c->colOff;
    A.parent = c->parent;
    A.diameter = (REAL *) c->diameter;
    A.selfweight = (REAL *) c->selfweight;
    A.selfdist = (REAL *) c->selfdist;
    A.outDist = (REAL *) c->outDist;
    A.nOutActive = c->nOutActive;
    A.mOutDist = (REAL *) c->dOutDistM;
    A.mNOut = c->dNOutM;
    A.outW = (REAL *) c->outW;
    A.outF = (REAL *) c->outF;
    A.outCD = c->hasDm ? (REAL *) c->outCD : nullptr;
    A.dmDist = c->hasDm ? (const REAL *) c->dm[0] : nullptr;
    A.dmCodeFreq = c->hasDm ? (const REAL *) c->dm[1] : nullptr;
    A.dmEigenval = c->hasDm ? (const REAL *) c->dm[2] : nullptr;
    A.dmEigentot = c->hasDm ? (const REAL *) c->dm[3] : nullptr;
    A.tmStat = c->hasTm ? (const REAL *) c->tm[0] : nullptr;
    A.tmStatInv = c->hasTm ? (const REAL *) c->tm[1] : nullptr;
    A.tmEigenval = c->hasTm ? (const REAL *) c->tm[2] : nullptr;
    A.tmCodeFreq = c->hasTm ? (const REAL *) c->tm[3] : nullptr;
    A.tmEigenInv = c->hasTm ? (const REAL *) c->tm[4] : nullptr;
    A.tmEigenInvT = c->hasTm ? (const REAL *) c->tm[5] : nullptr;
    A.rates = (const REAL *) c->rates;
    A.ratecat = c->ratecat;
    A.nRates = c->nRates;
    A.jcExact = c->jcExact ? 1 : 0;
    A.mlIs = c->mlIs;
    A.mlW = (REAL *) c->mlW;
    A.mlC = c->mlC;
    A.mlF = (REAL *) c->mlF;
    return A;
}

template <typename REAL>
static QueryBuf<REAL> qbuf(const vft_ctx *c, int which) {
    QueryBuf<REAL> q;
    q.w = (REAL *) c->qW[which];
    q.code = c->qC[which];
    q.f = (REAL *) c->qF[which];
    q.enc = c->qEnc[which];
    q.tab = c->qTab[which];
    return q;
}

// the staged query of a batch slot
template <typename REAL>
static QueryBuf<REAL> qbuf_slot(const vft_ctx *c, int slot) {
    if (slot == 0) return qbuf<REAL>(c, 0);
    const vft_ctx::SweepSlotHost &h = c->slots[(size_t) slot];
    QueryBuf<REAL> q;
    q.w = (REAL *) h.qW;
    q.code = h.qC;
    q.f = (REAL *) h.qF;
    q.enc = h.qEnc;
    q.tab = h.qTab;
    return q;
}

template <typename REAL>
static SweepOut<REAL> sweepout(const vft_ctx *c, int slot = 0) {
    const vft_ctx::SweepSlotHost &h = c->slots[(size_t) slot];
    SweepOut<REAL> o;
    o.dist = (REAL *) h.swDist;
    o.weight = (REAL *) h.swWeight;
    o.crit = (REAL *) h.swCrit;
    o.partMin = (REAL *) h.partMin;
    o.partMax = (REAL *) h.partMax;
    return o;
}

// precision x alphabet dispatch: BODY sees REAL and NC
#define VFT_DISPATCH(c, ...)                                                           \
    do {                                                                               \
        if ((c)->cfg.precision == 4) {                                                 \
            typedef float REAL;                                                        \
            if ((c)->cfg.n_codes == 4) { constexpr int NC = 4; __VA_ARGS__; }          \
            else { constexpr int NC = 20; __VA_ARGS__; }                               \
        } else {                                                                       \
            typedef double REAL;                                                       \
            if ((c)->cfg.n_codes == 4) { constexpr int NC = 4; __VA_ARGS__; }          \
            else { constexpr int NC = 20; __VA_ARGS__; }                               \
        }                                                                              \
    } while (0)

// statement-safe launch wrapper (hipLaunchKernelGGL is a do/while macro)
static vft_ctx *g_walkServerOwner = nullptr;   // the context whose walk server is resident (one per process)
static unsigned long long *g_wsTicks = nullptr;   // phase clock ticks of the walk server's workgroup 0 (written by -DVFT_WALK_TIMING builds only)
static int walk_server_retire(vft_ctx *c);
// tools (VFT_LAUNCH_TRACE=1, with AMD_SERIALIZE_KERNEL=3 so that a launch has finished before the next one is recorded): the kernel
// launched last, printed when the process is aborted - a GPU memory fault ends the process without saying whose it was
static const bool g_launchTrace = getenv("VFT_LAUNCH_TRACE") != nullptr;
static const void *volatile g_lastKernel = nullptr;
static volatile unsigned long long g_launchCount = 0;
static void launch_trace_abort(int) {
    const char *name = g_lastKernel ? hipKernelNameRefByPtr((const void *) g_lastKernel, nullptr) : "(none)";
    fprintf(stderr, "[launch trace] aborted; last kernel launched: %s (launch %llu); walk server %s\n", name ? name : "?", (unsigned long long) g_launchCount,
            g_walkServerOwner ? "resident" : "not running");
    _exit(134);
}
template <typename... KArgs, typename... Args>
static inline void launch(void (*k)(KArgs...), dim3 g, dim3 b, size_t shm, hipStream_t s, Args... args) {
    if (g_launchTrace) {
        g_lastKernel = (const void *) k;
        g_launchCount = g_launchCount + 1;
    }
    // a context whose walk server is resident owns its rows through the server: anything else launched on that context's stream
    // first retires the server (the refinement code stops it itself; this is the safety net)
    if (g_walkServerOwner && g_walkServerOwner->stream == s) walk_server_retire(g_walkServerOwner);
    hipLaunchKernelGGL(k, g, b, shm, s, static_cast<KArgs>(args)...);
}

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned) ((a + b - 1) / b); }

// Completion signal in mapped host memory: waiting for the stream by spinning on a word a trailing 1-thread kernel
// writes costs ~10 us; hipStreamSynchronize costs ~80 us here, and a join makes several round trips.
__global__ void k_signal(unsigned long long *flag, unsigned long long seq) {
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int wait_flag(vft_ctx *c, unsigned long long seq, int first = 0, int count = 1);
static int wait_stream(vft_ctx *c) {
    const unsigned long long seq = ++c->signalSeq;
    launch(k_signal, dim3(1), dim3(1), 0, c->stream, c->dFlag, seq);
    LAUNCHCHK(c);
    return wait_flag(c, seq);
}
// spins on the host-mapped flag until the stream has published `seq`
// The spin is bounded: a flag that a kernel never raises (a bug in one of the hand-written completion protocols, a faulted
// kernel) comes back as VFT_ERR_TIMEOUT instead of hanging the caller - at once when the stream has drained without the flag
// moving, after waitLimitS seconds (vft_debug_option(VFT_DEBUG_WAIT_LIMIT_MS)) when it is still busy.
// (first, count: the words of the flag block that have to reach seq)
static int wait_flag(vft_ctx *c, unsigned long long seq, int first, int count) {
    volatile unsigned long long *f = c->hFlag + first;
    if (c->faultNoFlag) {   // test hook: wait for a value nobody will ever publish
        seq += 1ull << 40;
        c->faultNoFlag = false;
    }
    auto raised = [&]() {
        for (int k = 0; k < count; k++)
            if (__atomic_load_n(f + k, __ATOMIC_ACQUIRE) < seq) return false;
        return true;
    };
    std::chrono::steady_clock::time_point t0;
    for (long spins = 0;; spins++) {
        if (raised()) return VFT_OK;
        if (spins == 200000) t0 = std::chrono::steady_clock::now();   // (~ 0.1 s of spinning: start looking at the stream)
        if (spins >= 200000 && (spins & 0xFFFF) == 0) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {   // everything enqueued has run: the flag is as high as it will ever get
                if (raised()) return VFT_OK;
                return fail(c, VFT_ERR_TIMEOUT, "the stream has drained but the completion flag stands at %llu, not %llu",
                            (unsigned long long) *f, seq);
            }
            if (e != hipErrorNotReady) return fail(c, VFT_ERR_HIP, "stream error while waiting: %s", hipGetErrorString(e));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->waitLimitS)
                return fail(c, VFT_ERR_TIMEOUT, "no completion flag after %.0f s (the stream is still busy)", c->waitLimitS);
        }
        __builtin_ia32_pause();
    }
}

// Bump allocation in the mapped ring.  Wrapping waits for the stream: everything that was reading older slots is done.
static int io_alloc(vft_ctx *c, size_t bytes, char **host, char **dev) {
    bytes = (bytes + 255) & ~(size_t) 255;
    if (bytes > c->ioCap) return fail(c, VFT_ERR_INVALID, "request of %zu bytes exceeds the mapped I/O ring", bytes);
    if (c->ioHead + bytes > c->ioCap) {
        if (int r = wait_stream(c)) return r;
        c->ioHead = 0;
    }
    *host = c->hIO + c->ioHead;
    *dev = c->dIO + c->ioHead;
    c->ioHead += bytes;
    return VFT_OK;
}


static int raise_pair_kernel_lds(vft_ctx *c);   // defined next to the kernels it configures

// ---------------------------------------------------------------------------------------------- life cycle
extern "C" int vft_create(vft_ctx **out, const vft_config *cfg) {
    if (!out || !cfg) return VFT_ERR_INVALID;
    *out = nullptr;
    if (g_launchTrace) signal(SIGABRT, launch_trace_abort);
    vft_ctx *c = new (std::nothrow) vft_ctx();
    if (!c) return VFT_ERR_INVALID;
    c->cfg = *cfg;
    auto bail = [&](int code) {
        // keep the context alive so the caller can read the error text
        *out = c;
        return code;
    };
    if ((cfg->precision != 4 && cfg->precision != 8) || (cfg->n_codes != 4 && cfg->n_codes != 20) || cfg->n_seqs < 1 ||
        cfg->n_pos < 1 || cfg->max_nodes < cfg->n_seqs || cfg->max_nodes >= (1ll << 31))
        return bail(fail(c, VFT_ERR_INVALID, "vft_create: bad configuration"));
    // a tile's vector stream is addressed with 32-bit byte offsets (k_sweep_nt): 64 nodes x nPos x nCodes reals < 4 GiB
    if ((double) (cfg->n_pos + VFT_CHUNK) * VFT_TILE * cfg->n_codes * cfg->precision >= 4294967296.0)
        return bail(fail(c, VFT_ERR_INVALID, "vft_create: alignment too long (%lld columns)", (long long) cfg->n_pos));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return bail(fail(c, VFT_ERR_HIP, "vft_create: no HIP device (this backend has no CPU fallback)"));
    if (cfg->device < 0 || cfg->device >= ndev) return bail(fail(c, VFT_ERR_INVALID, "vft_create: bad device ordinal"));
    if (hipSetDevice(cfg->device) != hipSuccess) return bail(fail(c, VFT_ERR_HIP, "hipSetDevice failed"));
    c->rs = (size_t) cfg->precision;
    VftDims &d = c->d;
    d.nSeqs = cfg->n_seqs;
    d.nPos = cfg->n_pos;
    d.maxNodes = cfg->max_nodes;
    d.nCodes = cfg->n_codes;
    d.nChunk = (int32_t) ((cfg->n_pos + VFT_CHUNK - 1) / VFT_CHUNK);
    d.nPosPad = (int64_t) d.nChunk * VFT_CHUNK;
    d.firstProfTile = cfg->n_seqs / VFT_TILE;
    d.nTiles = (cfg->max_nodes + VFT_TILE - 1) / VFT_TILE;
    c->nLeafTiles = (cfg->n_seqs + VFT_TILE - 1) / VFT_TILE;
    c->nProfTiles = d.nTiles - d.firstProfTile;
    c->maxnode = cfg->n_seqs;
    c->shardLo = 0;
    c->shardHi = cfg->max_nodes;
    const size_t rs = c->rs;
    const int64_t N = d.nTiles * VFT_TILE;   // padded node count
    const int64_t nPosPad = (int64_t) d.nChunk * VFT_CHUNK;
#define CR(call)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return bail(fail(c, VFT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_))); \
    } while (0)
    CR(hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking));
    c->stream = c->ownStream;
    CR(dalloc(&c->leafT, (size_t) c->nLeafTiles * d.nChunk * VFT_TILE));
    CR(dallocb(&c->profW, (size_t) c->nProfTiles * d.nPosPad * VFT_TILE * rs));
    CR(dallocb(&c->profF, (size_t) c->nProfTiles * d.nPosPad * VFT_TILE * d.nCodes * rs));
    CR(dalloc(&c->profC, (size_t) c->nProfTiles * d.nChunk * VFT_TILE));
    CR(hipMemset(c->profW, 0, (size_t) c->nProfTiles * d.nPosPad * VFT_TILE * rs));
    CR(dalloc(&c->colMask, (size_t) c->nProfTiles * d.nPosPad));
    CR(hipMemset(c->colMask, 0, (size_t) c->nProfTiles * d.nPosPad * sizeof(ColMask)));
    CR(dalloc(&c->colOff, (size_t) c->nProfTiles * d.nPosPad));
    CR(hipMemset(c->colOff, 0, (size_t) c->nProfTiles * d.nPosPad * sizeof(ColOff)));
    c->hParent.assign((size_t) N, 0);
    for (int64_t i = 0; i < cfg->max_nodes; i++) c->hParent[(size_t) i] = -1;
    CR(dalloc(&c->tileMask, (size_t) d.nTiles));
    CR(hipMemset(c->profC, 0x7F, (size_t) c->nProfTiles * d.nChunk * VFT_TILE * sizeof(uint4)));
    CR(dalloc(&c->parent, (size_t) N));
    CR(dalloc(&c->nOutActive, (size_t) N));
    // padding ids beyond max_nodes are permanently inactive
    {
        std::vector<int32_t> par((size_t) N, 0);
        for (int64_t i = 0; i < cfg->max_nodes; i++) par[(size_t) i] = -1;
        CR(hipMemcpy(c->parent, par.data(), (size_t) N * 4, hipMemcpyHostToDevice));
    }
    CR(hipMemset(c->nOutActive, 0, (size_t) N * 4));
    void **reals[] = {&c->diameter, &c->selfweight, &c->selfdist, &c->outDist, &c->swDist, &c->swWeight, &c->swCrit};
    for (void **p : reals) {
        CR(dallocb(p, (size_t) N * rs));
        CR(hipMemset(*p, 0, (size_t) N * rs));
    }
    CR(dallocb(&c->outW, (size_t) d.nPos * rs));
    CR(dallocb(&c->outF, (size_t) d.nPos * d.nCodes * rs));
    CR(dallocb(&c->outCD, (size_t) d.nPos * d.nCodes * rs));
    for (int q = 0; q < 2; q++) {
        CR(dallocb(&c->qW[q], (size_t) nPosPad * rs));
        CR(dallocb(&c->qF[q], (size_t) nPosPad * d.nCodes * rs));
        CR(dalloc(&c->qC[q], (size_t) nPosPad));
        CR(dalloc(&c->qEnc[q], (size_t) d.nChunk));
        CR(dalloc(&c->qTab[q], (size_t) nPosPad * 5));
        CR(dallocb(&c->qPT[q], d.nCodes == 20 ? (size_t) nPosPad * 20 * rs : 1));
    }
    if (d.nCodes == 20) {
        const size_t need = (((size_t) nPosPad * 20 * rs + 15) & ~(size_t) 15) + (size_t) nPosPad * 8 + (size_t) 2 * VFT_CHUNK * VFT_TILE * 16;
        if (need <= (160u << 10) - 1024) {
            c->aaLds = need;
            if (need > (48u << 10)) {
                if (rs == 4) {
                    CR(hipFuncSetAttribute((const void *) k_sweep_aa<float, MODE_CRIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) need));
                    CR(hipFuncSetAttribute((const void *) k_sweep_aa<float, MODE_OUTDIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) need));
                } else {
                    CR(hipFuncSetAttribute((const void *) k_sweep_aa<double, MODE_CRIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) need));
                    CR(hipFuncSetAttribute((const void *) k_sweep_aa<double, MODE_OUTDIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) need));
                }
            }
        }
    }
    c->nPart = (int) cdiv(N, VFT_WG);
    CR(dallocb(&c->partMin, (size_t) c->nPart * 8));
    CR(dallocb(&c->partMax, (size_t) c->nPart * 8));
    CR(dalloc(&c->sel, 1));
    CR(hipMemset(c->sel, 0, sizeof(SelectState)));
    CR(dalloc(&c->slices, (size_t) VFT_NBINS));
    CR(hipMemset(c->slices, 0, (size_t) VFT_NBINS * 4));   // (the selection's histogram is zero between selections: k_select_rank's last workgroup leaves it so)
    CR(dalloc(&c->candKey, (size_t) VFT_CAND_CAP));
    CR(hipMemset(c->candKey, 0, (size_t) VFT_CAND_CAP * 8));
    CR(dalloc(&c->candId, (size_t) VFT_CAND_CAP));
    CR(hipMemset(c->candId, 0, (size_t) VFT_CAND_CAP * 4));
    c->hitsCap = VFT_CAND_CAP;
    CR(dallocb((void **) &c->dRes, sizeof(SelectHeader) + (size_t) c->hitsCap * sizeof(vft_hit_f64)));
    CR(hipHostMalloc((void **) &c->hRes, sizeof(SelectHeader) + (size_t) c->hitsCap * sizeof(vft_hit_f64), hipHostMallocMapped));
    CR(hipHostGetDevicePointer((void **) &c->hResDev, c->hRes, 0));
    memset(c->hRes, 0, sizeof(SelectHeader));
    {
        vft_ctx::SweepSlotHost s0;
        s0.swDist = c->swDist;
        s0.swWeight = c->swWeight;
        s0.swCrit = c->swCrit;
        s0.partMin = c->partMin;
        s0.partMax = c->partMax;
        s0.sel = c->sel;
        s0.slices = c->slices;
        s0.candKey = c->candKey;
        s0.candId = c->candId;
        s0.dRes = c->dRes;
        s0.hRes = c->hRes;
        s0.hResDev = c->hResDev;
        c->slots.assign(1, s0);
    }
    // [0]: the stream's completion flag
    CR(hipHostMalloc((void **) &c->hFlag, 512, hipHostMallocMapped));
    CR(hipHostGetDevicePointer((void **) &c->dFlag, c->hFlag, 0));
    memset(c->hFlag, 0, 512);
    CR(dalloc(&c->doneCtr, 65));   // [0]: top level / single-level users, [1..64]: slots of vft_publish_staged
    CR(hipMalloc(&c->pairStage, 3 * VFT_PAIR_STAGE_CAP * sizeof(double)));   // staging of short pair lists' results (vft_publish_staged)
    CR(hipMalloc(&c->pairIn, VFT_SMALL_BYTES_));   // device copy of a short list's inputs
    CR(hipMemset(c->doneCtr, 0, 65 * 4));
    CR(hipHostMalloc((void **) &c->hIO, c->ioCap, hipHostMallocMapped));
    CR(hipHostGetDevicePointer((void **) &c->dIO, c->hIO, 0));
    CR(hipHostMalloc(&c->hOutDist, (size_t) N * rs, hipHostMallocMapped));
    CR(hipHostGetDevicePointer(&c->dOutDistM, c->hOutDist, 0));
    CR(hipHostMalloc((void **) &c->hNOut, (size_t) N * 4, hipHostMallocMapped));
    CR(hipHostGetDevicePointer((void **) &c->dNOutM, c->hNOut, 0));
    memset(c->hOutDist, 0, (size_t) N * rs);
    memset(c->hNOut, 0, (size_t) N * 4);
    for (int i = 0; i < 4; i++) CR(dallocb(&c->dm[i], (size_t) 21 * 20 * 8));
    for (int i = 0; i < 6; i++) CR(dallocb(&c->tm[i], (size_t) 21 * 20 * 8));
    CR(dallocb(&c->rates, (size_t) VFT_MAXRATES * 8));
    CR(dalloc(&c->ratecat, (size_t) d.nPos));
    {
        // one rate category with rate 1.0 (Rates(1, nPos) in the NJ constructor, NJ.tcc:226)
        double one64 = 1.0;
        float one32 = 1.0f;
        CR(hipMemcpy(c->rates, rs == 4 ? (void *) &one32 : (void *) &one64, rs, hipMemcpyHostToDevice));
        CR(hipMemset(c->ratecat, 0, (size_t) d.nPos * 4));
        c->nRates = 1;
    }
    if (cfg->precision == 8) {
        c->minLen = 5e-9;
        c->minRel = 2.5e-9;
        c->fpostTol = 1e-20;
    }
    CR(hipEventCreate(&c->ev0));
    CR(hipEventCreate(&c->ev1));
    if (int r = raise_pair_kernel_lds(c)) return bail(r);
    // The hipMemsets above run on the NULL stream and need not have finished when they return (device memory), and the context's own
    // stream is a non-blocking one: nothing orders them before its first kernel.  Wait here, once.
    CR(hipDeviceSynchronize());
#undef CR
    *out = c;
    return VFT_OK;
}

extern "C" int vft_destroy(vft_ctx *c) {
    if (!c) return VFT_OK;
    (void) walk_server_retire(c);
    if (c->ws.stream) {
        hipStreamDestroy(c->ws.stream);
        if (c->ws.mailOnDevice) hipFree(c->ws.dMail);
        else hipHostFree(c->ws.hMail);
        hipHostFree(c->ws.hRes);
        hipFree(c->ws.dFlags);
    }
    if (c->blen) hipFree(c->blen);
    if (c->mlLongWs) hipFree(c->mlLongWs);
    if (c->opHist) hipFree(c->opHist);
    if (c->refDone) hipFree(c->refDone);
    if (c->pendBase) hipFree(c->pendBase);
    if (c->pendIdsDev) hipFree(c->pendIdsDev);
    if (c->mlIs) hipFree(c->mlIs);
    if (c->mlC) hipFree(c->mlC);
    if (c->mlW) hipFree(c->mlW);
    if (c->mlF) hipFree(c->mlF);
    if (c->mlEvals) hipFree(c->mlEvals);
    if (c->dMerge) hipFree(c->dMerge);
    if (c->hMerge) hipHostFree(c->hMerge);
    if (c->mqBuf) hipFree(c->mqBuf);
    for (size_t i = 1; i < c->slots.size(); i++) {   // slot 0 aliases members freed below
        vft_ctx::SweepSlotHost &h = c->slots[i];
        void *dev[] = {h.swDist, h.swWeight, h.swCrit, h.partMin, h.partMax, h.sel, h.slices, h.candKey, h.candId, h.dRes, h.qW, h.qF, h.qC, h.qEnc, h.qTab};
        for (void *p: dev)
            if (p) hipFree(p);
        if (h.hRes) hipHostFree(h.hRes);
    }
    if (c->ownStream) hipStreamSynchronize(c->ownStream);
    void *ptrs[] = {c->tileMask, c->colMask, c->colOff, c->leafT, c->profC, c->profW, c->profF, c->parent, c->nOutActive, c->diameter, c->selfweight,
                    c->selfdist, c->outDist, c->outW, c->outF, c->outCD, c->qW[0], c->qW[1], c->qF[0], c->qF[1],
                    c->qC[0], c->qC[1], c->qEnc[0], c->qEnc[1], c->qTab[0], c->qTab[1], c->qPT[0], c->qPT[1], c->swDist, c->swWeight, c->swCrit,
                    c->partMin, c->partMax, c->sel, c->slices, c->candKey, c->candId, c->dRes, c->dm[0],
                    c->dm[1], c->dm[2], c->dm[3], c->tm[0], c->tm[1], c->tm[2], c->tm[3], c->tm[4], c->tm[5],
                    c->rates, c->ratecat, c->scratch};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (c->hRes) hipHostFree(c->hRes);
    if (c->hIO) hipHostFree(c->hIO);
    if (c->hFlag) hipHostFree(c->hFlag);
    if (c->doneCtr) hipFree(c->doneCtr);
    for (void *p : {c->thHits, c->thStD, c->thStC, (void *) c->thLen, (void *) c->thStJ, (void *) c->thMark, (void *) c->thDone, (void *) c->thSorted,
                    c->njState, c->njVisD, (void *) c->njVisJ, (void *) c->njTop, (void *) c->njAge, (void *) c->njLogDev, (void *) c->njClaim,
                    (void *) c->njLogNode, (void *) c->njLogStamp, c->njLogOut, (void *) c->njSlotI, c->njSlotR, (void *) c->njCandI, c->njCandR})
        if (p) hipFree(p);
    if (c->njLogHost) hipHostFree(c->njLogHost);
    if (c->njStatusHost) hipHostFree(c->njStatusHost);
    if (c->pairStage) hipFree(c->pairStage);
    if (c->pairIn) hipFree(c->pairIn);
    if (c->hOutDist) hipHostFree(c->hOutDist);
    if (c->hNOut) hipHostFree(c->hNOut);
    for (hipEvent_t e : c->kev) hipEventDestroy(e);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ownStream) hipStreamDestroy(c->ownStream);
    delete c;
    return VFT_OK;
}

extern "C" const char *vft_last_error(const vft_ctx *c) { return c ? c->err : "null context"; }

extern "C" int vft_set_stream(vft_ctx *c, void *s) {
    if (!c) return VFT_ERR_INVALID;
    // work queued on the stream being left (ring slots, the completion flag) must not race with the new one
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t) s : c->ownStream;
    return VFT_OK;
}

extern "C" int vft_synchronize(vft_ctx *c) {
    if (!c) return VFT_ERR_INVALID;
    return wait_stream(c);
}

extern "C" int vft_device_malloc(vft_ctx *c, int64_t bytes, void **p) {
    if (!c || !p || bytes < 0) return VFT_ERR_INVALID;
    HIPCHK(c, hipMalloc(p, bytes ? (size_t) bytes : 1));
    return VFT_OK;
}

extern "C" int vft_device_free(vft_ctx *c, void *p) {
    if (!c) return VFT_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(p));
    return VFT_OK;
}

extern "C" int vft_device_upload(vft_ctx *c, void *dst, const void *src, int64_t bytes) {
    if (!c || !dst || !src || bytes < 0) return VFT_ERR_INVALID;
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t) bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- inputs
extern "C" int vft_upload_leaves(vft_ctx *c, const uint8_t *codes) {
    if (!c || !codes) return VFT_ERR_INVALID;
    const VftDims &d = c->d;
    const size_t n = (size_t) c->nLeafTiles * d.nChunk * VFT_TILE * 16;
    std::vector<uint8_t> host(n, d.nCodes == 4 ? 0 : VFT_NOCODE_);
    for (int64_t i = 0; i < d.nSeqs; i++) {
        const int64_t tile = i >> 6;
        const int lane = (int) (i & 63);
        const uint8_t *row = codes + (size_t) i * d.nPos;
        for (int64_t p = 0; p < d.nPos; p++) {
            const uint8_t code = row[p];
            if (code != VFT_NOCODE && code >= d.nCodes)
                return fail(c, VFT_ERR_INVALID, "vft_upload_leaves: code %d at seq %lld pos %lld", (int) code,
                            (long long) i, (long long) p);
            host[(size_t) vft_leaf_idx(d, tile, (int) (p >> 4), lane) * 16 + (p & 15)] = vft_encode(code, d.nCodes);
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->leafT, host.data(), n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->leavesUp = true;
    return VFT_OK;
}

static int upload_tables(vft_ctx *c, void **dst, const void *const *src, const size_t *counts, int n) {
    for (int i = 0; i < n; i++) {
        if (!src[i]) return fail(c, VFT_ERR_INVALID, "model table %d is NULL", i);
        HIPCHK(c, hipMemcpyAsync(dst[i], src[i], counts[i] * c->rs, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_set_distance_matrix(vft_ctx *c, const void *distances, const void *codefreq, const void *eigenval,
                                       const void *eigentot) {
    if (!c) return VFT_ERR_INVALID;
    if (!distances) {
        c->hasDm = false;
        return VFT_OK;
    }
    const size_t nc = (size_t) c->d.nCodes;
    const void *src[4] = {distances, codefreq, eigenval, eigentot};
    const size_t cnt[4] = {nc * nc, nc * nc, nc, nc};
    int r = upload_tables(c, c->dm, src, cnt, 4);
    if (r) return r;
    c->hasDm = true;
    return VFT_OK;
}

extern "C" int vft_set_transition_matrix(vft_ctx *c, const void *stat, const void *statinv, const void *eigenval,
                                         const void *codefreq, const void *eigeninv, const void *eigeninvT) {
    if (!c) return VFT_ERR_INVALID;
    if (!stat) {
        if (c->d.nCodes != 4) return fail(c, VFT_ERR_INVALID, "Jukes-Cantor needs a nucleotide alphabet");
        c->hasTm = false;
        return VFT_OK;
    }
    const size_t nc = (size_t) c->d.nCodes;
    const void *src[6] = {stat, statinv, eigenval, codefreq, eigeninv, eigeninvT};
    const size_t cnt[6] = {nc, nc, nc, (nc + 1) * nc, nc * nc, nc * nc};
    int r = upload_tables(c, c->tm, src, cnt, 6);
    if (r) return r;
    c->hasTm = true;
    return VFT_OK;
}

extern "C" int vft_set_rates(vft_ctx *c, const void *rates, int32_t nRates, const int64_t *ratecat) {
    if (!c || !rates || !ratecat || nRates < 1 || nRates > VFT_MAXRATES)
        return fail(c, VFT_ERR_INVALID, "vft_set_rates: bad arguments");
    std::vector<int32_t> rc((size_t) c->d.nPos);
    for (int64_t p = 0; p < c->d.nPos; p++) {
        if (ratecat[p] < 0 || ratecat[p] >= nRates) return fail(c, VFT_ERR_INVALID, "vft_set_rates: category out of range");
        rc[(size_t) p] = (int32_t) ratecat[p];
    }
    HIPCHK(c, hipMemcpyAsync(c->rates, rates, (size_t) nRates * c->rs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ratecat, rc.data(), rc.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nRates = nRates;
    return VFT_OK;
}

extern "C" int vft_set_jc_exact(vft_ctx *c, int32_t on) {
    if (!c) return VFT_ERR_INVALID;
    c->jcExact = on != 0;
    return VFT_OK;
}

extern "C" int vft_set_ml_limits(vft_ctx *c, double minLen, double minRel, double fpostTol) {
    if (!c) return VFT_ERR_INVALID;
    c->minLen = minLen;
    c->minRel = minRel;
    c->fpostTol = fpostTol;
    return VFT_OK;
}

static int range_ok(vft_ctx *c, int64_t first, int64_t count) {
    if (first < 0 || count < 0 || first + count > c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "node range out of bounds");
    return VFT_OK;
}

#define VFT_SMALL_BYTES (256u << 10)

// dst[first .. first+count) = src (host), stream-ordered, no host synchronisation for small ranges
template <typename T>
static int store_range(vft_ctx *c, T *dst, T *mirror, const T *src, int64_t first, int64_t count) {
    const size_t bytes = (size_t) count * sizeof(T);
    if (bytes <= VFT_SMALL_BYTES) {
        char *h, *d;
        if (int r = io_alloc(c, bytes, &h, &d)) return r;
        memcpy(h, src, bytes);
        launch((k_store_range<T>), dim3(cdiv(count, 256)), dim3(256), 0, c->stream, dst, mirror, (const T *) d, first, count);
        LAUNCHCHK(c);
        return VFT_OK;
    }
    HIPCHK(c, hipMemcpyAsync(dst + first, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_set_parents(vft_ctx *c, int64_t first, int64_t count, const int64_t *parent) {
    if (!c || !parent) return VFT_ERR_INVALID;
    if (int r = range_ok(c, first, count)) return r;
    std::vector<int32_t> p((size_t) count);
    for (int64_t i = 0; i < count; i++) {
        p[(size_t) i] = parent[i] < 0 ? -1 : (int32_t) parent[i];
        c->hParent[(size_t) (first + i)] = p[(size_t) i];
    }
    return store_range<int32_t>(c, c->parent, nullptr, p.data(), first, count);
}

extern "C" int vft_set_node_scalars(vft_ctx *c, int64_t first, int64_t count, const void *diameter,
                                    const void *selfweight, const void *selfdist) {
    if (!c) return VFT_ERR_INVALID;
    if (int r = range_ok(c, first, count)) return r;
    const void *src[3] = {diameter, selfweight, selfdist};
    void *dst[3] = {c->diameter, c->selfweight, c->selfdist};
    for (int k = 0; k < 3; k++) {
        if (!src[k]) continue;
        int r;
        if (c->rs == 4) r = store_range<float>(c, (float *) dst[k], nullptr, (const float *) src[k], first, count);
        else r = store_range<double>(c, (double *) dst[k], nullptr, (const double *) src[k], first, count);
        if (r) return r;
    }
    return VFT_OK;
}

extern "C" int vft_get_node_scalars(vft_ctx *c, int64_t first, int64_t count, void *diameter, void *selfweight,
                                    void *selfdist) {
    if (!c) return VFT_ERR_INVALID;
    if (int r = range_ok(c, first, count)) return r;
    const size_t rs = c->rs;
    if (diameter) HIPCHK(c, hipMemcpyAsync(diameter, (char *) c->diameter + first * rs, count * rs, hipMemcpyDeviceToHost, c->stream));
    if (selfweight) HIPCHK(c, hipMemcpyAsync(selfweight, (char *) c->selfweight + first * rs, count * rs, hipMemcpyDeviceToHost, c->stream));
    if (selfdist) HIPCHK(c, hipMemcpyAsync(selfdist, (char *) c->selfdist + first * rs, count * rs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

static int32_t clamp_i32(int64_t v) { return v > 0x7FFFFFFF ? 0x7FFFFFFF : (v < 0 ? 0 : (int32_t) v); }

extern "C" int vft_set_out_distances(vft_ctx *c, int64_t first, int64_t count, const void *outDist,
                                     const int64_t *nOutActive) {
    if (!c) return VFT_ERR_INVALID;
    if (int r = range_ok(c, first, count)) return r;
    // bulk host-side sets must not race with refresh kernels still in flight that write the mirrors; a single
    // fresh node (the join loop's newnode) cannot be touched by an earlier kernel
    if (count > 1) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (outDist) {
        int r;
        if (c->rs == 4) r = store_range<float>(c, (float *) c->outDist, nullptr, (const float *) outDist, first, count);
        else r = store_range<double>(c, (double *) c->outDist, nullptr, (const double *) outDist, first, count);
        if (r) return r;
        memcpy((char *) c->hOutDist + first * c->rs, outDist, (size_t) count * c->rs);   // host side of the mirror
    }
    if (nOutActive) {
        std::vector<int32_t> n32((size_t) count);
        for (int64_t i = 0; i < count; i++) {
            n32[(size_t) i] = clamp_i32(nOutActive[i]);
            if ((int64_t) n32[(size_t) i] > c->maxStamp) c->maxStamp = n32[(size_t) i];
        }
        if (int r = store_range<int32_t>(c, c->nOutActive, nullptr, n32.data(), first, count)) return r;
        memcpy(c->hNOut + first, n32.data(), (size_t) count * 4);
    }
    return VFT_OK;
}

extern "C" int vft_out_distance_mirror(vft_ctx *c, const void **outDist, const int32_t **nOutActive) {
    if (!c || !outDist || !nOutActive) return VFT_ERR_INVALID;
    *outDist = c->hOutDist;
    *nOutActive = c->hNOut;
    return VFT_OK;
}

extern "C" int vft_get_out_distances(vft_ctx *c, int64_t first, int64_t count, void *outDist, int64_t *nOutActive) {
    if (!c) return VFT_ERR_INVALID;
    if (int r = range_ok(c, first, count)) return r;
    if (outDist) HIPCHK(c, hipMemcpyAsync(outDist, (char *) c->outDist + first * c->rs, count * c->rs, hipMemcpyDeviceToHost, c->stream));
    std::vector<int32_t> n32((size_t) count);
    if (nOutActive) HIPCHK(c, hipMemcpyAsync(n32.data(), c->nOutActive + first, (size_t) count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nOutActive)
        for (int64_t i = 0; i < count; i++) nOutActive[i] = n32[(size_t) i];
    return VFT_OK;
}

extern "C" int vft_set_max_node(vft_ctx *c, int64_t maxnode) {
    if (!c || maxnode < c->d.nSeqs || maxnode > c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_set_max_node: out of range");
    c->maxnode = maxnode;
    return VFT_OK;
}

extern "C" int vft_join_nodes(vft_ctx *c, int64_t i, int64_t j, int64_t newnode, double diameter, int64_t staleStamp) {
    if (!c) return VFT_ERR_INVALID;
    if (i < 0 || j < 0 || i == j || i >= c->maxnode || j >= c->maxnode || newnode < c->d.nSeqs || newnode >= c->d.maxNodes)
        return fail(c, VFT_ERR_INVALID, "vft_join_nodes: bad ids (%lld, %lld -> %lld)", (long long) i, (long long) j, (long long) newnode);
    if (newnode >= c->maxnode) c->maxnode = newnode + 1;
    c->hParent[(size_t) i] = c->hParent[(size_t) j] = (int32_t) newnode;
    const int32_t stamp = clamp_i32(staleStamp);
    if ((int64_t) stamp > c->maxStamp) c->maxStamp = stamp;
    // host side of the mirrors: no earlier kernel can touch a node that did not exist
    if (c->rs == 4) ((float *) c->hOutDist)[newnode] = 0.f;
    else ((double *) c->hOutDist)[newnode] = 0.0;
    c->hNOut[newnode] = stamp;
    if (c->rs == 4) launch((k_join_nodes<float>), dim3(1), dim3(1), 0, c->stream, arena<float>(c), i, j, newnode, (float) diameter, stamp);
    else launch((k_join_nodes<double>), dim3(1), dim3(1), 0, c->stream, arena<double>(c), i, j, newnode, diameter, stamp);
    LAUNCHCHK(c);
    return VFT_OK;
}

static int ensure_ml_rows(vft_ctx *c);
static int flush_pending(vft_ctx *c);
struct CommitPlan;

extern "C" int vft_set_shard(vft_ctx *c, int64_t lo, int64_t hi) {
    if (!c || lo < 0 || hi < lo || hi > c->d.maxNodes || (lo % VFT_TILE) != 0)
        return fail(c, VFT_ERR_INVALID, "vft_set_shard: need 0 <= lo <= hi <= max_nodes and lo %% 64 == 0");
    c->shardLo = lo;
    c->shardHi = hi;
    if (!c->refreshAll) c->maxStamp = (int64_t) 1 << 62;   // the bound was about the previous range
    return VFT_OK;
}

extern "C" int vft_set_shard_mode(vft_ctx *c, int32_t refreshAll) {
    if (!c) return VFT_ERR_INVALID;
    c->refreshAll = refreshAll != 0;
    c->maxStamp = (int64_t) 1 << 62;
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- profiles
// Writing nodes = producing kernel (codes -> profC, columns -> stash) + k_tile_commit per touched tile.
struct CommitPlan {
    int64_t chunk;      // nodes per producing launch
    size_t stashB;      // stash bytes (chunk nodes)
    size_t metaB;       // order[] + segFirst[] when they do not fit the mapped ring
    size_t stride;      // commit scratch per tile
    size_t totalB;      // stashB + metaB + VFT_COMMIT_SEGS * stride
};
#define VFT_COMMIT_SEGS 256
static CommitPlan commit_plan(const vft_ctx *c, int64_t n) {
    CommitPlan p;
    const size_t perNode = (size_t) c->d.nPos * (size_t) (c->d.nCodes + 1) * c->rs;
    int64_t chunk = (int64_t) ((256u << 20) / perNode);
    chunk = chunk < 64 ? 64 : chunk > 16384 ? 16384 : chunk;
    p.chunk = n < chunk ? (n > 0 ? n : 1) : chunk;
    p.stashB = (((size_t) p.chunk * perNode) + 255) & ~(size_t) 255;
    p.metaB = (((size_t) (2 * p.chunk + 2) * 4) + 255) & ~(size_t) 255;
    p.stride = vft_commit_scratch_bytes(c->d, c->rs);
    const int64_t segs = p.chunk < VFT_COMMIT_SEGS ? p.chunk : VFT_COMMIT_SEGS;
    p.totalB = p.stashB + p.metaB + (size_t) segs * p.stride;
    return p;
}

// Dense ML rows of the internal nodes (vft_layout.h): allocated by the first ML-phase write.
static int ensure_ml_rows(vft_ctx *c) {
    if (c->mlIs) return VFT_OK;
    const size_t nodes = (size_t) (c->d.maxNodes - c->d.nSeqs), cols = nodes * (size_t) c->d.nPos;
    HIPCHK(c, hipMalloc((void **) &c->mlW, cols * c->rs));
    HIPCHK(c, hipMalloc((void **) &c->mlF, cols * (size_t) c->d.nCodes * c->rs));
    HIPCHK(c, hipMalloc((void **) &c->mlC, cols));
    HIPCHK(c, hipMalloc((void **) &c->mlIs, nodes));
    HIPCHK(c, hipMemsetAsync(c->mlIs, 0, nodes, c->stream));
    return VFT_OK;
}

// a node rewritten through the tile streams no longer has a dense ML row
__global__ void k_clear_ml_rows(uint8_t *mlIs, const int64_t *nodes, int64_t n, int64_t nSeqs) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) mlIs[nodes[k] - nSeqs] = 0;
}

// hNodes/dNodes: the cnt node ids of this launch (host copy / device copy, batch order); base: device memory laid
// out as [stash | meta | commit scratch] according to `plan`.
static int commit_nodes(vft_ctx *c, const CommitPlan &plan, const int64_t *hNodes, const int64_t *dNodes, int64_t cnt,
                        char *base, bool keepRows = false) {
    std::vector<std::pair<int64_t, int32_t>> tk((size_t) cnt);
    for (int64_t k = 0; k < cnt; k++) tk[(size_t) k] = std::make_pair(hNodes[k] >> 6, (int32_t) k);
    std::stable_sort(tk.begin(), tk.end(),
                     [](const std::pair<int64_t, int32_t> &x, const std::pair<int64_t, int32_t> &y) { return x.first < y.first; });
    std::vector<int32_t> meta((size_t) cnt);
    std::vector<int32_t> segFirst;
    for (int64_t i = 0; i < cnt; i++) {
        meta[(size_t) i] = tk[(size_t) i].second;
        if (i == 0 || tk[(size_t) i].first != tk[(size_t) i - 1].first) segFirst.push_back((int32_t) i);
    }
    segFirst.push_back((int32_t) cnt);
    const int64_t nSeg = (int64_t) segFirst.size() - 1;
    meta.insert(meta.end(), segFirst.begin(), segFirst.end());
    const size_t bytes = meta.size() * 4;
    char *dMeta;
    bool viaScratch = false;
    if (bytes <= VFT_SMALL_BYTES) {
        char *h;
        if (int r = io_alloc(c, bytes, &h, &dMeta)) return r;
        memcpy(h, meta.data(), bytes);
    } else {
        dMeta = base + plan.stashB;
        HIPCHK(c, hipMemcpyAsync(dMeta, meta.data(), bytes, hipMemcpyHostToDevice, c->stream));
        viaScratch = true;
    }
    const int32_t *dOrder = (const int32_t *) dMeta, *dSeg = dOrder + cnt;
    char *cs = base + plan.stashB + plan.metaB;
    if (c->mlIs && !keepRows) {
        launch(k_clear_ml_rows, dim3(cdiv(cnt, 256)), dim3(256), 0, c->stream, c->mlIs, dNodes, cnt, c->d.nSeqs);
        c->allRows = false;
    }
    for (int64_t g0 = 0; g0 < nSeg; g0 += VFT_COMMIT_SEGS) {
        const int64_t g = nSeg - g0 < VFT_COMMIT_SEGS ? nSeg - g0 : VFT_COMMIT_SEGS;
        VFT_DISPATCH(c, (launch((k_tile_commit<REAL, NC>), dim3((unsigned) g), dim3(VFT_COMMIT_WG), 0, c->stream,
                                arena<REAL>(c), dNodes, dOrder, dSeg + g0, (const REAL *) base, cs, plan.stride)));
        LAUNCHCHK(c);
    }
    if (viaScratch) HIPCHK(c, hipStreamSynchronize(c->stream));   // meta[] is a local
    return VFT_OK;
}

#define VFT_PEND_MAX 64
// rebuilds the tile streams of the nodes joined by vft_join_fused since the last rebuild (their stash slots are intact)
static int flush_pending(vft_ctx *c) {
    if (c->pend.empty()) return VFT_OK;
    CommitPlan plan = commit_plan(c, VFT_PEND_MAX);
    const int64_t cnt = (int64_t) c->pend.size();
    // the rows k_join_fused wrote stay valid (NJ-phase profiles never change): pair lists keep reading them - one
    // coalesced stage per column instead of the tile streams' dependent mask -> offset -> stream gathers
    int r = commit_nodes(c, plan, c->pend.data(), c->pendIdsDev, cnt, c->pendBase, true);
    c->pend.clear();
    return r;
}

static int internal_ok(vft_ctx *c, int64_t node) {
    if (node < c->d.nSeqs || node >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "node %lld is not an internal node", (long long) node);
    return VFT_OK;
}

extern "C" int vft_profile_upload(vft_ctx *c, int64_t node, const void *w, const uint8_t *codes, const void *f) {
    if (!c || !w || !codes || !f) return VFT_ERR_INVALID;
    if (int r = internal_ok(c, node)) return r;
    const VftDims &d = c->d;
    const size_t rs = c->rs, wB = d.nPos * rs, fB = d.nPos * d.nCodes * rs, cB = (size_t) d.nPos;
    const size_t inB = (wB + fB + cB + 255) & ~(size_t) 255;
    const CommitPlan plan = commit_plan(c, 1);
    if (int r = ensure_scratch(c, inB + plan.totalB + 256)) return r;
    char *s = (char *) c->scratch;
    char *base = s + inB;
    HIPCHK(c, hipMemcpyAsync(s, w, wB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + wB, f, fB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + wB + fB, codes, cB, hipMemcpyHostToDevice, c->stream));
    char *hId, *dId;
    if (int r = io_alloc(c, 8, &hId, &dId)) return r;
    memcpy(hId, &node, 8);
    VFT_DISPATCH(c, (launch((k_profile_scatter<REAL, NC>), dim3(cdiv(d.nPos, 256)), dim3(256), 0, c->stream,
                                        arena<REAL>(c), node, (const REAL *) s, (const uint8_t *) (s + wB + fB),
                                        (const REAL *) (s + wB), (REAL *) base)));
    LAUNCHCHK(c);
    if (int r = commit_nodes(c, plan, &node, (const int64_t *) dId, 1, base)) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_profile_download(vft_ctx *c, int64_t node, void *w, uint8_t *codes, void *f) {
    if (!c || !w || !codes || !f) return VFT_ERR_INVALID;
    if (node < 0 || node >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "bad node");
    const VftDims &d = c->d;
    const size_t rs = c->rs, wB = d.nPos * rs, fB = d.nPos * d.nCodes * rs, cB = (size_t) d.nPos;
    if (int r = ensure_scratch(c, wB + fB + cB + 64)) return r;
    char *s = (char *) c->scratch;
    VFT_DISPATCH(c, (launch((k_profile_gather<REAL, NC>), dim3(cdiv(d.nPos, 256)), dim3(256), 0, c->stream,
                                        arena<REAL>(c), node, (REAL *) s, (uint8_t *) (s + wB + fB), (REAL *) (s + wB))));
    LAUNCHCHK(c);
    HIPCHK(c, hipMemcpyAsync(w, s, wB, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(f, s + wB, fB, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(codes, s + wB + fB, cB, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

template <typename REAL, int NC>
__global__ void k_nvectors(Arena<REAL> A, int64_t first, int64_t count, int64_t *out) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int64_t v = first + t;
    int64_t n = 0;
    if (v >= A.d.nSeqs) {
        for (int64_t p = 0; p < A.d.nPos; p++) {
            Col<REAL, NC> c;
            vft_load_col<REAL, NC>(A, v, p, c);
            n += c.vec ? 1 : 0;
        }
    }
    out[t] = n;
}

extern "C" int vft_profile_nvectors(vft_ctx *c, int64_t first, int64_t count, int64_t *nvec) {
    if (!c || !nvec) return VFT_ERR_INVALID;
    if (first < 0 || count < 0 || first + count > c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "node range out of bounds");
    if (count == 0) return VFT_OK;
    if (int r = flush_pending(c)) return r;
    if (int r = ensure_scratch(c, (size_t) count * 8)) return r;
    VFT_DISPATCH(c, launch((k_nvectors<REAL, NC>), dim3(cdiv(count, 64)), dim3(64), 0, c->stream, arena<REAL>(c), first, count,
                           (int64_t *) c->scratch));
    LAUNCHCHK(c);
    HIPCHK(c, hipMemcpyAsync(nvec, c->scratch, (size_t) count * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_selfdist(Arena<REAL> A, const int64_t *nodes, int64_t n) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int64_t t = (int64_t) blockIdx.x * VFT_PW_WAVES + (threadIdx.x >> 6);
    if (t >= n) return;
    REAL d, w;
    vft_pair_wave<REAL, NC>(A, nodes[t], nodes[t], false, vft_pw_lds(pwLds, A.d.nPosPad, 0),
                            vft_pw_lds(pwLds, A.d.nPosPad, 1), d, w);
    if ((threadIdx.x & 63) != 0) return;
    A.selfdist[nodes[t]] = d;
    A.selfweight[nodes[t]] = w;
}

// one node, whole workgroup (the join loop's case)
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_selfdist_one(Arena<REAL> A, const int64_t *nodes) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int64_t v = nodes[0];
    REAL d, w;
    vft_pair_block<REAL, NC>(A, v, v, false, pwLds, pwLds + A.d.nPosPad, d, w);
    if (threadIdx.x != 0) return;
    A.selfdist[v] = d;
    A.selfweight[v] = w;
}

// dynamic LDS of the wave-per-item kernels
static size_t pw_lds_bytes(const vft_ctx *c) { return (size_t) c->pwWaves * 2 * c->d.nPosPad * sizeof(double); }

// long alignments: those kernels stage 2 doubles per column and wave, which can exceed the default dynamic-LDS limit
static int raise_pair_kernel_lds(vft_ctx *c) {
    // 2 x nPosPad doubles per item: 4 items per workgroup up to 2560 columns, 2 up to 5120, 1 up to 10240
    c->pwWaves = 4;
    while (c->pwWaves > 1 && pw_lds_bytes(c) > (160u << 10)) c->pwWaves >>= 1;
    const size_t bytes = pw_lds_bytes(c);
    {   // how many workgroups of the single-launch pair list are resident at once (its pair workgroups wait for refresh
        // workgroups of the same grid, so the host only uses it while the whole grid fits)
        int dev = 0, cus = 0, perCU = 0;
        HIPCHK(c, hipGetDevice(&dev));
        HIPCHK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const size_t lds = bytes / c->pwWaves;
        hipError_t e = hipErrorUnknown;
        VFT_DISPATCH(c, (e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void *) k_pairs_refresh_fused<REAL, NC>, VFT_WG, lds)));
        c->fusedLimit = (e == hipSuccess && perCU > 0 && cus > 0) ? std::min(4096, perCU * cus) : 256;
    }
    if (bytes <= (48u << 10)) return VFT_OK;
    if (bytes > (160u << 10)) return fail(c, VFT_ERR_INVALID, "alignment too long for the pair kernels' LDS staging (%lld columns, limit 10240)", (long long) c->d.nPos);
    VFT_DISPATCH(c, {
        HIPCHK(c, hipFuncSetAttribute((const void *) k_pairs_fused<REAL, NC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_pairs_fused<REAL, NC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_pairs_block<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_pairs_refresh_fused<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_refresh_list<REAL, NC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_refresh_list<REAL, NC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_out_distances<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_out_distance_one<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_selfdist<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_selfdist_one<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        HIPCHK(c, hipFuncSetAttribute((const void *) k_sweep_wave<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
    });
    return VFT_OK;
}

extern "C" int vft_average_profiles(vft_ctx *c, int64_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                                    const double *bionj) {
    if (!c || n < 0 || !out || !a || !b) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    for (int64_t k = 0; k < n; k++) {
        if (int r = internal_ok(c, out[k])) return r;
        if (a[k] < 0 || a[k] >= c->d.maxNodes || b[k] < 0 || b[k] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "bad child id");
    }
    const bool rows = c->rowMode;   // tree-refinement phase: plain rows, no tile re-pack, no self distances
    if (rows)
        if (int r = ensure_ml_rows(c)) return r;
    CommitPlan plan = commit_plan(c, n);
    if (rows) {
        plan.chunk = 32768;
        plan.totalB = 0;
    }
    const int64_t chunk = plan.chunk;
    const size_t idB = (((size_t) n * 8) + 255) & ~(size_t) 255;
    const bool smallIds = 4 * idB <= VFT_SMALL_BYTES;
    if (int r = ensure_scratch(c, (smallIds ? 0 : 4 * idB) + plan.totalB + 512)) return r;
    char *s, *base;
    if (smallIds) {
        char *h;
        if (int r = io_alloc(c, 4 * idB, &h, &s)) return r;
        memcpy(h, out, (size_t) n * 8);
        memcpy(h + idB, a, (size_t) n * 8);
        memcpy(h + 2 * idB, b, (size_t) n * 8);
        if (bionj) memcpy(h + 3 * idB, bionj, (size_t) n * 8);
        base = (char *) c->scratch;
    } else {
        s = (char *) c->scratch;
        base = s + 4 * idB;
        HIPCHK(c, hipMemcpyAsync(s, out, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + idB, a, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + 2 * idB, b, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
        if (bionj) HIPCHK(c, hipMemcpyAsync(s + 3 * idB, bionj, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
    }
    base += (256 - ((uintptr_t) base & 255)) & 255;
    for (int64_t k0 = 0; k0 < n; k0 += chunk) {
        const int64_t cnt = n - k0 < chunk ? n - k0 : chunk;
        VFT_DISPATCH(c, {
            const dim3 grid(cdiv(c->d.nPos, 128), (unsigned) cnt);
            launch((k_average<REAL, NC>), grid, dim3(128), 0, c->stream, arena<REAL>(c), (const int64_t *) s + k0,
                   (const int64_t *) (s + idB) + k0, (const int64_t *) (s + 2 * idB) + k0,
                   bionj ? (const double *) (s + 3 * idB) + k0 : (const double *) nullptr, c->fpostTol,
                   rows ? (REAL *) nullptr : (REAL *) base);
        });
        LAUNCHCHK(c);
        if (!rows)
            if (int r = commit_nodes(c, plan, out + k0, (const int64_t *) s + k0, cnt, base)) return r;
    }
    if (rows) {
        if (!smallIds) HIPCHK(c, hipStreamSynchronize(c->stream));
        return VFT_OK;
    }
    if (n == 1) {
        VFT_DISPATCH(c, launch((k_selfdist_one<REAL, NC>), dim3(1), dim3(VFT_WG), pw_lds_bytes(c) / c->pwWaves, c->stream,
                               arena<REAL>(c), (const int64_t *) s));
    } else {
        VFT_DISPATCH(c, launch((k_selfdist<REAL, NC>), dim3(cdiv(n, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c), c->stream,
                               arena<REAL>(c), (const int64_t *) s, n));
    }
    LAUNCHCHK(c);
    // id lists in the mapped ring are protected by its wrap-around synchronisation; only the scratch path must wait
    if (!smallIds) HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// n unweighted averages in order, later ones may read earlier results (k_average_chain); profile-rows mode only.
// nChains independent chains go down as one launch (blockIdx.y): chainOff[nChains + 1] splits the n ops (NULL: one chain).
static int average_chains(vft_ctx *c, int32_t nChains, const int32_t *chainOff, int32_t n, const int64_t *out, const int64_t *a,
                          const int64_t *b, const char *who) {
    if (n == 0) return VFT_OK;
    if (!c->rowMode) return fail(c, VFT_ERR_STATE, "%s needs vft_set_profile_rows(ctx, 1)", who);
    for (int32_t k = 0; k < n; k++) {
        if (int r = internal_ok(c, out[k])) return r;
        if (a[k] < 0 || a[k] >= c->d.maxNodes || b[k] < 0 || b[k] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "bad child id");
    }
    if (int r = ensure_ml_rows(c)) return r;
    const size_t idB = (size_t) n * 8, dB = ((size_t) n + 7) & ~(size_t) 7, offB = chainOff ? (size_t) (nChains + 1) * 4 : 0;
    char *h, *s;
    if (int r = io_alloc(c, 3 * idB + dB + offB, &h, &s)) return r;
    memcpy(h, out, idB);
    memcpy(h + idB, a, idB);
    memcpy(h + 2 * idB, b, idB);
    uint8_t *direct = (uint8_t *) (h + 3 * idB);
    for (int32_t ch = 0; ch < nChains; ch++) {
        const int32_t k0 = chainOff ? chainOff[ch] : 0, k1 = chainOff ? chainOff[ch + 1] : n;
        for (int32_t k = k0; k < k1; k++) {
            uint8_t d = 0;
            for (int32_t j = k0; j < k; j++) {
                if (out[j] == a[k]) d |= 1;
                if (out[j] == b[k]) d |= 2;
            }
            direct[k] = d;
        }
    }
    if (chainOff) memcpy(h + 3 * idB + dB, chainOff, offB);
    VFT_DISPATCH(c, launch((k_average_chain<REAL, NC>), dim3(cdiv(c->d.nPos, 128), (unsigned) nChains), dim3(128), 0, c->stream, arena<REAL>(c),
                           (const int64_t *) s, (const int64_t *) (s + idB), (const int64_t *) (s + 2 * idB),
                           (const uint8_t *) (s + 3 * idB), n, c->fpostTol, chainOff ? (const int32_t *) (s + 3 * idB + dB) : nullptr));
    if (!c->allRows) launch(k_mark_rows, dim3(cdiv(n, 64)), dim3(64), 0, c->stream, c->mlIs, (const int64_t *) s, n, c->d.nSeqs);
    LAUNCHCHK(c);
    return VFT_OK;
}

static int chains_ok(vft_ctx *c, int32_t nChains, const int32_t *chainOff, const char *who) {
    if (nChains < 1 || nChains > 65535 || chainOff[0] != 0) return fail(c, VFT_ERR_INVALID, "%s: bad chain table", who);
    for (int32_t ch = 0; ch < nChains; ch++)
        if (chainOff[ch + 1] < chainOff[ch] || chainOff[ch + 1] - chainOff[ch] > 256)
            return fail(c, VFT_ERR_INVALID, "%s: a chain holds 0..256 ops", who);
    if (chainOff[nChains] > 4096) return fail(c, VFT_ERR_INVALID, "%s: at most 4096 ops per call", who);
    return VFT_OK;
}

extern "C" int vft_average_chain(vft_ctx *c, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b) {
    if (!c || n < 0 || !out || !a || !b) return VFT_ERR_INVALID;
    if (n > 256) return fail(c, VFT_ERR_INVALID, "vft_average_chain: at most 256 averages per call");
    return average_chains(c, 1, nullptr, n, out, a, b, "vft_average_chain");
}

extern "C" int vft_average_chains(vft_ctx *c, int32_t nChains, const int32_t *chainOff, const int64_t *out, const int64_t *a,
                                  const int64_t *b) {
    if (!c || !chainOff || !out || !a || !b) return VFT_ERR_INVALID;
    if (int r = chains_ok(c, nChains, chainOff, "vft_average_chains")) return r;
    return average_chains(c, nChains, chainOff, chainOff[nChains], out, a, b, "vft_average_chains");
}

// ---------------------------------------------------------------------------------------------- the walk server (vft_kernels_walk.h)
// Host side: the mailbox (self-tagged 8-byte granules), the answers, start / stop.  One server per process (g_walkServerOwner).
static inline void ws_put(vft_ctx *c, uint32_t seq, int g, uint32_t data) {
    volatile unsigned long long *slot = c->ws.hMail + (size_t) (seq % VFT_WS_RING) * VFT_WS_GRAN;
    slot[g] = ((unsigned long long) seq << 32) | (unsigned long long) data;
}
// all six workgroups have answered command `seq` (isDist: with a distance, copied to dist[6] when not null)
static bool ws_answered(vft_ctx *c, uint32_t seq, bool wide) {
    const volatile unsigned long long *slot = c->ws.hRes + (size_t) (seq % VFT_WS_RING) * VFT_WS_RESG;
    for (int w = 0; w < VFT_WS_NWG; w++) {
        if ((uint32_t) (slot[2 * w] >> 32) != seq) return false;
        if (wide && (uint32_t) (slot[2 * w + 1] >> 32) != seq) return false;
    }
    return true;
}
static int ws_wait(vft_ctx *c, uint32_t seq, bool wide) {
    std::chrono::steady_clock::time_point t0;
    for (long spins = 0;; spins++) {
        if (ws_answered(c, seq, wide)) {
            if ((int32_t) (seq - c->ws.acked) > 0) c->ws.acked = seq;   // (the workgroups answer in order)
            return VFT_OK;
        }
        if (spins == 200000) t0 = std::chrono::steady_clock::now();
        if (spins >= 200000 && (spins & 0xFFFF) == 0) {
            const hipError_t e = hipStreamQuery(c->ws.stream);
            if (e == hipSuccess) {   // the server has left
                if (ws_answered(c, seq, wide)) continue;
                c->ws.up = false;
                if (g_walkServerOwner == c) g_walkServerOwner = nullptr;
                return fail(c, VFT_ERR_TIMEOUT, "the walk server has stopped (status %llu %llu %llu %llu %llu %llu) with command %u unanswered",
                            c->ws.hStatus[0], c->ws.hStatus[1], c->ws.hStatus[2], c->ws.hStatus[3], c->ws.hStatus[4], c->ws.hStatus[5], seq);
            }
            if (e != hipErrorNotReady) return fail(c, VFT_ERR_HIP, "walk server stream: %s", hipGetErrorString(e));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->waitLimitS)
                return fail(c, VFT_ERR_TIMEOUT, "no answer from the walk server to command %u after %.0f s", seq, c->waitLimitS);
        }
    }
}
// one command: nOps averages (<= VFT_WS_MAXOPS) and, with q, the six distances of the quartet; returns its sequence number
static int ws_command(vft_ctx *c, int32_t nOps, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *q, uint32_t *seqOut) {
    const uint32_t seq = ++c->ws.seq;
    const bool wide = c->rs == 8;
    // the slot of seq is the slot of seq - RING, and the answers share a ring of the same length: stay well inside it
    while ((int32_t) (seq - c->ws.acked) > VFT_WS_RING / 2)   // (the first granule of every workgroup's answer carries the tag, with or without distances)
        if (int r = ws_wait(c, c->ws.acked + 1, false)) return r;
    const bool noWait = c->ws.acked == seq - 1;   // every earlier answer has been seen: no workgroup is still reading rows
    for (int t = 0; t < 4; t++) ws_put(c, seq, 1 + t, q ? (uint32_t) (int32_t) q[t] : 0u);
    for (int32_t k = 0; k < nOps; k++) {
        ws_put(c, seq, 5 + 3 * k, (uint32_t) (int32_t) out[k]);
        ws_put(c, seq, 6 + 3 * k, (uint32_t) (int32_t) a[k]);
        ws_put(c, seq, 7 + 3 * k, (uint32_t) (int32_t) b[k]);
    }
    ws_put(c, seq, 0, VFT_WS_CMD_WORK | ((uint32_t) nOps << 8) | (q ? 1u << 16 : 0u) | (noWait ? 1u << 17 : 0u) | (c->ws.scoredist ? 1u << 21 : 0u));
    if (c->ws.mailOnDevice) __builtin_ia32_sfence();
    *seqOut = seq;
    return VFT_OK;
}

extern "C" int vft_walk_server_start(vft_ctx *c) {
    if (!c) return VFT_ERR_INVALID;
    if (c->ws.up) return VFT_OK;
    if (c->ws.disabled) return fail(c, VFT_ERR_STATE, "vft_walk_server_start: switched off (VFT_DEBUG_NO_WALK_SERVER)");
    if (!c->rowMode || !c->allRows) return fail(c, VFT_ERR_STATE, "vft_walk_server_start: every internal profile must be a plain row (vft_set_profile_rows)");
    if (c->d.maxNodes > 0x7FFFFFFF || c->d.nPos * 20 * 8 > 0x7FFFFFFF) return fail(c, VFT_ERR_STATE, "vft_walk_server_start: ids and row offsets are 32-bit");
    if (g_walkServerOwner && g_walkServerOwner != c) return fail(c, VFT_ERR_STATE, "vft_walk_server_start: another context's server is resident");
    if ((c->d.nCodes == 20) != c->hasDm) return fail(c, VFT_ERR_STATE, "vft_walk_server_start: built for nucleotides without and proteins with a distance matrix");
    const size_t lds = (size_t) 2 * ((c->d.nPos + 31) / 32 * 32) * sizeof(double);   // (the server's ordered sums add whole thirty-twos)
    const size_t staticLds = (c->d.nCodes == 20 ? (size_t) 840 * c->rs : 8) + 1024;
    if (lds + staticLds > (160u << 10)) return fail(c, VFT_ERR_STATE, "vft_walk_server_start: alignment too long for one workgroup per pair");
    vft_ctx::WalkServerHost &W = c->ws;
    // (`allocated` is raised behind the LAST allocation and the synchronisation below: a start that failed anywhere before that -
    // the flags, the tick block, the mailbox, its device pointer - must not be launched on null pointers by the next call)
    if (W.stream && !W.allocated) return fail(c, VFT_ERR_STATE, "vft_walk_server_start: an earlier start failed half-way (the walks keep the plain calls)");
    if (!W.stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking));
        HIPCHK(c, hipHostMalloc((void **) &W.hRes, (size_t) VFT_WS_RING * VFT_WS_RESG * 8 + 512, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void **) &W.dRes, W.hRes, 0));
        memset(W.hRes, 0, (size_t) VFT_WS_RING * VFT_WS_RESG * 8 + 512);
        W.hStatus = W.hRes + (size_t) VFT_WS_RING * VFT_WS_RESG;
        W.dStatus = W.dRes + (size_t) VFT_WS_RING * VFT_WS_RESG;
        HIPCHK(c, dalloc(&W.dFlags, 64));
        HIPCHK(c, hipMemset(W.dFlags, 0, 64 * 8));
        if (!g_wsTicks) {   // (per process: tools read it after the contexts are gone)
            HIPCHK(c, dalloc(&g_wsTicks, 16));
            HIPCHK(c, hipMemset(g_wsTicks, 0, 16 * 8));
        }
        const size_t mailBytes = (size_t) VFT_WS_RING * VFT_WS_GRAN * 8;
        int largeBar = 0;
        if (W.wantDeviceMail) (void) hipDeviceGetAttribute(&largeBar, hipDeviceAttributeIsLargeBar, c->cfg.device);
        if (W.wantDeviceMail && largeBar) {   // device memory the CPU writes through the PCIe aperture: the polls stay on the device
            HIPCHK(c, hipExtMallocWithFlags((void **) &W.dMail, mailBytes, hipDeviceMallocFinegrained));
            HIPCHK(c, hipMemset(W.dMail, 0, mailBytes));
            W.hMail = W.dMail;
            W.mailOnDevice = true;
        } else {
            HIPCHK(c, hipHostMalloc((void **) &W.hMail, mailBytes, hipHostMallocMapped));
            HIPCHK(c, hipHostGetDevicePointer((void **) &W.dMail, W.hMail, 0));
            memset(W.hMail, 0, mailBytes);
        }
        HIPCHK(c, hipDeviceSynchronize());
        W.allocated = true;
    }
    if (int r = wait_stream(c)) return r;   // everything queued on the context's stream has written its rows
    for (int w = 0; w < VFT_WS_NWG; w++) W.hStatus[w] = 0;
    WalkServerArgs S{};
    S.mail = W.dMail;
    S.res = W.dRes;
    S.flags = W.dFlags;
    S.status = W.dStatus;
    S.ticks = g_wsTicks;
    S.tol = c->fpostTol;
    S.firstSeq = W.seq + 1;
    S.stride = W.stride;
    S.idleTicks = 60000000000ll;  // ten minutes without a command: the host is stuck (a host that has gone away takes its queues with it)
    S.flagTicks = 500000000ll;    // 5 s for another workgroup's flag
    W.acked = W.seq;
    VFT_DISPATCH(c, {
        if (lds > (48u << 10)) HIPCHK(c, hipFuncSetAttribute((const void *) (k_walk_server<REAL, NC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipLaunchKernelGGL((k_walk_server<REAL, NC>), dim3((VFT_WS_NWG - 1) * W.stride + 1), dim3(VFT_WS_WG_OF(NC)), lds, W.stream, arena<REAL>(c), S);
    });
    LAUNCHCHK(c);
    W.up = true;
    g_walkServerOwner = c;
    return VFT_OK;
}

static int walk_server_retire(vft_ctx *c) {
    vft_ctx::WalkServerHost &W = c->ws;
    if (!W.up) return VFT_OK;
    W.up = false;
    if (g_walkServerOwner == c) g_walkServerOwner = nullptr;
    const uint32_t seq = ++W.seq;
    ws_put(c, seq, 0, VFT_WS_CMD_STOP);
    if (W.mailOnDevice) __builtin_ia32_sfence();
    HIPCHK(c, hipStreamSynchronize(W.stream));
    for (int w = 0; w < VFT_WS_NWG; w++)
        if (W.hStatus[w] != 1) return fail(c, VFT_ERR_TIMEOUT, "walk server: workgroup %d ended with status %llu", w, W.hStatus[w]);
    return VFT_OK;
}
extern "C" int vft_walk_server_stop(vft_ctx *c) {
    if (!c) return VFT_ERR_INVALID;
    return walk_server_retire(c);
}
extern "C" int vft_walk_server_ticks(int64_t *out, int32_t n) {
    if (!out || n < 0 || n > 16) return VFT_ERR_INVALID;
    for (int32_t k = 0; k < n; k++) out[k] = 0;
    if (!g_wsTicks) return VFT_OK;
    unsigned long long t[16];
    if (hipMemcpy(t, g_wsTicks, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess) return VFT_ERR_HIP;
    for (int32_t k = 0; k < n; k++) out[k] = (int64_t) t[k];
    return VFT_OK;
}

static int walk_ids_ok(vft_ctx *c, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *q, const char *who);

// a step (or, with q == NULL, averages alone) handed to the server; *ticket answers vft_walk_collect.  Steps with more than
// VFT_WS_MAXOPS averages go down as several commands, the distances with the last.
extern "C" int vft_walk_submit(vft_ctx *c, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *q, uint32_t *ticket) {
    if (!c || n < 0 || (n > 0 && (!out || !a || !b)) || !ticket) return VFT_ERR_INVALID;
    if (!c->ws.up) return fail(c, VFT_ERR_STATE, "vft_walk_submit: the walk server is not running (vft_walk_server_start)");
    if (int r = walk_ids_ok(c, n, out, a, b, q, "vft_walk_submit")) return r;
    int32_t first = 0;
    uint32_t seq = 0;
    do {
        const int32_t m = n - first > VFT_WS_MAXOPS ? VFT_WS_MAXOPS : n - first;
        const bool last = first + m == n;
        if (int r = ws_command(c, m, out + first, a + first, b + first, last ? q : nullptr, &seq)) return r;
        first += m;
    } while (first < n);
    *ticket = seq;
    return VFT_OK;
}
// Both continuations of an SPR chain behind the step that is running (vft_kernels_walk.h "DUAL command"): alternative 0 is taken when the
// comparison of the PREVIOUS command's distances says "swap B and C" (criteria[1] < criteria[2]), alternative 1 otherwise.  q0 / q1 == NULL:
// that alternative is not a device step - if it is the one taken the command counts as an empty one.  n0 + n1 <= VFT_WS_DUAL_MAXOPS
// (VFT_ERR_INVALID beyond: the caller waits for the answer instead).  The command before must be a step with distances.
extern "C" int vft_walk_submit_dual(vft_ctx *c, int32_t n0, const int64_t *out0, const int64_t *a0, const int64_t *b0, const int64_t *q0,
                                    int32_t n1, const int64_t *out1, const int64_t *a1, const int64_t *b1, const int64_t *q1,
                                    int32_t scoredist, uint32_t *ticket) {
    if (!c || n0 < 0 || n1 < 0 || !ticket || (n0 > 0 && (!out0 || !a0 || !b0)) || (n1 > 0 && (!out1 || !a1 || !b1))) return VFT_ERR_INVALID;
    if (!c->ws.up) return fail(c, VFT_ERR_STATE, "vft_walk_submit_dual: the walk server is not running (vft_walk_server_start)");
    c->ws.scoredist = scoredist != 0;   // (the step in front was sent with the flavour of the call before: the caller announces it with vft_walk_scoredist)
    if (!q0) n0 = 0;
    if (!q1) n1 = 0;
    if (n0 + n1 > VFT_WS_DUAL_MAXOPS) return fail(c, VFT_ERR_INVALID, "vft_walk_submit_dual: %d + %d averages do not fit one command", (int) n0, (int) n1);
    if (q0)
        if (int r = walk_ids_ok(c, n0, out0, a0, b0, q0, "vft_walk_submit_dual")) return r;
    if (q1)
        if (int r = walk_ids_ok(c, n1, out1, a1, b1, q1, "vft_walk_submit_dual")) return r;
    const uint32_t seq = ++c->ws.seq;
    while ((int32_t) (seq - c->ws.acked) > VFT_WS_RING / 2)
        if (int r = ws_wait(c, c->ws.acked + 1, false)) return r;
    int g = 1;
    for (int t = 0; t < 4; t++) ws_put(c, seq, g++, q0 ? (uint32_t) (int32_t) q0[t] : 0u);
    for (int32_t k = 0; k < n0; k++) {
        ws_put(c, seq, g++, (uint32_t) (int32_t) out0[k]);
        ws_put(c, seq, g++, (uint32_t) (int32_t) a0[k]);
        ws_put(c, seq, g++, (uint32_t) (int32_t) b0[k]);
    }
    for (int t = 0; t < 4; t++) ws_put(c, seq, g++, q1 ? (uint32_t) (int32_t) q1[t] : 0u);
    for (int32_t k = 0; k < n1; k++) {
        ws_put(c, seq, g++, (uint32_t) (int32_t) out1[k]);
        ws_put(c, seq, g++, (uint32_t) (int32_t) a1[k]);
        ws_put(c, seq, g++, (uint32_t) (int32_t) b1[k]);
    }
    // (bit 17 - "no workgroup is still reading rows" - stays clear: the step before is running)
    ws_put(c, seq, 0, VFT_WS_CMD_WORK | ((uint32_t) n0 << 8) | (1u << VFT_WS_DUAL_BIT) | (q0 ? 0u : 1u << 19) | (q1 ? 0u : 1u << 20) |
                      (scoredist ? 1u << 21 : 0u) | ((uint32_t) n1 << 24));
    if (c->ws.mailOnDevice) __builtin_ia32_sfence();
    *ticket = seq;
    return VFT_OK;
}
// logCorrect's flavour of the walk that follows (0: Jukes-Cantor, 1: scoredist-like): every step's workgroups log-correct their distance
// with it for a dual command that may follow
extern "C" int vft_walk_scoredist(vft_ctx *c, int32_t scoredist) {
    if (!c) return VFT_ERR_INVALID;
    c->ws.scoredist = scoredist != 0;
    return VFT_OK;
}
// which alternative the workgroups took for the dual command `ticket` (waits for its first answers): *alt = 0 / 1, *skipped = 1 when
// that alternative was not a device step
extern "C" int vft_walk_dual_choice(vft_ctx *c, uint32_t ticket, int32_t *alt, int32_t *skipped) {
    if (!c || !alt || !skipped) return VFT_ERR_INVALID;
    if ((int32_t) (c->ws.seq - ticket) >= VFT_WS_RING) return fail(c, VFT_ERR_STATE, "vft_walk_dual_choice: ticket %u is too old", ticket);
    if (int r = ws_wait(c, ticket, false)) return r;
    const volatile unsigned long long *slot = c->ws.hRes + (size_t) (ticket % VFT_WS_RING) * VFT_WS_RESG;
    for (long spins = 0; (uint32_t) (slot[VFT_WS_RES_CHOICE] >> 32) != ticket; spins++)   // (workgroup 0 writes it before its answer: a few polls at most)
        if (spins > 100000000) return fail(c, VFT_ERR_TIMEOUT, "vft_walk_dual_choice: command %u carries no choice (not a dual command?)", ticket);
    const uint32_t v = (uint32_t) slot[VFT_WS_RES_CHOICE];
    *alt = (int32_t) ((v >> 1) & 1u);
    *skipped = (int32_t) ((v >> 2) & 1u);
    return VFT_OK;
}

// waits for the answer to a ticket; dist[6] (numeric_t) when the step asked for distances (NULL: an acknowledgement is waited for)
extern "C" int vft_walk_collect(vft_ctx *c, uint32_t ticket, void *dist) {
    if (!c) return VFT_ERR_INVALID;
    const bool wide = c->rs == 8 && dist != nullptr;
    // The answers share a ring of VFT_WS_RING slots: a ticket that old has had its slot rewritten (ws_command's flow control runs the
    // sequence numbers at most VFT_WS_RING / 2 ahead of the acknowledged ones, not of the tickets a caller still holds) - say so at
    // once instead of spinning for a tag that will never show again.
    if ((int32_t) (c->ws.seq - ticket) >= VFT_WS_RING)
        return fail(c, VFT_ERR_STATE, "vft_walk_collect: ticket %u is %d commands old, its answer slot has been reused (collect within %d submissions)",
                    ticket, (int) (c->ws.seq - ticket), VFT_WS_RING - 1);
    if (int r = ws_wait(c, ticket, wide)) return r;
    if (dist) {
        const volatile unsigned long long *slot = c->ws.hRes + (size_t) (ticket % VFT_WS_RING) * VFT_WS_RESG;
        for (int w = 0; w < VFT_WS_NWG; w++) {
            if (c->rs == 4) {
                const uint32_t v = (uint32_t) slot[2 * w];
                memcpy((char *) dist + 4 * w, &v, 4);
            } else {
                const unsigned long long v = (slot[2 * w] & 0xFFFFFFFFull) | (slot[2 * w + 1] << 32);
                memcpy((char *) dist + 8 * w, &v, 8);
            }
        }
    }
    return VFT_OK;
}
static int walk_ids_ok(vft_ctx *c, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *q, const char *who) {
    for (int32_t k = 0; k < n; k++) {
        if (int r = internal_ok(c, out[k])) return r;
        if (a[k] < 0 || a[k] >= c->d.maxNodes || b[k] < 0 || b[k] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "%s: bad child id", who);
    }
    if (q)
        for (int t = 0; t < 4; t++)
            if (q[t] < 0 || q[t] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "%s: quartet member out of range", who);
    return VFT_OK;
}

extern "C" int vft_walk_step(vft_ctx *c, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *q, void *dist) {
    if (!c || n < 0 || (n > 0 && (!out || !a || !b)) || !q || !dist) return VFT_ERR_INVALID;
    if (!c->ws.up) return fail(c, VFT_ERR_STATE, "vft_walk_step: the walk server is not running (vft_walk_server_start); the caller makes the two plain calls");
    uint32_t ticket;
    if (int r = vft_walk_submit(c, n, out, a, b, q, &ticket)) return r;
    return vft_walk_collect(c, ticket, dist);
}

// differ[k] = 1 when the profiles of nodes a[k] and b[k] (rows or tile streams, internal or leaf) are not bit-identical.  Waits.
extern "C" int vft_profiles_differ(vft_ctx *c, int64_t n, const int64_t *a, const int64_t *b, int32_t *differ) {
    if (!c || n < 0 || !a || !b || !differ) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (n > 4096) return fail(c, VFT_ERR_INVALID, "vft_profiles_differ: at most 4096 pairs per call");
    for (int64_t k = 0; k < n; k++)
        if (a[k] < 0 || a[k] >= c->maxnode || b[k] < 0 || b[k] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "vft_profiles_differ: pair %lld out of range", (long long) k);
    const size_t idB = (size_t) n * 8, flB = ((size_t) n * 4 + 255) & ~(size_t) 255;
    char *h, *s;
    if (int r = io_alloc(c, 2 * idB + flB, &h, &s)) return r;
    memcpy(h, a, idB);
    memcpy(h + idB, b, idB);
    memset(h + 2 * idB, 0, flB);
    VFT_DISPATCH(c, launch((k_rows_differ<REAL, NC>), dim3(cdiv(c->d.nPos, 128), (unsigned) n), dim3(128), 0, c->stream, arena<REAL>(c),
                           (const int64_t *) s, (const int64_t *) (s + idB), (int32_t *) (s + 2 * idB)));
    LAUNCHCHK(c);
    if (int w = wait_stream(c)) return w;
    memcpy(differ, h + 2 * idB, (size_t) n * 4);
    return VFT_OK;
}

extern "C" int vft_get_n_codes(vft_ctx *c, int32_t *nCodes) {
    if (!c || !nCodes) return VFT_ERR_INVALID;
    *nCodes = c->cfg.n_codes;
    return VFT_OK;
}

extern "C" int vft_get_max_nodes(vft_ctx *c, int64_t *maxNodes) {
    if (!c || !maxNodes) return VFT_ERR_INVALID;
    *maxNodes = c->d.maxNodes;
    return VFT_OK;
}

extern "C" int vft_join_fused(vft_ctx *c, int64_t i, int64_t j, int64_t newnode, double diameter, int64_t staleStamp,
                              int64_t nActiveOld, int32_t updateOutProfile) {
    if (!c) return VFT_ERR_INVALID;
    if (i < 0 || j < 0 || i == j || i >= c->maxnode || j >= c->maxnode || newnode < c->d.nSeqs || newnode >= c->d.maxNodes || nActiveOld < 2)
        return fail(c, VFT_ERR_INVALID, "vft_join_fused: bad arguments (%lld, %lld -> %lld)", (long long) i, (long long) j, (long long) newnode);
    if (c->rowMode) return fail(c, VFT_ERR_STATE, "vft_join_fused belongs to the NJ phase (before vft_set_profile_rows)");
    if (int r = ensure_ml_rows(c)) return r;
    if (!c->pendBase) {
        const CommitPlan plan = commit_plan(c, VFT_PEND_MAX);
        HIPCHK(c, hipMalloc((void **) &c->pendBase, plan.totalB + 512));
        HIPCHK(c, hipMalloc((void **) &c->pendIdsDev, VFT_PEND_MAX * sizeof(int64_t)));
    }
    if ((int64_t) c->pend.size() == VFT_PEND_MAX)
        if (int r = flush_pending(c)) return r;
    // host-side bookkeeping of vft_join_nodes
    if (newnode >= c->maxnode) c->maxnode = newnode + 1;
    c->hParent[(size_t) i] = c->hParent[(size_t) j] = (int32_t) newnode;
    const int32_t stamp = clamp_i32(staleStamp);
    if ((int64_t) stamp > c->maxStamp) c->maxStamp = stamp;
    if (c->rs == 4) ((float *) c->hOutDist)[newnode] = 0.f;
    else ((double *) c->hOutDist)[newnode] = 0.0;
    c->hNOut[newnode] = stamp;
    const int32_t slot = (int32_t) c->pend.size();
    c->pend.push_back(newnode);
    const size_t lds = (size_t) 2 * c->d.nPosPad * sizeof(double);
    VFT_DISPATCH(c, {
        if (lds > (48u << 10))
            HIPCHK(c, hipFuncSetAttribute((const void *) k_join_fused<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        launch((k_join_fused<REAL, NC>), dim3(1), dim3(VFT_WG_PROF), lds, c->stream, arena<REAL>(c), i, j, newnode, (REAL) diameter, stamp,
               nActiveOld, updateOutProfile, c->fpostTol, (REAL *) c->pendBase, c->pendIdsDev, slot);
    });
    LAUNCHCHK(c);
    return VFT_OK;
}

// After the NJ phase nothing sweeps the internal profiles any more: vft_average_profiles then writes plain rows (the
// layout of the ML phase, vft_layout.h) instead of re-packing a tile per node, and skips the self distances only the
// out-distances of the NJ phase need.  Pair distances, split supports and the ML kernels read either layout.
extern "C" int vft_set_profile_rows(vft_ctx *c, int32_t on) {
    if (!c) return VFT_ERR_INVALID;
    if (int r = flush_pending(c)) return r;
    c->rowMode = on != 0;
    c->allRows = false;
    if (!on) return VFT_OK;
    // move every existing internal profile into its row and flag ALL internal ids (also the ones written later, e.g.
    // up-profile slots): from here on a row write needs no flag bookkeeping
    if (int r = ensure_ml_rows(c)) return r;
    const int64_t cnt = c->maxnode - c->d.nSeqs;
    for (int64_t k0 = 0; k0 < cnt; k0 += 32768) {
        const int64_t n = cnt - k0 < 32768 ? cnt - k0 : 32768;
        VFT_DISPATCH(c, launch((k_rows_from_tiles<REAL, NC>), dim3(cdiv(c->d.nPos, 128), (unsigned) n), dim3(128), 0, c->stream,
                               arena<REAL>(c), c->d.nSeqs + k0));
    }
    HIPCHK(c, hipMemsetAsync(c->mlIs, 1, (size_t) (c->d.maxNodes - c->d.nSeqs), c->stream));
    LAUNCHCHK(c);
    c->allRows = true;
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- out-profile
extern "C" int vft_out_profile_full(vft_ctx *c, int64_t n, const int64_t *ids) {
    if (!c || n < 1 || !ids) return VFT_ERR_INVALID;
    if (int r = flush_pending(c)) return r;
    // fast path: the list is exactly the active nodes below maxnode in ascending order (what the join loop passes,
    // NJ.tcc:3017-3031): k_outprofile_chain, with the leaves of a matrix-free nucleotide alignment folded into a histogram
    bool tiled = !c->genericOutProfile;   // tests (vft_debug_option): the one-thread-per-column kernel as the reference
    if (tiled) {
        int64_t k = 0;
        for (int64_t v = 0; v < c->maxnode && tiled; v++) {
            if (c->hParent[(size_t) v] >= 0) continue;
            if (k >= n || ids[k] != v) tiled = false;
            k++;
        }
        if (k != n) tiled = false;
    }
    if (tiled) {
        const int64_t nTiles = (c->maxnode + 63) / 64;
        launch(k_tile_active_masks, dim3(cdiv(nTiles, 4)), dim3(256), 0, c->stream, (const int32_t *) c->parent, c->maxnode,
               c->tileMask, nTiles);
        const bool leafHist = c->cfg.n_codes == 4 && !c->hasDm;
        if (leafHist) {
            const size_t hb = (size_t) c->d.nPosPad * 4 * sizeof(unsigned int);
            if (!c->opHist) HIPCHK(c, hipMalloc((void **) &c->opHist, hb));
            HIPCHK(c, hipMemsetAsync(c->opHist, 0, hb, c->stream));
            const int64_t nLeafTiles = (c->d.nSeqs + 63) / 64;
            const unsigned gx = (unsigned) std::min<int64_t>(cdiv(nLeafTiles, 16), 256);
            launch(k_leaf_hist, dim3(gx, (unsigned) c->d.nChunk), dim3(256), 0, c->stream, (const uint4 *) c->leafT, c->d,
                   (const unsigned long long *) c->tileMask, nLeafTiles, c->opHist);
        }
        const int64_t firstTile = leafHist ? c->d.firstProfTile : 0;
        VFT_DISPATCH(c, (launch((k_outprofile_chain<REAL, NC>), dim3((unsigned) (c->d.nPosPad / OpCols<NC>::value)),
                                dim3(OpThreads<NC>::value), 0, c->stream, arena<REAL>(c), (const unsigned long long *) c->tileMask,
                                firstTile, nTiles, n, c->fpostTol, leafHist ? (const unsigned int *) c->opHist : (const unsigned int *) nullptr)));
        LAUNCHCHK(c);
        return VFT_OK;
    }
    for (int64_t k = 0; k < n; k++)
        if (ids[k] < 0 || ids[k] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "vft_out_profile_full: node %lld out of range", (long long) ids[k]);
    if (int r = ensure_scratch(c, (size_t) n * 8)) return r;
    HIPCHK(c, hipMemcpyAsync(c->scratch, ids, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
    VFT_DISPATCH(c, (launch((k_outprofile_full<REAL, NC>), dim3(cdiv(c->d.nPos, 64)), dim3(64), 0, c->stream,
                                        arena<REAL>(c), (const int64_t *) c->scratch, n, c->fpostTol)));
    LAUNCHCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// outProfile in parts (vft_kernels_profile.h: k_outprofile_partial / k_outprofile_finish).  vft_out_profile_partial: the raw sums of
// `ids` (one block of the active list) with the in-weight of a list of n_total nodes, nPos x (1 + nCodes) numbers, to the host.
// vft_out_profile_finish: n_parts such blocks (host, block after block) added in order, normalised, installed as the out-profile.
extern "C" int vft_out_profile_partial(vft_ctx *c, int64_t nTotal, int64_t n, const int64_t *ids, void *part) {
    if (!c || nTotal < 1 || n < 0 || n > nTotal || (n > 0 && !ids) || !part) return VFT_ERR_INVALID;
    if (int r = flush_pending(c)) return r;
    for (int64_t k = 0; k < n; k++)
        if (ids[k] < 0 || ids[k] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "vft_out_profile_partial: node %lld out of range", (long long) ids[k]);
    const size_t pb = (size_t) c->d.nPos * (size_t) (1 + c->d.nCodes) * c->rs;
    if (int r = ensure_scratch(c, (size_t) (n > 0 ? n : 1) * 8 + pb + 64)) return r;
    char *dIds = (char *) c->scratch, *dPart = dIds + (((size_t) (n > 0 ? n : 1) * 8 + 63) / 64) * 64;
    if (n > 0) HIPCHK(c, hipMemcpyAsync(dIds, ids, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
    VFT_DISPATCH(c, (launch((k_outprofile_partial<REAL, NC>), dim3(cdiv(c->d.nPos, 64)), dim3(64), 0, c->stream,
                            arena<REAL>(c), (const int64_t *) dIds, n, nTotal, (REAL *) dPart)));
    LAUNCHCHK(c);
    HIPCHK(c, hipMemcpyAsync(part, dPart, pb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_out_profile_finish(vft_ctx *c, int32_t nParts, const void *parts) {
    if (!c || nParts < 1 || !parts) return VFT_ERR_INVALID;
    const size_t pb = (size_t) c->d.nPos * (size_t) (1 + c->d.nCodes) * c->rs;
    if (int r = ensure_scratch(c, (size_t) nParts * pb)) return r;
    HIPCHK(c, hipMemcpyAsync(c->scratch, parts, (size_t) nParts * pb, hipMemcpyHostToDevice, c->stream));
    VFT_DISPATCH(c, (launch((k_outprofile_finish<REAL, NC>), dim3(cdiv(c->d.nPos, 64)), dim3(64), 0, c->stream,
                            arena<REAL>(c), (const REAL *) c->scratch, nParts, c->fpostTol)));
    LAUNCHCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (the scratch block may be reused by the next call)
    return VFT_OK;
}

extern "C" int vft_out_profile_update(vft_ctx *c, int64_t old1, int64_t old2, int64_t newn, int64_t nActiveOld) {
    if (!c || nActiveOld < 2) return VFT_ERR_INVALID;
    if (old1 < 0 || old1 >= c->d.maxNodes || old2 < 0 || old2 >= c->d.maxNodes || newn < c->d.nSeqs || newn >= c->d.maxNodes)
        return fail(c, VFT_ERR_INVALID, "vft_out_profile_update: bad ids (%lld, %lld -> %lld)", (long long) old1, (long long) old2, (long long) newn);
    VFT_DISPATCH(c, (launch((k_outprofile_update<REAL, NC>), dim3(cdiv(c->d.nPos, 64)), dim3(64), 0, c->stream,
                                        arena<REAL>(c), old1, old2, newn, nActiveOld, c->fpostTol)));
    LAUNCHCHK(c);
    return VFT_OK;
}

extern "C" int vft_out_profile_upload(vft_ctx *c, const void *w, const void *f, const void *cd) {
    if (!c || !w || !f) return VFT_ERR_INVALID;
    const size_t rs = c->rs;
    HIPCHK(c, hipMemcpyAsync(c->outW, w, c->d.nPos * rs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->outF, f, c->d.nPos * c->d.nCodes * rs, hipMemcpyHostToDevice, c->stream));
    if (cd) HIPCHK(c, hipMemcpyAsync(c->outCD, cd, c->d.nPos * c->d.nCodes * rs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_out_profile_download(vft_ctx *c, void *w, void *f, void *cd) {
    if (!c) return VFT_ERR_INVALID;
    const size_t rs = c->rs;
    if (w) HIPCHK(c, hipMemcpyAsync(w, c->outW, c->d.nPos * rs, hipMemcpyDeviceToHost, c->stream));
    if (f) HIPCHK(c, hipMemcpyAsync(f, c->outF, c->d.nPos * c->d.nCodes * rs, hipMemcpyDeviceToHost, c->stream));
    if (cd) HIPCHK(c, hipMemcpyAsync(cd, c->outCD, c->d.nPos * c->d.nCodes * rs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// geometry of the two launches of an nt sweep over [s.lo, s.hi): with a profile query k_sweep_nt_table takes the
// leaves (VFT_LEAF_SPAN per workgroup) and k_sweep_nt starts at the tile that holds the first internal node; with a
// leaf query k_sweep_nt takes everything.  Returns the total number of workgroups (= min/max partials).
static unsigned sweep_nt_grid(vft_ctx *c, SweepArgs &s, bool tablePath) {
    const int64_t leafEnd = c->d.nSeqs < s.hi ? c->d.nSeqs : s.hi;
    s.leafEnd = leafEnd;
    s.nLeafWG = (tablePath && leafEnd > s.lo) ? (int32_t) cdiv(leafEnd - s.lo, VFT_LEAF_SPAN) : 0;
    s.heavyLo = s.nLeafWG ? (leafEnd / VFT_TILE) * VFT_TILE : s.lo;
    const int64_t rest = s.hi > s.heavyLo ? s.hi - s.heavyLo : 0;
    return (unsigned) s.nLeafWG + (rest > 0 ? cdiv(rest, VFT_WG) : 0u);
}

static void kernel_event(vft_ctx *c);
// the two launches of one nt sweep (vft_kernels_nj.h); events: before, between, after
template <typename REAL, int MODE>
static void launch_sweep_nt(vft_ctx *c, const SweepArgs &s, unsigned grid, const QueryBuf<REAL> &Q, bool timed, int slot = 0) {
    const unsigned nHeavy = grid - (unsigned) s.nLeafWG;
    if (MODE == MODE_CRIT && s.nLeafWG && nHeavy) {   // a profile seed's criteria: both kinds of workgroups in one launch
        if (timed) kernel_event(c);
        launch((k_sweep_nt_both<REAL>), dim3(grid), dim3(VFT_WG), 0, c->stream, arena<REAL>(c), Q, s, sweepout<REAL>(c, slot));
        if (timed) kernel_event(c);
        if (timed) kernel_event(c);
        return;
    }
    if (timed) kernel_event(c);
    if (nHeavy) launch((k_sweep_nt<REAL, MODE>), dim3(nHeavy), dim3(VFT_WG), 0, c->stream, arena<REAL>(c),
                       Q, s, sweepout<REAL>(c, slot));
    if (timed) kernel_event(c);
    if (s.nLeafWG) launch((k_sweep_nt_table<REAL, (MODE == MODE_OUTDIST ? MODE_OUTDIST : MODE_CRIT)>), dim3((unsigned) s.nLeafWG), dim3(VFT_WG), 0, c->stream,
                          arena<REAL>(c), Q, s, sweepout<REAL>(c, slot));
    if (timed) kernel_event(c);
}

// amino acids with a distance matrix: query tables + the tile kernels of vft_kernels_aa.h over [s.lo, s.hi).
// Returns the number of workgroups (= min / max partials of a MODE_CRIT launch).
template <typename REAL, int MODE>
static unsigned launch_sweep_aa(vft_ctx *c, const SweepArgs &s, int whichQuery, int slot) {
    const int64_t nPosPad = (int64_t) c->d.nChunk * VFT_CHUNK;
    REAL *wq = (REAL *) c->qW[whichQuery], *qvec = (REAL *) c->qF[whichQuery], *ptab = (REAL *) c->qPT[whichQuery];
    launch((k_aa_query_prep<REAL>), dim3(cdiv(nPosPad * 20, 256)), dim3(256), 0, c->stream, arena<REAL>(c), s.query, wq, qvec, ptab);
    const int64_t leafEnd = c->d.nSeqs < s.hi ? c->d.nSeqs : s.hi;
    const int64_t leafTile0 = s.lo / VFT_TILE;
    const int64_t nLeafTiles = leafEnd > s.lo ? (leafEnd + VFT_TILE - 1) / VFT_TILE - leafTile0 : 0;
    const int32_t nLeafWG = (int32_t) ((nLeafTiles + VFT_AA_WG / 64 - 1) / (VFT_AA_WG / 64));
    const int64_t intTile0 = leafTile0 > c->d.firstProfTile ? leafTile0 : c->d.firstProfTile;
    const int64_t intTileEnd = (s.hi + VFT_TILE - 1) / VFT_TILE;
    const int64_t nIntTiles = (s.hi > c->d.nSeqs && intTileEnd > intTile0) ? intTileEnd - intTile0 : 0;
    const unsigned grid = (unsigned) nLeafWG + (unsigned) nIntTiles;
    if (grid == 0) return 0;
    AaQuery<REAL> Q;
    Q.wq = wq;
    Q.qvec = qvec;
    Q.ptab = ptab;
    launch((k_sweep_aa<REAL, MODE>), dim3(grid), dim3(VFT_AA_WG), c->aaLds, c->stream, arena<REAL>(c), Q, s, sweepout<REAL>(c, slot),
           leafTile0, nLeafWG, intTile0);
    return grid;
}

// launches the out-distance refresh over [lo,hi) (ids == nullptr) or over a device id list
static int launch_out_distances(vft_ctx *c, const int64_t *dIds, int64_t n, int64_t nActive, int64_t nDiffAllow,
                                double totdiam, bool force) {
    SweepArgs s{};
    s.query = -1;
    s.lo = (c->refreshAll && !dIds) ? 0 : c->shardLo;
    s.hi = (c->refreshAll && !dIds) ? c->maxnode : (c->shardHi < c->maxnode ? c->shardHi : c->maxnode);
    s.nActive = nActive;
    s.nDiffAllow = nDiffAllow;
    s.totdiam = totdiam;
    s.queryIsLeaf = 0;
    s.force = force ? 1 : 0;
    if (dIds) {
        VFT_DISPATCH(c, (launch((k_out_distances<REAL, NC>), dim3(cdiv(n, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c),
                                c->stream, arena<REAL>(c), dIds, n, s)));
        LAUNCHCHK(c);
        return VFT_OK;
    }
    if (s.hi <= s.lo) return VFT_OK;
    if (int r = flush_pending(c)) return r;   // the sweep kernels read tile streams
    // after this pass every active node of the shard carries a stamp of at most nActive (forced) / nActive + nDiffAllow
    {
        const int64_t bound = force ? nActive : nActive + nDiffAllow;
        if (force) c->maxStamp = bound;
        else if (bound < c->maxStamp) c->maxStamp = bound;
    }
    const int64_t span = s.hi - s.lo;
    if (c->cfg.n_codes == 4 && !c->hasDm) {
        const int64_t nPosPad = (int64_t) c->d.nChunk * VFT_CHUNK;
        const unsigned grid = sweep_nt_grid(c, s, true);
        if (c->cfg.precision == 4) {
            launch((k_outprofile_as_query<float, 4>), dim3(cdiv(nPosPad, 256)), dim3(256), 0, c->stream,
                               arena<float>(c), qbuf<float>(c, 1));
            launch_sweep_nt<float, MODE_OUTDIST>(c, s, grid, qbuf<float>(c, 1), false);
        } else {
            launch((k_outprofile_as_query<double, 4>), dim3(cdiv(nPosPad, 256)), dim3(256), 0, c->stream,
                               arena<double>(c), qbuf<double>(c, 1));
            launch_sweep_nt<double, MODE_OUTDIST>(c, s, grid, qbuf<double>(c, 1), false);
        }
    } else if (c->cfg.n_codes == 20 && c->hasDm && c->aaLds) {
        if (c->cfg.precision == 4) launch_sweep_aa<float, MODE_OUTDIST>(c, s, 1, 0);
        else launch_sweep_aa<double, MODE_OUTDIST>(c, s, 1, 0);
    } else {
        VFT_DISPATCH(c, (launch((k_out_distances<REAL, NC>), dim3(cdiv(span, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c),
                                c->stream, arena<REAL>(c), (const int64_t *) nullptr, span, s)));
    }
    LAUNCHCHK(c);
    return VFT_OK;
}

extern "C" int vft_out_distances(vft_ctx *c, int64_t n, const int64_t *ids, int64_t nActive, double totdiam) {
    if (!c || nActive < 2) return VFT_ERR_INVALID;
    if (ids) {
        if (n <= 0) return VFT_OK;
        if (nActive > c->maxStamp) c->maxStamp = nActive;   // forced refreshes stamp the listed nodes with nActive
        if (n == 1) {
            if (ids[0] < 0 || ids[0] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "vft_out_distances: bad node");
            SweepArgs s{};
            s.nActive = nActive;
            s.totdiam = totdiam;
            s.force = 1;
            VFT_DISPATCH(c, launch((k_out_distance_one<REAL, NC>), dim3(1), dim3(VFT_WG), pw_lds_bytes(c) / c->pwWaves, c->stream,
                                   arena<REAL>(c), ids[0], s));
            LAUNCHCHK(c);
            return VFT_OK;
        }
        char *h, *d;
        if ((size_t) n * 8 <= VFT_SMALL_BYTES) {
            if (int r = io_alloc(c, (size_t) n * 8, &h, &d)) return r;
            memcpy(h, ids, (size_t) n * 8);
            return launch_out_distances(c, (const int64_t *) d, n, nActive, 0, totdiam, true);
        }
        if (int r = ensure_scratch(c, (size_t) n * 8)) return r;
        HIPCHK(c, hipMemcpyAsync(c->scratch, ids, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
        if (int r = launch_out_distances(c, (const int64_t *) c->scratch, n, nActive, 0, totdiam, true)) return r;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return VFT_OK;
    }
    return launch_out_distances(c, nullptr, 0, nActive, 0, totdiam, true);
}

// ---------------------------------------------------------------------------------------------- sweep
static void kernel_event(vft_ctx *c) {   // three per sweep: before / between / after its two kernels
    if (!c->timeKernels) return;
    if (c->kevUsed == c->kev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->kev.push_back(e);
    }
    hipEventRecord(c->kev[c->kevUsed++], c->stream);
}

// makes sure the context owns `count` sets of sweep-result / selection buffers
static int ensure_slots(vft_ctx *c, int count) {
    const size_t rs = c->rs;
    const int64_t N = ((c->d.maxNodes + VFT_TILE - 1) / VFT_TILE + 4) * VFT_TILE;   // same padding as vft_create
    bool added = false;
    while ((int) c->slots.size() < count) {
        vft_ctx::SweepSlotHost h;
        void **reals[] = {&h.swDist, &h.swWeight, &h.swCrit};
        for (void **p: reals) {
            HIPCHK(c, hipMalloc(p, (size_t) N * rs));
            HIPCHK(c, hipMemset(*p, 0, (size_t) N * rs));   // (as slot 0's arrays in vft_create)
        }
        const size_t nPartCap = (size_t) cdiv(N, VFT_WG);
        HIPCHK(c, hipMalloc(&h.partMin, nPartCap * 8));
        HIPCHK(c, hipMalloc(&h.partMax, nPartCap * 8));
        HIPCHK(c, hipMalloc((void **) &h.sel, sizeof(SelectState)));
        HIPCHK(c, hipMalloc((void **) &h.slices, (size_t) VFT_NBINS * 4));
        HIPCHK(c, hipMemset(h.slices, 0, (size_t) VFT_NBINS * 4));
        HIPCHK(c, hipMalloc((void **) &h.candKey, (size_t) VFT_CAND_CAP * 8));
        HIPCHK(c, hipMalloc((void **) &h.candId, (size_t) VFT_CAND_CAP * 4));
        HIPCHK(c, hipMemset(h.candKey, 0, (size_t) VFT_CAND_CAP * 8));   // (never read before written since k_select_rank skips overflowed collections; zeroed all the same)
        HIPCHK(c, hipMemset(h.candId, 0, (size_t) VFT_CAND_CAP * 4));
        HIPCHK(c, hipMemset(h.sel, 0, sizeof(SelectState)));
        const size_t resB = sizeof(SelectHeader) + (size_t) c->hitsCap * sizeof(vft_hit_f64);
        HIPCHK(c, hipMalloc((void **) &h.dRes, resB));
        HIPCHK(c, hipHostMalloc((void **) &h.hRes, resB, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void **) &h.hResDev, h.hRes, 0));
        memset(h.hRes, 0, sizeof(SelectHeader));
        if (c->cfg.n_codes == 4 && !c->slots.empty()) {   // (slot 0 is made by vft_create and uses the context's staging)
            const size_t nPosPad = (size_t) c->d.nChunk * VFT_CHUNK;
            HIPCHK(c, hipMalloc(&h.qW, nPosPad * rs));
            HIPCHK(c, hipMalloc(&h.qF, nPosPad * 4 * rs));
            HIPCHK(c, hipMalloc((void **) &h.qC, nPosPad));
            HIPCHK(c, hipMalloc((void **) &h.qEnc, (size_t) c->d.nChunk * sizeof(uint4)));
            HIPCHK(c, hipMalloc((void **) &h.qTab, nPosPad * 5 * sizeof(double2)));
        }
        c->slots.push_back(h);
        added = true;
    }
    // The hipMemsets of the new slots run on the NULL stream; the context's stream is non-blocking, so nothing orders them before the
    // sweep that is about to write into those arrays.  (Found in round 6 under `rocprofv3 --pmc`, whose serialised kernels let a
    // late memset zero a slot's criteria AFTER its first sweep: "more than 8192 hits tied at the k-th criterion".)
    if (added) HIPCHK(c, hipDeviceSynchronize());
    return VFT_OK;
}

// waits until the host headers of slots [s0, s0 + count) carry `seq` (written by the last workgroup of k_select_rank); bounded like
// wait_flag: VFT_ERR_TIMEOUT when the stream has drained without them, or after waitLimitS
static int wait_headers(vft_ctx *c, int s0, int count, unsigned long long seq) {
    if (c->faultNoFlag) {   // test hook: wait for a value nobody will ever publish
        seq += 1ull << 40;
        c->faultNoFlag = false;
    }
    auto raised = [&]() {
        for (int s = s0; s < s0 + count; s++)
            if ((unsigned long long) __atomic_load_n(&((const SelectHeader *) c->slots[(size_t) s].hRes)->pad2, __ATOMIC_ACQUIRE) != seq) return false;
        return true;
    };
    std::chrono::steady_clock::time_point t0;
    for (long spins = 0;; spins++) {
        if (raised()) return VFT_OK;
        if (spins == 200000) t0 = std::chrono::steady_clock::now();
        if (spins >= 200000 && (spins & 0xFFFF) == 0) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {
                if (raised()) return VFT_OK;
                return fail(c, VFT_ERR_TIMEOUT, "the stream has drained but a selection never published its result");
            }
            if (e != hipErrorNotReady) return fail(c, VFT_ERR_HIP, "stream error while waiting: %s", hipGetErrorString(e));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->waitLimitS)
                return fail(c, VFT_ERR_TIMEOUT, "no selection result after %.0f s (the stream is still busy)", c->waitLimitS);
        }
    }
}

// Top-k selection of K seeds at once (their sweep results sit in slots 0..K-1): one set of launches, blockIdx.y =
// seed, one host synchronisation.  A seed whose threshold digit overflows the candidate buffer (rare) is finished
// alone, one digit deeper, exactly as a single selection would.
template <typename REAL, typename HIT>
static int run_select(vft_ctx *c, int K, const int64_t *queries, int64_t lo, int64_t hi, int32_t k) {
    std::vector<SelSlot> hs((size_t) K);
    for (int s = 0; s < K; s++) {
        const vft_ctx::SweepSlotHost &h = c->slots[(size_t) s];
        SelSlot &d = hs[(size_t) s];
        d.crit = h.swCrit;
        d.dist = h.swDist;
        d.weight = h.swWeight;
        d.partMin = h.partMin;
        d.partMax = h.partMax;
        d.sel = h.sel;
        d.slices = h.slices;
        d.candKey = h.candKey;
        d.candId = h.candId;
        d.hits = h.dRes + sizeof(SelectHeader);
        d.hdr = (SelectHeader *) h.dRes;
        d.hostHdr = (SelectHeader *) h.hResDev;
        d.hostHits = h.hResDev + sizeof(SelectHeader);
        d.query = queries[s];
        d.nPart = h.nPart;
        d.pad = 0;
    }
    char *hS, *dS;
    if (int r = io_alloc(c, hs.size() * sizeof(SelSlot), &hS, &dS)) return r;
    memcpy(hS, hs.data(), hs.size() * sizeof(SelSlot));
    const SelSlot *slots = (const SelSlot *) dS;
    // hist (round one: with the key range) -> collect (every workgroup finds the threshold digit in the seed's histogram itself) -> rank; the rank sort's last workgroup of every seed finishes the
    // selection and writes `seq` into the seed's host header - the host waits for those words, no kernel behind the rank sort
    auto round = [&](const SelSlot *sl, unsigned ny, int first, int s0) -> int {
        const unsigned long long seq = ++c->signalSeq;
        launch((k_select_hist<REAL>), dim3(VFT_SEL_WGS, ny), dim3(VFT_WG), 0, c->stream, sl, lo, hi, first);
        launch((k_select_collect<REAL>), dim3(VFT_SEL_WGS, ny), dim3(VFT_WG), 0, c->stream, sl, lo, hi, (unsigned int) k);
        launch((k_select_rank<REAL, HIT>), dim3(VFT_CAND_CAP * VFT_RANK_LANES / VFT_WG, ny), dim3(VFT_WG), 0, c->stream, sl, k, lo, hi, (long long) seq);
        LAUNCHCHK(c);
        return wait_headers(c, s0, (int) ny, seq);
    };
    if (int r = round(slots, (unsigned) K, 1, 0)) return r;
    for (int s = 0; s < K; s++) {
        const SelectHeader *h = (const SelectHeader *) c->slots[(size_t) s].hRes;
        // rare: the threshold bin alone has more candidates than the rank sort takes; narrow the key range to it
        for (int level = 0; h->overflow; level++) {
            if (level == VFT_MAX_LEVEL)
                return fail(c, VFT_ERR_STATE, "top-k select: more than %d hits tied at the k-th criterion", VFT_CAND_CAP);
            launch(k_select_refine, dim3(1, 1), dim3(1), 0, c->stream, slots + s);
            if (int r = round(slots + s, 1u, 0, s)) return r;
        }
    }
    return VFT_OK;
}

// the lazy refresh, query staging and sweep kernels of ONE seed; its results go to the buffers of `slot`
static int sweep_one(vft_ctx *c, int slot, int64_t query, int64_t nActive, int64_t nDiffAllow, double totdiam, bool staged = false) {
    if (int r = flush_pending(c)) return r;
    const int64_t lo = c->shardLo, hi = c->shardHi < c->maxnode ? c->shardHi : c->maxnode;
    // 1. lazy out-distance refresh of every stale active target and of the query (NJ.tcc:1092-1098) - skipped when the
    //    host's bookkeeping proves that nothing can be stale (seed after seed of setAllLeafTopHits, for instance)
    if (c->maxStamp - nActive > nDiffAllow)
        if (int r = launch_out_distances(c, nullptr, 0, nActive, nDiffAllow, totdiam, false)) return r;
    if ((int64_t) c->hNOut[query] - nActive > nDiffAllow) {   // the mirror can only lag towards "staler"
        if (nActive > c->maxStamp) c->maxStamp = nActive;     // the refresh stamps the query with nActive
        SweepArgs s{};
        s.nActive = nActive;
        s.nDiffAllow = nDiffAllow;
        s.totdiam = totdiam;
        s.force = 0;
        VFT_DISPATCH(c, launch((k_out_distance_one<REAL, NC>), dim3(1), dim3(VFT_WG), pw_lds_bytes(c) / c->pwWaves, c->stream,
                               arena<REAL>(c), query, s));
        LAUNCHCHK(c);
    }
    // 2. the sweep itself
    SweepArgs s{};
    s.query = query;
    s.lo = lo;
    s.hi = hi;
    s.nActive = nActive;
    s.nDiffAllow = nDiffAllow;
    s.totdiam = totdiam;
    s.queryIsLeaf = query < c->d.nSeqs ? 1 : 0;
#ifdef VFT_ABLATE   // tools-only build (tools/ablate_sweep.py): never in the product library
    if (const char *dbg = getenv("VFT_SWEEP_ABLATE")) s.pad = atoi(dbg);
#endif
    const int64_t span = hi > lo ? hi - lo : 0;
    const bool ntPath = c->cfg.n_codes == 4 && !c->hasDm;
    const unsigned grid = ntPath ? sweep_nt_grid(c, s, !s.queryIsLeaf) : cdiv(span > 0 ? span : 1, VFT_WG);
    c->nPart = (int) grid;
    c->slots[(size_t) slot].nPart = (int) grid;
    const int64_t nPosPad = (int64_t) c->d.nChunk * VFT_CHUNK;
    if (c->cfg.n_codes == 4 && !c->hasDm) {
        // (staged: vft_sweep_batch has extracted the queries of all its seeds in one launch, each into its slot's buffers)
        if (c->cfg.precision == 4) {
            const QueryBuf<float> Q = staged ? qbuf_slot<float>(c, slot) : qbuf<float>(c, 0);
            if (!staged) launch((k_extract_query<float, 4>), dim3(cdiv(nPosPad, 256)), dim3(256), 0, c->stream, arena<float>(c), query, Q);
            if (s.queryIsLeaf) launch_sweep_nt<float, MODE_CRIT_LEAFQ>(c, s, grid, Q, true, slot);
            else launch_sweep_nt<float, MODE_CRIT>(c, s, grid, Q, true, slot);
        } else {
            const QueryBuf<double> Q = staged ? qbuf_slot<double>(c, slot) : qbuf<double>(c, 0);
            if (!staged) launch((k_extract_query<double, 4>), dim3(cdiv(nPosPad, 256)), dim3(256), 0, c->stream, arena<double>(c), query, Q);
            if (s.queryIsLeaf) launch_sweep_nt<double, MODE_CRIT_LEAFQ>(c, s, grid, Q, true, slot);
            else launch_sweep_nt<double, MODE_CRIT>(c, s, grid, Q, true, slot);
        }
    } else {
        // amino acids / distance matrix.  Measured on C5 (50k x 300, f64): lane-per-target costs ~nPos dependent
        // latencies whatever the target count (~0.9 ms), wave-per-target ~50 us per 2048 targets: the crossover is
        // at ~32k targets.  (Neither is tuned: the aa arena wants its own kernel)
        kernel_event(c);
        if (c->cfg.n_codes == 20 && c->hasDm && c->aaLds && span > 0) {
            const unsigned g = c->cfg.precision == 4 ? launch_sweep_aa<float, MODE_CRIT>(c, s, 0, slot) : launch_sweep_aa<double, MODE_CRIT>(c, s, 0, slot);
            c->nPart = (int) g;
            c->slots[(size_t) slot].nPart = (int) g;
        } else if (span <= 32768) {
            const unsigned wgrid = (unsigned) std::min<int64_t>(cdiv(span > 0 ? span : 1, c->pwWaves), 256 * 16);
            c->nPart = (int) wgrid;
            c->slots[(size_t) slot].nPart = (int) wgrid;
            VFT_DISPATCH(c, (launch((k_sweep_wave<REAL, NC>), dim3(wgrid), dim3(64 * c->pwWaves), pw_lds_bytes(c), c->stream,
                                    arena<REAL>(c), s, sweepout<REAL>(c, slot))));
        } else {
            VFT_DISPATCH(c, (launch((k_sweep_generic<REAL, NC>), dim3(grid), dim3(VFT_WG), 0, c->stream,
                                    arena<REAL>(c), s, sweepout<REAL>(c, slot))));
        }
        kernel_event(c);
        kernel_event(c);
    }
    LAUNCHCHK(c);
    return VFT_OK;
}

// S leaf seeds - or S profile seeds - of a batch (the seeds at positions slotOf[0 .. S-1] of it) in ONE pass over the targets
// (k_sweep_nt_leafq_multi / k_sweep_nt_profq_multi): results in the buffers of those slots, bit for bit those of S sweep_one calls.  The caller has checked that no lazy refresh is due (sweep_one's step 1) and has staged the queries.
template <typename REAL, int S>
static int sweep_group(vft_ctx *c, bool leafSeeds, const int *slotOf, const int64_t *queries, int64_t nActive, int64_t nDiffAllow, double totdiam, const void *mq) {
    SweepArgs s{};
    s.query = queries[slotOf[0]];
    s.lo = c->shardLo;
    s.hi = c->shardHi < c->maxnode ? c->shardHi : c->maxnode;
    s.nActive = nActive;
    s.nDiffAllow = nDiffAllow;
    s.totdiam = totdiam;
    s.queryIsLeaf = leafSeeds ? 1 : 0;
    const unsigned grid = sweep_nt_grid(c, s, !leafSeeds);   // (= the workgroups, and min / max partials, of ONE seed's sweep)
    MultiLeafQ<REAL, S> M;
    for (int q = 0; q < S; q++) {
        M.Q[q] = qbuf_slot<REAL>(c, slotOf[q]);
        M.O[q] = sweepout<REAL>(c, slotOf[q]);
        M.query[q] = queries[slotOf[q]];
        c->slots[(size_t) slotOf[q]].nPart = (int) grid;
    }
    M.mq = (const REAL *) mq;
    c->nPart = (int) grid;
    kernel_event(c);
    if (grid && leafSeeds) launch((k_sweep_nt_leafq_multi<REAL, S>), dim3(grid), dim3(VFT_WG), 0, c->stream, arena<REAL>(c), M, s);
    // profile seeds: heavy and table workgroups alike take all S seeds
    if (grid && !leafSeeds) launch((k_sweep_nt_profq_multi<REAL, S>), dim3(grid), dim3(VFT_WG), 0, c->stream, arena<REAL>(c), M, s);
    kernel_event(c);
    kernel_event(c);
    if (c->timeKernels) c->kevSweeps += S - 1;   // (every triple of events counts as one sweep; this one covers S)
    LAUNCHCHK(c);
    return VFT_OK;
}

static int sweep_args_ok(vft_ctx *c, int64_t query, int64_t nActive, int32_t k) {
    if (!c->leavesUp) return fail(c, VFT_ERR_STATE, "vft_sweep before vft_upload_leaves");
    if (query < 0 || query >= c->maxnode || nActive < 3 || k < 0 || k > c->hitsCap)
        return fail(c, VFT_ERR_INVALID, "vft_sweep: bad arguments (query %lld, nActive %lld, k %d)", (long long) query,
                    (long long) nActive, (int) k);
    return VFT_OK;
}

extern "C" int vft_sweep(vft_ctx *c, int64_t query, int64_t nActive, int64_t nDiffAllow, double totdiam, int32_t k,
                         void *hits, void *dHitsOut, int64_t *bestJ) {
    if (!c) return VFT_ERR_INVALID;
    if (int r = sweep_args_ok(c, query, nActive, k)) return r;
    if (int r = sweep_one(c, 0, query, nActive, nDiffAllow, totdiam)) return r;
    if (k == 0) return VFT_OK;
    // top-k in the reference's sort order; header + hits come back through the host-mapped block
    if (bestJ && k < 2) return fail(c, VFT_ERR_INVALID, "vft_sweep: best_j needs k >= 2");
    const int64_t lo = c->shardLo, hi = c->shardHi < c->maxnode ? c->shardHi : c->maxnode;
    int r;
    if (c->cfg.precision == 4) r = run_select<float, vft_hit_f32>(c, 1, &query, lo, hi, k);
    else r = run_select<double, vft_hit_f64>(c, 1, &query, lo, hi, k);
    if (r) return r;
    const size_t hb = (size_t) k * (c->cfg.precision == 4 ? sizeof(vft_hit_f32) : sizeof(vft_hit_f64));
    if (dHitsOut) HIPCHK(c, hipMemcpyAsync(dHitsOut, c->dRes + sizeof(SelectHeader), hb, hipMemcpyDeviceToDevice, c->stream));
    if (hits) memcpy(hits, c->hRes + sizeof(SelectHeader), hb);
    if (bestJ) *bestJ = (int64_t) ((const SelectHeader *) c->hRes)->bestJ;
    return VFT_OK;
}

// Several seeds per call (setAllLeafTopHits sweeps seed after seed without anything changing in between, so a driver
// can run the next unvisited seeds speculatively; a multi-GPU run then needs one exchange per batch instead of one per
// seed - SURVEY section 8e).  The same kernels as vft_sweep, seed after seed on the stream, then ONE batched top-k
// selection and ONE host synchronisation.  Results are exactly those of n_seeds vft_sweep calls.
// the seeds of a batch that share a pass over the targets: `n` batch positions (2 or 4) of one kind
struct SeedGroup {
    int pos[4];
    int n;
    bool leaf;
};

extern "C" int vft_sweep_batch(vft_ctx *c, int32_t nSeeds, const int64_t *queries, int64_t nActive, int64_t nDiffAllow,
                               double totdiam, int32_t k, void *hits, void *dHitsOut, int64_t *bestJ) {
    if (!c || !queries || nSeeds < 1 || nSeeds > 64 || k < 1) return VFT_ERR_INVALID;
    if (bestJ && k < 2) return fail(c, VFT_ERR_INVALID, "vft_sweep_batch: best_j needs k >= 2");
    for (int s = 0; s < nSeeds; s++)
        if (int r = sweep_args_ok(c, queries[s], nActive, k)) return r;
    if (int r = ensure_slots(c, nSeeds)) return r;
    // nucleotides without a distance matrix: the queries of all seeds are staged by ONE launch, each into its slot's buffers
    const bool staged = c->cfg.n_codes == 4 && !c->hasDm && nSeeds > 1;
    // The leaf seeds of the batch share passes over the targets, four (or two) per launch, and so do its profile seeds - every seed has
    // buffers of its own, so the order the sweeps run in does not matter as long as no lazy refresh is due (sweep_one's step 1 changes out-distances that later
    // sweeps read: then the seeds go one by one, in order).  The groups are made before the staging launch: the profile seeds of a
    // group are also staged into the group's interleaved buffer (QuerySlot::mq).
    std::vector<SeedGroup> groups;
    if (staged && !c->noMultiSweep && !(c->maxStamp - nActive > nDiffAllow)) {
        bool fresh = true;
        for (int s = 0; s < nSeeds && fresh; s++) fresh = !((int64_t) c->hNOut[queries[s]] - nActive > nDiffAllow);
        const int maxGroup = c->multiMax ? c->multiMax : 4;
        for (int kind = 0; kind < 2 && fresh; kind++) {   // the leaf seeds, then the profile seeds: fours while they last, then a pair; a last single goes alone
            SeedGroup g{};
            g.leaf = kind == 0;
            int nLeft = 0;
            auto isKind = [&](int s) { return (queries[s] < c->d.nSeqs) == (kind == 0); };
            for (int s = 0; s < nSeeds; s++) nLeft += isKind(s);
            for (int s = 0; s < nSeeds; s++) {
                if (!isKind(s)) continue;
                g.pos[g.n++] = s;
                nLeft--;
                if (g.n == maxGroup || (g.n == 2 && nLeft < 2)) {
                    groups.push_back(g);
                    g.n = 0;
                }
            }
        }
    }
    const int64_t nPosPad = (int64_t) c->d.nChunk * VFT_CHUNK;
    const size_t mqGroupBytes = (size_t) nPosPad * VFT_MQ_STRIDE(4) * c->rs;
    if (!groups.empty() && !c->mqBuf) {   // (32 groups: the profile seeds of the largest batch, two per group)
        HIPCHK(c, hipMalloc(&c->mqBuf, 32 * mqGroupBytes));
        c->mqGroupBytes = mqGroupBytes;
    }
    if (staged) {
        if (int r = flush_pending(c)) return r;
        // where a seed's query goes in its group's interleaved buffer (profile-seed groups only)
        std::vector<int> mqGroup((size_t) nSeeds, -1), mqIdx((size_t) nSeeds, 0), mqS((size_t) nSeeds, 0);
        for (size_t g = 0; g < groups.size(); g++)
            if (!groups[g].leaf)
                for (int q = 0; q < groups[g].n; q++) {
                    mqGroup[(size_t) groups[g].pos[q]] = (int) g;
                    mqIdx[(size_t) groups[g].pos[q]] = q;
                    mqS[(size_t) groups[g].pos[q]] = groups[g].n;
                }
        char *hQ, *dQ;
        if (c->cfg.precision == 4) {
            if (int r = io_alloc(c, (size_t) nSeeds * sizeof(QuerySlot<float>), &hQ, &dQ)) return r;
            for (int s = 0; s < nSeeds; s++) {
                QuerySlot<float> qs;
                qs.node = queries[s];
                qs.q = qbuf_slot<float>(c, s);
                qs.mq = mqGroup[(size_t) s] >= 0 ? (float *) ((char *) c->mqBuf + (size_t) mqGroup[(size_t) s] * c->mqGroupBytes) : nullptr;
                qs.mqIdx = mqIdx[(size_t) s];
                qs.mqS = mqS[(size_t) s];
                memcpy(hQ + (size_t) s * sizeof(qs), &qs, sizeof(qs));
            }
            launch((k_extract_query_batch<float, 4>), dim3(cdiv(nPosPad, 256), (unsigned) nSeeds), dim3(256), 0, c->stream, arena<float>(c), (const QuerySlot<float> *) dQ);
        } else {
            if (int r = io_alloc(c, (size_t) nSeeds * sizeof(QuerySlot<double>), &hQ, &dQ)) return r;
            for (int s = 0; s < nSeeds; s++) {
                QuerySlot<double> qs;
                qs.node = queries[s];
                qs.q = qbuf_slot<double>(c, s);
                qs.mq = mqGroup[(size_t) s] >= 0 ? (double *) ((char *) c->mqBuf + (size_t) mqGroup[(size_t) s] * c->mqGroupBytes) : nullptr;
                qs.mqIdx = mqIdx[(size_t) s];
                qs.mqS = mqS[(size_t) s];
                memcpy(hQ + (size_t) s * sizeof(qs), &qs, sizeof(qs));
            }
            launch((k_extract_query_batch<double, 4>), dim3(cdiv(nPosPad, 256), (unsigned) nSeeds), dim3(256), 0, c->stream, arena<double>(c), (const QuerySlot<double> *) dQ);
        }
        LAUNCHCHK(c);
    }
    std::vector<char> done((size_t) nSeeds, 0);
    for (size_t g = 0; g < groups.size(); g++) {
        const SeedGroup &G = groups[g];
        const void *mq = G.leaf ? nullptr : (const void *) ((char *) c->mqBuf + g * c->mqGroupBytes);
        int r;
        if (c->cfg.precision == 4) r = G.n == 4 ? sweep_group<float, 4>(c, G.leaf, G.pos, queries, nActive, nDiffAllow, totdiam, mq) : sweep_group<float, 2>(c, G.leaf, G.pos, queries, nActive, nDiffAllow, totdiam, mq);
        else r = G.n == 4 ? sweep_group<double, 4>(c, G.leaf, G.pos, queries, nActive, nDiffAllow, totdiam, mq) : sweep_group<double, 2>(c, G.leaf, G.pos, queries, nActive, nDiffAllow, totdiam, mq);
        if (r) return r;
        for (int q = 0; q < G.n; q++) done[(size_t) G.pos[q]] = 1;
    }
    for (int s = 0; s < nSeeds; s++)
        if (!done[(size_t) s])
            if (int r = sweep_one(c, s, queries[s], nActive, nDiffAllow, totdiam, staged)) return r;
    const int64_t lo = c->shardLo, hi = c->shardHi < c->maxnode ? c->shardHi : c->maxnode;
    int r;
    if (c->cfg.precision == 4) r = run_select<float, vft_hit_f32>(c, nSeeds, queries, lo, hi, k);
    else r = run_select<double, vft_hit_f64>(c, nSeeds, queries, lo, hi, k);
    if (r) return r;
    const size_t hb = (size_t) k * (c->cfg.precision == 4 ? sizeof(vft_hit_f32) : sizeof(vft_hit_f64));
    for (int s = 0; s < nSeeds; s++) {
        const vft_ctx::SweepSlotHost &h = c->slots[(size_t) s];
        if (dHitsOut)
            HIPCHK(c, hipMemcpyAsync((char *) dHitsOut + (size_t) s * hb, h.dRes + sizeof(SelectHeader), hb, hipMemcpyDeviceToDevice, c->stream));
        if (hits) memcpy((char *) hits + (size_t) s * hb, h.hRes + sizeof(SelectHeader), hb);
        if (bestJ) bestJ[s] = (int64_t) ((const SelectHeader *) h.hRes)->bestJ;
    }
    return VFT_OK;
}

// the k records of slot `slot`'s last selection where the device left them: the host-mapped result block (no copy); valid until the
// context's next sweep
extern "C" int vft_sweep_batch_view(vft_ctx *c, int32_t slot, const void **hits, int64_t *bestJ) {
    if (!c || slot < 0 || (size_t) slot >= c->slots.size() || !hits) return VFT_ERR_INVALID;
    const vft_ctx::SweepSlotHost &h = c->slots[(size_t) slot];
    *hits = h.hRes + sizeof(SelectHeader);
    if (bestJ) *bestJ = (int64_t) ((const SelectHeader *) h.hRes)->bestJ;
    return VFT_OK;
}

// nSeeds == 1: the single-list entry point (results through slot 0's blocks); nSeeds > 1: [lists][nSeeds][k] in,
// [nSeeds][k] out through the batch blocks
template <typename REAL, typename HIT>
static int merge_hits_impl(vft_ctx *c, const void *dAll, int32_t nLists, int32_t nSeeds, int32_t k, void *hits, void *dOut) {
    const int32_t n = nLists * k;
    const size_t total = (size_t) nSeeds * k;
    HIT *dHits, *hHitsDev;
    const char *hHost;
    if (nSeeds == 1) {
        dHits = (HIT *) (c->dRes + sizeof(SelectHeader));
        hHitsDev = (HIT *) (c->hResDev + sizeof(SelectHeader));
        hHost = c->hRes + sizeof(SelectHeader);
    } else {
        const size_t need = total * sizeof(vft_hit_f64);
        if (need > c->mergeBytes) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->dMerge) hipFree(c->dMerge);
            if (c->hMerge) hipHostFree(c->hMerge);
            c->dMerge = c->hMerge = c->hMergeDev = nullptr;
            c->mergeBytes = 0;
            HIPCHK(c, hipMalloc((void **) &c->dMerge, need));
            HIPCHK(c, hipHostMalloc((void **) &c->hMerge, need, hipHostMallocMapped));
            HIPCHK(c, hipHostGetDevicePointer((void **) &c->hMergeDev, c->hMerge, 0));
            c->mergeBytes = need;
        }
        dHits = (HIT *) c->dMerge;
        hHitsDev = (HIT *) c->hMergeDev;
        hHost = c->hMerge;
    }
    launch((k_fill_empty_hits<REAL, HIT>), dim3(cdiv((int64_t) total, 256)), dim3(256), 0, c->stream, dHits, hHitsDev, (int32_t) total);
    launch((k_merge_hits<REAL, HIT>), dim3(cdiv((int64_t) n * VFT_RANK_LANES, VFT_WG), (unsigned) nSeeds), dim3(VFT_WG), 0,
           c->stream, (const HIT *) dAll, n, k, dHits, hHitsDev, nSeeds);
    LAUNCHCHK(c);
    if (dOut) HIPCHK(c, hipMemcpyAsync(dOut, dHits, total * sizeof(HIT), hipMemcpyDeviceToDevice, c->stream));
    if (hits) {
        if (int r = wait_stream(c)) return r;
        memcpy(hits, hHost, total * sizeof(HIT));
    }
    return VFT_OK;
}

extern "C" int vft_merge_hits(vft_ctx *c, const void *dAll, int32_t nLists, int32_t k, void *hits, void *dOut) {
    if (!c || !dAll || nLists < 1 || k < 1 || k > c->hitsCap) return fail(c, VFT_ERR_INVALID, "vft_merge_hits: bad arguments");
    if (c->cfg.precision == 4) return merge_hits_impl<float, vft_hit_f32>(c, dAll, nLists, 1, k, hits, dOut);
    return merge_hits_impl<double, vft_hit_f64>(c, dAll, nLists, 1, k, hits, dOut);
}

extern "C" int vft_merge_hits_batch(vft_ctx *c, const void *dAll, int32_t nLists, int32_t nSeeds, int32_t k, void *hits,
                                    void *dOut) {
    if (!c || !dAll || nLists < 1 || nSeeds < 1 || nSeeds > 64 || k < 1 || k > c->hitsCap)
        return fail(c, VFT_ERR_INVALID, "vft_merge_hits_batch: bad arguments");
    if (c->cfg.precision == 4) return merge_hits_impl<float, vft_hit_f32>(c, dAll, nLists, nSeeds, k, hits, dOut);
    return merge_hits_impl<double, vft_hit_f64>(c, dAll, nLists, nSeeds, k, hits, dOut);
}

extern "C" int vft_sweep_info(vft_ctx *c, int64_t info[2]) {
    if (!c || !info) return VFT_ERR_INVALID;
    const SelectHeader *h = (const SelectHeader *) c->hRes;
    info[0] = h->nCand;
    info[1] = h->shift;
    return VFT_OK;
}

// the same two numbers for slot `slot` of the last vft_sweep_batch (diagnostics: how many candidates its selection ranked)
extern "C" int vft_sweep_batch_info(vft_ctx *c, int32_t slot, int64_t info[2]) {
    if (!c || !info || slot < 0 || (size_t) slot >= c->slots.size()) return VFT_ERR_INVALID;
    const SelectHeader *h = (const SelectHeader *) c->slots[(size_t) slot].hRes;
    info[0] = h->nCand;
    info[1] = h->shift;
    return VFT_OK;
}

extern "C" int vft_sweep_results(vft_ctx *c, int64_t first, int64_t count, void *dist, void *weight, void *crit) {
    if (!c) return VFT_ERR_INVALID;
    if (int r = range_ok(c, first, count)) return r;
    const size_t rs = c->rs;
    if (dist) HIPCHK(c, hipMemcpyAsync(dist, (char *) c->swDist + first * rs, count * rs, hipMemcpyDeviceToHost, c->stream));
    if (weight) HIPCHK(c, hipMemcpyAsync(weight, (char *) c->swWeight + first * rs, count * rs, hipMemcpyDeviceToHost, c->stream));
    if (crit) HIPCHK(c, hipMemcpyAsync(crit, (char *) c->swCrit + first * rs, count * rs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// two ranges of 16-byte pieces in one launch (host-mapped memory -> device memory)
static __global__ void k_copy16x2(uint4 *dstA, const uint4 *srcA, int64_t nA, uint4 *dstB, const uint4 *srcB, int64_t nB) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nA) dstA[t] = srcA[t];
    else if (t < nA + nB) dstB[t - nA] = srcB[t - nA];
}

static int pair_distances(vft_ctx *c, int64_t n, const int64_t *pi, const int64_t *pj, int64_t nActive, int64_t nDiffAllow,
                          double totdiam, void *dist, void *weight, void *crit, bool raw, int64_t nForce, const int64_t *forceIds);
extern "C" int vft_pair_distances(vft_ctx *c, int64_t n, const int64_t *pi, const int64_t *pj, int64_t nActive,
                                  int64_t nDiffAllow, double totdiam, void *dist, void *weight, void *crit) {
    return pair_distances(c, n, pi, pj, nActive, nDiffAllow, totdiam, dist, weight, crit, false, 0, nullptr);
}
extern "C" int vft_pair_distances_refresh(vft_ctx *c, int64_t n, const int64_t *pi, const int64_t *pj, int64_t nForce,
                                          const int64_t *forceIds, int64_t nActive, int64_t nDiffAllow, double totdiam,
                                          void *dist, void *weight, void *crit) {
    if (nForce < 0 || (nForce > 0 && !forceIds)) return VFT_ERR_INVALID;
    if (n == 0 && nForce > 0) return vft_out_distances(c, nForce, forceIds, nActive, totdiam);
    return pair_distances(c, n, pi, pj, nActive, nDiffAllow, totdiam, dist, weight, crit, false, nForce, forceIds);
}
extern "C" int vft_profile_distances(vft_ctx *c, int64_t n, const int64_t *pi, const int64_t *pj, void *dist, void *weight) {
    return pair_distances(c, n, pi, pj, 3, 0, 0.0, dist, weight, nullptr, true, 0, nullptr);
}
static int pair_distances(vft_ctx *c, int64_t n, const int64_t *pi, const int64_t *pj, int64_t nActive, int64_t nDiffAllow,
                          double totdiam, void *dist, void *weight, void *crit, bool raw, int64_t nForce, const int64_t *forceIds) {
    if (!c || n < 0 || !pi || !pj) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    static const bool smallTimes = getenv("VFT_API_PROFILE") != nullptr;   // tools only: where a short list's time goes
    static double tsPrep = 0, tsLaunch = 0, tsWait = 0, tsOut = 0;
    static int64_t tsCalls = 0, tsPairs = 0;
    std::chrono::steady_clock::time_point ts0, ts1, ts2, ts3;
    if (smallTimes) ts0 = std::chrono::steady_clock::now();
    for (int64_t t = 0; t < nForce; t++)
        if (forceIds[t] < 0 || forceIds[t] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "forced node %lld out of range", (long long) t);
    if (!c->leavesUp) return fail(c, VFT_ERR_STATE, "vft_pair_distances before vft_upload_leaves");
    for (int64_t t = 0; t < n; t++)
        if (pi[t] < 0 || pi[t] >= c->maxnode || pj[t] < 0 || pj[t] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "pair %lld out of range", (long long) t);
    const size_t rs = c->rs, idB = (((size_t) n * 8) + 255) & ~(size_t) 255, oB = (((size_t) n * rs) + 255) & ~(size_t) 255;
    // setCriterion's lazy refresh (NJ.tcc:1092-1098) for the nodes the list names: the DISTINCT stale ones, found on the
    // host-mapped stamp mirror (it can only lag towards "staler": kernels in flight make nodes fresher, host-side sets
    // update it at once; the kernel looks at the real stamp again).  A node is typically named by hundreds of pairs
    // (the new node of a join by all of its candidates): one workgroup / wave per NAME recomputed the same value that
    // many times - 0.85 ms of a 0.94 ms call at 60 000 pairs.
    std::vector<int64_t> &stale = c->staleIds;
    stale.clear();
    int64_t nForced = 0;
    if (!raw) {
        if (c->staleMark.size() != (size_t) c->d.maxNodes) c->staleMark.assign((size_t) c->d.maxNodes, 0u);
        if (++c->staleEpoch == 0u) {
            std::fill(c->staleMark.begin(), c->staleMark.end(), 0u);
            c->staleEpoch = 1u;
        }
        const uint32_t ep = c->staleEpoch;
        uint32_t *mark = c->staleMark.data();
        // forced refreshes first (setOutDistance, NJ.tcc:1012-1015: recomputed unless the stamp IS n_active); the kernel
        // applies that rule to the first nForced entries of the list and the lazy one to the rest
        if (c->staleIdx.size() != (size_t) c->d.maxNodes) c->staleIdx.assign((size_t) c->d.maxNodes, -1);
        int32_t *sidx = c->staleIdx.data();   // position in the list, valid where mark == ep
        for (int64_t t = 0; t < nForce; t++) {
            const int64_t v = forceIds[t];
            if ((int64_t) c->hNOut[v] != nActive && mark[v] != ep) {
                mark[v] = ep;
                sidx[v] = (int32_t) stale.size();
                stale.push_back(v);
            }
        }
        nForced = (int64_t) stale.size();
        for (int64_t t = 0; t < n; t++) {
            const int64_t a = pi[t], b = pj[t];
            if ((int64_t) c->hNOut[a] - nActive > nDiffAllow && mark[a] != ep) {
                mark[a] = ep;
                sidx[a] = (int32_t) stale.size();
                stale.push_back(a);
            }
            if ((int64_t) c->hNOut[b] - nActive > nDiffAllow && mark[b] != ep) {
                mark[b] = ep;
                sidx[b] = (int32_t) stale.size();
                stale.push_back(b);
            }
        }
    }
    const int64_t nStale = (int64_t) stale.size();
    const size_t sB = (((size_t) nStale * 8) + 255) & ~(size_t) 255;
    const size_t wB = (((size_t) n * 8) + 255) & ~(size_t) 255;   // per-pair wait indices of the single-launch variant
    const bool small = 2 * idB + 3 * oB + sB + wB <= VFT_SMALL_BYTES;
    // (the pair workgroups of the single-launch variant spin on tags of refresh workgroups of the same grid: only while the
    //  whole grid is resident at once, whatever order the dispatcher picks)
    bool fusedRefresh = small && nStale > 0 && n <= 2048 && n + nStale <= c->fusedLimit && !c->noFusedRefresh;
    if (fusedRefresh && !c->refDone) {
        HIPCHK(c, hipMalloc((void **) &c->refDone, 4096 * sizeof(unsigned int)));
        HIPCHK(c, hipMemsetAsync(c->refDone, 0, 4096 * sizeof(unsigned int), c->stream));
    }
    char *hBase = nullptr, *dBase = nullptr;
    if (small) {
        if (int r = io_alloc(c, 2 * idB + 3 * oB + sB + wB, &hBase, &dBase)) return r;
        memcpy(hBase, pi, (size_t) n * 8);
        memcpy(hBase + idB, pj, (size_t) n * 8);
        if (nStale) memcpy(hBase + 2 * idB + 3 * oB, stale.data(), (size_t) nStale * 8);
        if (fusedRefresh) {   // which refresh workgroup each end of a pair has to wait for (-1: none)
            int32_t *wt = (int32_t *) (hBase + 2 * idB + 3 * oB + sB);
            const uint32_t ep = c->staleEpoch;
            const uint32_t *mark = c->staleMark.data();
            const int32_t *sidx = c->staleIdx.data();
            for (int64_t t = 0; t < n; t++) {
                wt[2 * t] = mark[pi[t]] == ep ? sidx[pi[t]] : -1;
                wt[2 * t + 1] = mark[pj[t]] == ep ? sidx[pj[t]] : -1;
            }
        }
    } else {
        if (int r = ensure_scratch(c, 2 * idB + 3 * oB + sB + 64)) return r;
        dBase = (char *) c->scratch;
        HIPCHK(c, hipMemcpyAsync(dBase, pi, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dBase + idB, pj, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
        if (nStale)
            HIPCHK(c, hipMemcpyAsync(dBase + 2 * idB + 3 * oB, stale.data(), (size_t) nStale * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (smallTimes) ts1 = std::chrono::steady_clock::now();
    int64_t *dI = (int64_t *) dBase, *dJ = (int64_t *) (dBase + idB);
    const int64_t *dStale = (const int64_t *) (dBase + 2 * idB + 3 * oB);
    const int32_t *dWait = (const int32_t *) (dBase + 2 * idB + 3 * oB + sB);
    char *o = dBase + 2 * idB;
    if (small && n >= 128 && !c->noPairStaging) {
        // One workgroup per pair reading ITS ids from the host-mapped ring is two 8-byte PCIe reads per pair - 4 000 tiny
        // non-posted reads at 2 000 pairs, ~35 ns per pair of a call that otherwise takes ~25 us.  A copy kernel fetches the
        // inputs with 16-byte-per-lane loads first; the pair kernels then read device memory.
        char *in = (char *) c->pairIn;
        const size_t nA = 2 * idB / 16, nB = (sB + wB) / 16;
        launch(k_copy16x2, dim3(cdiv((int64_t) (nA + nB), 256)), dim3(256), 0, c->stream, (uint4 *) in, (const uint4 *) dBase, (int64_t) nA,
               (uint4 *) (in + 2 * idB), (const uint4 *) (dBase + 2 * idB + 3 * oB), (int64_t) nB);
        dI = (int64_t *) in;
        dJ = (int64_t *) (in + idB);
        dStale = (const int64_t *) (in + 2 * idB);
        dWait = (const int32_t *) (in + 2 * idB + sB);
    }
    SweepArgs sa{};
    sa.nActive = nActive;
    sa.nDiffAllow = nDiffAllow;
    sa.totdiam = totdiam;
    sa.force = raw ? 1 : 0;
    // two launches: lazy refresh of the stale nodes, then distances + criteria + the completion flag (lists that fit
    // the mapped ring)
    const unsigned long long seq = small ? ++c->signalSeq : 0ull;
    if (nStale && nActive > c->maxStamp) c->maxStamp = nActive;   // refreshed nodes are stamped with nActive
    // threads per pair of the workgroup-per-pair kernels: one column per thread for short lists (lowest latency); from 1024
    // pairs on the list fills the chip and half the wavefronts per pair finish sooner
    const unsigned pairThreads = c->pairWG > 0 ? (unsigned) c->pairWG : (n > 1024 && c->d.nPos <= 1024 ? 128u : (unsigned) VFT_WG);
    if (fusedRefresh) {   // short list that needs refreshes: one launch, pair workgroups wait on the stamps of flagged ends
        const size_t lds = pw_lds_bytes(c) / c->pwWaves;
        VFT_DISPATCH(c, (launch((k_pairs_refresh_fused<REAL, NC>), dim3((unsigned) (nStale + n)), dim3(pairThreads), lds, c->stream,
                                arena<REAL>(c), dStale, nStale, nForced, (const int64_t *) dI, (const int64_t *) dJ,
                                dWait, n, sa, (REAL *) o, (REAL *) (o + oB),
                                (REAL *) (o + 2 * oB), c->refDone, c->doneCtr, c->dFlag, seq, (REAL *) c->pairStage, (int64_t) VFT_PAIR_STAGE_CAP)));
    } else if (n <= 2048) {   // short list: a workgroup per pair (all of them resident at once)
        const size_t lds = pw_lds_bytes(c) / c->pwWaves;
        if (nStale)
            VFT_DISPATCH(c, (launch((k_refresh_list<REAL, NC, true>), dim3((unsigned) nStale), dim3(VFT_WG), lds, c->stream,
                                    arena<REAL>(c), dStale, nStale, nForced, sa)));
        VFT_DISPATCH(c, (launch((k_pairs_fused<REAL, NC, true>), dim3((unsigned) n), dim3(pairThreads), lds, c->stream,
                                arena<REAL>(c), dI, dJ, n, sa, (REAL *) o, (REAL *) (o + oB), (REAL *) (o + 2 * oB),
                                c->doneCtr, small ? c->dFlag : (unsigned long long *) nullptr, seq, (REAL *) c->pairStage,
                                (int64_t) VFT_PAIR_STAGE_CAP)));
    } else {
        if (nStale)
            VFT_DISPATCH(c, (launch((k_refresh_list<REAL, NC, false>), dim3(cdiv(nStale, c->pwWaves)), dim3(64 * c->pwWaves),
                                    pw_lds_bytes(c), c->stream, arena<REAL>(c), dStale, nStale, nForced, sa)));
        VFT_DISPATCH(c, (launch((k_pairs_fused<REAL, NC, false>), dim3(cdiv(n, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c),
                                c->stream, arena<REAL>(c), dI, dJ, n, sa, (REAL *) o, (REAL *) (o + oB),
                                (REAL *) (o + 2 * oB), c->doneCtr, small ? c->dFlag : (unsigned long long *) nullptr, seq,
                                (REAL *) nullptr, (int64_t) 0)));
    }
    LAUNCHCHK(c);
    if (small) {
        // results were written straight into mapped host memory; the kernel's last wave raises the flag
        if (smallTimes) ts2 = std::chrono::steady_clock::now();
        if (int r = wait_flag(c, seq)) return r;
        if (smallTimes) ts3 = std::chrono::steady_clock::now();
        const char *ho = hBase + 2 * idB;
        if (dist) memcpy(dist, ho, (size_t) n * rs);
        if (weight) memcpy(weight, ho + oB, (size_t) n * rs);
        if (crit) memcpy(crit, ho + 2 * oB, (size_t) n * rs);
        if (smallTimes && n >= 1000) {
            const auto ts4 = std::chrono::steady_clock::now();
            tsPrep += std::chrono::duration<double>(ts1 - ts0).count();
            tsLaunch += std::chrono::duration<double>(ts2 - ts1).count();
            tsWait += std::chrono::duration<double>(ts3 - ts2).count();
            tsOut += std::chrono::duration<double>(ts4 - ts3).count();
            tsPairs += n;
            if ((++tsCalls & 127) == 0)
                fprintf(stderr, "[vft api] short pair lists (n >= 1000): %lld calls, %.0f pairs avg; per call: prep %.1f us, launch %.1f us, wait %.1f us, copy-out %.1f us\n",
                        (long long) tsCalls, (double) tsPairs / tsCalls, 1e6 * tsPrep / tsCalls, 1e6 * tsLaunch / tsCalls, 1e6 * tsWait / tsCalls, 1e6 * tsOut / tsCalls);
        }
        return VFT_OK;
    }
    static const bool stageTimes = getenv("VFT_API_PROFILE") != nullptr;   // tools only: where a long list's time goes
    static double tKernel = 0, tCopy = 0;
    static int64_t nCalls = 0, nPairs = 0;
    std::chrono::steady_clock::time_point t0;
    if (stageTimes) {
        t0 = std::chrono::steady_clock::now();
        HIPCHK(c, hipStreamSynchronize(c->stream));
        tKernel += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
    }
    if (dist) HIPCHK(c, hipMemcpyAsync(dist, o, (size_t) n * rs, hipMemcpyDeviceToHost, c->stream));
    if (weight) HIPCHK(c, hipMemcpyAsync(weight, o + oB, (size_t) n * rs, hipMemcpyDeviceToHost, c->stream));
    if (crit) HIPCHK(c, hipMemcpyAsync(crit, o + 2 * oB, (size_t) n * rs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (stageTimes) {
        tCopy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        nCalls++;
        nPairs += n;
        if ((nCalls & 255) == 0)
            fprintf(stderr, "[vft api] long pair lists: %lld calls, %lld pairs, H2D + kernels %.3f s, D2H %.3f s\n", (long long) nCalls,
                    (long long) nPairs, tKernel, tCopy);
    }
    return VFT_OK;
}

// the cross product of two id lists in device memory -> dist[nA][nB] (device): lane per pair for 4-state alphabets without a
// distance matrix (k_pairs_block_tiled), a wavefront per pair otherwise
static int launch_pairs_block(vft_ctx *c, const int64_t *dA, int64_t nA, const int64_t *dB, int64_t nB, void *dOut) {
    const size_t tiledLds = (size_t) VFT_PB_A * (size_t) c->d.nPos * (5 * c->rs + 4) + 64;   // [A][nPos] x (4 frequencies + weight + code)
    if (c->cfg.n_codes == 4 && !c->hasDm && tiledLds <= (144u << 10)) {
        if (tiledLds > (48u << 10) && tiledLds > c->pbLdsSet) {
            if (c->rs == 4) HIPCHK(c, hipFuncSetAttribute((const void *) k_pairs_block_tiled<float, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) tiledLds));
            else HIPCHK(c, hipFuncSetAttribute((const void *) k_pairs_block_tiled<double, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) tiledLds));
            c->pbLdsSet = tiledLds;
        }
        const dim3 grid(cdiv(nB, VFT_WG), cdiv(nA, VFT_PB_A));
        if (c->rs == 4) launch((k_pairs_block_tiled<float, 4>), grid, dim3(VFT_WG), tiledLds, c->stream, arena<float>(c), dA, nA, dB, nB, (float *) dOut);
        else launch((k_pairs_block_tiled<double, 4>), grid, dim3(VFT_WG), tiledLds, c->stream, arena<double>(c), dA, nA, dB, nB, (double *) dOut);
    } else {
        VFT_DISPATCH(c, (launch((k_pairs_block<REAL, NC>), dim3(cdiv(nA * nB, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c), c->stream,
                                arena<REAL>(c), dA, nA, dB, nB, (REAL *) dOut)));
    }
    LAUNCHCHK(c);
    return VFT_OK;
}

extern "C" int vft_block_distances(vft_ctx *c, int64_t nA, const int64_t *a, int64_t nB, const int64_t *b, int64_t nActive,
                                   int64_t nDiffAllow, double totdiam, void *dist) {
    if (!c || nA < 0 || nB < 0 || !a || !b || !dist) return VFT_ERR_INVALID;
    if (nA == 0 || nB == 0) return VFT_OK;
    if (!c->leavesUp) return fail(c, VFT_ERR_STATE, "vft_block_distances before vft_upload_leaves");
    for (int64_t t = 0; t < nA; t++)
        if (a[t] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "vft_block_distances: a[%lld] out of range", (long long) t);
    for (int64_t t = 0; t < nB; t++)
        if (b[t] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "vft_block_distances: b[%lld] out of range", (long long) t);
    // the distinct stale nodes among both lists (as in pair_distances)
    std::vector<int64_t> &stale = c->staleIds;
    stale.clear();
    if (c->staleMark.size() != (size_t) c->d.maxNodes) c->staleMark.assign((size_t) c->d.maxNodes, 0u);
    if (++c->staleEpoch == 0u) {
        std::fill(c->staleMark.begin(), c->staleMark.end(), 0u);
        c->staleEpoch = 1u;
    }
    for (int pass = 0; pass < 2; pass++) {
        const int64_t *ids = pass ? b : a, cnt = pass ? nB : nA;
        for (int64_t t = 0; t < cnt; t++) {
            const int64_t v = ids[t];
            if (v >= 0 && (int64_t) c->hNOut[v] - nActive > nDiffAllow && c->staleMark[(size_t) v] != c->staleEpoch) {
                c->staleMark[(size_t) v] = c->staleEpoch;
                stale.push_back(v);
            }
        }
    }
    const int64_t nStale = (int64_t) stale.size();
    const size_t rs = c->rs, idB = (((size_t) (nA + nB + nStale) * 8) + 255) & ~(size_t) 255;
    const size_t oB = (((size_t) nA * (size_t) nB * rs) + 255) & ~(size_t) 255;
    if (int r = ensure_scratch(c, idB + oB + 256)) return r;
    char *sBase = (char *) c->scratch;
    int64_t *dA = (int64_t *) sBase, *dB = dA + nA, *dStale = dB + nB;
    // ids through the mapped ring when they fit (no staging copy), otherwise plain copies
    if (idB <= VFT_SMALL_BYTES) {
        char *h, *d;
        if (int r = io_alloc(c, idB, &h, &d)) return r;
        memcpy(h, a, (size_t) nA * 8);
        memcpy(h + (size_t) nA * 8, b, (size_t) nB * 8);
        if (nStale) memcpy(h + (size_t) (nA + nB) * 8, stale.data(), (size_t) nStale * 8);
        dA = (int64_t *) d;
        dB = dA + nA;
        dStale = dB + nB;
    } else {
        HIPCHK(c, hipMemcpyAsync(dA, a, (size_t) nA * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dB, b, (size_t) nB * 8, hipMemcpyHostToDevice, c->stream));
        if (nStale) HIPCHK(c, hipMemcpyAsync(dStale, stale.data(), (size_t) nStale * 8, hipMemcpyHostToDevice, c->stream));
    }
    SweepArgs sa{};
    sa.nActive = nActive;
    sa.nDiffAllow = nDiffAllow;
    sa.totdiam = totdiam;
    if (nStale) {
        if (nActive > c->maxStamp) c->maxStamp = nActive;
        VFT_DISPATCH(c, (launch((k_refresh_list<REAL, NC, false>), dim3(cdiv(nStale, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c),
                                c->stream, arena<REAL>(c), (const int64_t *) dStale, nStale, (int64_t) 0, sa)));
    }
    char *o = sBase + idB;
    if (int r = launch_pairs_block(c, (const int64_t *) dA, nA, (const int64_t *) dB, nB, o)) return r;
    HIPCHK(c, hipMemcpyAsync(dist, o, (size_t) nA * (size_t) nB * rs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_leaf_block_distances(vft_ctx *c, int64_t nA, const int64_t *a, int64_t nB, const int64_t *b, int64_t nActive,
                                        int64_t nDiffAllow, double totdiam, void *dist, void *weight, void *crit) {
    if (!c || nA < 0 || nB < 0 || !a || !b || !dist || !weight || !crit) return VFT_ERR_INVALID;
    if (nA == 0 || nB == 0) return VFT_OK;
    if (!c->leavesUp) return fail(c, VFT_ERR_STATE, "vft_leaf_block_distances before vft_upload_leaves");
    if (c->cfg.n_codes != 4 || c->hasDm) return fail(c, VFT_ERR_STATE, "vft_leaf_block_distances: nucleotides without a distance matrix only");
    for (int64_t t = 0; t < nA; t++)
        if (a[t] < 0 || a[t] >= c->d.nSeqs) return fail(c, VFT_ERR_INVALID, "vft_leaf_block_distances: a[%lld] is not a leaf", (long long) t);
    for (int64_t t = 0; t < nB; t++)
        if (b[t] >= c->d.nSeqs) return fail(c, VFT_ERR_INVALID, "vft_leaf_block_distances: b[%lld] is not a leaf", (long long) t);
    const size_t rs = c->rs, idB = (((size_t) (nA + nB) * 8) + 255) & ~(size_t) 255;
    const size_t oB = (((size_t) nA * (size_t) nB * rs) + 255) & ~(size_t) 255;
    if (int r = ensure_scratch(c, idB + 3 * oB + 256)) return r;
    char *s = (char *) c->scratch;
    int64_t *dA = (int64_t *) s, *dB = dA + nA;
    HIPCHK(c, hipMemcpyAsync(dA, a, (size_t) nA * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dB, b, (size_t) nB * 8, hipMemcpyHostToDevice, c->stream));
    // lazy refresh of whatever is stale (the host-mapped stamp mirror can only lag towards "staler")
    bool anyStale = false;
    for (int64_t t = 0; t < nA && !anyStale; t++) anyStale = (int64_t) c->hNOut[a[t]] - nActive > nDiffAllow;
    for (int64_t t = 0; t < nB && !anyStale; t++) anyStale = b[t] >= 0 && (int64_t) c->hNOut[b[t]] - nActive > nDiffAllow;
    if (anyStale) {
        std::vector<int64_t> ids;
        for (int64_t t = 0; t < nA; t++) ids.push_back(a[t]);
        for (int64_t t = 0; t < nB; t++)
            if (b[t] >= 0) ids.push_back(b[t]);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        int64_t *dIds = (int64_t *) (s + idB + 3 * oB);
        if (int r = ensure_scratch(c, idB + 3 * oB + 256 + ids.size() * 8)) return r;
        s = (char *) c->scratch;   // (ensure_scratch may have moved it: re-derive everything)
        dA = (int64_t *) s;
        dB = dA + nA;
        dIds = (int64_t *) (s + idB + 3 * oB);
        HIPCHK(c, hipMemcpyAsync(dA, a, (size_t) nA * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dB, b, (size_t) nB * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dIds, ids.data(), ids.size() * 8, hipMemcpyHostToDevice, c->stream));
        if (nActive > c->maxStamp) c->maxStamp = nActive;
        if (int r = launch_out_distances(c, dIds, (int64_t) ids.size(), nActive, nDiffAllow, totdiam, false)) return r;
        HIPCHK(c, hipStreamSynchronize(c->stream));   // ids is a local
    }
    SweepArgs sa{};
    sa.nActive = nActive;
    sa.nDiffAllow = nDiffAllow;
    sa.totdiam = totdiam;
    char *o = s + idB;
    const dim3 grid(cdiv(nB, 64), cdiv(nA, 64));
    if (c->cfg.precision == 4)
        launch((k_leaf_block<float>), grid, dim3(VFT_WG), 0, c->stream, arena<float>(c), (const int64_t *) dA, nA, (const int64_t *) dB, nB, sa,
               (float *) o, (float *) (o + oB), (float *) (o + 2 * oB));
    else
        launch((k_leaf_block<double>), grid, dim3(VFT_WG), 0, c->stream, arena<double>(c), (const int64_t *) dA, nA, (const int64_t *) dB, nB, sa,
               (double *) o, (double *) (o + oB), (double *) (o + 2 * oB));
    LAUNCHCHK(c);
    const size_t bytes = (size_t) nA * (size_t) nB * rs;
    HIPCHK(c, hipMemcpyAsync(dist, o, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(weight, o + oB, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(crit, o + 2 * oB, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- top-hit lists
template <typename REAL>
static TopHits<REAL> tophits(const vft_ctx *c) {
    TopHits<REAL> T;
    T.hits = (ThHit<REAL> *) c->thHits;
    T.len = c->thLen;
    T.m = c->thM;
    T.cap = c->thCap;
    T.nLists = c->thLists;
    T.stJ = c->thStJ;
    T.stD = (REAL *) c->thStD;
    T.stC = (REAL *) c->thStC;
    T.mark = c->thMark;
    T.doneCtr = c->thDone;
    T.sorted = c->thSorted;
    return T;
}

extern "C" int vft_tophits_create(vft_ctx *c, int32_t m, int64_t nLists) {
    if (!c || m < 1 || nLists < 1 || nLists > c->d.maxNodes) return VFT_ERR_INVALID;
    if (c->thHits) return fail(c, VFT_ERR_STATE, "vft_tophits_create: lists exist already");
    const size_t hitB = c->rs == 4 ? sizeof(ThHit<float>) : sizeof(ThHit<double>);
    int P = 1;
    while (P < 2 * m) P <<= 1;
    c->thLds = std::max(pw_lds_bytes(c) / c->pwWaves, (size_t) P * sizeof(ThKey));
    if (c->thLds > (160u << 10) - 8192) return fail(c, VFT_ERR_INVALID, "vft_tophits_create: lists of %d entries do not fit the LDS of the merge kernel", (int) m);
    c->thM = m;
    c->thCap = 2 * m + 64;
    c->thLists = nLists;
    HIPCHK(c, hipMalloc(&c->thHits, (size_t) nLists * (size_t) m * hitB));
    HIPCHK(c, hipMalloc((void **) &c->thLen, (size_t) nLists * 4));
    HIPCHK(c, hipMemsetAsync(c->thLen, 0, (size_t) nLists * 4, c->stream));
    HIPCHK(c, hipMalloc((void **) &c->thStJ, (size_t) c->thCap * 4));
    HIPCHK(c, hipMalloc(&c->thStD, (size_t) c->thCap * c->rs));
    HIPCHK(c, hipMalloc(&c->thStC, (size_t) c->thCap * c->rs));
    HIPCHK(c, hipMalloc((void **) &c->thMark, (size_t) c->d.maxNodes * 4));
    HIPCHK(c, hipMemsetAsync(c->thMark, 0, (size_t) c->d.maxNodes * 4, c->stream));
    HIPCHK(c, hipMalloc((void **) &c->thSorted, (size_t) (c->thCap + 1) * 4));
    HIPCHK(c, hipMalloc((void **) &c->thDone, 65 * 4));
    HIPCHK(c, hipMemsetAsync(c->thDone, 0, 65 * 4, c->stream));
    c->thTag = 0;
    if (c->thLds > (48u << 10))
        VFT_DISPATCH(c, {
            HIPCHK(c, hipFuncSetAttribute((const void *) k_th_best<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->thLds));
            HIPCHK(c, hipFuncSetAttribute((const void *) k_th_join<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->thLds));
        });
    return VFT_OK;
}

extern "C" int vft_tophits_upload(vft_ctx *c, int64_t count, const int64_t *nodes, const int32_t *lens, const void *packed) {
    if (!c || count < 0 || !nodes || !lens || !packed) return VFT_ERR_INVALID;
    if (!c->thHits) return fail(c, VFT_ERR_STATE, "vft_tophits_upload before vft_tophits_create");
    if (count == 0) return VFT_OK;
    for (int64_t t = 0; t < count; t++)
        if (nodes[t] < 0 || nodes[t] >= c->thLists || lens[t] < 0 || lens[t] > c->thM) return fail(c, VFT_ERR_INVALID, "vft_tophits_upload: list %lld out of range", (long long) t);
    const size_t hitB = c->rs == 4 ? sizeof(ThHit<float>) : sizeof(ThHit<double>);
    const size_t idB = (((size_t) count * 8) + 255) & ~(size_t) 255, lenB = (((size_t) count * 4) + 255) & ~(size_t) 255;
    const size_t pkB = (size_t) count * (size_t) c->thM * hitB;
    if (int r = ensure_scratch(c, idB + lenB + pkB + 256)) return r;
    char *sb = (char *) c->scratch;
    HIPCHK(c, hipMemcpyAsync(sb, nodes, (size_t) count * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(sb + idB, lens, (size_t) count * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(sb + idB + lenB, packed, pkB, hipMemcpyHostToDevice, c->stream));
    if (c->rs == 4)
        launch((k_th_scatter<float>), dim3((unsigned) count), dim3(256), 0, c->stream, tophits<float>(c), (const int64_t *) sb,
               (const int32_t *) (sb + idB), (const ThHit<float> *) (sb + idB + lenB));
    else
        launch((k_th_scatter<double>), dim3((unsigned) count), dim3(256), 0, c->stream, tophits<double>(c), (const int64_t *) sb,
               (const int32_t *) (sb + idB), (const ThHit<double> *) (sb + idB + lenB));
    LAUNCHCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the caller's buffers are pageable: the copies may still be reading them
    return VFT_OK;
}

extern "C" int vft_tophits_download(vft_ctx *c, int64_t node, int32_t *len, void *hits) {
    if (!c || !len || !hits) return VFT_ERR_INVALID;
    if (!c->thHits || node < 0 || node >= c->thLists) return fail(c, VFT_ERR_INVALID, "vft_tophits_download: no such list");
    const size_t hitB = c->rs == 4 ? sizeof(ThHit<float>) : sizeof(ThHit<double>);
    HIPCHK(c, hipMemcpyAsync(len, c->thLen + node, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hits, (char *) c->thHits + (size_t) node * (size_t) c->thM * hitB, (size_t) c->thM * hitB, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// setOutDistance(node) in front of a list walk: forced (recomputed unless the stamp IS n_active) or the lazy rule; skipped
// when the host-mapped stamp mirror shows that nothing would happen (it can only lag towards "staler")
static void tophits_refresh_own(vft_ctx *c, int64_t node, const SweepArgs &sa, bool force) {
    const int64_t st = (int64_t) c->hNOut[node];
    if (force ? st == sa.nActive : !(st - sa.nActive > sa.nDiffAllow)) return;
    SweepArgs s1 = sa;
    s1.force = force ? 1 : 0;
    if (sa.nActive > c->maxStamp) c->maxStamp = sa.nActive;
    VFT_DISPATCH(c, (launch((k_out_distance_one<REAL, NC>), dim3(1), dim3(VFT_WG), pw_lds_bytes(c) / c->pwWaves, c->stream,
                            arena<REAL>(c), node, s1)));
}

extern "C" int vft_tophits_best(vft_ctx *c, int64_t node, int32_t len, int64_t nActive, int64_t nDiffAllow, double totdiam,
                                int32_t forceNode, vft_tophits_best_t *out) {
    if (!c || !out) return VFT_ERR_INVALID;
    if (!c->thHits || node < 0 || node >= c->thLists || node >= c->maxnode || len < 0 || len > c->thM)
        return fail(c, VFT_ERR_INVALID, "vft_tophits_best: bad list");
    if (len == 0) {
        out->j = out->pos = -1;
        out->dist = out->criterion = 1e20;
        return VFT_OK;
    }
    SweepArgs sa{};
    sa.nActive = nActive;
    sa.nDiffAllow = nDiffAllow;
    sa.totdiam = totdiam;
    tophits_refresh_own(c, node, sa, forceNode != 0);
    if (nActive > c->maxStamp) c->maxStamp = nActive;   // the walk may refresh partners
    char *h, *d;
    if (int r = io_alloc(c, sizeof(ThBestOut), &h, &d)) return r;
    const unsigned long long seq = ++c->signalSeq;
    VFT_DISPATCH(c, (launch((k_th_best<REAL, NC>), dim3((unsigned) len), dim3(VFT_WG), c->thLds, c->stream, arena<REAL>(c),
                            tophits<REAL>(c), node, sa, (ThBestOut *) d, c->dFlag, seq)));
    LAUNCHCHK(c);
    if (int r = wait_flag(c, seq)) return r;
    const ThBestOut *o = (const ThBestOut *) h;
    out->j = o->j;
    out->pos = o->pos;
    out->dist = o->dist;
    out->criterion = o->crit;
    return VFT_OK;
}

extern "C" int vft_tophits_join(vft_ctx *c, int64_t newnode, int64_t c0, int32_t n0, int64_t c1, int32_t n1, int64_t nActive,
                                int64_t nDiffAllow, double totdiam, int32_t nSaveMax, int32_t need, int32_t ageOK,
                                vft_tophits_join_t *info, int32_t *j, void *dist, void *crit) {
    if (!c || !info || !j || !dist || !crit) return VFT_ERR_INVALID;
    if (!c->thHits || newnode < 0 || newnode >= c->thLists || newnode >= c->maxnode || c0 < 0 || c0 >= c->thLists || c1 < 0 ||
        c1 >= c->thLists || n0 < 0 || n0 > c->thM || n1 < 0 || n1 > c->thM || nSaveMax < 0 || nSaveMax > c->thM)
        return fail(c, VFT_ERR_INVALID, "vft_tophits_join: bad lists");
    const int32_t n = n0 + n1;
    if (n == 0) {
        info->n_unique = info->n_save = info->pad = 0;
        info->use_unique = (nActive - 1 == 0 || (ageOK && 0 >= need)) ? 1 : 0;
        return VFT_OK;
    }
    SweepArgs sa{};
    sa.nActive = nActive;
    sa.nDiffAllow = nDiffAllow;
    sa.totdiam = totdiam;
    tophits_refresh_own(c, newnode, sa, false);   // the new node's stamp is "unreasonably high" (NJ.tcc:254): the lazy rule fires
    if (nActive > c->maxStamp) c->maxStamp = nActive;
    const size_t rs = c->rs, aB = (((size_t) n * 8) + 255) & ~(size_t) 255;
    char *h, *d;
    if (int r = io_alloc(c, 256 + 3 * aB, &h, &d)) return r;
    const unsigned long long seq = ++c->signalSeq;
    if (++c->thTag == 0u) {   // (2^32 joins on one context: start the marks over)
        HIPCHK(c, hipMemsetAsync(c->thMark, 0, (size_t) c->d.maxNodes * 4, c->stream));
        c->thTag = 1u;
    }
    VFT_DISPATCH(c, (launch((k_th_join<REAL, NC>), dim3((unsigned) n), dim3(VFT_WG), c->thLds, c->stream, arena<REAL>(c),
                            tophits<REAL>(c), newnode, c0, n0, c1, sa, c->thTag, nSaveMax, need, ageOK, (ThJoinInfo *) d,
                            (int32_t *) (d + 256), (REAL *) (d + 256 + aB), (REAL *) (d + 256 + 2 * aB), c->dFlag, seq)));
    LAUNCHCHK(c);
    if (int r = wait_flag(c, seq)) return r;
    const ThJoinInfo *o = (const ThJoinInfo *) h;
    info->n_unique = o->nUnique;
    info->use_unique = o->useUnique;
    info->n_save = o->nSave;
    info->pad = 0;
    memcpy(j, h + 256, (size_t) o->nUnique * 4);
    memcpy(dist, h + 256 + aB, (size_t) o->nUnique * rs);
    memcpy(crit, h + 256 + 2 * aB, (size_t) o->nUnique * rs);
    return VFT_OK;
}

extern "C" int vft_tophits_refresh(vft_ctx *c, int64_t newnode, int32_t nHits, const int64_t *hitJ, const void *hitDist, int32_t nOwn,
                                   const void *ownList, int64_t nWork, const int64_t *work, const int32_t *nNew, int64_t nActive,
                                   int64_t nDiffAllow, double totdiam, int32_t *lens, void *first) {
    if (!c || nHits < 0 || !hitJ || !hitDist || nOwn < 0 || (nOwn > 0 && !ownList) || nWork < 0 || (nWork > 0 && (!work || !nNew || !lens || !first)))
        return VFT_ERR_INVALID;
    if (!c->thHits || newnode < 0 || newnode >= c->thLists || newnode >= c->maxnode || nOwn > c->thM)
        return fail(c, VFT_ERR_INVALID, "vft_tophits_refresh: bad list");
    int32_t maxNew = 0;
    for (int64_t t = 0; t < nWork; t++) {
        if (work[t] < 0 || work[t] >= c->thLists || work[t] >= c->maxnode || nNew[t] < 1 || nNew[t] > c->thM)
            return fail(c, VFT_ERR_INVALID, "vft_tophits_refresh: work node %lld out of range", (long long) t);
        maxNew = std::max(maxNew, nNew[t]);
    }
    for (int32_t u = 0; u < nHits; u++)
        if (hitJ[u] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "vft_tophits_refresh: hit %d out of range", (int) u);
    const size_t rs = c->rs, hitB = rs == 4 ? sizeof(ThHit<float>) : sizeof(ThHit<double>);
    const int32_t nB = std::min<int32_t>(nHits, 2 * maxNew);
    const int32_t E = ((c->thM + nB + 3) / 4) * 4;
    int32_t P = 2;
    while (P < E) P <<= 1;
    const size_t lds = pw_lds_bytes(c) / c->pwWaves + (size_t) P * sizeof(ThKey) + (size_t) E * 16 + (size_t) E * rs;
    if (lds > (160u << 10) - 1024) return fail(c, VFT_ERR_STATE, "vft_tophits_refresh: %zu bytes of LDS per node", lds);
    // input block: targets | distances | work | nNew | own list   (host-mapped ring, then one copy to device memory)
    auto pad = [](size_t b) { return (b + 255) & ~(size_t) 255; };
    const size_t oT = 0, oD = oT + pad((size_t) nB * 8), oW = oD + pad((size_t) nB * rs), oN = oW + pad((size_t) nWork * 8),
                 oO = oN + pad((size_t) nWork * 4), inB = oO + pad((size_t) nOwn * hitB);
    const size_t oLen = 0, oFirst = pad((size_t) nWork * 4), outB = oFirst + pad((size_t) nWork * hitB);
    char *h, *d;
    if (int r = io_alloc(c, inB + outB, &h, &d)) return r;
    memcpy(h + oT, hitJ, (size_t) nB * 8);
    memcpy(h + oD, hitDist, (size_t) nB * rs);
    if (nWork) memcpy(h + oW, work, (size_t) nWork * 8);
    if (nWork) memcpy(h + oN, nNew, (size_t) nWork * 4);
    if (nOwn) memcpy(h + oO, ownList, (size_t) nOwn * hitB);
    const size_t blockB = pad((size_t) nWork * (size_t) nB * rs);
    if (int r = ensure_scratch(c, inB + blockB + 256)) return r;
    char *sIn = (char *) c->scratch, *sBlock = sIn + inB;
    launch(k_copy16x2, dim3(cdiv((int64_t) (inB / 16), 256)), dim3(256), 0, c->stream, (uint4 *) sIn, (const uint4 *) d, (int64_t) (inB / 16),
           (uint4 *) nullptr, (const uint4 *) nullptr, (int64_t) 0);
    SweepArgs sa{};
    sa.nActive = nActive;
    sa.nDiffAllow = nDiffAllow;
    sa.totdiam = totdiam;
    if (c->rs == 4) launch((k_th_store<float>), dim3(1), dim3(256), 0, c->stream, tophits<float>(c), newnode, (const ThHit<float> *) (sIn + oO), nOwn);
    else launch((k_th_store<double>), dim3(1), dim3(256), 0, c->stream, tophits<double>(c), newnode, (const ThHit<double> *) (sIn + oO), nOwn);
    if (nWork == 0) {
        LAUNCHCHK(c);
        return VFT_OK;   // (stream-ordered: the next walk sees the list)
    }
    // setCriterion's lazy refresh for every node the block names (none after NJ.tcc:4451-4464; kept for callers that skip it)
    {
        std::vector<int64_t> &stale = c->staleIds;
        stale.clear();
        if (c->staleMark.size() != (size_t) c->d.maxNodes) c->staleMark.assign((size_t) c->d.maxNodes, 0u);
        if (++c->staleEpoch == 0u) {
            std::fill(c->staleMark.begin(), c->staleMark.end(), 0u);
            c->staleEpoch = 1u;
        }
        for (int pass = 0; pass < 2; pass++) {
            const int64_t *ids = pass ? hitJ : work, cnt = pass ? nB : nWork;
            for (int64_t t = 0; t < cnt; t++) {
                const int64_t v = ids[t];
                if (v >= 0 && (int64_t) c->hNOut[v] - nActive > nDiffAllow && c->staleMark[(size_t) v] != c->staleEpoch) {
                    c->staleMark[(size_t) v] = c->staleEpoch;
                    stale.push_back(v);
                }
            }
        }
        if (!stale.empty()) {
            const int64_t nStale = (int64_t) stale.size();
            char *hs, *ds;
            if (int r = io_alloc(c, (size_t) nStale * 8, &hs, &ds)) return r;
            memcpy(hs, stale.data(), (size_t) nStale * 8);
            if (nActive > c->maxStamp) c->maxStamp = nActive;
            VFT_DISPATCH(c, (launch((k_refresh_list<REAL, NC, false>), dim3(cdiv(nStale, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c),
                                    c->stream, arena<REAL>(c), (const int64_t *) ds, nStale, (int64_t) 0, sa)));
        }
    }
    if (int r = launch_pairs_block(c, (const int64_t *) (sIn + oW), nWork, (const int64_t *) (sIn + oT), (int64_t) nB, sBlock)) return r;
    if (lds > c->thRefreshLds) {
        VFT_DISPATCH(c, HIPCHK(c, hipFuncSetAttribute((const void *) k_th_refresh<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)));
        c->thRefreshLds = lds;
    }
    const unsigned long long seq = ++c->signalSeq;
    char *dOut = d + inB;
    VFT_DISPATCH(c, (launch((k_th_refresh<REAL, NC>), dim3((unsigned) nWork), dim3(VFT_WG), lds, c->stream, arena<REAL>(c), tophits<REAL>(c),
                            newnode, (const int64_t *) (sIn + oW), (const int32_t *) (sIn + oN), (const int64_t *) (sIn + oT),
                            (const REAL *) (sIn + oD), nB, (const REAL *) sBlock, sa, P, E, (int32_t *) (dOut + oLen),
                            (ThHit<REAL> *) (dOut + oFirst), c->dFlag, seq)));
    LAUNCHCHK(c);
    if (int r = wait_flag(c, seq)) return r;
    memcpy(lens, h + inB + oLen, (size_t) nWork * 4);
    memcpy(first, h + inB + oFirst, (size_t) nWork * hitB);
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- join engine
template <typename REAL>
static NjEngine<REAL> njengine(const vft_ctx *c) {
    NjEngine<REAL> E;
    E.st = (NjState<REAL> *) c->njState;
    E.visJ = c->njVisJ;
    E.visD = (REAL *) c->njVisD;
    E.topvis = c->njTop;
    E.age = c->njAge;
    E.logDev = c->njLogDev;
    E.logHost = c->njLogHostDev;
    E.hostStatus = c->njStatusDev;
    E.m = c->njCfg.m;
    E.nTop = c->njCfg.n_top;
    E.need = c->njCfg.need;
    E.ageLimit = c->njCfg.age_limit;
    E.fastest = c->njCfg.fastest;
    E.staleStamp = clamp_i32(c->njCfg.stale_stamp);
    E.staleOutLimit = c->njCfg.stale_out_limit;
    E.tol = c->fpostTol;
    E.stash = (REAL *) c->pendBase;
    E.pendIds = c->pendIdsDev;
    E.refClaim = c->njClaim;
    E.logNode = c->njLogNode;
    E.logStamp = c->njLogStamp;
    E.logOut = (REAL *) c->njLogOut;
    E.slotI = c->njSlotI;
    E.slotR = (REAL *) c->njSlotR;
    E.candI = c->njCandI;
    E.candR = (REAL *) c->njCandR;
    E.nTopPad = c->njTopPad;
    E.capPad = c->njCapPad;
    return E;
}

template <typename REAL>
static __global__ void k_nj_set_state(NjState<REAL> *st, long long nActive, long long maxnode, double totdiam, int32_t tvAge, int32_t clearHalt,
                                      volatile long long *hostStatus) {
    if (nActive >= 0) st->nActive = nActive;
    if (maxnode >= 0) st->maxnode = maxnode;
    if (totdiam == totdiam) st->totdiam = totdiam;
    if (tvAge >= 0) st->tvAge = tvAge;
    if (clearHalt) {
        st->halt = 0;
        hostStatus[0] = (long long) ((unsigned long long) st->joinsDone & 0x7FFFFFFFull);
    }
}

template <typename REAL>
static __global__ void k_nj_nodes_set(NjEngine<REAL> E, const int64_t *nodes, const int32_t *j, const REAL *dist, int64_t n, int32_t age) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t v = nodes[t];
    if (j) {
        E.visJ[v] = j[t];
        E.visD[v] = dist[t];
    }
    if (age >= 0) E.age[v] = age;
}

static __global__ void k_copy_i32(int32_t *dst, const int32_t *src, int64_t n) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[t];
}

static int nj_engine_create_impl(vft_ctx *c, const vft_nj_engine_config *cfg) {
    if (!c || !cfg || cfg->m < 1 || cfg->n_top < 1) return VFT_ERR_INVALID;
    if (!c->thHits || cfg->m != c->thM) return fail(c, VFT_ERR_STATE, "vft_nj_engine_create: vft_tophits_create(m) first");
    if (c->njState) return fail(c, VFT_ERR_STATE, "vft_nj_engine_create: the engine exists already");
    if (c->rowMode) return fail(c, VFT_ERR_STATE, "vft_nj_engine_create belongs to the NJ phase");
    c->njCfg = *cfg;
    const size_t rs = c->rs, stB = c->rs == 4 ? sizeof(NjState<float>) : sizeof(NjState<double>);
    const size_t pairLds = pw_lds_bytes(c) / c->pwWaves;
    int P = 2;
    while (P < 2 * cfg->m) P <<= 1;
    c->njP = P;
    c->njScanLds = 0;
    // k_nj_glue_scan: 2 x pair staging | keys | slot criteria | distances by staging index | slot cache | stale list | pass list
    c->njTailLds = 2 * pairLds + (size_t) P * sizeof(ThKey) + (size_t) cfg->n_top * 8 + (size_t) P * rs + (size_t) cfg->n_top * rs +
                   (size_t) cfg->n_top * 12 + (size_t) (2 * cfg->n_top + 2 * P + 2) * 4 + (size_t) P * 4 + (size_t) P * rs + 64;
    if (c->njTailLds > (160u << 10) - (16u << 10) || 2 * pairLds > (160u << 10) - (16u << 10))
        return fail(c, VFT_ERR_STATE, "vft_nj_engine_create: lists or alignment too long for the glue kernels' LDS");
    // a thread of k_nj_glue_scan holds VFT_NJ_BATCH slots of the top-visible list and as many ranks of the merge
    c->njTailThreads = std::max(cfg->n_top, P) <= VFT_NJ_BATCH * VFT_NJ_TAIL && !c->wideGlue ? VFT_NJ_TAIL : 1024;
    if (std::max(cfg->n_top, P) > VFT_NJ_BATCH * 1024) return fail(c, VFT_ERR_STATE, "vft_nj_engine_create: lists too long for the glue kernel");
    if (int r = ensure_ml_rows(c)) return r;
    if (!c->pendBase) {
        const CommitPlan plan = commit_plan(c, VFT_PEND_MAX);
        HIPCHK(c, hipMalloc((void **) &c->pendBase, plan.totalB + 512));
        HIPCHK(c, hipMalloc((void **) &c->pendIdsDev, VFT_PEND_MAX * sizeof(int64_t)));
    }
    const size_t nodes = (size_t) c->d.maxNodes, joins = (size_t) c->d.nSeqs;
    HIPCHK(c, hipMalloc(&c->njState, stB));
    HIPCHK(c, hipMemsetAsync(c->njState, 0, stB, c->stream));
    HIPCHK(c, hipMalloc((void **) &c->njVisJ, nodes * 4));
    HIPCHK(c, hipMemsetAsync(c->njVisJ, 0xFF, nodes * 4, c->stream));
    HIPCHK(c, hipMalloc(&c->njVisD, nodes * rs));
    HIPCHK(c, hipMemsetAsync(c->njVisD, 0, nodes * rs, c->stream));
    HIPCHK(c, hipMalloc((void **) &c->njTop, (size_t) cfg->n_top * 4));
    HIPCHK(c, hipMemsetAsync(c->njTop, 0xFF, (size_t) cfg->n_top * 4, c->stream));
    HIPCHK(c, hipMalloc((void **) &c->njAge, nodes * 4));
    HIPCHK(c, hipMemsetAsync(c->njAge, 0, nodes * 4, c->stream));
    HIPCHK(c, hipMalloc((void **) &c->njLogDev, joins * sizeof(NjJoinRec)));
    HIPCHK(c, hipMalloc((void **) &c->njClaim, nodes * 4));
    HIPCHK(c, hipMemsetAsync(c->njClaim, 0, nodes * 4, c->stream));
    HIPCHK(c, hipMalloc((void **) &c->njLogNode, (size_t) (cfg->m + 64) * 4));
    HIPCHK(c, hipMalloc((void **) &c->njLogStamp, (size_t) (cfg->m + 64) * 4));
    HIPCHK(c, hipMalloc(&c->njLogOut, (size_t) (cfg->m + 64) * rs));
    c->njTopPad = (cfg->n_top + 63) & ~63;
    c->njCapPad = (c->thCap + 63) & ~63;
    HIPCHK(c, hipMalloc((void **) &c->njSlotI, (size_t) 6 * c->njTopPad * 4));
    HIPCHK(c, hipMalloc(&c->njSlotR, (size_t) 3 * c->njTopPad * rs));
    HIPCHK(c, hipMalloc((void **) &c->njCandI, (size_t) 5 * c->njCapPad * 4));
    HIPCHK(c, hipMalloc(&c->njCandR, (size_t) 5 * c->njCapPad * rs));
    HIPCHK(c, hipMemsetAsync(c->njSlotI, 0, (size_t) 6 * c->njTopPad * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->njSlotR, 0, (size_t) 3 * c->njTopPad * rs, c->stream));
    HIPCHK(c, hipMemsetAsync(c->njCandI, 0, (size_t) 5 * c->njCapPad * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->njCandR, 0, (size_t) 5 * c->njCapPad * rs, c->stream));
    HIPCHK(c, hipHostMalloc((void **) &c->njLogHost, joins * sizeof(NjJoinRec), hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void **) &c->njLogHostDev, c->njLogHost, 0));
    memset(c->njLogHost, 0xFF, joins * sizeof(NjJoinRec));   // an unwritten record fails vft_nj_engine_adopt's check (newnode = -1)
    HIPCHK(c, hipHostMalloc((void **) &c->njStatusHost, 64, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void **) &c->njStatusDev, c->njStatusHost, 0));
    memset(c->njStatusHost, 0, 64);
    VFT_DISPATCH(c, {
        if (c->njTailLds > (48u << 10)) {
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_glue_scan<REAL, NC, VFT_NJ_TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->njTailLds));
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_glue_scan<REAL, NC, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->njTailLds));
        }
        if ((size_t) P * sizeof(ThKey) > (48u << 10))
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_merge_rank<REAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ((size_t) P * sizeof(ThKey))));
        if (2 * pairLds > (48u << 10)) {
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_glue_best<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pairLds));
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_best_pairs<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pairLds));
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_best_pairs2<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pairLds));
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_glue_join<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (2 * pairLds)));
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_refresh_new<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pairLds));
            HIPCHK(c, hipFuncSetAttribute((const void *) k_nj_merge_pairs<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) pairLds));
        }
    });
    return VFT_OK;
}

// every buffer of the engine, released: after a failed creation the context must not look as if it had one
static void nj_engine_free(vft_ctx *c) {
    void **dev[] = {&c->njState, &c->njVisD, (void **) &c->njVisJ, (void **) &c->njTop, (void **) &c->njAge, (void **) &c->njLogDev, (void **) &c->njClaim,
                    (void **) &c->njLogNode, (void **) &c->njLogStamp, &c->njLogOut, (void **) &c->njSlotI, &c->njSlotR, (void **) &c->njCandI, &c->njCandR};
    for (void **p: dev)
        if (*p) {
            hipFree(*p);
            *p = nullptr;
        }
    if (c->njLogHost) hipHostFree(c->njLogHost);
    if (c->njStatusHost) hipHostFree(c->njStatusHost);
    c->njLogHost = c->njLogHostDev = nullptr;
    c->njStatusHost = c->njStatusDev = nullptr;
}

extern "C" int vft_nj_engine_create(vft_ctx *c, const vft_nj_engine_config *cfg) {
    const bool had = c && c->njState;   // ("the engine exists already" must leave that engine alone)
    const int r = nj_engine_create_impl(c, cfg);
    if (r != VFT_OK && c && !had && c->njState) {
        hipStreamSynchronize(c->stream);
        nj_engine_free(c);
    }
    return r;
}

#define NJ_ENGINE_OK(c) do { if (!(c)) return VFT_ERR_INVALID; if (!(c)->njState) return fail((c), VFT_ERR_STATE, "no join engine (vft_nj_engine_create)"); } while (0)

extern "C" int vft_nj_engine_set_state(vft_ctx *c, int64_t nActive, int64_t maxnode, double totdiam, int32_t tvAge) {
    NJ_ENGINE_OK(c);
    if (c->rs == 4) launch((k_nj_set_state<float>), dim3(1), dim3(1), 0, c->stream, (NjState<float> *) c->njState, (long long) nActive, (long long) maxnode, totdiam, tvAge, 0, c->njStatusDev);
    else launch((k_nj_set_state<double>), dim3(1), dim3(1), 0, c->stream, (NjState<double> *) c->njState, (long long) nActive, (long long) maxnode, totdiam, tvAge, 0, c->njStatusDev);
    LAUNCHCHK(c);
    return VFT_OK;
}

template <typename REAL>
static int nj_get_state(vft_ctx *c, int64_t *nActive, int64_t *maxnode, double *totdiam, int32_t *tvAge, int64_t *joinsDone, int32_t *halt,
                        int32_t *haltJoin, int32_t *nUnique) {
    NjState<REAL> st;
    HIPCHK(c, hipMemcpyAsync(&st, c->njState, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nActive) *nActive = st.nActive;
    if (maxnode) *maxnode = st.maxnode;
    if (totdiam) *totdiam = st.totdiam;
    if (tvAge) *tvAge = st.tvAge;
    if (joinsDone) *joinsDone = st.joinsDone;
    if (halt) *halt = st.halt;
    if (haltJoin) *haltJoin = st.haltJoin;
    if (nUnique) *nUnique = st.nUnique;
    return VFT_OK;
}
extern "C" int vft_nj_engine_get_state(vft_ctx *c, int64_t *nActive, int64_t *maxnode, double *totdiam, int32_t *tvAge, int64_t *joinsDone,
                                       int32_t *halt, int32_t *haltJoin, int32_t *nUnique) {
    NJ_ENGINE_OK(c);
    return c->rs == 4 ? nj_get_state<float>(c, nActive, maxnode, totdiam, tvAge, joinsDone, halt, haltJoin, nUnique)
                      : nj_get_state<double>(c, nActive, maxnode, totdiam, tvAge, joinsDone, halt, haltJoin, nUnique);
}

extern "C" int vft_nj_engine_visible_set(vft_ctx *c, int64_t first, int64_t count, const int32_t *j, const void *dist) {
    NJ_ENGINE_OK(c);
    if (first < 0 || count < 0 || first + count > c->d.maxNodes || !j || !dist) return VFT_ERR_INVALID;
    HIPCHK(c, hipMemcpyAsync(c->njVisJ + first, j, (size_t) count * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((char *) c->njVisD + (size_t) first * c->rs, dist, (size_t) count * c->rs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_nj_engine_visible_get(vft_ctx *c, int64_t first, int64_t count, int32_t *j, void *dist) {
    NJ_ENGINE_OK(c);
    if (first < 0 || count < 0 || first + count > c->d.maxNodes || !j || !dist) return VFT_ERR_INVALID;
    HIPCHK(c, hipMemcpyAsync(j, c->njVisJ + first, (size_t) count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dist, (char *) c->njVisD + (size_t) first * c->rs, (size_t) count * c->rs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_nj_engine_nodes_set(vft_ctx *c, int64_t n, const int64_t *nodes, const int32_t *j, const void *dist, int32_t age) {
    NJ_ENGINE_OK(c);
    if (n < 0 || (n > 0 && !nodes) || (j && !dist)) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    for (int64_t t = 0; t < n; t++)
        if (nodes[t] < 0 || nodes[t] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_nj_engine_nodes_set: node %lld out of range", (long long) t);
    auto pad = [](size_t b) { return (b + 255) & ~(size_t) 255; };
    const size_t oN = 0, oJ = pad((size_t) n * 8), oD = oJ + pad((size_t) n * 4), tot = oD + pad((size_t) n * c->rs);
    char *h, *d;
    if (tot <= VFT_SMALL_BYTES) {
        if (int r = io_alloc(c, tot, &h, &d)) return r;
        memcpy(h + oN, nodes, (size_t) n * 8);
        if (j) memcpy(h + oJ, j, (size_t) n * 4);
        if (j) memcpy(h + oD, dist, (size_t) n * c->rs);
    } else {
        if (int r = ensure_scratch(c, tot + 256)) return r;
        d = (char *) c->scratch;
        HIPCHK(c, hipMemcpyAsync(d + oN, nodes, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
        if (j) HIPCHK(c, hipMemcpyAsync(d + oJ, j, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
        if (j) HIPCHK(c, hipMemcpyAsync(d + oD, dist, (size_t) n * c->rs, hipMemcpyHostToDevice, c->stream));
    }
    if (c->rs == 4)
        launch((k_nj_nodes_set<float>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, njengine<float>(c), (const int64_t *) (d + oN),
               j ? (const int32_t *) (d + oJ) : (const int32_t *) nullptr, (const float *) (d + oD), n, age);
    else
        launch((k_nj_nodes_set<double>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, njengine<double>(c), (const int64_t *) (d + oN),
               j ? (const int32_t *) (d + oJ) : (const int32_t *) nullptr, (const double *) (d + oD), n, age);
    LAUNCHCHK(c);
    if (tot > VFT_SMALL_BYTES) HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_nj_engine_topvisible_set(vft_ctx *c, const int32_t *nodes) {
    NJ_ENGINE_OK(c);
    if (!nodes) return VFT_ERR_INVALID;
    const size_t bytes = (size_t) c->njCfg.n_top * 4;
    char *h, *d;
    if (int r = io_alloc(c, bytes, &h, &d)) return r;
    memcpy(h, nodes, bytes);
    launch(k_copy_i32, dim3(cdiv(c->njCfg.n_top, 256)), dim3(256), 0, c->stream, c->njTop, (const int32_t *) d, (int64_t) c->njCfg.n_top);
    LAUNCHCHK(c);
    return VFT_OK;
}

extern "C" int vft_nj_engine_topvisible_get(vft_ctx *c, int32_t *nodes) {
    NJ_ENGINE_OK(c);
    if (!nodes) return VFT_ERR_INVALID;
    HIPCHK(c, hipMemcpyAsync(nodes, c->njTop, (size_t) c->njCfg.n_top * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_nj_engine_reset_candidates(vft_ctx *c, int64_t nActive, double totdiam, int32_t k, void *hits, int64_t *nVisible) {
    NJ_ENGINE_OK(c);
    if (!hits || !nVisible || k < 1 || k > c->hitsCap || nActive < 1) return VFT_ERR_INVALID;
    const int64_t maxnode = c->maxnode;
    if (int r = ensure_scratch(c, (size_t) maxnode * 8 + 512)) return r;
    unsigned int *dCnt = (unsigned int *) c->scratch;            // [0] stale nodes, [1] nodes with a usable visible hit
    int64_t *dList = (int64_t *) ((char *) c->scratch + 256);
    char *h, *d;
    if (int r = io_alloc(c, 256, &h, &d)) return r;
    SweepArgs sa{};
    sa.nActive = nActive;
    sa.nDiffAllow = (int64_t) ((double) nActive * c->njCfg.stale_out_limit);
    sa.totdiam = totdiam;
    HIPCHK(c, hipMemsetAsync(dCnt, 0, 8, c->stream));
    if (++c->thTag == 0u) {
        HIPCHK(c, hipMemsetAsync(c->thMark, 0, (size_t) c->d.maxNodes * 4, c->stream));
        c->thTag = 1u;
    }
    const unsigned grid = cdiv(maxnode, VFT_WG);
    // 1. the lazy refreshes (setCriterion inside getVisible, NJ.tcc:1092-1098) of exactly the nodes the reference touches
    if (c->rs == 4) launch((k_nj_reset_stale<float>), dim3(grid), dim3(VFT_WG), 0, c->stream, arena<float>(c), njengine<float>(c), sa, maxnode, c->thMark, c->thTag, dList, dCnt);
    else launch((k_nj_reset_stale<double>), dim3(grid), dim3(VFT_WG), 0, c->stream, arena<double>(c), njengine<double>(c), sa, maxnode, c->thMark, c->thTag, dList, dCnt);
    launch(k_nj_publish_u32, dim3(1), dim3(64), 0, c->stream, (const unsigned int *) dCnt, (unsigned int *) d, 1);
    LAUNCHCHK(c);
    if (int r = wait_stream(c)) return r;
    const int64_t nStale = (int64_t) *(volatile unsigned int *) h;
    if (nStale > 0) {
        if (nActive > c->maxStamp) c->maxStamp = nActive;
        VFT_DISPATCH(c, (launch((k_refresh_list<REAL, NC, false>), dim3(cdiv(nStale, c->pwWaves)), dim3(64 * c->pwWaves), pw_lds_bytes(c), c->stream,
                                arena<REAL>(c), (const int64_t *) dList, nStale, (int64_t) 0, sa)));
    }
    // 2. criteria as a sweep-shaped result in slot 0, 3. the selection of the sweeps
    if (int r = ensure_slots(c, 1)) return r;
    c->nPart = (int) grid;
    c->slots[0].nPart = (int) grid;
    if (c->rs == 4) launch((k_nj_reset_crit<float>), dim3(grid), dim3(VFT_WG), 0, c->stream, arena<float>(c), njengine<float>(c), sa, maxnode, sweepout<float>(c, 0), dCnt + 1);
    else launch((k_nj_reset_crit<double>), dim3(grid), dim3(VFT_WG), 0, c->stream, arena<double>(c), njengine<double>(c), sa, maxnode, sweepout<double>(c, 0), dCnt + 1);
    launch(k_nj_publish_u32, dim3(1), dim3(64), 0, c->stream, (const unsigned int *) dCnt, (unsigned int *) d, 2);
    LAUNCHCHK(c);
    const int64_t query = -1;
    int r;
    if (c->cfg.precision == 4) r = run_select<float, vft_hit_f32>(c, 1, &query, 0, maxnode, k);
    else r = run_select<double, vft_hit_f64>(c, 1, &query, 0, maxnode, k);
    if (r) return r;
    *nVisible = (int64_t) ((volatile unsigned int *) h)[1];
    memcpy(hits, c->hRes + sizeof(SelectHeader), (size_t) k * (c->cfg.precision == 4 ? sizeof(vft_hit_f32) : sizeof(vft_hit_f64)));
    return VFT_OK;
}

template <typename REAL, int NC>
static int nj_enqueue(vft_ctx *c, int64_t joinIndex, int32_t phases, int32_t updateOut) {
    const Arena<REAL> A = arena<REAL>(c);
    NjEngine<REAL> E = njengine<REAL>(c);
    const TopHits<REAL> T = tophits<REAL>(c);
    const size_t pairLds = pw_lds_bytes(c) / c->pwWaves;
    const long long ji = (long long) joinIndex;
    const int64_t newnode = c->d.nSeqs + joinIndex;   // (ids are handed out in join order)
    if (phases & VFT_NJ_PHASE_JOIN) {
        if (newnode >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_nj_engine_enqueue: join %lld out of range", (long long) joinIndex);
        // Room in the pending stash BEFORE anything is launched (a refused call must not have enqueued half a join).  Every
        // pending node must really exist before its tile is rebuilt: the joins enqueued so far have run - or an event has
        // stopped them, which the caller handles first.
        if ((c->pend.empty() || c->pend.back() != newnode) && (int64_t) c->pend.size() == VFT_PEND_MAX) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if ((((unsigned long long) __atomic_load_n(&c->njStatusHost[0], __ATOMIC_ACQUIRE) >> 31) & 7ull) != 0)
                return fail(c, VFT_ERR_HALTED, "vft_nj_engine_enqueue: the engine has halted (handle the event first)");
            if (int r = flush_pending(c)) return r;
            E = njengine<REAL>(c);
        }
    }
    // one hill-climbing round up to its last comparison, which k_nj_glue_join makes
    auto round = [&]() {
        launch((k_nj_best_pairs<REAL, NC>), dim3((unsigned) c->thM), dim3(VFT_WG), pairLds, c->stream, A, E, T, 0);
        launch((k_nj_glue_best<REAL, NC>), dim3(1), dim3(VFT_WG), pairLds, c->stream, A, E, T);
        launch((k_nj_best_pairs<REAL, NC>), dim3((unsigned) c->thM), dim3(VFT_WG), pairLds, c->stream, A, E, T, 1);
    };
    // ... and both walks of a round in one launch, the second one speculatively (k_nj_best_pairs2): what follows a search
    auto roundSpec = [&]() {
        if (++c->njClaimTag == 0u) {
            hipMemsetAsync(c->njClaim, 0, (size_t) c->d.maxNodes * 4, c->stream);
            c->njClaimTag = 1u;
        }
        launch((k_nj_best_pairs2<REAL, NC>), dim3((unsigned) (2 * c->thM)), dim3(VFT_WG), pairLds, c->stream, A, E, T, c->njClaimTag);
    };
    auto glueScan = [&](long long done, long long next) {
        if (c->njTailThreads == VFT_NJ_TAIL)
            launch((k_nj_glue_scan<REAL, NC, VFT_NJ_TAIL>), dim3(1), dim3(VFT_NJ_TAIL), c->njTailLds, c->stream, A, E, T, done, next, c->njP);
        else launch((k_nj_glue_scan<REAL, NC, 1024>), dim3(1), dim3(1024), c->njTailLds, c->stream, A, E, T, done, next, c->njP);
    };
    if (phases & VFT_NJ_PHASE_SEARCH) {
        // (the slot records of the top-visible list: k_nj_merge_rank's second half alone)
        launch((k_nj_merge_rank<REAL>), dim3((unsigned) cdiv(c->njCfg.n_top, VFT_WG)), dim3(VFT_WG), (size_t) c->njP * sizeof(ThKey), c->stream, A, E, T, 0);
        glueScan(-1ll, ji);
        if (!c->njCfg.fastest) roundSpec();
    }
    if (phases & VFT_NJ_PHASE_CLIMB) round();
    if (phases & VFT_NJ_PHASE_JOIN) {
        // host-side bookkeeping of vft_join_fused (a join that is enqueued again after a halt keeps its slot)
        if (c->pend.empty() || c->pend.back() != newnode) c->pend.push_back(newnode);
        if (newnode >= c->maxnode) c->maxnode = newnode + 1;
        if ((int64_t) E.staleStamp > c->maxStamp) c->maxStamp = E.staleStamp;
        const int32_t slot = (int32_t) c->pend.size() - 1;
        // (the round in front of this join was a speculative double walk unless the caller asked for a classic round)
        launch((k_nj_glue_join<REAL, NC>), dim3(1), dim3(VFT_WG_PROF), 2 * pairLds, c->stream, arena<REAL>(c), E, T, ji, updateOut, slot, 1,
               (phases & VFT_NJ_PHASE_CLIMB) ? 0 : 1);
    }
    if (phases & VFT_NJ_PHASE_MERGE) {
        if (++c->thTag == 0u) {
            HIPCHK(c, hipMemsetAsync(c->thMark, 0, (size_t) c->d.maxNodes * 4, c->stream));
            c->thTag = 1u;
        }
        // (the join kernel computes the new node's out-distance itself unless the caller recomputes the out-profile in between)
        if (!updateOut) launch((k_nj_refresh_new<REAL, NC>), dim3(1), dim3(VFT_WG), pairLds, c->stream, A, E);
        launch((k_nj_merge_pairs<REAL, NC>), dim3((unsigned) (2 * c->thM)), dim3(VFT_WG), pairLds, c->stream, A, E, T, ji, c->thTag);
        {
            const int rankBlocks = (int) cdiv(2 * c->thM, VFT_NJ_RANK_PER_WG), prepBlocks = (int) cdiv(c->njCfg.n_top, VFT_WG);
            launch((k_nj_merge_rank<REAL>), dim3((unsigned) (rankBlocks + prepBlocks)), dim3(VFT_WG), (size_t) c->njP * sizeof(ThKey), c->stream, A, E, T, rankBlocks);
        }
        const bool chain = (phases & VFT_NJ_PHASE_NEXT) != 0;
        glueScan(ji, chain ? ji + 1 : -1ll);
        if (chain && !c->njCfg.fastest) roundSpec();
    }
    LAUNCHCHK(c);
    return VFT_OK;
}

extern "C" int vft_nj_engine_enqueue(vft_ctx *c, int64_t joinIndex, int32_t phases, int32_t updateOut) {
    NJ_ENGINE_OK(c);
    if (joinIndex < 0 || joinIndex >= c->d.nSeqs) return VFT_ERR_INVALID;
    int r = VFT_OK;
    VFT_DISPATCH(c, (r = nj_enqueue<REAL, NC>(c, joinIndex, phases, updateOut)));
    return r;
}

extern "C" int vft_nj_engine_poll(vft_ctx *c, int64_t *joinsDone, int32_t *halt, int32_t *haltJoin) {
    NJ_ENGINE_OK(c);
    volatile long long *s = c->njStatusHost;
    const unsigned long long w = (unsigned long long) __atomic_load_n(&s[0], __ATOMIC_ACQUIRE);   // vft_nj_publish: one word
    if (joinsDone) *joinsDone = (int64_t) (w & 0x7FFFFFFFull);
    if (halt) *halt = (int32_t) ((w >> 31) & 7ull);
    if (haltJoin) *haltJoin = (int32_t) (w >> 34);
    return VFT_OK;
}

extern "C" int vft_nj_engine_resume(vft_ctx *c, int64_t nextJoin) {
    NJ_ENGINE_OK(c);
    // joins enqueued behind the event have not run: their pending slots and node ids are handed out again
    const int64_t firstFree = c->d.nSeqs + nextJoin;
    while (!c->pend.empty() && c->pend.back() >= firstFree) c->pend.pop_back();
    if (c->maxnode > firstFree) c->maxnode = firstFree;
    c->njStatusHost[0] &= 0x7FFFFFFFll;   // (the halt bits; the device clears them again when the stream gets there)
    if (c->rs == 4) launch((k_nj_set_state<float>), dim3(1), dim3(1), 0, c->stream, (NjState<float> *) c->njState, -1ll, -1ll, (double) NAN, -1, 1, c->njStatusDev);
    else launch((k_nj_set_state<double>), dim3(1), dim3(1), 0, c->stream, (NjState<double> *) c->njState, -1ll, -1ll, (double) NAN, -1, 1, c->njStatusDev);
    LAUNCHCHK(c);
    return VFT_OK;
}

#ifdef VFT_NJ_TIMING
extern "C" int vft_nj_engine_ticks(vft_ctx *c, unsigned long long *ticks /* 16 */) {   // tools-only build
    hipStreamSynchronize(c->stream);
    return hipMemcpyFromSymbol(ticks, HIP_SYMBOL(vftNjTicks), 16 * sizeof(unsigned long long)) == hipSuccess ? VFT_OK : VFT_ERR_HIP;
}
#endif

extern "C" int vft_nj_engine_log(vft_ctx *c, const vft_nj_join_t **log) {
    NJ_ENGINE_OK(c);
    if (!log) return VFT_ERR_INVALID;
    *log = (const vft_nj_join_t *) c->njLogHost;
    return VFT_OK;
}

extern "C" int vft_nj_engine_adopt(vft_ctx *c, int64_t from, int64_t to) {
    NJ_ENGINE_OK(c);
    if (from < 0 || to > c->d.nSeqs || from > to) return VFT_ERR_INVALID;
    for (int64_t k = from; k < to; k++) {
        const NjJoinRec &r = c->njLogHost[k];
        if (r.i < 0 || r.j < 0 || r.i >= c->d.maxNodes || r.j >= c->d.maxNodes || r.newnode != (int32_t) (c->d.nSeqs + k))
            return fail(c, VFT_ERR_STATE, "vft_nj_engine_adopt: join %lld has not been logged", (long long) k);
        c->hParent[(size_t) r.i] = c->hParent[(size_t) r.j] = r.newnode;
        if ((int64_t) r.newnode >= c->maxnode) c->maxnode = (int64_t) r.newnode + 1;
    }
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- likelihood
// the batch operators under a 20-state matrix model: a quad of lanes per column (vft_kernels_ml.h, k_pair_loglk_quad / k_posterior_quad)
static bool quad_pair_ok(const vft_ctx *c) { return c->d.nCodes == 20 && c->hasTm && c->d.nPos <= 512; }
static bool quad_posterior_ok(const vft_ctx *c) { return c->d.nCodes == 20 && c->hasTm; }
template <typename REAL>
static void launch_pair_loglk_quad(vft_ctx *c, int64_t n, const int64_t *dA, const int64_t *dB, const double *dLen, double *dOut, double *dSite) {
    if (c->d.nPos <= 320)
        launch((k_pair_loglk_quad<REAL, 5>), dim3((unsigned) n), dim3(256), 0, c->stream, arena<REAL>(c), dA, dB, dLen, n, c->minRel, dOut, dSite);
    else
        launch((k_pair_loglk_quad<REAL, 8>), dim3((unsigned) n), dim3(256), 0, c->stream, arena<REAL>(c), dA, dB, dLen, n, c->minRel, dOut, dSite);
}
static void launch_pair_loglk_any(vft_ctx *c, int64_t n, const int64_t *dA, const int64_t *dB, const double *dLen, double *dOut, double *dSite) {
    if (quad_pair_ok(c)) {
        if (c->cfg.precision == 4) launch_pair_loglk_quad<float>(c, n, dA, dB, dLen, dOut, dSite);
        else launch_pair_loglk_quad<double>(c, n, dA, dB, dLen, dOut, dSite);
        return;
    }
    VFT_DISPATCH(c, (launch((k_pair_loglk<REAL, NC>), dim3((unsigned) n), dim3(VFT_ML_WG), 0, c->stream, arena<REAL>(c), dA, dB, dLen, n, c->minRel, dOut, dSite)));
}
// posteriors into dense ML rows (stash-free): one workgroup per node with quads, or the whole-column kernel
static void launch_posterior_rows(vft_ctx *c, int64_t cnt, const int64_t *dOut, const int64_t *dA, const int64_t *dB, const double *dL1, const double *dL2) {
    if (quad_posterior_ok(c)) {
        if (c->cfg.precision == 4) launch((k_posterior_quad<float>), dim3((unsigned) cnt), dim3(256), 0, c->stream, arena<float>(c), dOut, dA, dB, dL1, dL2, c->minLen, c->minRel);
        else launch((k_posterior_quad<double>), dim3((unsigned) cnt), dim3(256), 0, c->stream, arena<double>(c), dOut, dA, dB, dL1, dL2, c->minLen, c->minRel);
        return;
    }
    VFT_DISPATCH(c, {
        const dim3 grid(cdiv(c->d.nPos, VFT_ML_WG), (unsigned) cnt);
        launch((k_posterior<REAL, NC>), grid, dim3(VFT_ML_WG), 0, c->stream, arena<REAL>(c), dOut, dA, dB, dL1, dL2, c->minLen, c->minRel, (REAL *) nullptr);
    });
}

extern "C" int vft_pair_loglk(vft_ctx *c, int64_t n, const int64_t *a, const int64_t *b, const double *length,
                              double *loglk, double *siteLk) {
    if (!c || n < 0 || !a || !b || !length || !loglk) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (!c->hasTm && c->d.nCodes != 4) return fail(c, VFT_ERR_STATE, "amino-acid likelihoods need vft_set_transition_matrix");
    for (int64_t k = 0; k < n; k++)
        if (a[k] < 0 || a[k] >= c->d.maxNodes || b[k] < 0 || b[k] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_pair_loglk: pair %lld out of range", (long long) k);
    const size_t idB = (size_t) n * 8, sB = siteLk ? (size_t) n * c->d.nPos * 8 : 0;
    if (4 * idB <= 2 * VFT_SMALL_BYTES && !siteLk) {
        // a batch of a few thousand pairs (treeLogLk of a 10 000-taxon tree, the per-operator tables): ids and lengths travel through
        // the host-mapped ring and the totals come back through it - three staged copies, a DMA read-back and a stream
        // synchronisation were 80 of the call's 130 us at 8 192 pairs
        char *h, *d;
        if (int r = io_alloc(c, 4 * idB, &h, &d)) return r;
        memcpy(h, a, idB);
        memcpy(h + idB, b, idB);
        memcpy(h + 2 * idB, length, idB);
        launch_pair_loglk_any(c, n, (const int64_t *) d, (const int64_t *) (d + idB), (const double *) (d + 2 * idB), (double *) (d + 3 * idB), (double *) nullptr);
        LAUNCHCHK(c);
        if (int r = wait_stream(c)) return r;
        memcpy(loglk, h + 3 * idB, idB);
        return VFT_OK;
    }
    if (int r = ensure_scratch(c, 4 * idB + sB + 64)) return r;
    char *s = (char *) c->scratch;
    HIPCHK(c, hipMemcpyAsync(s, a, idB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + idB, b, idB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + 2 * idB, length, idB, hipMemcpyHostToDevice, c->stream));
    double *dOut = (double *) (s + 3 * idB);
    double *dSite = siteLk ? (double *) (s + 4 * idB) : nullptr;
    launch_pair_loglk_any(c, n, (const int64_t *) s, (const int64_t *) (s + idB), (const double *) (s + 2 * idB), dOut, dSite);
    LAUNCHCHK(c);
    HIPCHK(c, hipMemcpyAsync(loglk, dOut, idB, hipMemcpyDeviceToHost, c->stream));
    if (siteLk) HIPCHK(c, hipMemcpyAsync(siteLk, dSite, sB, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_posterior_profiles(vft_ctx *c, int64_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                                      const double *len1, const double *len2) {
    if (!c || n < 0 || !out || !a || !b || !len1 || !len2) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (!c->hasTm && c->d.nCodes != 4) return fail(c, VFT_ERR_STATE, "amino-acid posteriors need vft_set_transition_matrix");
    for (int64_t k = 0; k < n; k++) {
        if (int r = internal_ok(c, out[k])) return r;
        if (a[k] < 0 || a[k] >= c->d.maxNodes || b[k] < 0 || b[k] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_posterior_profiles: child %lld out of range", (long long) k);
    }
    const CommitPlan plan = commit_plan(c, n);
    const int64_t chunk = plan.chunk;
    const size_t idB = (size_t) n * 8;
    if (int r = ensure_scratch(c, 5 * idB + plan.totalB + 512)) return r;
    char *s = (char *) c->scratch;
    char *base = s + 5 * idB;
    base += (256 - ((uintptr_t) base & 255)) & 255;
    const bool viaRing = 5 * idB <= 2 * VFT_SMALL_BYTES;   // (five staged copies cost more than the kernel at a few thousand nodes)
    if (viaRing) {
        char *h, *d;
        if (int r = io_alloc(c, 5 * idB, &h, &d)) return r;
        memcpy(h, out, idB);
        memcpy(h + idB, a, idB);
        memcpy(h + 2 * idB, b, idB);
        memcpy(h + 3 * idB, len1, idB);
        memcpy(h + 4 * idB, len2, idB);
        launch(k_copy16x2, dim3(cdiv((int64_t) (5 * idB + 15) / 16, 256)), dim3(256), 0, c->stream, (uint4 *) s, (const uint4 *) d, (int64_t) (5 * idB + 15) / 16,
               (uint4 *) nullptr, (const uint4 *) nullptr, (int64_t) 0);
    } else {
    HIPCHK(c, hipMemcpyAsync(s, out, idB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + idB, a, idB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + 2 * idB, b, idB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + 3 * idB, len1, idB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + 4 * idB, len2, idB, hipMemcpyHostToDevice, c->stream));
    }
    // Row mode with every internal profile a row (the ML stage, vft_set_profile_rows): the results go to the nodes' dense rows - what every
    // reader takes from then on - with no stash and no tile commit, as vft_posterior_profiles_blen writes them
    const bool rows = c->rowMode && c->allRows && c->mlIs != nullptr;
    for (int64_t k0 = 0; k0 < n; k0 += chunk) {
        const int64_t cnt = n - k0 < chunk ? n - k0 : chunk;
        if (rows) {
            launch_posterior_rows(c, cnt, (const int64_t *) s + k0, (const int64_t *) (s + idB) + k0, (const int64_t *) (s + 2 * idB) + k0,
                                  (const double *) (s + 3 * idB) + k0, (const double *) (s + 4 * idB) + k0);
            LAUNCHCHK(c);
            continue;
        }
        VFT_DISPATCH(c, {
            const dim3 grid(cdiv(c->d.nPos, VFT_ML_WG), (unsigned) cnt);
            launch((k_posterior<REAL, NC>), grid, dim3(VFT_ML_WG), 0, c->stream, arena<REAL>(c), (const int64_t *) s + k0,
                   (const int64_t *) (s + idB) + k0, (const int64_t *) (s + 2 * idB) + k0,
                   (const double *) (s + 3 * idB) + k0, (const double *) (s + 4 * idB) + k0, c->minLen, c->minRel,
                   (REAL *) base);
        });
        LAUNCHCHK(c);
        if (int r = commit_nodes(c, plan, out + k0, (const int64_t *) s + k0, cnt, base)) return r;
    }
    if (viaRing) return wait_stream(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- ML branch lengths
// branchlength[] of the tree on the device (numeric_t, NJ.h: branchlength): the ML length optimiser reads and writes it
// in place, so a whole traversal is queued without a host round trip.
static int ensure_blen(vft_ctx *c) {
    if (c->blen) return VFT_OK;
    HIPCHK(c, hipMalloc(&c->blen, (size_t) c->d.maxNodes * c->rs));
    HIPCHK(c, hipMemsetAsync(c->blen, 0, (size_t) c->d.maxNodes * c->rs, c->stream));
    return VFT_OK;
}

extern "C" int vft_branch_lengths_set(vft_ctx *c, int64_t first, int64_t count, const void *values) {
    if (!c || !values || first < 0 || count < 0 || first + count > c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_branch_lengths_set: bad range");
    if (int r = ensure_blen(c)) return r;
    HIPCHK(c, hipMemcpyAsync((char *) c->blen + (size_t) first * c->rs, values, (size_t) count * c->rs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

extern "C" int vft_branch_lengths_get(vft_ctx *c, int64_t first, int64_t count, void *values) {
    if (!c || !values || first < 0 || count < 0 || first + count > c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_branch_lengths_get: bad range");
    if (int r = ensure_blen(c)) return r;
    HIPCHK(c, hipMemcpyAsync(values, (char *) c->blen + (size_t) first * c->rs, (size_t) count * c->rs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// branchlength[] entries by index list, through the host-mapped ring (the lanes of the subtree schedule exchange the lengths their
// rank's share of a batch optimised, host/MLLengths.h "lanes across ranks")
template <typename REAL>
__global__ void k_blen_gather(const REAL *blen, const int64_t *idx, REAL *out, int64_t n) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = blen[idx[k]];
}
template <typename REAL>
__global__ void k_blen_scatter(REAL *blen, const int64_t *idx, const REAL *in, int64_t n) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) blen[idx[k]] = in[k];
}
static int blen_indexed(vft_ctx *c, int64_t n, const int64_t *idx, void *values, bool gather) {
    if (!c || n < 0 || (n > 0 && (!idx || !values))) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (n > 65536) return fail(c, VFT_ERR_INVALID, "vft_branch_lengths_%s: at most 65536 entries per call", gather ? "gather" : "scatter");
    for (int64_t k = 0; k < n; k++)
        if (idx[k] < 0 || idx[k] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "vft_branch_lengths_%s: index out of range", gather ? "gather" : "scatter");
    if (int r = ensure_blen(c)) return r;
    const size_t ib = ((size_t) n * 8 + 255) & ~(size_t) 255, vb = ((size_t) n * c->rs + 255) & ~(size_t) 255;
    char *h, *d;
    if (int r = io_alloc(c, ib + vb, &h, &d)) return r;
    memcpy(h, idx, (size_t) n * 8);
    if (!gather) memcpy(h + ib, values, (size_t) n * c->rs);
    if (c->cfg.precision == 4) {
        if (gather) launch((k_blen_gather<float>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const float *) c->blen, (const int64_t *) d, (float *) (d + ib), n);
        else launch((k_blen_scatter<float>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (float *) c->blen, (const int64_t *) d, (const float *) (d + ib), n);
    } else {
        if (gather) launch((k_blen_gather<double>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const double *) c->blen, (const int64_t *) d, (double *) (d + ib), n);
        else launch((k_blen_scatter<double>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (double *) c->blen, (const int64_t *) d, (const double *) (d + ib), n);
    }
    LAUNCHCHK(c);
    if (!gather) return VFT_OK;   // stream-ordered: the kernels that read the lengths queue behind it
    if (int r = wait_stream(c)) return r;
    memcpy(values, h + ib, (size_t) n * c->rs);
    return VFT_OK;
}
extern "C" int vft_branch_lengths_gather(vft_ctx *c, int64_t n, const int64_t *idx, void *values) { return blen_indexed(c, n, idx, values, true); }
extern "C" int vft_branch_lengths_scatter(vft_ctx *c, int64_t n, const int64_t *idx, const void *values) {
    return blen_indexed(c, n, idx, const_cast<void *>(values), false);
}

template <typename REAL>
__global__ void k_gather_lengths(const REAL *blen, const int64_t *li1, const int64_t *li2, double *l1, double *l2, int64_t n) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    l1[k] = (double) blen[li1[k]];
    l2[k] = (double) blen[li2[k]];
}

// posteriorProfile for a batch of independent nodes with the two branch lengths taken from the device array
// (recomputeMLProfiles level by level, up-profiles of a traversal).  Stream-ordered: does not wait.
extern "C" int vft_posterior_profiles_blen(vft_ctx *c, int64_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                                           const int64_t *lenIdxA, const int64_t *lenIdxB) {
    if (!c || n < 0 || !out || !a || !b || !lenIdxA || !lenIdxB) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (!c->hasTm && c->d.nCodes != 4) return fail(c, VFT_ERR_STATE, "amino-acid posteriors need vft_set_transition_matrix");
    for (int64_t k = 0; k < n; k++) {
        if (int r = internal_ok(c, out[k])) return r;
        if (lenIdxA[k] < 0 || lenIdxA[k] >= c->d.maxNodes || lenIdxB[k] < 0 || lenIdxB[k] >= c->d.maxNodes || a[k] < 0 ||
            a[k] >= c->d.maxNodes || b[k] < 0 || b[k] >= c->d.maxNodes)
            return fail(c, VFT_ERR_INVALID, "vft_posterior_profiles_blen: index out of range");
    }
    if (int r = ensure_blen(c)) return r;
    if (int r = ensure_ml_rows(c)) return r;
    const size_t idB = (size_t) n * 8;
    const bool smallIds = 7 * idB <= VFT_SMALL_BYTES;
    char *s;
    if (smallIds) {
        char *h;
        if (int r = io_alloc(c, 7 * idB, &h, &s)) return r;
        memcpy(h, out, idB);
        memcpy(h + idB, a, idB);
        memcpy(h + 2 * idB, b, idB);
        memcpy(h + 3 * idB, lenIdxA, idB);
        memcpy(h + 4 * idB, lenIdxB, idB);
    } else {
        if (int r = ensure_scratch(c, 7 * idB + 512)) return r;
        s = (char *) c->scratch;
        HIPCHK(c, hipMemcpyAsync(s, out, idB, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + idB, a, idB, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + 2 * idB, b, idB, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + 3 * idB, lenIdxA, idB, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + 4 * idB, lenIdxB, idB, hipMemcpyHostToDevice, c->stream));
    }
    double *l1 = (double *) (s + 5 * idB), *l2 = (double *) (s + 6 * idB);
    if (c->cfg.precision == 4)
        launch((k_gather_lengths<float>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const float *) c->blen,
               (const int64_t *) (s + 3 * idB), (const int64_t *) (s + 4 * idB), l1, l2, n);
    else
        launch((k_gather_lengths<double>), dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const double *) c->blen,
               (const int64_t *) (s + 3 * idB), (const int64_t *) (s + 4 * idB), l1, l2, n);
    // the results go to the nodes' dense ML rows: no stash, no tile commit
    const int64_t chunk = 32768;
    for (int64_t k0 = 0; k0 < n; k0 += chunk) {
        const int64_t cnt = n - k0 < chunk ? n - k0 : chunk;
        launch_posterior_rows(c, cnt, (const int64_t *) s + k0, (const int64_t *) (s + idB) + k0, (const int64_t *) (s + 2 * idB) + k0, (const double *) l1 + k0,
                              (const double *) l2 + k0);
        LAUNCHCHK(c);
    }
    if (!smallIds) HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// n posteriorProfile calls in order in one launch (k_posterior_chain), lengths from the device's branchlength[];
// nChains independent chains as one launch (blockIdx.y), split by chainOff[nChains + 1] (NULL: one chain)
static int posterior_chains(vft_ctx *c, int32_t nChains, const int32_t *chainOff, int32_t n, const int64_t *out, const int64_t *a,
                            const int64_t *b, const int64_t *lenIdxA, const int64_t *lenIdxB, const char *who) {
    if (n == 0) return VFT_OK;
    if (!c->hasTm && c->d.nCodes != 4) return fail(c, VFT_ERR_STATE, "amino-acid posteriors need vft_set_transition_matrix");
    for (int32_t k = 0; k < n; k++) {
        if (int r = internal_ok(c, out[k])) return r;
        if (a[k] < 0 || a[k] >= c->d.maxNodes || b[k] < 0 || b[k] >= c->d.maxNodes || lenIdxA[k] < 0 || lenIdxA[k] >= c->d.maxNodes ||
            lenIdxB[k] < 0 || lenIdxB[k] >= c->d.maxNodes)
            return fail(c, VFT_ERR_INVALID, "%s: index out of range", who);
    }
    if (int r = ensure_blen(c)) return r;
    if (int r = ensure_ml_rows(c)) return r;
    const size_t idB = (size_t) n * 8, dB = ((size_t) n + 7) & ~(size_t) 7, offB = chainOff ? (size_t) (nChains + 1) * 4 : 0;
    char *h, *s;
    if (int r = io_alloc(c, 5 * idB + dB + offB, &h, &s)) return r;
    memcpy(h, out, idB);
    memcpy(h + idB, a, idB);
    memcpy(h + 2 * idB, b, idB);
    memcpy(h + 3 * idB, lenIdxA, idB);
    memcpy(h + 4 * idB, lenIdxB, idB);
    uint8_t *direct = (uint8_t *) (h + 5 * idB);
    for (int32_t ch = 0; ch < nChains; ch++) {
        const int32_t k0 = chainOff ? chainOff[ch] : 0, k1 = chainOff ? chainOff[ch + 1] : n;
        for (int32_t k = k0; k < k1; k++) {
            uint8_t d = 0;
            for (int32_t j = k0; j < k; j++) {
                if (out[j] == a[k]) d |= 1;
                if (out[j] == b[k]) d |= 2;
            }
            direct[k] = d;
        }
    }
    if (chainOff) memcpy(h + 5 * idB + dB, chainOff, offB);
    VFT_DISPATCH(c, launch((k_posterior_chain<REAL, NC>), dim3(cdiv(c->d.nPos, VFT_ML_WG), (unsigned) nChains), dim3(VFT_ML_WG), 0, c->stream,
                           arena<REAL>(c), (const int64_t *) s, (const int64_t *) (s + idB), (const int64_t *) (s + 2 * idB),
                           (const int64_t *) (s + 3 * idB), (const int64_t *) (s + 4 * idB), (const uint8_t *) (s + 5 * idB), n,
                           (const REAL *) c->blen, c->minLen, c->minRel, chainOff ? (const int32_t *) (s + 5 * idB + dB) : nullptr));
    if (!c->allRows) launch(k_mark_rows, dim3(cdiv(n, 64)), dim3(64), 0, c->stream, c->mlIs, (const int64_t *) s, n, c->d.nSeqs);
    LAUNCHCHK(c);
    return VFT_OK;
}

extern "C" int vft_posterior_chain_blen(vft_ctx *c, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                                        const int64_t *lenIdxA, const int64_t *lenIdxB) {
    if (!c || n < 0 || !out || !a || !b || !lenIdxA || !lenIdxB) return VFT_ERR_INVALID;
    if (n > 256) return fail(c, VFT_ERR_INVALID, "vft_posterior_chain_blen: at most 256 posteriors per call");
    return posterior_chains(c, 1, nullptr, n, out, a, b, lenIdxA, lenIdxB, "vft_posterior_chain_blen");
}

extern "C" int vft_posterior_chains_blen(vft_ctx *c, int32_t nChains, const int32_t *chainOff, const int64_t *out, const int64_t *a,
                                         const int64_t *b, const int64_t *lenIdxA, const int64_t *lenIdxB) {
    if (!c || !chainOff || !out || !a || !b || !lenIdxA || !lenIdxB) return VFT_ERR_INVALID;
    if (int r = chains_ok(c, nChains, chainOff, "vft_posterior_chains_blen")) return r;
    return posterior_chains(c, nChains, chainOff, chainOff[nChains], out, a, b, lenIdxA, lenIdxB, "vft_posterior_chains_blen");
}

static inline int mlopt_wg(const vft_ctx *c) { return c->d.nCodes == 20 ? MlOptWG<20>::value : MlOptWG<4>::value; }

// Columns per thread of the two line-search kernels (0: the alignment is too long for them).  Proteins under a matrix model up
// to 512 columns: a quad of lanes per column (`quad`; one or four passes of 128 columns); otherwise whole columns per thread.
static inline int mlopt_cpt(const vft_ctx *c, bool &quad) {
    quad = false;
    if (c->d.nCodes == 20 && c->hasTm && c->d.nPos <= 8 * (MlLineWG<20, true>::value / 4)) {
        quad = true;
        const int64_t per = cdiv(c->d.nPos, MlLineWG<20, true>::value / 4);
        return per <= 2 ? 2 : per <= 5 ? 5 : 8;
    }
    const int64_t per = cdiv(c->d.nPos, mlopt_wg(c));
    if (c->d.nCodes == 20) return per <= 4 ? 4 : 0;
    return per <= 1 ? 1 : per <= 4 ? 4 : per <= 8 ? 8 : 0;   // (two columns per thread run the four-column kernel: fewer instantiations)
}

// workspaces of the line-search kernels for long alignments (vft_kernels_ml_long.h): nWG x stride bytes, grown as needed
static int ensure_ml_long_ws(vft_ctx *c, size_t nWG, int nRows, size_t *stride) {
    *stride = (vft_ml_long_ws_bytes(c->d.nPos, c->d.nCodes, c->rs, nRows) + 255) & ~(size_t) 255;
    const size_t bytes = nWG * *stride;
    if (bytes <= c->mlLongWsBytes) return VFT_OK;
    if (c->mlLongWs) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->mlLongWs));
        c->mlLongWs = nullptr;
        c->mlLongWsBytes = 0;
    }
    HIPCHK(c, hipMalloc((void **) &c->mlLongWs, bytes));
    c->mlLongWsBytes = bytes;
    return VFT_OK;
}

template <typename REAL, int NC>
static int ml_optimize_launch(vft_ctx *c, int64_t n, int cpt, bool quad, const int64_t *dIds, const int64_t *dLi, const int64_t *dRec,
                              double ftol, double atol) {
    if (cpt == 0 || c->mlLong) {   // more columns than the register-resident instances hold: the workspace kernel
        size_t stride;
        if (int r = ensure_ml_long_ws(c, (size_t) n, 1, &stride)) return r;
        launch((k_ml_node_lengths_long<REAL, NC>), dim3((unsigned) n), dim3(MlOptWG<NC>::value), 0, c->stream, arena<REAL>(c), dIds, dLi, dRec,
               (REAL *) c->blen, c->minLen, c->minRel, ftol, atol, c->mlEvals, c->mlLongWs, stride);
        return VFT_OK;
    }
#define VFT_MLOPT_GO(CPT, QUAD)                                                                                         \
    launch((k_ml_node_lengths<REAL, NC, CPT, QUAD>), dim3((unsigned) n), dim3(MlLineWG<NC, QUAD>::value), 0, c->stream, arena<REAL>(c), \
           dIds, dLi, dRec, (REAL *) c->blen, c->minLen, c->minRel, ftol, atol, c->mlEvals)
    if constexpr (NC == 20) {
        if (quad && cpt == 2) VFT_MLOPT_GO(2, true);
        else if (quad && cpt == 5) VFT_MLOPT_GO(5, true);
        else if (quad && cpt == 8) VFT_MLOPT_GO(8, true);
        else if (!quad && cpt == 4) VFT_MLOPT_GO(4, false);
        else return fail(c, VFT_ERR_INVALID, "vft_ml_optimize_splits: alignment too long for the in-kernel optimiser");
    } else {
        if (cpt == 1) VFT_MLOPT_GO(1, false);
        else if (cpt == 4) VFT_MLOPT_GO(4, false);
        else if (cpt == 8) VFT_MLOPT_GO(8, false);
        else return fail(c, VFT_ERR_INVALID, "vft_ml_optimize_splits: alignment too long for the in-kernel optimiser");
    }
#undef VFT_MLOPT_GO
    return VFT_OK;
}

// The inner step of optimizeAllBranchLengths (NJ.tcc:5025-5064) for n independent splits: ids[3k..3k+2] are the three
// profiles around split k (two children + the up-profile, or the three children of the root), len_idx[3k..] the
// branch-length slots they own, recompute[k] the node whose posterior profile is rebuilt afterwards from ids[3k],
// ids[3k+1] and their new lengths (-1: none, the root; all entries of a call must agree on that).  Stream-ordered.
extern "C" int vft_ml_optimize_splits(vft_ctx *c, int64_t n, const int64_t *ids, const int64_t *lenIdx,
                                      const int64_t *recompute, double ftol, double atol) {
    if (!c || n < 0 || !ids || !lenIdx || !recompute) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (!c->hasTm && c->d.nCodes != 4) return fail(c, VFT_ERR_STATE, "amino-acid likelihoods need vft_set_transition_matrix");
    const bool rec = recompute[0] >= 0;
    for (int64_t k = 0; k < n; k++) {
        if ((recompute[k] >= 0) != rec) return fail(c, VFT_ERR_INVALID, "vft_ml_optimize_splits: mixed recompute flags");
        if (rec)
            if (int r = internal_ok(c, recompute[k])) return r;
        for (int t = 0; t < 3; t++)
            if (ids[3 * k + t] < 0 || ids[3 * k + t] >= c->maxnode || lenIdx[3 * k + t] < 0 || lenIdx[3 * k + t] >= c->d.maxNodes)
                return fail(c, VFT_ERR_INVALID, "vft_ml_optimize_splits: split %lld out of range", (long long) k);
    }
    if (int r = ensure_blen(c)) return r;
    if (!c->mlEvals) {
        HIPCHK(c, hipMalloc((void **) &c->mlEvals, sizeof(unsigned int)));
        HIPCHK(c, hipMemsetAsync(c->mlEvals, 0, sizeof(unsigned int), c->stream));
    }
    if (int r = ensure_ml_rows(c)) return r;
    bool quad;
    const int cpt = mlopt_cpt(c, quad);
    const size_t idB = (size_t) n * 8;
    if (7 * idB > VFT_SMALL_BYTES) return fail(c, VFT_ERR_INVALID, "vft_ml_optimize_splits: too many splits per call");
    char *h, *s;
    if (int r = io_alloc(c, 7 * idB, &h, &s)) return r;
    memcpy(h, ids, 3 * idB);
    memcpy(h + 3 * idB, lenIdx, 3 * idB);
    memcpy(h + 6 * idB, recompute, idB);
    int r = VFT_OK;
    VFT_DISPATCH(c, (r = ml_optimize_launch<REAL, NC>(c, n, cpt, quad, (const int64_t *) s, (const int64_t *) (s + 3 * idB),
                                                      (const int64_t *) (s + 6 * idB), ftol, atol)));
    if (r) return r;
    LAUNCHCHK(c);
    return VFT_OK;
}

template <typename REAL, int NC>
static int ml_quartet_launch(vft_ctx *c, int64_t n, int cpt, bool quad, const int64_t *dIds, const int64_t *dLi, double ftol, double atol,
                             double closeLimit, int mlAccuracy, int mode, double *dLoglk, double *dSite, double *dLen,
                             QuartetNNIResult *dNni, QuartetNNIState *dState = nullptr) {
    if (cpt == 0 || c->mlLong) {   // more columns than the register-resident instances hold: the workspace kernel
        const unsigned gy = mode == 2 ? 3u : 1u;
        size_t stride;
        if (int r = ensure_ml_long_ws(c, (size_t) n * gy, 3, &stride)) return r;
        launch((k_ml_quartet_long<REAL, NC>), dim3((unsigned) n, gy), dim3(MlOptWG<NC>::value), 0, c->stream, arena<REAL>(c), dIds, dLi,
               (REAL *) c->blen, c->minLen, c->minRel, ftol, atol, closeLimit, mlAccuracy, mode, dLoglk, dSite, dLen, dNni, dState, c->mlEvals,
               c->mlLongWs, stride);
        return VFT_OK;
    }
#define VFT_MLQ_GO(CPT, QUAD)                                                                                           \
    launch((k_ml_quartet<REAL, NC, CPT, QUAD>), dim3((unsigned) n, mode == 2 ? 3u : 1u), dim3(MlLineWG<NC, QUAD>::value), 0, c->stream, \
           arena<REAL>(c), dIds, dLi, (REAL *) c->blen, c->minLen, c->minRel, ftol, atol, closeLimit, mlAccuracy, mode,  \
           dLoglk, dSite, dLen, dNni, dState, c->mlEvals)
    if constexpr (NC == 20) {
        if (quad && cpt == 2) VFT_MLQ_GO(2, true);
        else if (quad && cpt == 5) VFT_MLQ_GO(5, true);
        else if (quad && cpt == 8) VFT_MLQ_GO(8, true);
        else if (!quad && cpt == 4) VFT_MLQ_GO(4, false);
        else return fail(c, VFT_ERR_INVALID, "alignment too long for the in-kernel quartet optimiser");
    } else {
        if (cpt == 1) VFT_MLQ_GO(1, false);
        else if (cpt == 4) VFT_MLQ_GO(4, false);
        else if (cpt == 8) VFT_MLQ_GO(8, false);   // nucleotides up to 2048 columns (16S-length alignments)
        else return fail(c, VFT_ERR_INVALID, "alignment too long for the in-kernel quartet optimiser");
    }
#undef VFT_MLQ_GO
    return VFT_OK;
}

static int quartet_args_ok(vft_ctx *c, int64_t n, const int64_t *ids, const int64_t *lenIdx, const char *who) {
    for (int64_t k = 0; k < n; k++) {
        for (int t = 0; t < 4; t++)
            if (ids[4 * k + t] < 0 || ids[4 * k + t] >= c->maxnode) return fail(c, VFT_ERR_INVALID, "%s: quartet %lld out of range", who, (long long) k);
        for (int t = 0; t < 5; t++)
            if (lenIdx[5 * k + t] < 0 || lenIdx[5 * k + t] >= c->d.maxNodes) return fail(c, VFT_ERR_INVALID, "%s: quartet %lld out of range", who, (long long) k);
    }
    return VFT_OK;
}

// MLQuartetNNI (NJ.tcc:4885-5004) for n independent quartets (DoNNI evaluates one at a time: n = 1): results come back
// through mapped host memory; the chosen pairing's branch lengths are written to the device's branchlength[].
static int ml_quartet_nni(vft_ctx *c, int64_t n, const int64_t *ids, const int64_t *lenIdx, double ftol, double atol,
                          double closeLimit, int32_t mlAccuracy, int32_t flags, vft_quartet_nni *results) {
    if (!c || n < 1 || !ids || !lenIdx || !results) return VFT_ERR_INVALID;
    if (!c->hasTm && c->d.nCodes != 4) return fail(c, VFT_ERR_STATE, "amino-acid likelihoods need vft_set_transition_matrix");
    if (int r = quartet_args_ok(c, n, ids, lenIdx, "vft_ml_quartet_nni")) return r;
    if (int r = ensure_blen(c)) return r;
    if (!c->mlEvals) {
        HIPCHK(c, hipMalloc((void **) &c->mlEvals, sizeof(unsigned int)));
        HIPCHK(c, hipMemsetAsync(c->mlEvals, 0, sizeof(unsigned int), c->stream));
    }
    bool quad;
    const int cpt = mlopt_cpt(c, quad);
    const size_t idB = (size_t) n * 8, resB = (size_t) n * sizeof(QuartetNNIResult);
    static_assert(sizeof(QuartetNNIResult) == sizeof(vft_quartet_nni), "result record layout");
    const size_t stB = ((size_t) n * sizeof(QuartetNNIState) + 255) & ~(size_t) 255;
    if (9 * idB + resB + stB > VFT_SMALL_BYTES) return fail(c, VFT_ERR_INVALID, "vft_ml_quartet_nni: too many quartets per call");
    const size_t off = (9 * idB + 255) & ~(size_t) 255;
    char *h, *s;
    if (int r = io_alloc(c, off + ((resB + 255) & ~(size_t) 255), &h, &s)) return r;
    memcpy(h, ids, 4 * idB);
    memcpy(h + 4 * idB, lenIdx, 5 * idB);
    if (int r = ensure_scratch(c, stB + 512)) return r;
    QuartetNNIState *dState = (QuartetNNIState *) c->scratch;
    const int64_t *dIds = (const int64_t *) s, *dLi = (const int64_t *) (s + 4 * idB);
    // init -> {round (a workgroup per pairing) -> decide} x rounds -> verdict, all queued; one wait at the end
    const int nRounds = mlAccuracy < 2 ? 2 : mlAccuracy;
    // In mode 2 the quartet kernel uses its close_limit for the star-topology test alone ("is the internal branch worth more
    // than close_limit?", NJ.tcc:1691-1700; the alternatives are weighed by k_ml_nni_decide): a limit no likelihood
    // difference reaches switches the test off without another kernel instance.
    const double starLimit = (flags & VFT_QUARTET_NO_STAR_TEST) ? 1.0e300 : closeLimit;
    const dim3 g1(cdiv(n, 64)), b1(64);
    if (c->cfg.precision == 4) launch((k_ml_nni_init<float>), g1, b1, 0, c->stream, dLi, (const float *) c->blen, dState, n);
    else launch((k_ml_nni_init<double>), g1, b1, 0, c->stream, dLi, (const double *) c->blen, dState, n);
    for (int round = 0; round < nRounds; round++) {
        int r = VFT_OK;
        VFT_DISPATCH(c, (r = ml_quartet_launch<REAL, NC>(c, n, cpt, quad, dIds, dLi, ftol, atol, starLimit, mlAccuracy, 2, nullptr, nullptr,
                                                         nullptr, nullptr, dState)));
        if (r) return r;
        launch(k_ml_nni_decide, g1, b1, 0, c->stream, dState, n, c->minLen, closeLimit, (int) mlAccuracy, round == nRounds - 1 ? 1 : 0);
    }
    if (c->cfg.precision == 4)
        launch((k_ml_nni_verdict<float>), g1, b1, 0, c->stream, (const QuartetNNIState *) dState, dLi, (float *) c->blen, (QuartetNNIResult *) (s + off), n);
    else
        launch((k_ml_nni_verdict<double>), g1, b1, 0, c->stream, (const QuartetNNIState *) dState, dLi, (double *) c->blen, (QuartetNNIResult *) (s + off), n);
    LAUNCHCHK(c);
    if (int w = wait_stream(c)) return w;
    memcpy(results, h + off, resB);
    return VFT_OK;
}

extern "C" int vft_ml_quartet_nni(vft_ctx *c, int64_t n, const int64_t *ids, const int64_t *lenIdx, double ftol, double atol,
                                  double closeLimit, int32_t mlAccuracy, vft_quartet_nni *results) {
    return ml_quartet_nni(c, n, ids, lenIdx, ftol, atol, closeLimit, mlAccuracy, 0, results);
}

extern "C" int vft_ml_quartet_nni_flags(vft_ctx *c, int64_t n, const int64_t *ids, const int64_t *lenIdx, double ftol, double atol,
                                        double closeLimit, int32_t mlAccuracy, int32_t flags, vft_quartet_nni *results) {
    return ml_quartet_nni(c, n, ids, lenIdx, ftol, atol, closeLimit, mlAccuracy, flags, results);
}

// testSplitsML (NJ.tcc:6800-6999) for n independent splits: ids[4k..] = A, B (children), C, D (sibling side / up-profile)
// of setupABCD, len_idx[5k..] = the branchlength[] slots of A, B, C, D and the split's own branch.  loglk[3k..] receives
// the quartet log-likelihoods of AB|CD (current lengths), AC|BD and AD|BC (lengths optimised); with n_boot > 0 and
// col ([n_boot][n_pos] resampled column indices) support[k] = SHSupport (NJ.tcc:1126-1165), the caller zeroes the
// supports of bad splits.  lengths (may be NULL): the optimised [2][5] lengths of the two alternatives.
extern "C" int vft_ml_split_tests(vft_ctx *c, int64_t n, const int64_t *ids, const int64_t *lenIdx, double ftol, double atol,
                                  double closeLimit, int32_t alwaysSecondPass, double *loglk, int32_t nBoot, const int32_t *col,
                                  double *support, double *lengths) {
    if (!c || n < 0 || !ids || !lenIdx || !loglk || (nBoot > 0 && (!col || !support))) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (!c->hasTm && c->d.nCodes != 4) return fail(c, VFT_ERR_STATE, "amino-acid likelihoods need vft_set_transition_matrix");
    if (int r = quartet_args_ok(c, n, ids, lenIdx, "vft_ml_split_tests")) return r;
    if (int r = ensure_blen(c)) return r;
    if (!c->mlEvals) {
        HIPCHK(c, hipMalloc((void **) &c->mlEvals, sizeof(unsigned int)));
        HIPCHK(c, hipMemsetAsync(c->mlEvals, 0, sizeof(unsigned int), c->stream));
    }
    const int64_t nPos = c->d.nPos;
    bool quad;
    const int cpt = mlopt_cpt(c, quad);
    if (nBoot > 0 && nPos > 65535) return fail(c, VFT_ERR_INVALID, "vft_ml_split_tests: alignment too long for the SH resampling kernel");
    // k_sh_support keeps a split's 3 x nPos per-site log-likelihoods in the LDS while they fit (160 KB per workgroup on gfx950: 6 800 columns)
    const size_t shBytes = (size_t) 3 * nPos * sizeof(double);
    const bool shLds = shBytes <= (size_t) 150 << 10;
    if (nBoot > 0 && shLds && shBytes > 48000 && shBytes > c->shLdsSet) {
        HIPCHK(c, hipFuncSetAttribute((const void *) k_sh_support, hipFuncAttributeMaxDynamicSharedMemorySize, (int) shBytes));
        c->shLdsSet = shBytes;
    }
    // chunks of splits: per split 3 x nPos site log-likelihoods
    int64_t chunk = (int64_t) ((512u << 20) / ((size_t) 3 * nPos * sizeof(double)));
    chunk = chunk < 1 ? 1 : chunk > n ? n : chunk;
    const size_t idB = (size_t) chunk * 8;
    const size_t colB = nBoot > 0 ? (((size_t) nBoot * nPos * 2 + 255) & ~(size_t) 255) : 0;
    const size_t siteB = (size_t) chunk * 3 * nPos * sizeof(double);
    // scratch: ids[4] | lenIdx[5] | loglk[3] | lengths[10] | support[1] | colT | site
    const size_t headB = (23 * idB + 255) & ~(size_t) 255;
    if (int r = ensure_scratch(c, headB + colB + siteB + 512)) return r;
    char *s = (char *) c->scratch;
    uint16_t *dCol = (uint16_t *) (s + headB);
    double *dSite = (double *) (s + headB + colB);
    if (nBoot > 0) {
        std::vector<uint16_t> colT((size_t) nBoot * nPos);
        for (int32_t b = 0; b < nBoot; b++)
            for (int64_t j = 0; j < nPos; j++) {
                const int32_t v = col[(size_t) b * nPos + j];
                if (v < 0 || v >= nPos) return fail(c, VFT_ERR_INVALID, "vft_ml_split_tests: resampled column out of range");
                colT[(size_t) j * nBoot + b] = (uint16_t) v;
            }
        HIPCHK(c, hipMemcpyAsync(dCol, colT.data(), colT.size() * 2, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int64_t k0 = 0; k0 < n; k0 += chunk) {
        const int64_t cnt = n - k0 < chunk ? n - k0 : chunk;
        const size_t cB = (size_t) cnt * 8;
        int64_t *dIds = (int64_t *) s, *dLi = (int64_t *) (s + 4 * idB);
        double *dLoglk = (double *) (s + 9 * idB), *dLen = (double *) (s + 12 * idB), *dSup = (double *) (s + 22 * idB);
        HIPCHK(c, hipMemcpyAsync(dIds, ids + 4 * k0, 4 * cB, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dLi, lenIdx + 5 * k0, 5 * cB, hipMemcpyHostToDevice, c->stream));
        int r = VFT_OK;
        VFT_DISPATCH(c, (r = ml_quartet_launch<REAL, NC>(c, cnt, cpt, quad, dIds, dLi, ftol, atol, closeLimit, alwaysSecondPass ? 2 : 1, 0,
                                                         dLoglk, dSite, dLen, nullptr)));
        if (r) return r;
        LAUNCHCHK(c);
        if (nBoot > 0) {
            launch(k_sh_support, dim3((unsigned) cnt), dim3(256), shLds ? shBytes : (size_t) 0, c->stream, (const double *) dSite,
                   (const double *) dLoglk, (const uint16_t *) dCol, nPos, nBoot, dSup, shLds ? 1 : 0);
            LAUNCHCHK(c);
            HIPCHK(c, hipMemcpyAsync(support + k0, dSup, cB, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipMemcpyAsync(loglk + 3 * k0, dLoglk, 3 * cB, hipMemcpyDeviceToHost, c->stream));
        if (lengths) HIPCHK(c, hipMemcpyAsync(lengths + 10 * k0, dLen, 10 * cB, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VFT_OK;
}

// number of likelihood evaluations the optimiser has made since the last call (the reference's nLkCompute share)
extern "C" int vft_ml_eval_count(vft_ctx *c, int64_t *evals) {
    if (!c || !evals) return VFT_ERR_INVALID;
    *evals = 0;
    if (!c->mlEvals) return VFT_OK;
    unsigned int v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, c->mlEvals, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->mlEvals, 0, sizeof(unsigned int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *evals = (int64_t) v;
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- supports
extern "C" int vft_split_supports(vft_ctx *c, int64_t n, const int64_t *a, const int64_t *b, const int64_t *cc,
                                  const int64_t *d, int32_t nBoot, const int32_t *col, double *support) {
    if (!c || n < 0 || !a || !b || !cc || !d || nBoot < 1 || !col || !support) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    for (int64_t k = 0; k < n; k++)
        if (a[k] < 0 || a[k] >= c->maxnode || b[k] < 0 || b[k] >= c->maxnode || cc[k] < 0 || cc[k] >= c->maxnode || d[k] < 0 ||
            d[k] >= c->maxnode)
            return fail(c, VFT_ERR_INVALID, "vft_split_supports: quartet %lld out of range", (long long) k);
    const int64_t nPos = c->d.nPos;
    const size_t lds = (size_t) 12 * nPos * sizeof(double);
    if (lds > (160u << 10)) return fail(c, VFT_ERR_INVALID, "vft_split_supports: alignment too long (%lld columns, limit 1706)", (long long) nPos);
    // the resample table, transposed to [nPos][nBoot]
    std::vector<int32_t> colT((size_t) nPos * nBoot);
    for (int32_t r = 0; r < nBoot; r++)
        for (int64_t i = 0; i < nPos; i++) {
            const int32_t v = col[(size_t) r * nPos + i];
            if (v < 0 || v >= nPos) return fail(c, VFT_ERR_INVALID, "vft_split_supports: column index out of range");
            colT[(size_t) i * nBoot + r] = v;
        }
    // Nodes go in chunks small enough that even if EVERY resample of every node of a chunk were a near-tie the flagged
    // buffer could not overflow (zero-distance quartets of near-duplicate sequences tie in all their resamples).
    const unsigned int flagCap = 1u << 22;   // 4M records = 235 MB
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t) flagCap / nBoot));
    const size_t idB = (((size_t) chunk * 8) + 255) & ~(size_t) 255, colB = ((colT.size() * 4) + 255) & ~(size_t) 255;
    const size_t cntB = (((size_t) chunk * 4) + 255) & ~(size_t) 255;
    const size_t flagB = (size_t) flagCap * VFT_SUPPORT_REC * 8;
    if (int r = ensure_scratch(c, 4 * idB + colB + cntB + 256 + flagB + 512)) return r;
    char *s = (char *) c->scratch;
    HIPCHK(c, hipMemcpyAsync(s + 4 * idB, colT.data(), colT.size() * 4, hipMemcpyHostToDevice, c->stream));
    unsigned int *dCnt = (unsigned int *) (s + 4 * idB + colB);
    unsigned int *dNFlag = (unsigned int *) (s + 4 * idB + colB + cntB);
    double *dFlag = (double *) (s + 4 * idB + colB + cntB + 256);
    const int32_t scoredist = (c->cfg.n_codes == 4 && !c->hasDm) ? 0 : 1;   // logCorrect's choice, NJ.tcc:324
    // margins below eps are decided on the host with its own libm (the device's log may differ in the last bit);
    // a distance is < 3, so errors of a few 1e-16 cannot reach 1e-9
    const double eps = 1e-9;
    VFT_DISPATCH(c, {
        if (lds > (48u << 10))
            HIPCHK(c, hipFuncSetAttribute((const void *) k_split_support<REAL, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    });
    std::vector<unsigned int> cnt((size_t) chunk);
    std::vector<double> rec;
    for (int64_t k0 = 0; k0 < n; k0 += chunk) {
        const int64_t m = std::min<int64_t>(chunk, n - k0);
        HIPCHK(c, hipMemcpyAsync(s, a + k0, (size_t) m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + idB, b + k0, (size_t) m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + 2 * idB, cc + k0, (size_t) m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(s + 3 * idB, d + k0, (size_t) m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(dNFlag, 0, 4, c->stream));
        VFT_DISPATCH(c, (launch((k_split_support<REAL, NC>), dim3((unsigned) m), dim3(VFT_SUPPORT_WG), lds, c->stream,
                                arena<REAL>(c), (const int64_t *) s, (const int64_t *) (s + idB), (const int64_t *) (s + 2 * idB),
                                (const int64_t *) (s + 3 * idB), m, (const int32_t *) (s + 4 * idB), nBoot, scoredist, eps, dCnt,
                                dNFlag, dFlag, flagCap)));
        LAUNCHCHK(c);
        unsigned int nFlag = 0;
        HIPCHK(c, hipMemcpyAsync(cnt.data(), dCnt, (size_t) m * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&nFlag, dNFlag, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (nFlag > flagCap) return fail(c, VFT_ERR_STATE, "vft_split_supports: flagged-resample buffer overflow (%u)", nFlag);
        if (nFlag) {
            rec.resize((size_t) nFlag * VFT_SUPPORT_REC);
            HIPCHK(c, hipMemcpy(rec.data(), dFlag, rec.size() * 8, hipMemcpyDeviceToHost));
            for (unsigned int f = 0; f < nFlag; f++) {
                const double *r = &rec[(size_t) f * VFT_SUPPORT_REC];
                double dd[6];
                for (int j = 0; j < 6; j++) {   // logCorrect (NJ.tcc:322-330) with the host's log
                    double x = r[1 + j];
                    if (scoredist) x = x < 0.99 ? -1.3 * std::log(1.0 - x) : 3.0;
                    else x = x < 0.74 ? -0.75 * std::log(1.0 - x * 4.0 / 3.0) : 3.0;
                    dd[j] = x < 3.0 ? x : 3.0;
                }
                const double s1 = dd[1] + dd[4] - dd[0] - dd[5], s2 = dd[2] + dd[3] - dd[0] - dd[5];
                if (s1 > 0 && s2 > 0) cnt[(size_t) r[0]]++;
            }
        }
        for (int64_t k = 0; k < m; k++) support[k0 + k] = (double) cnt[(size_t) k] / (double) nBoot;
    }
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- diagnostics
__global__ void k_debug_log(const double *x, double *out, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = vft_glibc_log(x[i]);
}

extern "C" int vft_debug_option(vft_ctx *c, int32_t option, int64_t value) {
    if (!c) return VFT_ERR_INVALID;
    switch (option) {
        case VFT_DEBUG_NO_FUSED_REFRESH: c->noFusedRefresh = value != 0; break;
        case VFT_DEBUG_PAIR_THREADS: c->pairWG = (int) value; break;
        case VFT_DEBUG_NO_PAIR_STAGING: c->noPairStaging = value != 0; break;
        case VFT_DEBUG_GENERIC_OUTPROFILE: c->genericOutProfile = value != 0; break;
        case VFT_DEBUG_FAULT_NO_FLAG: c->faultNoFlag = value != 0; break;
        case VFT_DEBUG_WAIT_LIMIT_MS: c->waitLimitS = value > 0 ? (double) value / 1000.0 : 120.0; break;
        case VFT_DEBUG_WIDE_GLUE: c->wideGlue = value != 0; break;
        case VFT_DEBUG_ML_LONG: c->mlLong = value != 0; break;
        case VFT_DEBUG_NO_WALK_SERVER: c->ws.disabled = value != 0; break;
        case VFT_DEBUG_WALK_DEVICE_MAILBOX: c->ws.wantDeviceMail = value != 0; break;
        case VFT_DEBUG_WALK_SERVER_STRIDE: c->ws.stride = value == 1 ? 1 : 8; break;
        case VFT_DEBUG_NO_MULTI_SWEEP:
            c->noMultiSweep = value == 1;
            c->multiMax = value == 2 || value == 4 ? (int) value : 0;
            break;
        case VFT_DEBUG_POISON_SELECTION:   // the candidate buffers of every slot filled with 0x7f bytes: what recycled memory looks like
            for (vft_ctx::SweepSlotHost &h: c->slots) {
                HIPCHK(c, hipMemset(h.candKey, 0x7f, (size_t) VFT_CAND_CAP * 8));
                HIPCHK(c, hipMemset(h.candId, 0x7f, (size_t) VFT_CAND_CAP * 4));
            }
            HIPCHK(c, hipDeviceSynchronize());   // (NULL-stream memsets vs the context's non-blocking stream)
            break;
        default: return fail(c, VFT_ERR_INVALID, "vft_debug_option: unknown option %d", (int) option);
    }
    return VFT_OK;
}

extern "C" int vft_debug_log(vft_ctx *c, int64_t n, const double *x, double *out) {
    if (!c || n < 0 || !x || !out) return VFT_ERR_INVALID;
    if (n == 0) return VFT_OK;
    if (int r = ensure_scratch(c, (size_t) n * 16)) return r;
    double *dx = (double *) c->scratch, *dy = dx + n;
    HIPCHK(c, hipMemcpyAsync(dx, x, (size_t) n * 8, hipMemcpyHostToDevice, c->stream));
    launch(k_debug_log, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const double *) dx, dy, n);
    LAUNCHCHK(c);
    HIPCHK(c, hipMemcpyAsync(out, dy, (size_t) n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VFT_OK;
}

// ---------------------------------------------------------------------------------------------- timing
extern "C" int vft_timer_start(vft_ctx *c) {
    if (!c) return VFT_ERR_INVALID;
    c->kevUsed = 0;
    c->kevSweeps = 0;
    c->timeKernels = true;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return VFT_OK;
}

extern "C" int vft_timer_stop_ms(vft_ctx *c, float *ms) {
    if (!c || !ms) return VFT_ERR_INVALID;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    c->timeKernels = false;
    return VFT_OK;
}

static int sweep_kernel_ms(vft_ctx *c, int which, float *avgMs, int64_t *launches) {
    if (!c || !avgMs || !launches) return VFT_ERR_INVALID;
    double total = 0;
    int64_t n = 0;
    for (size_t i = 0; i + 2 < c->kevUsed; i += 3) {
        float ms = 0;
        HIPCHK(c, hipEventSynchronize(c->kev[i + which + 1]));
        HIPCHK(c, hipEventElapsedTime(&ms, c->kev[i + which], c->kev[i + which + 1]));
        total += ms;
        n++;
    }
    *avgMs = n ? (float) (total / (double) n) : 0.f;
    *launches = n;
    return VFT_OK;
}
extern "C" int vft_sweep_kernel_ms(vft_ctx *c, float *avgMs, int64_t *launches) { return sweep_kernel_ms(c, 0, avgMs, launches); }
// the sweeps the launches timed since vft_timer_start stand for (a multi-seed pass is one launch for several sweeps)
extern "C" int vft_sweep_kernel_sweeps(vft_ctx *c, int64_t *sweeps) {
    if (!c || !sweeps) return VFT_ERR_INVALID;
    *sweeps = (int64_t) (c->kevUsed / 3) + c->kevSweeps;
    return VFT_OK;
}
extern "C" int vft_sweep_table_kernel_ms(vft_ctx *c, float *avgMs, int64_t *launches) {
    return sweep_kernel_ms(c, 1, avgMs, launches);
}
