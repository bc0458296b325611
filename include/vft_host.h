/*
 * vft_host.h — C entry point of the C++ host NJ driver (veryfasttree_amd/host/NJDriver.h), for callers that are not
 * C++ (tests, bench).  The driver itself is what a VeryFastTree build with `-ext HIP` would run in place of
 * NeighbourJoining::fastNJ (src/NeighbourJoining.tcc:2796-3155); it only talks to the device through vft_hip.h.
 */
#ifndef VFT_HOST_H
#define VFT_HOST_H
#include "vft_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Multi-GPU runs of the NJ driver: one process per GPU, every rank runs the SAME driver on the same input and keeps the
   whole NJ state (profiles, out-distances, top-hit lists) - join decisions are replicated and bit-identical - while the
   two bulk distance computations are split over the ranks and exchanged:
     - the one-vs-all sweeps (setBestHit, NJ.tcc:3571-3646): rank r sweeps its share of the target ids and selects its
       local top-2m, one all-gather of the k-record lists (device memory: RCCL over xGMI), every rank merges them under the
       reference's (criterion asc, id desc) order (vft_merge_hits);
     - the close-neighbour blocks of setAllLeafTopHits (vft_leaf_block_distances): computed whole on every rank.  They are
       integer seqDist counts, ~5 s of a million-sequence run on one GPU; split by rows (vft_nj_options.debug_flags &
       VFT_NJ_SHARD_LEAF_BLOCKS, round 3's behaviour) their results are 25 GB of host-buffer all-gathers at that size - more
       than the split saves.
   The driver is transport-agnostic: the caller supplies the buffers and one collective.  allgather(user, bytes, device)
   must gather `bytes` bytes from every rank's send buffer into every rank's recv buffer in rank order - d_send / d_recv
   (device memory, capacity d_cap / world * d_cap) when device != 0, h_send / h_recv (host) otherwise - and return 0. */
typedef struct vft_comm {
    int32_t rank, world;
    int (*allgather)(void *user, int64_t bytes, int32_t device);
    void *user;
    void *d_send, *d_recv;
    int64_t d_cap;
    void *h_send, *h_recv;
    int64_t h_cap;
} vft_comm;

typedef struct {
    int32_t fastest;            /* -fastest (main.cpp:339-343): also sets tophits_refresh = 0.5 in the caller */
    int32_t use_tophits_2nd;    /* Options::useTopHits2nd: on with -fastest at one thread (VeryFastTree.cpp:87-91) */
    double tophits_mult;        /* Options::tophitsMult      (1.0) */
    double tophits_close;       /* Options::tophitsClose     (-1 = log2N/(log2N+2)) */
    double tophits_refresh;     /* Options::tophitsRefresh   (0.8) */
    double topvisible_mult;     /* Options::topvisibleMult   (1.5) */
    double stale_out_limit;     /* Options::staleOutLimit    (0.01) */
    double f_reset_out_profile; /* Options::fResetOutProfile (0.02) */
    int32_t n_reset_out_profile;/* Options::nResetOutProfile (200) */
    int32_t tophits2_safety;    /* Options::tophits2Safety   (3) */
    double tophits2_mult;       /* Options::tophits2Mult     (1.0) */
    double tophits2_refresh;    /* Options::tophits2Refresh  (0.6) */
    int32_t scoredist;          /* logCorrect (NJ.tcc:322-330): 0 = Jukes-Cantor (nucleotides without a matrix),
                                   1 = scoredist-like (amino acids, or any alphabet with a distance matrix) */
    int32_t mllen;              /* vft_nj_ml_newick: 0 = none; 1 = `-mllen -nocat` (ML branch lengths on the NJ topology,
                                   Jukes-Cantor, constant rates); n >= 2 = `-mllen -cat n` (CAT approximation with n rate
                                   categories fitted after the first round, setMLRates NJ.tcc:5429-5488; the
                                   reference's default is 20) */
    int32_t me_nni;             /* 1 = minimum-evolution NNI rounds after fastNJ (DoNNI with useML = false, NJ.tcc:5797-6200;
                                   round(4 log2 N) rounds or until a round changes nothing, VeryFastTreeImpl.tcc:160-185):
                                   the reference's default without SPR moves (`-spr 0`); 0 = `-nome` */
    int32_t ml_nni;             /* vft_nj_ml_newick: n >= 1 = maximum-likelihood NNI rounds (DoNNI with useML = true; up to
                                   round(2 log2 N) rounds, VeryFastTreeImpl.tcc:311-393) under Jukes-Cantor with n rate
                                   categories (1 = `-nocat`, 20 = the default CAT approximation); 0 = `-noml` / `-mllen` */
    int32_t spr;                /* with me_nni: rounds of minimum-evolution SPR moves between the NNI rounds (SPR,
                                   NJ.tcc:6185-6404; the reference's default is 2, 0 = `-spr 0`) */
    int32_t gtr;                /* with mllen / ml_nni: 1 = `-gtr` (the six GTR rates and the base frequencies are fitted
                                   after the first ML round, setMLGtr NJ.tcc:6436-6500; Jukes-Cantor until then) */
    int32_t aa_model;           /* amino-acid contexts (n_codes = 20): 0 = the caller installs its own matrices
                                   (vft_set_distance_matrix / vft_set_transition_matrix); 1 = JTT92 (the reference's
                                   default), 2 = WAG01 (`-wag`), 3 = LG08 (`-lg`) (VeryFastTreeImpl.tcc:96-108).  With a
                                   model the driver installs the BLOSUM45-derived distance matrix for the NJ /
                                   minimum-evolution phase (the reference's default for proteins, DistanceMatrix.tcc:33)
                                   with scoredist log-correction, and before the ML stage re-averages every profile in
                                   the model's eigen-basis (transMatToDistanceMat + recomputeProfiles,
                                   VeryFastTreeImpl.tcc:253-256, 517-542) and installs the transition matrix */
    const vft_comm *comm;       /* NULL = one GPU; otherwise see vft_comm above (top-hits NJ phase only) */
    int32_t threads;            /* the reference's `-threads T` at its default -threads-level 3: 0 / 1 = the one-thread order
                                   (everything above); T > 1 = the refinement stages follow the schedule of a T-thread run -
                                   NNI rounds and ML length rounds over the subtrees of treePartitioning (NJ.tcc:5540-5750,
                                   :6108-6160, :5083-5112), whose walks this backend advances in lockstep as batches of
                                   quartets / splits (host/MLLengths.h "the subtree schedule"), and no star-topology test in
                                   the serial part of an ML NNI round (NJ.tcc:4902-4948).  The tree is byte-identical to
                                   `VeryFastTree -threads T` whenever the NJ phase of that run joins in the one-thread
                                   order (its tree does not depend on the thread count).  The caller keeps
                                   use_tophits_2nd = 0 with T > 1, as the reference does (VeryFastTree.cpp:87-91). */
    int32_t debug_flags;        /* tests and tools only - which of two equivalent loops runs (no environment variable selects one):
                                   VFT_NJ_DEBUG_HOST_JOINS 1  the host-driven join loop instead of the join engine
                                   VFT_NJ_DEBUG_HOST_LISTS 2  top-hit lists on the host (round 2's walks)
                                   VFT_NJ_DEBUG_HOST_RESET 4  resetTopVisible entirely on the host
                                   VFT_NJ_DEBUG_LEVEL_LENGTHS 16  ML length rounds as one batch per tree height - NOT the reference's
                                                              order in any of its modes (measurements only)
                                   VFT_NJ_DEBUG_NO_WALK_SERVER 128  the SPR / one-thread NNI walks with one launch per step
                                                              (vft_walk_step) instead of the resident walk server (same tree)
                                   VFT_NJ_DEBUG_SEED_BY_SEED 256  setAllLeafTopHits' seed sweeps one vft_sweep each instead of eight
                                                              unvisited seeds ahead per vft_sweep_batch (same lists) */
    int32_t gamma;              /* `-gamma` (VeryFastTreeImpl.tcc:391-394, NJ.tcc:297-308, :5261-5357): after the CAT tree and its supports
                                   are final, fit the shape of a discretised Gamma over the ml_nni / mllen rate categories and a
                                   multiplier of the rates to the per-site likelihoods, and multiply every branch length by
                                   1 / multiplier; vft_nj_last_gamma returns the "Gamma(20) LogLk" line's three numbers */
    int32_t out_profile_parts;  /* 0 = every full out-profile recomputation (NJ.tcc:3012-3033) adds the active nodes up in id order on one
                                   GPU, as the reference's one-thread run does; P >= 2 = in P blocks of the active list (the
                                   one true reduction of the NJ phase) - block b is summed by rank b % world (vft_out_profile_partial), the
                                   raw sums are all-gathered through vft_comm's host buffers and added in block order on every rank
                                   (vft_out_profile_finish).  Like the reference's threaded outProfile (per-thread partial sums merged with
                                   vector_add, NJ.tcc:763-783) this rounds differently from the one-thread order - the join order may
                                   differ from the one-thread reference's in the last bits' wake - but it depends on P only: the same tree
                                   on 1, 2, 4 or 8 ranks. */
    int32_t pad_;
} vft_nj_options;
#define VFT_NJ_DEBUG_HOST_JOINS 1
#define VFT_NJ_DEBUG_HOST_LISTS 2
#define VFT_NJ_DEBUG_HOST_RESET 4
#define VFT_NJ_DEBUG_LEVEL_LENGTHS 16
#define VFT_NJ_DEBUG_NO_WALK_SERVER 128
#define VFT_NJ_DEBUG_SEED_BY_SEED 256
#define VFT_NJ_DEBUG_NO_WALK_DUAL 512   /* SPR chains: every step waits for the host's verdict instead of handing both continuations of a step
                                           to the walk server (vft_walk_submit_dual); same tree (tests compare) */
#define VFT_NJ_SHARD_LEAF_BLOCKS 64   /* with comm: split the close-neighbour blocks by rows and all-gather the results (see vft_comm) */

/* Runs the NJ phase on a context created for (n_seqs, n_pos, nt, max_nodes = 2*n_seqs) with nothing uploaded yet.
   joins: up to n_seqs-3 rows of (i, j, newnode) with i < j, in join order; criterion[k] = the join's criterion.
   max_joins < 0 means all.  Returns VFT_OK or an error code; err (may be NULL) receives the message. */
int vft_nj_run(vft_ctx *ctx, const uint8_t *codes, int64_t n_seqs, int64_t n_pos, int32_t precision,
               const vft_nj_options *opt, int64_t max_joins, int64_t *joins, double *criterion, int64_t *n_joins,
               char *err, int32_t err_len);

/* The whole NJ phase and its tree: fastNJ to the end (root over the last 3 nodes, NJ.tcc:3098-3120) printed the way
   the reference prints it (printNJ, NJ.tcc:2706-2794; no supports): what the reference logs as "NJ\t<tree>".
   me_lengths != 0: also updateBranchLengths (NJ.tcc:6514-6595, log-corrected minimum-evolution lengths) before
   printing - the "ME_Lengths" tree, which is the final output of `-noml -nome -nosupport`; the context must then have
   been created with max_nodes >= 3 * n_seqs (the up-profiles live on the device).
   n_bootstrap > 0: also the local-bootstrap supports of the internal splits (reliabilityNJ, NJ.tcc:3191-3238; the
   reference's default is 1000), printed as ")0.987:length"; needs max_nodes >= 3 * n_seqs as well.
   codes: the n_seqs UNIQUE sequences in first-occurrence order; unique_first[u] = alignment row of unique sequence u;
   aln_next[k] = next alignment row with the same sequence or -1 (Uniquify, Alignment.cpp:494-526); names: n_all
   NUL-terminated names back to back.  out may be NULL to query the length (out_len, without the terminator). */
int vft_nj_newick(vft_ctx *ctx, const uint8_t *codes, int64_t n_seqs, int64_t n_pos, int32_t precision,
                  const vft_nj_options *opt, int32_t me_lengths, int32_t n_bootstrap, const int64_t *unique_first,
                  const int64_t *aln_next, int64_t n_all,
                  const char *names, char *out, int64_t out_cap, int64_t *out_len, char *err, int32_t err_len);

/* vft_nj_newick followed - when opt->mllen is set - by the reference's `-mllen -nocat` stage for nucleotides under
   Jukes-Cantor (VeryFastTreeImpl.tcc:249-311: rounds of optimizeAllBranchLengths + treeLogLk until the lengths settle);
   the printed lengths are then the ML ones, as in the reference's output.  me_lengths must be set (the reference always
   runs updateBranchLengths first).  loglk[loglk_cap] (may be NULL) receives "TreeLogLk Length<k>" of each round,
   n_rounds the number of rounds run; rates[rates_cap] / n_rates / ratecat[n_pos] (each may be NULL) the fitted rate
   categories - the reference's "Rates" and (0-based) "SiteCategories" log lines; with opt->gtr, gtr_out[10] (may be
   NULL) = the fitted rates ac ag at cg ct gt and frequencies A C G T ("GTR rates" / "GTR Frequencies"). */
int vft_nj_ml_newick(vft_ctx *ctx, const uint8_t *codes, int64_t n_seqs, int64_t n_pos, int32_t precision,
                     const vft_nj_options *opt, int32_t me_lengths, int32_t n_bootstrap, const int64_t *unique_first,
                     const int64_t *aln_next, int64_t n_all, const char *names, char *out, int64_t out_cap,
                     int64_t *out_len, double *loglk, int32_t loglk_cap, int32_t *n_rounds, double *rates,
                     int32_t rates_cap, int32_t *n_rates, int32_t *ratecat, double *gtr_out, char *err, int32_t err_len);

/* CRC-32 (zlib's) of the join order of the last vft_nj_run / vft_nj_newick / vft_nj_ml_newick of this process, one value per
   chunk of *chunk (10 000) joins - the joins behind the last complete chunk, if any, as one shorter chunk at the end -, each join as
   three little-endian int32 (i, j, new node) - the reference's `Join`
   trace lines (-verbose 3, NJ.tcc:2925) reduced the same way pin the join order of a run whose tree is all the caller asked
   for (tests/golden/bb_c4_prefix.npz).  crcs[cap] may be NULL. */
int vft_nj_last_join_crcs(int64_t *chunk, int64_t *n_joins, uint32_t *crcs, int64_t cap, int64_t *n_crcs);

/* The reference's treePartitioning(penalty) (NJ.tcc:5540-5750) for `threads` threads on a tree given as child[n_nodes][3] (-1 = none;
   the root has three children): the roots of the subtrees its threads walk, in its hand-out order (round-robin over the threads'
   lists), window = -threads-ptw (0: the default 50).  Pure host code - what MLLengths::doNNIThreaded / optimizeRoundThreaded make
   their lanes from (penalty 2 / 1).  out[cap] may be NULL to query *n_out. */
int vft_tree_partitioning(int64_t n_nodes, const int64_t *child, int64_t root, int32_t penalty, int32_t threads, int32_t window,
                          int64_t *out, int64_t cap, int64_t *n_out, double *speedup);

/* Where the wall-clock of the last vft_nj_newick / vft_nj_ml_newick of this process went.  seconds[8]: the NJ phase with its root;
   the minimum-evolution NNI + SPR rounds, of which the SPR rounds; ME branch lengths + local supports; the whole ML stage, of which
   the ML NNI rounds, the SH-like supports, the model fits (CAT rates, GTR).  counts[4]: lockstep steps of the subtree schedule
   (opt.threads > 1) and the quartets / splits judged in them, SPR chain steps evaluated, SPR moves made.  Either may be NULL. */
int vft_nj_last_stage_seconds(double *seconds, int64_t *counts);
/* out[2]: the SPR chains of the last tree - dual commands sent (both continuations of a chain step handed to the walk server,
   vft_walk_submit_dual) and continuations the resident workgroups ran without waiting for the host's verdict */
int vft_nj_last_walk_dual(int64_t *out);
/* out[2]: with a vft_comm of several ranks, what the lanes of the subtree schedule exchanged during the last tree of this process - the
   number of all-gathers (one per lockstep step with a batch to judge) and the bytes this rank received in them */
int vft_nj_last_lane_exchange(int64_t *out);
/* the layout of that exchange (host/MLLengths.h laneShare / laneRecord), exported for the CPU test that runs it over gloo: out[0] = the
   padded share every rank sends, out[1], out[2] = rank `rank`'s items [k0, k1) of n_items, out[3] = where item `item` sits in the gathered
   buffer (in records) */
int vft_nj_lane_share(int64_t n_items, int32_t world, int32_t rank, int64_t item, int64_t *out);
/* the layout of the out-profile blocks' exchange (vft_nj_options.out_profile_parts; host/NJDriver.h outProfileBlock), exported for the CPU
   test that runs it over gloo: out[0] = the rank that sums block `block` of `parts`, out[1] = the block's slot in that rank's share,
   out[2] = slots per share, out[3], out[4] = the block's entries [i0, i1) of an active list of n nodes */
int vft_nj_out_profile_block(int32_t parts, int32_t world, int32_t block, int64_t n, int64_t *out);

/* out[3]: what `-gamma` (vft_nj_options.gamma) found for the last tree of this process - the Gamma(nCat) log-likelihood, the shape
   alpha, the factor every branch length was multiplied by (the reference's "Gamma(20) LogLk = .. alpha = .. rescaling lengths by ..") */
int vft_nj_last_gamma(double *out);

/* The first n values of the random stream the bootstrap columns are drawn from (Knuth's ran_array at its default
   seed, as the reference uses it, Knuth.cpp:95-111): exported so that tests can pin the host generator. */
void vft_knuth_stream(double *out, int64_t n);

/* Maximum-likelihood branch lengths on a fixed topology (`-mllen`: VeryFastTreeImpl.tcc:263-311 without the rate /
   GTR re-estimation): optionally recomputeMLProfiles (NJ.tcc:3516), then `rounds` calls of optimizeAllBranchLengths
   (NJ.tcc:5065), each followed by treeLogLk (NJ.tcc:5160).  The model (vft_set_rates, vft_set_transition_matrix,
   vft_set_ml_limits) and the leaf / internal profiles must be on the device; the context needs max_nodes >= n_nodes +
   n_seqs (up-profiles).  parent[n_nodes] (-1 at the root), child[n_nodes][3] (-1 = none; the root has three);
   branchlength: numeric_t[n_nodes], in/out.  recompute_first: bit 0 = recomputeMLProfiles before the first round; bit 1 =
   level-parallel rounds (MLLengths::optimizeRoundParallel: every tree height as one batch - not the one-thread order, so
   lengths differ within the search tolerance, the way the reference's threaded mode differs).  ftol = MLFTolBranchLength, atol = MLMinBranchLengthTolerance
   (Constants.h:26-30).  n_leaf_gaps >= 0 applies treeLogLk's Jukes-Cantor correction for that many gap characters in
   the leaves, < 0 none (matrix models).  loglk[rounds] (may be NULL) receives the tree log-likelihood after each
   round, evals (may be NULL) the number of pairLogLk evaluations the line searches made. */
int vft_ml_lengths(vft_ctx *ctx, int64_t n_seqs, int64_t n_nodes, int64_t n_pos, int32_t precision, const int64_t *parent,
                   const int64_t *child, int64_t root, void *branchlength, int32_t recompute_first, int32_t rounds,
                   double ftol, double atol, int64_t n_leaf_gaps, double *loglk, int64_t *evals, char *err, int32_t err_len);

/* The GTR model's likelihood tables as the reference builds them (TransitionMatrix.tcc:26-60, 160-232): rates = ac ag
   at cg ct gt, freq = A C G T; out: stat[4], statinv[4], eigenval[4], codefreq[5][4] (last row = gap), eigeninv[4][4],
   eigeninvT[4][4]: numeric_t values (precision 4 / 8 bytes) held in double.  Exported for tests. */
int vft_gtr_tables(const double *rates, const double *freq, int32_t precision, double *stat, double *statinv, double *eigenval, double *codefreq,
                   double *eigeninv, double *eigeninvT);

/* The built-in amino-acid models as the reference builds them (createTransitionMatrixJTT92 / WAG01 / LG08,
   TransitionMatrix.tcc:14-24, 158-232): model 1 = JTT, 2 = WAG, 3 = LG; out: stat[20], statinv[20], eigenval[20],
   codefreq[21][20] (last row = gap), eigeninv[20][20], eigeninvT[20][20] - numeric_t values (precision 4 / 8 bytes) held
   in double, what vft_set_transition_matrix takes after narrowing.  Exported for tests and for callers that drive the
   C ABI themselves. */
int vft_aa_model_tables(int32_t model, int32_t precision, double *stat, double *statinv, double *eigenval, double *codefreq,
                        double *eigeninv, double *eigeninvT);
/* The default protein distance matrix (matrixBLOSUM45 + setupDistanceMatrix, DistanceMatrix.tcc:33-36, 102-155):
   distances[20][20], codefreq[20][20], eigenval[20], eigentot[20] as vft_set_distance_matrix takes them. */
int vft_blosum45_tables(int32_t precision, double *distances, double *codefreq, double *eigenval, double *eigentot);
/* transMatToDistanceMat (VeryFastTreeImpl.tcc:517-542) for a built-in model: the tables recomputeProfiles averages
   with before the ML stage (distances and eigenvalues are zero, as in the reference). */
int vft_aa_model_as_distance_tables(int32_t model, int32_t precision, double *distances, double *codefreq, double *eigenval,
                                    double *eigentot);

#ifdef __cplusplus
}
#endif
#endif
