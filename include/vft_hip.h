/*
 * vft_hip.h — C ABI of the MI355X (gfx950) profile-operations backend for VeryFastTree.
 *
 * This is the drop-in boundary.  The reference selects a backend through the template slot
 * `Operations<Precision>` (src/NeighbourJoining.h:19-22,256; registration src/impl/VeryFastTreeFloatCuda.cpp:1-7,
 * src/VeryFastTree.cpp:305-314) whose ten methods work on one 4- or 20-element vector at a time
 * (src/operations/BasicOperations.h:20-39).  That granularity cannot feed a GPU
 * (src/operations/CudaOperations.cu:19-27 pays an allocation and two copies per 4 floats), so this backend keeps
 * the slot (veryfasttree_amd/host/HipOperations.h satisfies the trait) and lifts the hot loops that wrap those
 * calls to whole-profile, batched entry points.  Each entry point below names the reference member it replaces.
 *
 * Conventions
 *   - plain C, opaque context, int return code (0 = VFT_OK); vft_last_error() gives the text.
 *   - "real" arrays are float when the context precision is 4, double when it is 8 (the reference's numeric_t).
 *   - node ids follow the reference: leaves 0..nSeqs-1 in unique-sequence order, internal nodes nSeqs..maxnodes-1
 *     in creation order (NJ.tcc:2904-2909).  Codes are the reference's: 0..nCodes-1, NOCODE = 127.
 *   - host-side dense profile = w[nPos], codes[nPos], f[nPos*nCodes]; f is read/written only for columns where
 *     the reference holds a vector (codes == NOCODE && w > 0, NJ.tcc:2040-2042).
 *   - every call is asynchronous on the context's stream unless it returns data to host memory, in which case
 *     it synchronises that stream before returning.  Small inputs (id lists, per-node scalars) are staged in a
 *     host-mapped ring, so they need no synchronisation either; host arrays passed in may be reused on return.  Pointers named d_* are DEVICE pointers supplied by the
 *     caller (e.g. torch tensors) and are written on the stream without synchronising.
 *   - there is no CPU fallback: without a HIP device vft_create fails.
 */
#ifndef VFT_HIP_H
#define VFT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFT_OK 0
#define VFT_ERR_INVALID 1   /* bad argument */
#define VFT_ERR_HIP 2       /* a HIP runtime call failed */
#define VFT_ERR_STATE 3     /* call sequence error (e.g. sweep before leaves were uploaded) */
#define VFT_ERR_TIMEOUT 4   /* a kernel did not raise its completion flag (the wait is bounded; the context is unusable afterwards) */
#define VFT_ERR_HALTED 5    /* vft_nj_engine_enqueue: the join engine has stopped at an event the caller handles first (poll, resume) */
#define VFT_NOCODE 127

typedef struct vft_ctx vft_ctx;

typedef struct {
    int32_t device;        /* HIP device ordinal */
    int32_t precision;     /* 4 = float, 8 = double: the reference's Precision template argument */
    int32_t n_codes;       /* 4 (nt) or 20 (aa): Options::nCodes */
    int32_t reserved;
    int64_t n_seqs;        /* unique sequences */
    int64_t n_pos;         /* alignment columns */
    int64_t max_nodes;     /* 2*n_seqs in the reference (NJ.tcc:229-231) */
} vft_config;

/* One hit of a one-vs-all sweep, in sorted order (the reference's Besthit, NJ.h:192-204, with 32-bit ids). */
typedef struct {
    int32_t j;             /* target node, -1 = empty */
    float dist;            /* numeric_t narrowed to float only for precision 4; see vft_sweep for precision 8 */
    float weight;
    float criterion;
} vft_hit_f32;

typedef struct {
    int64_t j;
    double dist, weight, criterion;
} vft_hit_f64;

/* ---- life cycle (replaces CudaOperations::configCuda, CudaOperations.cu:169-175) */
int vft_create(vft_ctx **out, const vft_config *cfg);
int vft_destroy(vft_ctx *ctx);
const char *vft_last_error(const vft_ctx *ctx);
/* hipStream_t to launch on (NULL = the context's own stream).  Lets a caller use torch's current stream. */
int vft_set_stream(vft_ctx *ctx, void *hip_stream);
int vft_synchronize(vft_ctx *ctx);

/* Plain device buffers for callers that do not bring their own allocator (the d_* arguments below). */
int vft_device_malloc(vft_ctx *ctx, int64_t bytes, void **d_ptr);
int vft_device_free(vft_ctx *ctx, void *d_ptr);
int vft_device_upload(vft_ctx *ctx, void *d_dst, const void *src, int64_t bytes);

/* ---- inputs */
/* Leaf profiles from codes[n_seqs][n_pos] (what seqsToProfiles builds, NJ.tcc:382-457). */
int vft_upload_leaves(vft_ctx *ctx, const uint8_t *codes);
/* Distance matrix used by the NJ phase for amino acids (DistanceMatrix.h:15-33); arrays are nCodes x nCodes
   row-major / nCodes long.  Not calling it means %-different distances (the -nt default, VeryFastTree.cpp:96-98). */
int vft_set_distance_matrix(vft_ctx *ctx, const void *distances, const void *codefreq, const void *eigenval,
                            const void *eigentot);
/* Transition matrix tables for the ML phase (TransitionMatrix.h:65-76).  codefreq has nCodes+1 rows, the last
   one being the NOCODE (gap) row.  Passing NULLs selects Jukes-Cantor (nt only). */
int vft_set_transition_matrix(vft_ctx *ctx, const void *stat, const void *statinv, const void *eigenval,
                              const void *codefreq, const void *eigeninv, const void *eigeninvT);
/* Rate categories (NJ.h Rates: rates[n_rates], ratecat[n_pos]). */
int vft_set_rates(vft_ctx *ctx, const void *rates, int32_t n_rates, const int64_t *ratecat);
/* ML tolerances that the reference reads from Options inside the hot loops (Constants.h:26-39). */
/* Jukes-Cantor likelihoods the reference's way to the last bit (NJ.tcc:1203-1262: `exp` of libm in pSame / pDiff, ONE running product over
   the columns in column order with the underflow rescaling, one `log` of libm at the end) - what the matrix models always get
   (vft_kernels_ml.h "ordered total").  ON by default since round 6: with it `VeryFastTree -nt -threads 64` on 100 000 sequences comes
   out byte for byte (tests/golden/thr_c4_100k_t64_crc.npz).  on = 0: per-thread running products summed over the threads, rounds 1-5's
   arithmetic - every column the same number, the total differs from the reference's in the last bits (every TreeLogLk line still to the
   printed digit; near-tie NNIs may fall the other way: 24 leaf placements of zero length among those 100 000 sequences); the ML NNI
   stage of a nucleotide run is ~2x faster that way (2.6 s instead of 5.6 s of a 46 s pipeline at that size). */
int vft_set_jc_exact(vft_ctx *ctx, int32_t on);
int vft_set_ml_limits(vft_ctx *ctx, double min_branch_length, double min_rel_branch_length,
                      double fpost_total_tolerance);

/* ---- per-node NJ state kept on the device (NJ.h:268-287).  `first..first+count` is a node-id range. */
int vft_set_parents(vft_ctx *ctx, int64_t first, int64_t count, const int64_t *parent);
int vft_set_node_scalars(vft_ctx *ctx, int64_t first, int64_t count, const void *diameter, const void *selfweight,
                         const void *selfdist);
int vft_get_node_scalars(vft_ctx *ctx, int64_t first, int64_t count, void *diameter, void *selfweight, void *selfdist);
int vft_set_out_distances(vft_ctx *ctx, int64_t first, int64_t count, const void *out_dist,
                          const int64_t *n_out_active);
int vft_get_out_distances(vft_ctx *ctx, int64_t first, int64_t count, void *out_dist, int64_t *n_out_active);
/* Host-mapped mirrors of outDistances[] / nOutDistActive[] (real / int32, indexed by node id): every refresh a
   kernel performs is also stored here, so host code can evaluate setCriterion (NJ.tcc:1099-1107) without a copy.
   Contents are valid after vft_synchronize() or after any call that returned data to host memory. */
int vft_out_distance_mirror(vft_ctx *ctx, const void **out_dist, const int32_t **n_out_active);
int vft_set_max_node(vft_ctx *ctx, int64_t maxnode);           /* NJ.h: maxnode, the next id to allocate */
/* The state change of one join in one stream-ordered launch (NJ.tcc:2904-2909, 3003-3007, 254): parent[i] = parent[j]
   = newnode, diameter[newnode], outDistances[newnode] = 0 with nOutDistActive[newnode] = stale_stamp ("unreasonably
   high"), maxnode = max(maxnode, newnode + 1).  Equivalent to vft_set_max_node + 2 x vft_set_parents +
   vft_set_node_scalars + vft_set_out_distances for that node. */
int vft_join_nodes(vft_ctx *ctx, int64_t i, int64_t j, int64_t newnode, double diameter, int64_t stale_stamp);

/* One join of the NJ loop in one launch (k_join_fused): vft_join_nodes + vft_average_profiles(1, newnode, i, j, unweighted)
   + the new node's self distance + - when update_out_profile != 0 - vft_out_profile_update(i, j, newnode, n_active_old)
   (NJ.tcc:2904-2909, 3003-3008, 3039-3042, 3034).  The new profile lives in the node's plain row until the tile streams
   are rebuilt - lazily, for all joined nodes since the last rebuild at once, before the next call that reads tile streams
   (sweeps, full out-profile, ...); results are those of the separate calls.  Stream-ordered. */
int vft_join_fused(vft_ctx *ctx, int64_t i, int64_t j, int64_t newnode, double diameter, int64_t stale_stamp,
                   int64_t n_active_old, int32_t update_out_profile);

/* ---- top-hit lists on the device (TopHits / TopHitsList / Hit, NJ.h:206-248)
 * The lists of fastNJ's top-hits heuristic live in HBM: `m` entries of {int32 j; numeric_t dist} (vft_tophit_f32 /
 * vft_tophit_f64) for each of the first n_lists node ids.  The three list walks of a join are one launch each and return
 * only their result (veryfasttree_amd/csrc/vft_kernels_tophits.h). */
typedef struct { int32_t j; float dist; } vft_tophit_f32;
typedef struct { int32_t j; int32_t pad; double dist; } vft_tophit_f64;   /* = struct {int32_t j; double dist;} */
typedef struct { int32_t j, pos; double dist, criterion; } vft_tophits_best_t;
typedef struct { int32_t n_unique, use_unique, n_save, pad; } vft_tophits_join_t;
int vft_tophits_create(vft_ctx *ctx, int32_t m, int64_t n_lists);
/* lists of `count` nodes in one piece: packed = count x m entries (list t at packed[t * m]), lens[t] of them valid */
int vft_tophits_upload(vft_ctx *ctx, int64_t count, const int64_t *nodes, const int32_t *lens, const void *packed);
int vft_tophits_download(vft_ctx *ctx, int64_t node, int32_t *len, void *hits /* m entries */);
/* getBestFromTopHits (NJ.tcc:4267-4298) over the first `len` entries of node's list: every entry re-targeted to the active
   ancestor of its partner (updateBestHit :1626-1648; new distance where the partner changed), lazy out-distance refreshes
   (setCriterion :1092-1098), criterion; the first strict minimum in list order.  force_node != 0: setOutDistance(node) first
   (:4273-4279).  The list itself is not changed.  out->j < 0: no usable entry. */
int vft_tophits_best(vft_ctx *ctx, int64_t node, int32_t len, int64_t n_active, int64_t n_diff_allow, double totdiam,
                     int32_t force_node, vft_tophits_best_t *out);
/* The merge of the lists of the two children c0 (n0 entries) and c1 (n1) of the freshly joined `newnode`
   (topHitJoin NJ.tcc:4319-4362 -> uniqueBestHits :4786-4833 -> sortSaveBestHits :4535-4578): one candidate per distinct
   active ancestor of a listed partner, its distance against newnode, lazy refreshes, criteria; sorted by (criterion
   ascending, partner id descending); used - i.e. its first min(n_unique, n_save_max) entries saved as newnode's list - when
   n_unique == n_active - 1 or (age_ok and n_unique >= need).  info: counts and the decision; j / dist / criterion: the
   n_unique sorted candidates (host arrays of n0 + n1 entries; dist / criterion in the context's precision).
   n_active: the count after the join. */
int vft_tophits_join(vft_ctx *ctx, int64_t newnode, int64_t c0, int32_t n0, int64_t c1, int32_t n1, int64_t n_active,
                     int64_t n_diff_allow, double totdiam, int32_t n_save_max, int32_t need, int32_t age_ok,
                     vft_tophits_join_t *info, int32_t *j, void *dist, void *criterion);

/* The top-hits refresh of a join (topHitJoin's else-branch, NJ.tcc:4440-4517) once `newnode` has been swept (vft_sweep):
   hit_j / hit_dist = the first n_hits (2m) records of its sorted hits (negative j: empty record); own_list = newnode's own
   new list (n_own entries, what sortSaveBestHits keeps of the sweep); work = the active nodes among its first m hits, each
   to receive the first 2 * n_new[t] swept hits: merged with the node's own re-targeted hits, one record per partner,
   distances recomputed where the reference recomputes them, sorted, the first n_new[t] saved (uniqueBestHits :4786-4833,
   sortSaveBestHits :4535-4578) - one workgroup per node, the m x 2m distances as one block.  lens[t] / first[t]: new length
   and first entry (vft_tophit_*) of every list.  Every out-distance must be fresh enough already (:4451-4464).
   VFT_ERR_STATE: lists too long for the merge kernel's LDS (the caller merges on the host). */
int vft_tophits_refresh(vft_ctx *ctx, int64_t newnode, int32_t n_hits, const int64_t *hit_j, const void *hit_dist, int32_t n_own,
                        const void *own_list, int64_t n_work, const int64_t *work, const int32_t *n_new, int64_t n_active,
                        int64_t n_diff_allow, double totdiam, int32_t *lens, void *first);

/* ---- the join loop on the device (veryfasttree_amd/csrc/vft_kernels_njengine.h)
 * fastNJ's loop over joins with top hits (NJ.tcc:2857-3047) as a stream of kernels whose arguments live in a device-resident
 * state block (nActive, maxnode, totdiam, the candidate join, the visible set, list ages): the caller enqueues the launches
 * of many joins without waiting, reads the join records from a host-mapped log, and handles the events a kernel flags by
 * raising `halt` (every later kernel then does nothing until vft_nj_engine_resume).  First-level top-hit lists only
 * (vft_tophits_create must have been called; lists, out-distances, profiles are the context's). */
typedef struct { int32_t i, j, newnode, pad; double dist, criterion, bl_i, bl_j, diameter; } vft_nj_join_t;
typedef struct {
    int32_t m, n_top;          /* list length, size of the top-visible list (NJ.tcc:2827-2839, 197-208) */
    int32_t need;              /* a merged list is used when it holds at least this many hits... (m * tophitsRefresh, NJ.tcc:4352) */
    int32_t age_limit;         /* ...and is at most this old (NJ.tcc:4342-4351) */
    int32_t fastest;           /* -fastest: no hill climbing (NJ.tcc:4218-4220) */
    int32_t pad;
    int64_t stale_stamp;       /* nOutDistActive of a new node ("unreasonably high", NJ.tcc:254) */
    double stale_out_limit;    /* Options.h:38 */
} vft_nj_engine_config;
#define VFT_NJ_HALT_NONE 0
#define VFT_NJ_HALT_RESET 1     /* topHitNJSearch of join halt_join wants resetTopVisible (NJ.tcc:4156-4206); the join has not happened */
#define VFT_NJ_HALT_REFRESH 2   /* topHitJoin of join halt_join wants a top-hits refresh of the new node (NJ.tcc:4440-4517); the join is logged */
#define VFT_NJ_HALT_CLIMB 3     /* the hill climbing of join halt_join changed the candidate in its last enqueued round */
#define VFT_NJ_PHASE_SEARCH 1   /* top-visible scan of join_index + one hill-climbing round (up to its last comparison) */
#define VFT_NJ_PHASE_CLIMB 2    /* one more hill-climbing round */
#define VFT_NJ_PHASE_JOIN 4     /* the round's last comparison; the join (record, tree arrays, profile, out-profile, out-distance) */
#define VFT_NJ_PHASE_MERGE 8    /* the new node's list from the children's, the visible-set updates */
#define VFT_NJ_PHASE_NEXT 16    /* with PHASE_MERGE: the same kernels go on with PHASE_SEARCH of join_index + 1 */
int vft_nj_engine_create(vft_ctx *ctx, const vft_nj_engine_config *cfg);
/* scalars of the loop (stream-ordered stores); a negative / NaN argument leaves the value alone */
int vft_nj_engine_set_state(vft_ctx *ctx, int64_t n_active, int64_t maxnode, double totdiam, int32_t top_visible_age);
/* waits for the stream; any pointer may be NULL */
int vft_nj_engine_get_state(vft_ctx *ctx, int64_t *n_active, int64_t *maxnode, double *totdiam, int32_t *top_visible_age,
                            int64_t *joins_done, int32_t *halt, int32_t *halt_join, int32_t *n_unique);
/* visible[first .. first + count) (j int32, dist numeric_t) */
int vft_nj_engine_visible_set(vft_ctx *ctx, int64_t first, int64_t count, const int32_t *j, const void *dist);
int vft_nj_engine_visible_get(vft_ctx *ctx, int64_t first, int64_t count, int32_t *j, void *dist);
/* visible[nodes[t]] = (j[t], dist[t]) and age[nodes[t]] = age for t < n (j == NULL: ages only) */
int vft_nj_engine_nodes_set(vft_ctx *ctx, int64_t n, const int64_t *nodes, const int32_t *j, const void *dist, int32_t age);
int vft_nj_engine_topvisible_set(vft_ctx *ctx, const int32_t *nodes /* n_top */);
int vft_nj_engine_topvisible_get(vft_ctx *ctx, int32_t *nodes /* n_top */);   /* waits for the stream */
/* the kernels of the given phases of join number join_index (0-based), no waiting.  update_out (PHASE_JOIN): the incremental
   out-profile update (NJ.tcc:3034-3035); 0 when the caller recomputes the out-profile after the join (then it also sets
   totdiam, and passes 0 with PHASE_MERGE as well: the new node's out-distance is then computed first).  A candidate that
   the enqueued hill-climbing round changed raises HALT_CLIMB. */
int vft_nj_engine_enqueue(vft_ctx *ctx, int64_t join_index, int32_t phases, int32_t update_out);
/* resetTopVisible (NJ.tcc:4728-4784), device part: the lazy out-distance refreshes of getVisible for every active node with
   a usable visible hit, its criterion, and the first k such nodes under (criterion ascending, node id descending) - the order
   the reference sorts by (positions are ascending node ids).  hits: k records vft_hit_* in host memory with j = the node,
   weight = its visible partner, dist = the visible hit's distance, criterion (j = -1: fewer than k); n_visible: how many
   nodes have a usable visible hit.  k <= 8192.  Waits for the stream. */
int vft_nj_engine_reset_candidates(vft_ctx *ctx, int64_t n_active, double totdiam, int32_t k, void *hits, int64_t *n_visible);
/* host-mapped status words, no synchronisation: joins completed (merge done or refresh requested), halt reason, its join */
int vft_nj_engine_poll(vft_ctx *ctx, int64_t *joins_done, int32_t *halt, int32_t *halt_join);
/* clears the halt; next_join = the first join the caller enqueues next (joins enqueued behind the event have not run) */
int vft_nj_engine_resume(vft_ctx *ctx, int64_t next_join);
/* the host-mapped join log (record k is valid once join k has run: joins_done > k, or halt_join == k after a REFRESH halt)
   and the host bookkeeping of the joins [from, to): parent mirror, node count */
int vft_nj_engine_log(vft_ctx *ctx, const vft_nj_join_t **log);
int vft_nj_engine_adopt(vft_ctx *ctx, int64_t from, int64_t to);

/* ---- profiles */
int vft_profile_upload(vft_ctx *ctx, int64_t node, const void *w, const uint8_t *codes, const void *f);
int vft_profile_download(vft_ctx *ctx, int64_t node, void *w, uint8_t *codes, void *f);
/* number of frequency vectors each node holds (Profile::nVectors, NJ.h:137): leaves have none */
int vft_profile_nvectors(vft_ctx *ctx, int64_t first, int64_t count, int64_t *nvec);
/* averageProfile (NJ.tcc:2067-2135) for a batch of independent joins: out[k] = avg(a[k], b[k], weight[k]);
   weight < 0 means unweighted (0.5).  Also stores the new node's self-distance/self-weight
   (profileDist(new,new), NJ.tcc:3039-3042) on the device. */
int vft_average_profiles(vft_ctx *ctx, int64_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                         const double *bionj_weight);

/* ---- out-profile (NJ.tcc:729-815, 943-1010) */
int vft_out_profile_full(vft_ctx *ctx, int64_t n_active, const int64_t *active_ids);
/* outProfile in parts - the multi-GPU shard of the one true reduction of the NJ phase (the reference's threaded
   outProfile sums per-thread partials and merges them with vector_add, NJ.tcc:763-783).  vft_out_profile_partial: the raw sums over
   `ids` (one block of the active list, n of its n_total nodes; weights with the in-weight 1 / n_total) as n_pos x (1 + n_codes)
   numbers of the context's precision - per column the weight, then the frequencies - written to the HOST buffer `part`.
   vft_out_profile_finish: n_parts such blocks (host memory, block after block) added in block order with numeric_t additions, the
   weight floored, the frequencies normalised, codeDist filled: the new out-profile.  The result depends on the partition (as the
   reference's does on its thread count), not on which rank computed which block. */
int vft_out_profile_partial(vft_ctx *ctx, int64_t n_total, int64_t n, const int64_t *ids, void *part);
int vft_out_profile_finish(vft_ctx *ctx, int32_t n_parts, const void *parts);
int vft_out_profile_update(vft_ctx *ctx, int64_t old1, int64_t old2, int64_t newnode, int64_t n_active_old);
int vft_out_profile_upload(vft_ctx *ctx, const void *w, const void *f, const void *codedist);
int vft_out_profile_download(vft_ctx *ctx, void *w, void *f, void *codedist);
/* setOutDistance (NJ.tcc:1012-1083) for a list of nodes (all active nodes when ids == NULL). */
int vft_out_distances(vft_ctx *ctx, int64_t n, const int64_t *ids, int64_t n_active, double totdiam);

/* The cross product of two node lists, any mix of leaves and internal nodes: dist[ia * n_b + ib] = the join distance of
 * (a[ia], b[ib]) - profileDist / seqDist minus the two diameters, setDistCriterion (NJ.tcc:1115-1124) without the
 * criterion - after the lazy out-distance refresh (setCriterion, NJ.tcc:1092-1098) of every listed node.  This is what
 * a top-hits refresh recomputes (NJ.tcc:4477-4515: transferBestHits to each of the m closest nodes): 3m ids in, m x 2m
 * distances out.  Negative ids and pairs of a node with itself are skipped (the content of their slots is unspecified).
 * dist: numeric_t[n_a * n_b], host memory. */
int vft_block_distances(vft_ctx *ctx, int64_t n_a, const int64_t *a, int64_t n_b, const int64_t *b, int64_t n_active,
                        int64_t n_diff_allow, double totdiam, void *dist);

/* ---- distances
 * vft_sweep = setBestHit (NJ.tcc:3571-3646): node `query` against every node < maxnode, each pair through
 * setDistCriterion (NJ.tcc:1115-1124: seqDist NJ.tcc:1601-1624 for leaf x leaf, profileDist NJ.tcc:1167-1190
 * otherwise, minus diameters) and setCriterion (NJ.tcc:1085-1107, including the lazy refresh of out-distances
 * staler than n_diff_allow), followed by the reference's sort (psort + CompareHitsByCriterion, Utils.h:126-146,
 * NJ.tcc:7301-7306: ascending criterion, ties by descending node id) truncated to the first `k` hits.
 * hits: k records of vft_hit_f32 / vft_hit_f64 (by context precision) in HOST memory, may be NULL.
 * d_hits: same records in DEVICE memory, may be NULL (no synchronisation when hits == NULL).
 * best_j: argmin excluding the query itself (the `bestjoin` out-parameter), may be NULL.
 */
int vft_sweep(vft_ctx *ctx, int64_t query, int64_t n_active, int64_t n_diff_allow, double totdiam, int32_t k,
              void *hits, void *d_hits, int64_t *best_j);
/* n_seeds sweeps in one call: hits = n_seeds x k records (host, may be NULL), d_hits likewise in device memory,
   best_j[n_seeds].  Results are exactly those of n_seeds vft_sweep calls in this order; the top-k selections of the
   batch share their launches and the call synchronises once.  For seeds that are independent of each other's results:
   the speculative next seeds of setAllLeafTopHits (NJ.tcc:3772-3800; NJDriver::seedSweep takes eight ahead), a multi-GPU exchange per
   batch.  Nucleotides without a distance matrix: while no lazy out-distance refresh is due, the leaf seeds of the batch share passes
   over the targets - four (two) seeds per launch, every (query, target) pair evaluated with vft_sweep's operations - and so do its
   profile seeds (csrc/vft_kernels_nj.h: k_sweep_nt_leafq_multi, k_sweep_nt_profq_multi; VFT_DEBUG_NO_MULTI_SWEEP: a launch per seed). */
int vft_sweep_batch(vft_ctx *ctx, int32_t n_seeds, const int64_t *queries, int64_t n_active, int64_t n_diff_allow,
                    double totdiam, int32_t k, void *hits, void *d_hits, int64_t *best_j);
/* The k records of seed number `slot` of the last vft_sweep_batch (slot 0: of the last vft_sweep) where the selection left them - the
   host-mapped result block, no copy; *hits is valid until the context's next sweep.  For callers that pass hits = NULL above and read
   the records in place (bench.py's step, NJDriver::sweep). */
int vft_sweep_batch_view(vft_ctx *ctx, int32_t slot, const void **hits, int64_t *best_j);
/* Restrict sweeps/out-distance passes to node ids [lo, hi): the shard a rank owns in a multi-GPU run
   (default [0, max_nodes)).  Hits keep global ids. */
int vft_set_shard(vft_ctx *ctx, int64_t lo, int64_t hi);
/* refresh_all != 0: the lazy out-distance refresh that precedes a sweep covers every node below maxnode instead of the
   shard.  A multi-GPU run that keeps the whole NJ state on every rank (host/NJDriver.h) needs that: out-distances are
   state, and every rank must refresh the same nodes at the same moments for the ranks to stay bit-identical; only the
   distances of the sweep itself are split.  With it vft_set_shard no longer forgets what it knows about staleness. */
int vft_set_shard_mode(vft_ctx *ctx, int32_t refresh_all);
/* Multi-GPU merge: d_all holds n_lists sorted lists of k records each (every rank's vft_sweep d_hits, all-gathered,
   DEVICE memory).  Produces the k best records under the same (criterion asc, id desc) order into hits (host, may be
   NULL) and d_out (device, may be NULL) — the result a single-rank sweep over the union of the shards would give. */
int vft_merge_hits(vft_ctx *ctx, const void *d_all, int32_t n_lists, int32_t k, void *hits, void *d_out);
/* The same for batches (vft_sweep_batch on every rank, one all-gather): d_all = [n_lists][n_seeds][k] records, the
   merged lists come back as [n_seeds][k] in hits (host, may be NULL) and d_out (device, may be NULL). */
int vft_merge_hits_batch(vft_ctx *ctx, const void *d_all, int32_t n_lists, int32_t n_seeds, int32_t k, void *hits,
                         void *d_out);
/* Diagnostics of the last sweep's top-k selection: info[0] = candidates that were rank-sorted, info[1] = extra
   refinement rounds that were needed (0 in the common case). */
int vft_sweep_info(vft_ctx *ctx, int64_t info[2]);
/* the same for slot `slot` of the last vft_sweep_batch */
int vft_sweep_batch_info(vft_ctx *ctx, int32_t slot, int64_t info[2]);
/* The full, unsorted result of the last sweep for ids [first, first+count): what `allhits[]` holds. */
int vft_sweep_results(vft_ctx *ctx, int64_t first, int64_t count, void *dist, void *weight, void *criterion);
/* setDistCriterion on an explicit pair list — transferBestHits / uniqueBestHits / getBestFromTopHits
   (NJ.tcc:4580-4613, 4786-4833, 4267-4298).  Outputs are host arrays of the context precision. */
int vft_pair_distances(vft_ctx *ctx, int64_t n, const int64_t *i, const int64_t *j, int64_t n_active,
                       int64_t n_diff_allow, double totdiam, void *dist, void *weight, void *criterion);
/* The same with n_force nodes whose out-distance is recomputed first unless it carries the stamp n_active (setOutDistance,
   NJ.tcc:1012-1015) - getBestFromTopHits' own node (NJ.tcc:4270) and the stale ends of the hits whose distance is known,
   which would otherwise be a vft_out_distances call and a second wait per list.  n == 0: vft_out_distances(force_ids). */
int vft_pair_distances_refresh(vft_ctx *ctx, int64_t n, const int64_t *i, const int64_t *j, int64_t n_force,
                               const int64_t *force_ids, int64_t n_active, int64_t n_diff_allow, double totdiam, void *dist,
                               void *weight, void *criterion);
/* setDistCriterion for the cross product of two lists of LEAVES (every id below n_seqs; nucleotides without a distance
   matrix): out[x * n_b + y] belongs to the pair (a[x], b[y]).  The close-neighbour transfers of setAllLeafTopHits
   (NJ.tcc:3957-3992 -> transferBestHits :4580-4613) are such blocks - up to m close neighbours x the seed's 2m best hits -
   and a pair list is the wrong shape for millions of integer seqDist counts (k_leaf_block, vft_kernels_nj.h).  Negative
   ids in b are allowed and give (1e20, 0, 1e20).  Out-distances staler than n_diff_allow are refreshed first, as in
   vft_pair_distances.  Outputs: host arrays of the context precision, n_a * n_b each. */
int vft_leaf_block_distances(vft_ctx *ctx, int64_t n_a, const int64_t *a, int64_t n_b, const int64_t *b, int64_t n_active,
                             int64_t n_diff_allow, double totdiam, void *dist, void *weight, void *criterion);
/* profileDist / seqDist itself (NJ.tcc:1167-1190, 1601-1624) for n pairs: the raw distance and weight, without the
   diameter correction and criterion of setDistCriterion.  The three distances that root the tree at the end of fastNJ
   (NJ.tcc:3110-3120) and every ME-phase distance are this call. */
int vft_profile_distances(vft_ctx *ctx, int64_t n, const int64_t *i, const int64_t *j, void *dist, void *weight);

/* ---- likelihood (ML phase)
 * pairLogLk (NJ.tcc:1192-1447) for n independent pairs; site_lk (n x n_pos doubles, host) may be NULL, when
 * given it receives the per-site likelihoods lkAB (what the reference multiplies into site_likelihoods[]).
 */
int vft_pair_loglk(vft_ctx *ctx, int64_t n, const int64_t *a, const int64_t *b, const double *length,
                   double *loglk, double *site_lk);
/* posteriorProfile (NJ.tcc:2137-2447) for n independent triples: out[k] = posterior(a[k], b[k], len1[k], len2[k]).
   A whole tree level of recomputeMLProfiles (NJ.tcc:3516-3539) is one call. */
int vft_posterior_profiles(vft_ctx *ctx, int64_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                           const double *len1, const double *len2);

/* Tree-refinement phase switch.  on != 0: vft_average_profiles writes plain per-node rows (vft_layout.h, "dense ML rows")
   instead of the tile streams and skips the self distances - for everything after fastNJ (NNIs, SPRs, up-profiles,
   branch lengths, supports), where single nodes are rewritten constantly and nothing sweeps.  vft_profile_distances,
   vft_split_supports and the likelihood calls read either layout; one-vs-all sweeps must not target nodes written in
   this mode. */
int vft_set_profile_rows(vft_ctx *ctx, int32_t on);

/* n unweighted averageProfile calls executed in order in ONE launch, where a later one may read an earlier one's output
   (recomputeProfile of a node, then of its parent; up-profiles down a path; NJ.tcc:3382-3473).  Needs
   vft_set_profile_rows(ctx, 1); n <= 256.  Stream-ordered. */
int vft_average_chain(vft_ctx *ctx, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b);

/* n_chains independent chains of averages in ONE launch: chain k = ops [chain_off[k], chain_off[k + 1]) of out / a / b
   (chain_off[0] = 0; a chain holds at most 256 ops, a call at most 4096).  Chains must not read what another chain of the
   same call writes.  The subtree schedule of the refinement stages (MLLengths::doNNIThreaded - the reference's
   `-threads T` traversal, NJ.tcc:6108-6160) queues one chain per subtree and step.  Stream-ordered. */
int vft_average_chains(vft_ctx *ctx, int32_t n_chains, const int32_t *chain_off, const int64_t *out, const int64_t *a,
                       const int64_t *b);

/* One step of a host-driven refinement walk (an SPR chain step, a minimum-evolution NNI of the one-thread order): the n unweighted
   averages queued since the last step, in order (as vft_average_chain; n may be 0), then the six raw profile distances AB AC AD BC BD CD
   of the quartet q[0..3] = A, B, C, D (as vft_profile_distances; chooseNNI, NJ.tcc:4836-4846) into dist[6] (numeric_t) - handed to the
   walk server below (= vft_walk_submit + vft_walk_collect); VFT_ERR_STATE while no server is running: the caller makes the two plain
   calls.  Results are bit-identical to them. */
int vft_walk_step(vft_ctx *ctx, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *q, void *dist);

/* The walk server: the same steps WITHOUT a launch each.  vft_walk_server_start leaves six workgroups resident on a stream of their own
   (csrc/vft_kernels_walk.h); from then on vft_walk_step hands its step over through a mailbox they poll (a round trip of ~2.5 us instead
   of ~11 for a launch and its completion wait) until vft_walk_server_stop - or any other call that launches on the context's stream,
   which retires the server first.  The walks of host/MLLengths.h (doSPR, the one-thread minimum-evolution NNIs: traverseSPR
   NJ.tcc:6185-6312, traverseNNI :5797-5990) start it around their loops.  vft_walk_submit / vft_walk_collect split a step in two so that a
   caller who does not need a step's distances to build the next step (the forced first NNI of an SPR chain, NJ.tcc:1820-1830) can have
   two steps in flight; q == NULL submits averages alone (collect with dist == NULL waits for their acknowledgement).  Results are
   bit-identical to vft_walk_step without the server.  VFT_ERR_STATE from start: rows missing, alignment too long for the staging, or the
   server switched off - the caller keeps the launch per step.  One server per process. */
int vft_walk_server_start(vft_ctx *ctx);
int vft_walk_server_stop(vft_ctx *ctx);
int vft_walk_submit(vft_ctx *ctx, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *q, uint32_t *ticket);
/* Both continuations of an SPR chain in one command (csrc/vft_kernels_walk.h "DUAL command"; findSPRSteps NJ.tcc:1805-1859): which NNI
   follows a chain step is one comparison of that step's own distances, so the step after it is built for either outcome and handed over
   while the step is still running; the resident workgroups evaluate `criteria[1] < criteria[2]` themselves (logCorrect, NJ.tcc:322-330,
   with glibc's log bit for bit) and take alternative 0 (B and C swapped) or 1 without a host round trip.  q0 / q1 == NULL: that
   alternative is not a device step; taken, the command is an empty one.  At most 18 averages in both alternatives together.  The command
   before it must be a step with distances.  vft_walk_dual_choice: what the workgroups chose (the caller, who makes the same
   comparison on the distances it collects, checks it). */
#define VFT_WALK_DUAL_MAX_AVERAGES 18
int vft_walk_submit_dual(vft_ctx *ctx, int32_t n0, const int64_t *out0, const int64_t *a0, const int64_t *b0, const int64_t *q0,
                         int32_t n1, const int64_t *out1, const int64_t *a1, const int64_t *b1, const int64_t *q1, int32_t scoredist,
                         uint32_t *ticket);
int vft_walk_dual_choice(vft_ctx *ctx, uint32_t ticket, int32_t *alt, int32_t *skipped);
/* logCorrect's flavour (NJ.tcc:322-330) of the steps that follow: 0 Jukes-Cantor, 1 scoredist-like.  Call before the first step of a walk
   that sends dual commands (the workgroups log-correct every step's distance for the comparison of a dual command behind it). */
int vft_walk_scoredist(vft_ctx *ctx, int32_t scoredist);
int vft_walk_collect(vft_ctx *ctx, uint32_t ticket, void *dist);
/* tools builds (-DVFT_WALK_TIMING): clock ticks (100 MHz) workgroup 0 of the servers of this process spent per phase -
   [0] waiting for a command, [1] averages, [2] waiting for the other workgroups' averages, [3] the pair's columns, [4] the ordered
   sums, [5] the answer, [6] steps with distances, [7] averages; zeros in a production build */
int vft_walk_server_ticks(int64_t *out, int32_t n);

/* differ[k] = 1 when the profiles of nodes a[k] and b[k] are not bit-identical (weights, codes, vectors), else 0; n <= 4096.  The
   speculative SPR rounds (host/MLLengths.h, doSPRSpeculative) ask whether an attempt that left the tree as it was also left the
   profiles it recomputed as they were.  Waits. */
int vft_profiles_differ(vft_ctx *ctx, int64_t n, const int64_t *a, const int64_t *b, int32_t *differ);
/* the max_nodes the context was created with (rows beyond the tree's nodes and up-profile slots serve as private scratch) */
int vft_get_max_nodes(vft_ctx *ctx, int64_t *max_nodes);
/* the alphabet size the context was created with (vft_config.n_codes: 4 or 20) */
int vft_get_n_codes(vft_ctx *ctx, int32_t *n_codes);

/* ---- ML branch lengths (optimizeAllBranchLengths, NJ.tcc:5006-5113)
 * branchlength[] (NJ.h) lives on the device as numeric_t[max_nodes]; set / get copy a range (get waits). */
int vft_branch_lengths_set(vft_ctx *ctx, int64_t first, int64_t count, const void *values);
int vft_branch_lengths_get(vft_ctx *ctx, int64_t first, int64_t count, void *values);
/* the same by index list: values[k] = branchlength[idx[k]] (gather; waits) / branchlength[idx[k]] = values[k] (scatter; stream-ordered);
   n <= 65536.  With several ranks the lanes of the subtree schedule exchange the lengths each rank's share of a batch optimised
   (host/MLLengths.h). */
int vft_branch_lengths_gather(vft_ctx *ctx, int64_t n, const int64_t *idx, void *values);
int vft_branch_lengths_scatter(vft_ctx *ctx, int64_t n, const int64_t *idx, const void *values);
/* vft_posterior_profiles with len1[k] = branchlength[len_idx_a[k]], len2[k] = branchlength[len_idx_b[k]] read on the
   device at execution time.  Stream-ordered (does not wait): recomputeMLProfiles' levels and the up-profiles of
   getUpProfile(useML = true) (NJ.tcc:3382-3434) queue behind the optimiser launches that produce their lengths. */
int vft_posterior_profiles_blen(vft_ctx *ctx, int64_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                                const int64_t *len_idx_a, const int64_t *len_idx_b);
/* n vft_posterior_profiles_blen calls executed in order in ONE launch, later ones may read earlier outputs (n <= 256). */
int vft_posterior_chain_blen(vft_ctx *ctx, int32_t n, const int64_t *out, const int64_t *a, const int64_t *b,
                             const int64_t *len_idx_a, const int64_t *len_idx_b);
/* The same for n_chains independent chains in one launch (see vft_average_chains). */
int vft_posterior_chains_blen(vft_ctx *ctx, int32_t n_chains, const int32_t *chain_off, const int64_t *out, const int64_t *a,
                              const int64_t *b, const int64_t *len_idx_a, const int64_t *len_idx_b);
/* The body of traverseOptimizeAllBranchLengths' loop (NJ.tcc:5025-5064) for n independent splits, one workgroup each:
   ids[3k..3k+2] = the three profiles around split k (children 0 and 1 + the up-profile, or the root's three children),
   len_idx[3k..3k+2] = the branchlength[] slots they own.  Two passes over the three branches; branch i gets
   MLPairOptimize(P_i, posteriorProfile(P_i+1, P_i+2)) = onedimenmin/Brent on pairLogLk (NJ.tcc:1790-1803, 7025-7178)
   with ftol = MLFTolBranchLength, atol = MLMinBranchLengthTolerance, limits [MLMinBranchLength (vft_set_ml_limits), 6].
   recompute[k] >= 0: afterwards that node's profile = posteriorProfile(ids[3k], ids[3k+1]) with the new lengths
   (recomputeProfile, NJ.tcc:3436); -1 for the root (all entries of one call alike).  Any n_pos the context was created for: up to 2 048 columns
   the search keeps its profiles in registers, beyond that in a workspace in device memory (csrc/vft_kernels_ml_long.h; the same numbers).
   Stream-ordered. */
int vft_ml_optimize_splits(vft_ctx *ctx, int64_t n, const int64_t *ids, const int64_t *len_idx, const int64_t *recompute,
                           double ftol, double atol);
/* testSplitsML (NJ.tcc:6800-6999) for n independent internal splits, one workgroup each: ids[4k..4k+3] = A, B, C, D of
   setupABCD (NJ.tcc:1942-1975), len_idx[5k..5k+4] = the branchlength[] slots of A, B, C, D and of the split itself.
   loglk[3k..3k+2] = quartet log-likelihood of AB|CD with the current lengths (MLQuartetLogLk, NJ.tcc:5412), of AC|BD
   and of AD|BC after MLQuartetOptimize (NJ.tcc:1650-1788), the better alternative optimised a second time when it is
   within close_limit (Constants::closeLogLkLimit = 5) of AB|CD or when always_second_pass (-mlacc 2).
   n_boot > 0: support[k] = SHSupport (NJ.tcc:1126-1165) over the resamples col[n_boot][n_pos] (resampleColumns,
   NJ.tcc:705-727); the caller sets the supports of bad splits to 0 (NJ.tcc:6956, 6992).  lengths (may be NULL):
   [n][2][5] optimised A, B, C, D, I lengths of the two alternatives (what an NNI would adopt).
   n_pos <= 1024; with n_boot > 0 also n_pos <= 2500. */
int vft_ml_split_tests(vft_ctx *ctx, int64_t n, const int64_t *ids, const int64_t *len_idx, double ftol, double atol,
                       double close_limit, int32_t always_second_pass, double *loglk, int32_t n_boot, const int32_t *col,
                       double *support, double *lengths);
/* MLQuartetNNI (NJ.tcc:4885-5004), the evaluation of one maximum-likelihood NNI, for n independent quartets (DoNNI
   evaluates one at a time): ids / len_idx as in vft_ml_split_tests.  Up to two rounds (ml_accuracy < 2; else
   ml_accuracy rounds) of MLQuartetOptimize for AB|CD - with the star-topology test - AC|BD and AD|BC, dropping
   alternatives that fall close_limit behind.  choice: 0 = keep AB|CD, 1 = AC|BD (swap B and C), 2 = AD|BC; criteria =
   the three log-likelihoods (-1e20 for the alternatives after a successful star test).  The winner's branch lengths
   are written to the device's branchlength[] as DoNNI assigns them (NJ.tcc:5889-5915). */
typedef struct vft_quartet_nni {
    double criteria[3];
    int32_t choice;
    int32_t star;
} vft_quartet_nni;
int vft_ml_quartet_nni(vft_ctx *ctx, int64_t n, const int64_t *ids, const int64_t *len_idx, double ftol, double atol,
                       double close_limit, int32_t ml_accuracy, vft_quartet_nni *results);
/* vft_ml_quartet_nni with flags.  VFT_QUARTET_NO_STAR_TEST: AB|CD is optimised without the star-topology test - what the
   reference does when MLQuartetNNI runs outside a parallel region in a run with threads > 1 (the `omp sections` variant,
   NJ.tcc:4925-4931, passes no bStarTest): the serial part of a threaded NNI round. */
#define VFT_QUARTET_NO_STAR_TEST 1
int vft_ml_quartet_nni_flags(vft_ctx *ctx, int64_t n, const int64_t *ids, const int64_t *len_idx, double ftol, double atol,
                             double close_limit, int32_t ml_accuracy, int32_t flags, vft_quartet_nni *results);
/* likelihood evaluations (pairLogLk calls of the reference) made by vft_ml_optimize_splits since the last query */
int vft_ml_eval_count(vft_ctx *ctx, int64_t *evals);

/* Local-bootstrap supports (splitSupport, NJ.tcc:607-702) of n splits (a[k], b[k]) | (c[k], d[k]): col holds n_boot
   resamples of n_pos column indices each ([n_boot][n_pos], host; resampleColumns NJ.tcc:705-727); support[k] = the
   fraction of resamples in which the split beats both alternative pairings of the four profiles. */
int vft_split_supports(vft_ctx *ctx, int64_t n, const int64_t *a, const int64_t *b, const int64_t *c, const int64_t *d,
                       int32_t n_boot, const int32_t *col, double *support);

/* Diagnostics: out[i] = log(x[i]) evaluated on the device the way the ML kernels evaluate the final logarithm of a
   float-precision matrix-model pairLogLk - glibc 2.35's algorithm (csrc/vft_glibc_log.h), not the device math library.
   x, out: host arrays of n doubles (positive, normal). */
int vft_debug_log(vft_ctx *ctx, int64_t n, const double *x, double *out);
/* Test / tool hooks: which of two equivalent kernel variants a context uses (results are identical; the tests compare
   them).  No environment variable selects kernels in this library. */
#define VFT_DEBUG_NO_FUSED_REFRESH 1   /* value != 0: pair lists with refreshes as two launches instead of one */
#define VFT_DEBUG_PAIR_THREADS 2       /* threads per pair of the short-list kernels (0 = the built-in choice) */
#define VFT_DEBUG_NO_PAIR_STAGING 3    /* value != 0: short pair lists read their ids from the mapped ring directly */
#define VFT_DEBUG_GENERIC_OUTPROFILE 4 /* value != 0: vft_out_profile_full always takes the one-thread-per-column kernel */
#define VFT_DEBUG_FAULT_NO_FLAG 5      /* value != 0: fault injection - the next wait for a completion flag waits for a value no kernel publishes */
#define VFT_DEBUG_WAIT_LIMIT_MS 6      /* the longest a wait for a completion flag may last while the stream is busy (default 120 000) */
#define VFT_DEBUG_WIDE_GLUE 7          /* value != 0: vft_nj_engine_create takes the 1 024-thread glue kernel (lists beyond 1 024 hits) at any size */
#define VFT_DEBUG_NO_WALK_SERVER 9      /* value != 0: vft_walk_server_start answers VFT_ERR_STATE - the walks keep one launch per step (tests compare) */
#define VFT_DEBUG_WALK_DEVICE_MAILBOX 10 /* value != 0: the server's mailbox in device memory written through the PCIe aperture (large-BAR boxes) instead of pinned host memory */
#define VFT_DEBUG_WALK_SERVER_STRIDE 11 /* 1: the server's six workgroups on six XCDs instead of one (placement is for speed only; tests run both) */
#define VFT_DEBUG_POISON_SELECTION 14    /* fills the selection's candidate buffers of every slot with 0x7f bytes - what a recycled allocation holds - before the next sweep (tests: a collection that overflows must not look at entries it never stored) */
#define VFT_DEBUG_ML_LONG 16            /* value != 0: the ML line searches (vft_ml_optimize_splits, vft_ml_quartet_nni*, vft_ml_split_tests) run the workspace kernels of alignments beyond 2 048 columns at any length (tests compare with the register-resident kernels) */
#define VFT_DEBUG_NO_MULTI_SWEEP 12     /* 1: vft_sweep_batch sweeps its seeds one launch each instead of four per pass over the targets (tests compare); 2 / 4: that many seeds per pass whatever the shard; 0: the built-in choice */
int vft_debug_option(vft_ctx *ctx, int32_t option, int64_t value);

/* ---- measurement helpers used by bench.py (HIP events on the context's stream) */
int vft_timer_start(vft_ctx *ctx);
int vft_timer_stop_ms(vft_ctx *ctx, float *ms);
/* average duration (ms) of the dominant kernel's launches since the last vft_timer_start: k_sweep_nt, the sweep over
   internal-profile targets (and over every target when the seed is a leaf) */
int vft_sweep_kernel_ms(vft_ctx *ctx, float *avg_ms, int64_t *launches);
/* same for the sweep's second launch, k_sweep_nt_table (leaf targets of a profile seed; 0 ms when it did not run) */
int vft_sweep_table_kernel_ms(vft_ctx *ctx, float *avg_ms, int64_t *launches);
/* the sweeps those launches stand for: vft_sweep_batch takes consecutive leaf seeds four (two) per pass over the targets
   (k_sweep_nt_leafq_multi), so a launch can be several seeds' sweeps (setAllLeafTopHits, NJ.tcc:3798-3880) */
int vft_sweep_kernel_sweeps(vft_ctx *ctx, int64_t *sweeps);

#ifdef __cplusplus
}
#endif
#endif
