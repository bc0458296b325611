"""ctypes binding of the C ABI (include/vft_hip.h) — the only way Python reaches the HIP backend.

There is no CPU fallback: if the shared library is missing or no HIP device is present, construction raises.
Method names follow the reference members they replace (src/NeighbourJoining.tcc), so that the parity tests read
like calls into the reference.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# VFT_LIB_DIR: a variant build of tools/ (veryfasttree_amd/build.py with VFT_EXTRA_HIPCC_FLAGS: both libraries in
# build/variants/<hash>/); VFT_HIP_LIB: another build of the HIP library alone.  Never set by tests or bench.py.
LIB_DIR = os.environ.get("VFT_LIB_DIR") or os.path.join(HERE, "lib")
LIB_PATH = os.environ.get("VFT_HIP_LIB") or os.path.join(LIB_DIR, "libvft_hip.so")
NOCODE = 127
# vft_nj_options.debug_flags (include/vft_host.h): tests and tools choose between equivalent loops explicitly
DEBUG_HOST_JOINS, DEBUG_HOST_LISTS, DEBUG_HOST_RESET, DEBUG_LEVEL_LENGTHS, DEBUG_NO_WALK_SERVER = 1, 2, 4, 16, 128
DEBUG_NO_WALK_DUAL = 512
SHARD_LEAF_BLOCKS = 64

P = C.c_void_p
I64 = C.c_int64
I32 = C.c_int32

EXPORTS = [
    "vft_device_malloc", "vft_device_free", "vft_device_upload", "vft_create", "vft_destroy", "vft_last_error", "vft_set_stream", "vft_synchronize", "vft_upload_leaves",
    "vft_set_distance_matrix", "vft_set_transition_matrix", "vft_set_rates", "vft_set_ml_limits", "vft_set_jc_exact", "vft_set_parents",
    "vft_set_node_scalars", "vft_get_node_scalars", "vft_set_out_distances", "vft_get_out_distances", "vft_out_distance_mirror", "vft_set_max_node",
    "vft_profile_upload", "vft_profile_download", "vft_profile_nvectors", "vft_average_profiles", "vft_out_profile_full", "vft_out_profile_partial", "vft_out_profile_finish",
    "vft_out_profile_update", "vft_out_profile_upload", "vft_out_profile_download", "vft_out_distances", "vft_sweep",
    "vft_sweep_batch", "vft_sweep_batch_view", "vft_set_shard", "vft_merge_hits", "vft_merge_hits_batch", "vft_sweep_info", "vft_sweep_batch_info", "vft_sweep_results", "vft_pair_distances", "vft_pair_loglk", "vft_posterior_profiles", "vft_set_profile_rows", "vft_average_chain", "vft_average_chains", "vft_profiles_differ", "vft_get_max_nodes", "vft_get_n_codes", "vft_walk_step", "vft_walk_server_start", "vft_walk_server_stop", "vft_walk_submit", "vft_walk_submit_dual", "vft_walk_dual_choice", "vft_walk_scoredist", "vft_walk_collect", "vft_walk_server_ticks", "vft_posterior_chains_blen", "vft_ml_quartet_nni_flags", "vft_branch_lengths_set", "vft_branch_lengths_get", "vft_branch_lengths_gather", "vft_branch_lengths_scatter", "vft_posterior_profiles_blen", "vft_posterior_chain_blen", "vft_ml_optimize_splits", "vft_ml_split_tests", "vft_ml_quartet_nni", "vft_ml_eval_count",
    "vft_join_nodes", "vft_profile_distances", "vft_split_supports", "vft_timer_start", "vft_timer_stop_ms", "vft_sweep_kernel_ms", "vft_sweep_table_kernel_ms", "vft_sweep_kernel_sweeps",
    "vft_debug_log", "vft_debug_option", "vft_tophits_create", "vft_tophits_upload", "vft_tophits_download", "vft_tophits_best", "vft_tophits_join", "vft_tophits_refresh", "vft_nj_engine_create", "vft_nj_engine_set_state", "vft_nj_engine_get_state", "vft_nj_engine_visible_set", "vft_nj_engine_visible_get", "vft_nj_engine_nodes_set", "vft_nj_engine_topvisible_set", "vft_nj_engine_topvisible_get", "vft_nj_engine_reset_candidates", "vft_nj_engine_enqueue", "vft_nj_engine_poll", "vft_nj_engine_resume", "vft_nj_engine_log", "vft_nj_engine_adopt", "vft_leaf_block_distances", "vft_set_shard_mode", "vft_join_fused", "vft_block_distances", "vft_pair_distances_refresh",
]


class VftError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("device", I32), ("precision", I32), ("n_codes", I32), ("reserved", I32), ("n_seqs", I64),
                ("n_pos", I64), ("max_nodes", I64)]


HIT_F32 = np.dtype([("j", np.int32), ("dist", np.float32), ("weight", np.float32), ("criterion", np.float32)])
HIT_F64 = np.dtype([("j", np.int64), ("dist", np.float64), ("weight", np.float64), ("criterion", np.float64)])

HOST_LIB_PATH = os.path.join(LIB_DIR, "libvft_host.so")
HOST_EXPORTS = ["vft_nj_run", "vft_nj_newick", "vft_nj_ml_newick", "vft_nj_last_join_crcs", "vft_nj_last_stage_seconds", "vft_nj_last_walk_dual", "vft_nj_last_lane_exchange", "vft_nj_lane_share", "vft_nj_out_profile_block", "vft_nj_last_gamma", "vft_tree_partitioning", "vft_knuth_stream", "vft_ml_lengths", "vft_gtr_tables",
                "vft_aa_model_tables", "vft_blosum45_tables", "vft_aa_model_as_distance_tables"]


class _NJOptions(C.Structure):
    _fields_ = [("fastest", I32), ("use_tophits_2nd", I32), ("tophits_mult", C.c_double), ("tophits_close", C.c_double),
                ("tophits_refresh", C.c_double), ("topvisible_mult", C.c_double), ("stale_out_limit", C.c_double),
                ("f_reset_out_profile", C.c_double), ("n_reset_out_profile", I32), ("tophits2_safety", I32),
                ("tophits2_mult", C.c_double), ("tophits2_refresh", C.c_double), ("scoredist", I32), ("mllen", I32), ("me_nni", I32), ("ml_nni", I32), ("spr", I32), ("gtr", I32), ("aa_model", I32), ("comm", P), ("threads", I32), ("debug_flags", I32), ("gamma", I32), ("out_profile_parts", I32), ("pad_", I32)]


_lib = None
_host_lib = None


def load_host_library():
    """libvft_host.so: the C++ host NJ driver (include/vft_host.h)."""
    global _host_lib
    if _host_lib is None:
        load_library()
        if not os.path.exists(HOST_LIB_PATH):
            raise VftError("host driver %s is missing: run build()" % HOST_LIB_PATH)
        _host_lib = C.CDLL(HOST_LIB_PATH)
    return _host_lib


AA_MODELS = {None: 0, "": 0, "jtt": 1, "wag": 2, "lg": 3}

_ALLGATHER = C.CFUNCTYPE(C.c_int, P, I64, I32)


class _Comm(C.Structure):
    _fields_ = [("rank", I32), ("world", I32), ("allgather", _ALLGATHER), ("user", P), ("d_send", P), ("d_recv", P),
                ("d_cap", I64), ("h_send", P), ("h_recv", P), ("h_cap", I64)]


class TorchComm:
    """vft_comm (include/vft_host.h) over torch.distributed: exchange buffers as torch tensors (device + pinned host) and
    ONE collective, all_gather_into_tensor - RCCL over xGMI with the nccl backend; with gloo (CPU tests, two ranks on one
    GPU) the device buffers travel through the host.  `dist` must be initialised; device = this rank's GPU ordinal."""

    def __init__(self, dist, device, d_cap=1 << 20, h_cap=64 << 20):
        import torch
        self.dist, self.torch = dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.gloo = dist.get_backend() != "nccl"
        dev = torch.device("cuda", device)
        self.d_send = torch.zeros(d_cap, dtype=torch.uint8, device=dev)
        self.d_recv = torch.zeros(self.world * d_cap, dtype=torch.uint8, device=dev)
        self.h_send = torch.zeros(h_cap, dtype=torch.uint8)
        self.h_recv = torch.zeros(self.world * h_cap, dtype=torch.uint8)
        self.calls = 0
        self.bytes = 0
        self.device_calls = self.device_bytes = 0   # of which through the device buffers (sweep lists)

        def allgather(_user, nbytes, device_side):
            try:
                self.calls += 1
                self.bytes += int(nbytes) * self.world
                if device_side:
                    self.device_calls += 1
                    self.device_bytes += int(nbytes) * self.world
                if device_side:
                    if self.gloo:
                        out = torch.zeros(self.world * nbytes, dtype=torch.uint8)
                        dist.all_gather_into_tensor(out, self.d_send[:nbytes].cpu())
                        self.d_recv[:self.world * nbytes].copy_(out)
                        torch.cuda.synchronize(dev)
                    else:
                        torch.cuda.synchronize(dev)   # the driver's stream is not torch's
                        dist.all_gather_into_tensor(self.d_recv[:self.world * nbytes], self.d_send[:nbytes])
                        torch.cuda.synchronize(dev)
                elif self.gloo:
                    dist.all_gather_into_tensor(self.h_recv[:self.world * nbytes], self.h_send[:nbytes])
                else:
                    out = torch.zeros(self.world * nbytes, dtype=torch.uint8, device=dev)
                    dist.all_gather_into_tensor(out, self.h_send[:nbytes].to(dev))
                    self.h_recv[:self.world * nbytes].copy_(out.cpu())
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                sys.stderr.write("TorchComm.allgather failed: %r\n" % (e,))
                return 1

        self._cb = _ALLGATHER(allgather)
        self.struct = _Comm(self.rank, self.world, self._cb, None, self.d_send.data_ptr(), self.d_recv.data_ptr(), d_cap,
                            self.h_send.data_ptr(), self.h_recv.data_ptr(), h_cap)

    def pointer(self):
        return C.cast(C.pointer(self.struct), P)


def nj_run(ops, codes, fastest=False, max_joins=-1, tophits_refresh=None, second_level=None, scoredist=False, aa_model=None,
           tophits_mult=1.0, comm=None, debug_flags=0, out_profile_parts=0):
    """fastNJ through the C++ host driver.  Returns (joins[n,3], criterion[n]).
    second_level defaults to `fastest`, as in the reference at one thread (-fastest turns -2nd on)."""
    lib = load_host_library()
    codes = np.ascontiguousarray(codes, np.uint8)
    n, L = codes.shape
    if second_level is None:
        second_level = fastest
    opt = _NJOptions(1 if fastest else 0, 1 if second_level else 0, float(tophits_mult), -1.0,
                     tophits_refresh if tophits_refresh is not None else (0.5 if fastest else 0.8), 1.5, 0.01, 0.02,
                     200, 3, 1.0, 0.6, 1 if scoredist else 0, 0, 0, 0, 0, 0, AA_MODELS[aa_model],
                     comm.pointer() if comm is not None else None, 1, int(debug_flags), 0, int(out_profile_parts), 0)
    joins = np.zeros((max(n - 3, 1), 3), np.int64)
    crit = np.zeros(max(n - 3, 1), np.float64)
    nj = I64(0)
    err = C.create_string_buffer(512)
    rc = lib.vft_nj_run(ops.ctx, _ptr(codes), I64(n), I64(L), I32(ops.dt.itemsize), C.byref(opt), I64(max_joins),
                        _ptr(joins), _ptr(crit), C.byref(nj), err, I32(512))
    if rc != 0:
        raise VftError(err.value.decode() or "vft_nj_run failed")
    return joins[:nj.value], crit[:nj.value]


def ml_lengths(ops, n_seqs, parent, child, root, branchlength, rounds=1, recompute_first=True, n_leaf_gaps=-1,
               ftol=0.001, atol=None, parallel=False):
    """`-mllen` on a fixed topology through the C++ host driver (vft_ml_lengths): returns (branchlength after the
    last round, loglk per round, likelihood evaluations).  atol defaults to MLMinBranchLengthTolerance of ops.dt."""
    lib = load_host_library()
    parent = np.ascontiguousarray(parent, np.int64)
    child = np.ascontiguousarray(child, np.int64)
    bl = np.ascontiguousarray(branchlength, ops.dt).copy()
    if atol is None:
        atol = 1.0e-4 if ops.dt == np.float32 else 1.0e-9
    loglk = np.zeros(max(rounds, 1), np.float64)
    evals = I64(0)
    err = C.create_string_buffer(512)
    rc = lib.vft_ml_lengths(ops.ctx, I64(n_seqs), I64(len(parent)), I64(ops.n_pos), I32(ops.dt.itemsize), _ptr(parent),
                            _ptr(child), I64(root), _ptr(bl), I32((1 if recompute_first else 0) | (2 if parallel else 0)), I32(rounds),
                            C.c_double(ftol), C.c_double(atol), I64(n_leaf_gaps), _ptr(loglk), C.byref(evals), err, I32(512))
    if rc != 0:
        raise VftError(err.value.decode() or "vft_ml_lengths failed")
    return bl, loglk[:rounds], evals.value

def gtr_tables(rates, freq, dtype=np.float32):
    """createGTR of the host driver (veryfasttree_amd/host/GtrModel.h): dict of stat, statinv, eigenval, codefreq[5,4],
    eigeninv[4,4], eigeninvT[4,4] in double."""
    lib = load_host_library()
    r, f = np.ascontiguousarray(rates, np.float64), np.ascontiguousarray(freq, np.float64)
    out = dict(stat=np.zeros(4), statinv=np.zeros(4), eigenval=np.zeros(4), codefreq=np.zeros((5, 4)), eigeninv=np.zeros((4, 4)),
               eigeninvT=np.zeros((4, 4)))
    rc = lib.vft_gtr_tables(_ptr(r), _ptr(f), I32(np.dtype(dtype).itemsize), _ptr(out["stat"]), _ptr(out["statinv"]), _ptr(out["eigenval"]), _ptr(out["codefreq"]),
                            _ptr(out["eigeninv"]), _ptr(out["eigeninvT"]))
    if rc != 0:
        raise VftError("vft_gtr_tables failed")
    return out


def aa_model_tables(model, dtype=np.float32):
    """The built-in amino-acid model `model` ("jtt" / "wag" / "lg") as the host driver builds it (host/AAModels.h): dict of
    stat, statinv, eigenval, codefreq[21,20], eigeninv[20,20], eigeninvT[20,20] in double."""
    lib = load_host_library()
    out = dict(stat=np.zeros(20), statinv=np.zeros(20), eigenval=np.zeros(20), codefreq=np.zeros((21, 20)),
               eigeninv=np.zeros((20, 20)), eigeninvT=np.zeros((20, 20)))
    rc = lib.vft_aa_model_tables(I32(AA_MODELS[model]), I32(np.dtype(dtype).itemsize), _ptr(out["stat"]), _ptr(out["statinv"]),
                                 _ptr(out["eigenval"]), _ptr(out["codefreq"]), _ptr(out["eigeninv"]), _ptr(out["eigeninvT"]))
    if rc != 0:
        raise VftError("vft_aa_model_tables failed")
    return out


def distance_tables(model=None, dtype=np.float32):
    """model None: the BLOSUM45-derived distance matrix (vft_blosum45_tables); else the model's transition matrix in the
    distance-matrix slots (vft_aa_model_as_distance_tables).  dict of distances, codefreq [20,20], eigenval, eigentot."""
    lib = load_host_library()
    out = dict(distances=np.zeros((20, 20)), codefreq=np.zeros((20, 20)), eigenval=np.zeros(20), eigentot=np.zeros(20))
    args = (_ptr(out["distances"]), _ptr(out["codefreq"]), _ptr(out["eigenval"]), _ptr(out["eigentot"]))
    if model is None:
        rc = lib.vft_blosum45_tables(I32(np.dtype(dtype).itemsize), *args)
    else:
        rc = lib.vft_aa_model_as_distance_tables(I32(AA_MODELS[model]), I32(np.dtype(dtype).itemsize), *args)
    if rc != 0:
        raise VftError("distance tables failed")
    return out


def uniquify(codes):
    """First-occurrence uniquify of alignment rows (Uniquify, Alignment.cpp:494-526).
    Returns (unique_first[u] = row of unique sequence u, aln_next[k] = next row with the same sequence or -1)."""
    codes = np.ascontiguousarray(codes, np.uint8)
    first_of, unique_first, last = {}, [], {}
    aln_next = np.full(len(codes), -1, np.int64)
    for k, row in enumerate(codes):
        key = row.tobytes()
        if key not in first_of:
            first_of[key] = k
            unique_first.append(k)
        else:
            aln_next[last[key]] = k
        last[key] = k
    return np.array(unique_first, np.int64), aln_next


def nj_newick(make_ops, codes_all, names, fastest=False, second_level=None, dtype=np.float32, me_lengths=False,
              unique=None, scoredist=False, n_bootstrap=0, mllen=0, return_loglk=False, return_rates=False, me_nni=False, ml_nni=0, spr=0, gtr=False, return_gtr=False,
              aa_model=None, comm=None, threads=1, debug_flags=0, gamma=False, out_profile_parts=0):
    """The NJ phase of the whole alignment `codes_all` (duplicates included) as the reference's "NJ" tree string.
    make_ops(n_unique, n_pos) -> HipProfileOps for the unique sequences (max_nodes >= 3 * n_unique with me_lengths:
    then the tree carries the minimum-evolution branch lengths, the final output of -noml -nome -nosupport)."""
    lib = load_host_library()
    codes_all = np.ascontiguousarray(codes_all, np.uint8)
    # unique = (unique_first, aln_next) when the caller has uniquified the sequence STRINGS like the reference does
    # (two different characters may share a code); otherwise rows with equal codes are taken as duplicates
    unique_first, aln_next = unique if unique is not None else uniquify(codes_all)
    unique_first = np.ascontiguousarray(unique_first, np.int64)
    aln_next = np.ascontiguousarray(aln_next, np.int64)
    codes = np.ascontiguousarray(codes_all[unique_first])
    n, L = codes.shape
    ops = make_ops(n, L)
    if second_level is None:
        second_level = fastest
    opt = _NJOptions(1 if fastest else 0, 1 if second_level else 0, 1.0, -1.0, 0.5 if fastest else 0.8, 1.5, 0.01, 0.02,
                     200, 3, 1.0, 0.6, 1 if scoredist else 0, int(mllen), 1 if me_nni else 0, int(ml_nni), int(spr), 1 if gtr else 0,
                     AA_MODELS[aa_model], comm.pointer() if comm is not None else None, int(threads), int(debug_flags), 1 if gamma else 0, int(out_profile_parts), 0)
    blob = b"".join(nm.encode() + b"\0" for nm in names)
    cap = 64 * len(names) + len(blob) + 1024
    out = C.create_string_buffer(cap)
    olen = I64(0)
    err = C.create_string_buffer(512)
    loglk = np.zeros(64, np.float64)
    n_rounds = I32(0)
    rates = np.zeros(64, np.float64)
    n_rates = I32(0)
    ratecat = np.zeros(L, np.int32)
    gtr_out = np.zeros(10, np.float64)
    rc = lib.vft_nj_ml_newick(ops.ctx, _ptr(codes), I64(n), I64(L), I32(np.dtype(dtype).itemsize), C.byref(opt),
                              I32(1 if me_lengths else 0), I32(n_bootstrap), _ptr(unique_first), _ptr(aln_next),
                              I64(len(codes_all)), blob, out, I64(cap), C.byref(olen), _ptr(loglk), I32(64),
                              C.byref(n_rounds), _ptr(rates), I32(64), C.byref(n_rates), _ptr(ratecat), _ptr(gtr_out), err, I32(512))
    if rc != 0:
        raise VftError(err.value.decode() or "vft_nj_ml_newick failed")
    if return_gtr:
        return out.value.decode(), loglk[:n_rounds.value], gtr_out[:6], gtr_out[6:]
    if return_rates:
        return out.value.decode(), loglk[:n_rounds.value], rates[:n_rates.value], ratecat
    if return_loglk:
        return out.value.decode(), loglk[:n_rounds.value]
    return out.value.decode()


STAGES = ("nj", "me_nni_spr", "of_which_spr", "me_lengths_supports", "ml_stage", "of_which_ml_nni", "of_which_sh_supports", "of_which_model_fits")


def last_gamma():
    """(Gamma(nCat) log-likelihood, alpha, length factor) of the last nj_newick(..., gamma=True): the reference's "Gamma(20) LogLk" line"""
    out = np.zeros(3, np.float64)
    load_host_library().vft_nj_last_gamma(_ptr(out))
    return tuple(float(x) for x in out)


def last_stage_seconds():
    """dict: wall-clock per stage of the last nj_newick of this process + the lane / SPR counters (vft_nj_last_stage_seconds)"""
    lib = load_host_library()
    sec = np.zeros(8, np.float64)
    cnt = np.zeros(4, np.int64)
    lib.vft_nj_last_stage_seconds(_ptr(sec), _ptr(cnt))
    out = {k: round(float(v), 2) for k, v in zip(STAGES, sec)}
    out.update(lane_steps=int(cnt[0]), lane_work=int(cnt[1]), spr_steps=int(cnt[2]), spr_moves=int(cnt[3]))
    dual = np.zeros(2, np.int64)
    lib.vft_nj_last_walk_dual(_ptr(dual))
    out.update(spr_dual_commands=int(dual[0]), spr_dual_continuations=int(dual[1]))
    return out


def last_lane_exchange():
    """(all-gathers, bytes received) of the lanes-across-ranks exchange of the last nj_newick of this process"""
    lib = load_host_library()
    out = np.zeros(2, np.int64)
    lib.vft_nj_last_lane_exchange(_ptr(out))
    return int(out[0]), int(out[1])


def last_join_crcs():
    """(chunk, n_joins, crcs): CRC-32 per chunk of joins of the last NJ run of this process, the joins behind the last complete chunk as
    one shorter chunk at the end (vft_nj_last_join_crcs)"""
    lib = load_host_library()
    chunk, nj, nc = I64(0), I64(0), I64(0)
    lib.vft_nj_last_join_crcs(C.byref(chunk), C.byref(nj), None, I64(0), C.byref(nc))
    crcs = np.zeros(max(nc.value, 1), np.uint32)
    lib.vft_nj_last_join_crcs(None, None, _ptr(crcs), I64(nc.value), None)
    return chunk.value, nj.value, crcs[:nc.value]


def load_library():
    """Load libvft_hip.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VftError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                           % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.vft_last_error.restype = C.c_char_p
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(P)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class HipProfileOps:
    """Device-resident profile arena + the batched profile operations of the hot path."""

    def __init__(self, n_seqs, n_pos, n_codes=4, dtype=np.float32, max_nodes=None, device=0):
        self.lib = load_library()
        self.dt = np.dtype(dtype)
        if self.dt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise VftError("dtype must be float32 or float64")
        self.n_seqs, self.n_pos, self.n_codes = int(n_seqs), int(n_pos), int(n_codes)
        self.max_nodes = int(max_nodes if max_nodes is not None else 2 * n_seqs)
        cfg = _Config(device, self.dt.itemsize, n_codes, 0, n_seqs, n_pos, self.max_nodes)
        self.ctx = P()
        rc = self.lib.vft_create(C.byref(self.ctx), C.byref(cfg))
        if rc != 0:
            msg = self.lib.vft_last_error(self.ctx).decode() if self.ctx else "vft_create failed"
            if self.ctx:
                self.lib.vft_destroy(self.ctx)
                self.ctx = None
            raise VftError(msg)
        self.hit_dtype = HIT_F32 if self.dt == np.float32 else HIT_F64

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vft_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise VftError(self.lib.vft_last_error(self.ctx).decode())

    def real(self, a):
        return np.ascontiguousarray(a, dtype=self.dt)

    # ---- plumbing
    def set_stream(self, stream_handle):
        self._chk(self.lib.vft_set_stream(self.ctx, P(stream_handle)))

    def synchronize(self):
        self._chk(self.lib.vft_synchronize(self.ctx))

    def set_shard(self, lo, hi):
        self._chk(self.lib.vft_set_shard(self.ctx, I64(lo), I64(hi)))

    def set_max_node(self, maxnode):
        self._chk(self.lib.vft_set_max_node(self.ctx, I64(maxnode)))

    # ---- inputs
    def upload_leaves(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        assert codes.shape == (self.n_seqs, self.n_pos)
        self._chk(self.lib.vft_upload_leaves(self.ctx, _ptr(codes)))

    def set_distance_matrix(self, distances, codefreq, eigenval, eigentot):
        a = [self.real(x) for x in (distances, codefreq, eigenval, eigentot)]
        self._chk(self.lib.vft_set_distance_matrix(self.ctx, *[_ptr(x) for x in a]))

    def set_transition_matrix(self, stat=None, statinv=None, eigenval=None, codefreq=None, eigeninv=None,
                              eigeninvT=None):
        if stat is None:
            self._chk(self.lib.vft_set_transition_matrix(self.ctx, None, None, None, None, None, None))
            return
        a = [self.real(x) for x in (stat, statinv, eigenval, codefreq, eigeninv, eigeninvT)]
        self._chk(self.lib.vft_set_transition_matrix(self.ctx, *[_ptr(x) for x in a]))

    def set_rates(self, rates, ratecat):
        rates = self.real(rates)
        ratecat = _i64(ratecat)
        self._chk(self.lib.vft_set_rates(self.ctx, _ptr(rates), I32(len(rates)), _ptr(ratecat)))

    def set_ml_limits(self, min_len, min_rel, fpost_tol):
        self._chk(self.lib.vft_set_ml_limits(self.ctx, C.c_double(min_len), C.c_double(min_rel), C.c_double(fpost_tol)))

    def set_parents(self, first, parent):
        parent = _i64(parent)
        self._chk(self.lib.vft_set_parents(self.ctx, I64(first), I64(len(parent)), _ptr(parent)))

    def set_node_scalars(self, first, diameter=None, selfweight=None, selfdist=None):
        arrs = [None if x is None else self.real(x) for x in (diameter, selfweight, selfdist)]
        n = max(len(x) for x in arrs if x is not None)
        self._chk(self.lib.vft_set_node_scalars(self.ctx, I64(first), I64(n), *[_ptr(x) for x in arrs]))

    def get_node_scalars(self, first, count):
        d, sw, sd = (np.zeros(count, self.dt) for _ in range(3))
        self._chk(self.lib.vft_get_node_scalars(self.ctx, I64(first), I64(count), _ptr(d), _ptr(sw), _ptr(sd)))
        return d, sw, sd

    def set_out_distances(self, first, out_dist, n_out_active):
        od = self.real(out_dist)
        na = _i64(n_out_active)
        self._chk(self.lib.vft_set_out_distances(self.ctx, I64(first), I64(len(od)), _ptr(od), _ptr(na)))

    def join_nodes(self, i, j, newnode, diameter, stale_stamp):
        """The state change of one join in one launch (NJ.tcc:2904-2909, 3003-3007, 254)."""
        self._chk(self.lib.vft_join_nodes(self.ctx, I64(i), I64(j), I64(newnode), C.c_double(diameter), I64(stale_stamp)))

    def join_fused(self, i, j, newnode, diameter, stale_stamp, n_active_old, update_out_profile=True):
        """joinNodes + averageProfile(newnode, i, j) + its self-distance + updateOutProfile in one launch (vft_join_fused)."""
        self._chk(self.lib.vft_join_fused(self.ctx, I64(i), I64(j), I64(newnode), C.c_double(diameter), I64(stale_stamp),
                                          I64(n_active_old), C.c_int32(1 if update_out_profile else 0)))

    def out_distance_mirror(self):
        """numpy views of the host-mapped mirrors of outDistances[] / nOutDistActive[] (vft_out_distance_mirror): what the
        host driver reads without a device call; valid while the context lives."""
        po, pn = C.c_void_p(), C.c_void_p()
        self._chk(self.lib.vft_out_distance_mirror(self.ctx, C.byref(po), C.byref(pn)))
        od = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_float if self.dt == np.float32 else C.c_double)), shape=(self.max_nodes,))
        na = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_int32)), shape=(self.max_nodes,))
        return od, na

    def get_out_distances(self, first, count):
        od = np.zeros(count, self.dt)
        na = np.zeros(count, np.int64)
        self._chk(self.lib.vft_get_out_distances(self.ctx, I64(first), I64(count), _ptr(od), _ptr(na)))
        return od, na

    # ---- profiles
    def profile_upload(self, node, prof):
        w, c, f = self.real(prof[0]), np.ascontiguousarray(prof[1], np.uint8), self.real(prof[2])
        self._chk(self.lib.vft_profile_upload(self.ctx, I64(node), _ptr(w), _ptr(c), _ptr(f)))

    def profile_download(self, node):
        w = np.zeros(self.n_pos, self.dt)
        c = np.zeros(self.n_pos, np.uint8)
        f = np.zeros((self.n_pos, self.n_codes), self.dt)
        self._chk(self.lib.vft_profile_download(self.ctx, I64(node), _ptr(w), _ptr(c), _ptr(f)))
        return w, c, f

    def profile_nvectors(self, first, count):
        out = np.zeros(count, np.int64)
        self._chk(self.lib.vft_profile_nvectors(self.ctx, I64(first), I64(count), _ptr(out)))
        return out

    def averageProfile(self, out, a, b, bionj_weight=None):
        """NJ.tcc:2067 for a batch: out[k] = average(a[k], b[k])."""
        out, a, b = _i64(out), _i64(a), _i64(b)
        bw = None if bionj_weight is None else np.ascontiguousarray(bionj_weight, np.float64)
        self._chk(self.lib.vft_average_profiles(self.ctx, I64(len(out)), _ptr(out), _ptr(a), _ptr(b), _ptr(bw)))

    # ---- out-profile
    def outProfile(self, active_ids):
        """NJ.tcc:729: out-profile of the listed nodes, accumulated in list order."""
        ids = _i64(active_ids)
        self._chk(self.lib.vft_out_profile_full(self.ctx, I64(len(ids)), _ptr(ids)))


    def out_profile_partial(self, n_total, ids):
        """raw sums of one block of the active list (vft_out_profile_partial): (n_pos, 1 + n_codes) array"""
        ids = _i64(ids)
        part = np.zeros((self.n_pos, 1 + self.n_codes), self.dt)
        self._chk(self.lib.vft_out_profile_partial(self.ctx, I64(n_total), I64(len(ids)), _ptr(ids), _ptr(part)))
        return part

    def out_profile_finish(self, parts):
        """the blocks added in order, normalised, installed (vft_out_profile_finish)"""
        parts = np.ascontiguousarray(np.stack(parts), self.dt)
        self._chk(self.lib.vft_out_profile_finish(self.ctx, I32(len(parts)), _ptr(parts)))

    def updateOutProfile(self, old1, old2, new, n_active_old):
        """NJ.tcc:943."""
        self._chk(self.lib.vft_out_profile_update(self.ctx, I64(old1), I64(old2), I64(new), I64(n_active_old)))

    def out_profile_upload(self, w, f, cd=None):
        w, f = self.real(w), self.real(f)
        cd = None if cd is None else self.real(cd)
        self._chk(self.lib.vft_out_profile_upload(self.ctx, _ptr(w), _ptr(f), _ptr(cd)))

    def out_profile_download(self, with_codedist=False):
        w = np.zeros(self.n_pos, self.dt)
        f = np.zeros((self.n_pos, self.n_codes), self.dt)
        cd = np.zeros((self.n_pos, self.n_codes), self.dt) if with_codedist else None
        self._chk(self.lib.vft_out_profile_download(self.ctx, _ptr(w), _ptr(f), _ptr(cd)))
        return (w, np.full(self.n_pos, NOCODE, np.uint8), f), cd

    def setOutDistance(self, ids, n_active, totdiam):
        """NJ.tcc:1012 for a list of nodes (None: every active node)."""
        ids = None if ids is None else _i64(ids)
        n = 0 if ids is None else len(ids)
        self._chk(self.lib.vft_out_distances(self.ctx, I64(n), _ptr(ids), I64(n_active), C.c_double(totdiam)))

    # ---- distances
    def setBestHit(self, query, n_active, n_diff_allow, totdiam, k, want_best=True, d_hits=None, want_hits=True):
        """NJ.tcc:3571 + the reference's sort, truncated to k hits.  Returns (hits, best_j)."""
        hits = np.zeros(k, self.hit_dtype) if want_hits else None
        best = I64(-1)
        self._chk(self.lib.vft_sweep(self.ctx, I64(query), I64(n_active), I64(n_diff_allow), C.c_double(totdiam),
                                     I32(k), _ptr(hits), P(d_hits) if d_hits else None,
                                     C.byref(best) if want_best else None))
        return hits, best.value

    def setBestHitBatch(self, queries, n_active, n_diff_allow, totdiam, k, d_hits=None, want_hits=True, view=False):
        """Several seeds per call (vft_sweep_batch): returns (hits[n_seeds, k], best_j[n_seeds]).
        view: the hit records are read where the device left them (host-mapped result blocks, vft_sweep_batch_view) - a list of
        n_seeds arrays that are valid until this context's next sweep."""
        q = _i64(queries)
        hits = np.zeros((len(q), k), self.hit_dtype) if want_hits and not view else None
        best = np.full(len(q), -1, np.int64)
        self._chk(self.lib.vft_sweep_batch(self.ctx, I32(len(q)), _ptr(q), I64(n_active), I64(n_diff_allow),
                                           C.c_double(totdiam), I32(k), _ptr(hits), P(d_hits) if d_hits else None,
                                           _ptr(best) if k >= 2 else None))
        if view and want_hits:
            nb = k * self.hit_dtype.itemsize
            hits = []
            for s in range(len(q)):
                p = P()
                self._chk(self.lib.vft_sweep_batch_view(self.ctx, I32(s), C.byref(p), None))
                hits.append(np.frombuffer((C.c_char * nb).from_address(p.value), dtype=self.hit_dtype))
        return hits, best

    def device_buffer(self, array):
        """Copy a host array into a fresh device buffer; returns the device address (free with device_free)."""
        a = np.ascontiguousarray(array)
        p = P()
        self._chk(self.lib.vft_device_malloc(self.ctx, I64(a.nbytes), C.byref(p)))
        self._chk(self.lib.vft_device_upload(self.ctx, p, _ptr(a), I64(a.nbytes)))
        return p.value

    def device_free(self, addr):
        self._chk(self.lib.vft_device_free(self.ctx, P(addr)))

    def merge_hits(self, d_all, n_lists, k, d_out=None):
        """Merge all-gathered per-shard hit lists (device pointer) into the global top-k, on the device.
        d_out (device address): leave the result there and do not wait (stream-ordered); otherwise return it."""
        if d_out is not None:
            self._chk(self.lib.vft_merge_hits(self.ctx, P(d_all), I32(n_lists), I32(k), None, P(d_out)))
            return None
        hits = np.zeros(k, self.hit_dtype)
        self._chk(self.lib.vft_merge_hits(self.ctx, P(d_all), I32(n_lists), I32(k), _ptr(hits), None))
        return hits

    def merge_hits_batch(self, d_all, n_lists, n_seeds, k, d_out=None, want_hits=True):
        """Merge all-gathered batches ([n_lists][n_seeds][k], device pointer): returns hits[n_seeds, k]."""
        hits = np.zeros((n_seeds, k), self.hit_dtype) if want_hits else None
        self._chk(self.lib.vft_merge_hits_batch(self.ctx, P(d_all), I32(n_lists), I32(n_seeds), I32(k), _ptr(hits),
                                                P(d_out) if d_out else None))
        return hits

    def sweep_info(self):
        info = (I64 * 2)()
        self._chk(self.lib.vft_sweep_info(self.ctx, info))
        return int(info[0]), int(info[1])

    def sweep_batch_info(self, slot):
        info = (I64 * 2)()
        self._chk(self.lib.vft_sweep_batch_info(self.ctx, I32(slot), info))
        return int(info[0]), int(info[1])

    def sweep_results(self, first, count):
        d, w, c = (np.zeros(count, self.dt) for _ in range(3))
        self._chk(self.lib.vft_sweep_results(self.ctx, I64(first), I64(count), _ptr(d), _ptr(w), _ptr(c)))
        return d, w, c

    def setDistCriterion(self, i, j, n_active, n_diff_allow, totdiam):
        """NJ.tcc:1115 over a pair list.  Returns (dist, weight, criterion)."""
        i, j = _i64(i), _i64(j)
        n = len(i)
        d, w, c = (np.zeros(n, self.dt) for _ in range(3))
        self._chk(self.lib.vft_pair_distances(self.ctx, I64(n), _ptr(i), _ptr(j), I64(n_active), I64(n_diff_allow),
                                              C.c_double(totdiam), _ptr(d), _ptr(w), _ptr(c)))
        return d, w, c

    def setDistCriterionRefresh(self, i, j, force_ids, n_active, n_diff_allow, totdiam):
        """vft_pair_distances_refresh: the pair list plus nodes whose out-distance is recomputed first unless it carries
        the stamp n_active (setOutDistance).  (dist, weight, criterion)."""
        i, j, f = _i64(i), _i64(j), _i64(force_ids)
        d, w, c = (np.zeros(len(i), self.dt) for _ in range(3))
        self._chk(self.lib.vft_pair_distances_refresh(self.ctx, I64(len(i)), _ptr(i), _ptr(j), I64(len(f)), _ptr(f), I64(n_active),
                                                      I64(n_diff_allow), C.c_double(totdiam), _ptr(d), _ptr(w), _ptr(c)))
        return d, w, c

    def leafBlockDistances(self, a, b, n_active, n_diff_allow, totdiam):
        """setDistCriterion for the cross product of two leaf lists (vft_leaf_block_distances): three [len(a), len(b)]
        arrays (dist, weight, criterion)."""
        a, b = _i64(a), _i64(b)
        d, w, c = (np.zeros((len(a), len(b)), self.dt) for _ in range(3))
        self._chk(self.lib.vft_leaf_block_distances(self.ctx, I64(len(a)), _ptr(a), I64(len(b)), _ptr(b), I64(n_active),
                                                    I64(n_diff_allow), C.c_double(totdiam), _ptr(d), _ptr(w), _ptr(c)))
        return d, w, c

    def blockDistances(self, a, b, n_active, n_diff_allow, totdiam):
        """Join distances of the cross product of two node lists (vft_block_distances), lazy out-distance refresh of
        every listed node included: a [len(a), len(b)] array; slots of negative ids / i == j are unspecified."""
        a, b = _i64(a), _i64(b)
        d = np.full((len(a), len(b)), np.nan, self.dt)
        self._chk(self.lib.vft_block_distances(self.ctx, I64(len(a)), _ptr(a), I64(len(b)), _ptr(b), I64(n_active),
                                               I64(n_diff_allow), C.c_double(totdiam), _ptr(d)))
        return d

    def profileDist(self, i, j):
        """NJ.tcc:1167 / 1601 over a pair list: raw (dist, weight), no diameter correction, no criterion."""
        i, j = _i64(i), _i64(j)
        d, w = np.zeros(len(i), self.dt), np.zeros(len(i), self.dt)
        self._chk(self.lib.vft_profile_distances(self.ctx, I64(len(i)), _ptr(i), _ptr(j), _ptr(d), _ptr(w)))
        return d, w

    # ---- likelihood
    # ---- ML branch lengths (device-resident branchlength[], dense ML rows)
    def branch_lengths_set(self, first, values):
        v = self.real(values)
        self._chk(self.lib.vft_branch_lengths_set(self.ctx, I64(first), I64(len(v)), _ptr(v)))

    def branch_lengths_get(self, first, count):
        v = np.zeros(count, self.dt)
        self._chk(self.lib.vft_branch_lengths_get(self.ctx, I64(first), I64(count), _ptr(v)))
        return v

    def posteriorProfileBlen(self, out, a, b, len_idx_a, len_idx_b):
        """posteriorProfile with the lengths read from the device branchlength[]; results go to dense ML rows.
        Stream-ordered (does not wait)."""
        out, a, b, la, lb = _i64(out), _i64(a), _i64(b), _i64(len_idx_a), _i64(len_idx_b)
        self._chk(self.lib.vft_posterior_profiles_blen(self.ctx, I64(len(out)), _ptr(out), _ptr(a), _ptr(b), _ptr(la), _ptr(lb)))

    def mlOptimizeSplits(self, ids, len_idx, recompute, ftol=0.001, atol=None):
        """vft_ml_optimize_splits: ids / len_idx are [n, 3]."""
        ids, li, rec = _i64(ids).reshape(-1, 3), _i64(len_idx).reshape(-1, 3), _i64(recompute)
        if atol is None:
            atol = 1.0e-4 if self.dt == np.float32 else 1.0e-9
        self._chk(self.lib.vft_ml_optimize_splits(self.ctx, I64(len(rec)), _ptr(ids), _ptr(li), _ptr(rec),
                                                  C.c_double(ftol), C.c_double(atol)))

    def mlSplitTests(self, ids, len_idx, always_second_pass=False, ftol=0.001, atol=None):
        """vft_ml_split_tests without resampling: ids [n, 4], len_idx [n, 5] -> (loglk [n, 3], lengths [n, 2, 5])."""
        ids, li = _i64(ids).reshape(-1, 4), _i64(len_idx).reshape(-1, 5)
        if atol is None:
            atol = 1.0e-4 if self.dt == np.float32 else 1.0e-9
        n = len(ids)
        loglk, lengths = np.zeros((n, 3)), np.zeros((n, 2, 5))
        self._chk(self.lib.vft_ml_split_tests(self.ctx, I64(n), _ptr(ids), _ptr(li), C.c_double(ftol), C.c_double(atol),
                                              C.c_double(5.0), I32(1 if always_second_pass else 0), _ptr(loglk), I32(0), None,
                                              None, _ptr(lengths)))
        return loglk, lengths

    def ml_eval_count(self):
        n = I64(0)
        self._chk(self.lib.vft_ml_eval_count(self.ctx, C.byref(n)))
        return n.value

    def pairLogLk(self, a, b, length, site_lk=False):
        """NJ.tcc:1192 for a batch of pairs.  Returns loglk[n] (and site likelihoods [n, n_pos])."""
        a, b = _i64(a), _i64(b)
        length = np.ascontiguousarray(length, np.float64)
        out = np.zeros(len(a), np.float64)
        site = np.zeros((len(a), self.n_pos), np.float64) if site_lk else None
        self._chk(self.lib.vft_pair_loglk(self.ctx, I64(len(a)), _ptr(a), _ptr(b), _ptr(length), _ptr(out), _ptr(site)))
        return (out, site) if site_lk else out

    def set_profile_rows(self, on=True):
        """vft_set_profile_rows: every internal profile as a dense row (what the ML stage works on); posteriors then write rows"""
        self._chk(self.lib.vft_set_profile_rows(self.ctx, I32(1 if on else 0)))

    def posteriorProfile(self, out, a, b, len1, len2):
        """NJ.tcc:2137 for a batch of triples."""
        out, a, b = _i64(out), _i64(a), _i64(b)
        l1 = np.ascontiguousarray(len1, np.float64)
        l2 = np.ascontiguousarray(len2, np.float64)
        self._chk(self.lib.vft_posterior_profiles(self.ctx, I64(len(out)), _ptr(out), _ptr(a), _ptr(b), _ptr(l1),
                                                  _ptr(l2)))

    # ---- top-hit lists on the device (include/vft_hip.h, vft_tophits_*)
    @property
    def tophit_dtype(self):
        return np.dtype([("j", np.int32), ("dist", np.float32)]) if self.dt == np.float32 else \
            np.dtype([("j", np.int32), ("pad", np.int32), ("dist", np.float64)])

    def tophits_create(self, m, n_lists=None):
        self.th_m = int(m)
        self._chk(self.lib.vft_tophits_create(self.ctx, C.c_int32(m), I64(self.max_nodes if n_lists is None else n_lists)))

    def tophits_upload(self, nodes, lists):
        """lists: one (j[], dist[]) pair per node"""
        nodes = _i64(nodes)
        packed = np.zeros((len(nodes), self.th_m), self.tophit_dtype)
        lens = np.zeros(len(nodes), np.int32)
        for t, (j, d) in enumerate(lists):
            lens[t] = len(j)
            packed["j"][t, :len(j)] = j
            packed["dist"][t, :len(j)] = d
        self._chk(self.lib.vft_tophits_upload(self.ctx, I64(len(nodes)), _ptr(nodes), _ptr(lens), _ptr(packed)))

    def tophits_download(self, node):
        buf = np.zeros(self.th_m, self.tophit_dtype)
        n = C.c_int32(0)
        self._chk(self.lib.vft_tophits_download(self.ctx, I64(node), C.byref(n), _ptr(buf)))
        return buf["j"][:n.value].copy(), buf["dist"][:n.value].copy()

    def tophits_best(self, node, length, n_active, n_diff_allow, totdiam, force_node=True):
        out = np.zeros(1, np.dtype([("j", np.int32), ("pos", np.int32), ("dist", np.float64), ("criterion", np.float64)]))
        self._chk(self.lib.vft_tophits_best(self.ctx, I64(node), C.c_int32(length), I64(n_active), I64(n_diff_allow), C.c_double(totdiam),
                                            C.c_int32(1 if force_node else 0), _ptr(out)))
        return int(out["j"][0]), int(out["pos"][0]), self.dt.type(out["dist"][0]), self.dt.type(out["criterion"][0])

    def tophits_join(self, newnode, c0, n0, c1, n1, n_active, n_diff_allow, totdiam, n_save_max, need, age_ok):
        info = np.zeros(4, np.int32)
        j = np.zeros(n0 + n1 + 1, np.int32)
        d = np.zeros(n0 + n1 + 1, self.dt)
        cr = np.zeros(n0 + n1 + 1, self.dt)
        self._chk(self.lib.vft_tophits_join(self.ctx, I64(newnode), I64(c0), C.c_int32(n0), I64(c1), C.c_int32(n1), I64(n_active),
                                            I64(n_diff_allow), C.c_double(totdiam), C.c_int32(n_save_max), C.c_int32(need),
                                            C.c_int32(1 if age_ok else 0), _ptr(info), _ptr(j), _ptr(d), _ptr(cr)))
        nu = int(info[0])
        return dict(n_unique=nu, use_unique=bool(info[1]), n_save=int(info[2]), j=j[:nu], dist=d[:nu], criterion=cr[:nu])

    def debug_option(self, option, value):
        """test hook (include/vft_hip.h VFT_DEBUG_*): 1 no fused refresh, 2 threads per pair, 3 no pair staging, 4 generic out-profile"""
        self._chk(self.lib.vft_debug_option(self.ctx, C.c_int32(option), I64(value)))

    def debug_log(self, x):
        """log(x) as the ML kernels evaluate it on the device (glibc's algorithm): diagnostics."""
        x = np.ascontiguousarray(x, np.float64)
        out = np.zeros_like(x)
        self._chk(self.lib.vft_debug_log(self.ctx, I64(x.size), _ptr(x), _ptr(out)))
        return out

    # ---- measurement
    def timer_start(self):
        self._chk(self.lib.vft_timer_start(self.ctx))

    def timer_stop_ms(self):
        ms = C.c_float(0)
        self._chk(self.lib.vft_timer_stop_ms(self.ctx, C.byref(ms)))
        return ms.value

    def sweep_kernel_ms(self):
        ms, n = C.c_float(0), I64(0)
        self._chk(self.lib.vft_sweep_kernel_ms(self.ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def sweep_kernel_sweeps(self):
        n = I64(0)
        self._chk(self.lib.vft_sweep_kernel_sweeps(self.ctx, C.byref(n)))
        return n.value

    def sweep_table_kernel_ms(self):
        ms, n = C.c_float(0), I64(0)
        self._chk(self.lib.vft_sweep_table_kernel_ms(self.ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value
