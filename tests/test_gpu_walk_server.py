"""The walk server (csrc/vft_kernels_walk.h, vft_walk_server_start / _stop, vft_walk_submit / _collect): the steps of a refinement walk taken
by six resident workgroups from a mailbox instead of one launch each.  Every test drives one context through the server and an identical
second context through the plain calls the server replaces (vft_average_chain + vft_profile_distances) and asks for the same bits: the
six distances of every step and every profile row the steps wrote."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_chains import make_state, ptr, same

pytestmark = pytest.mark.gpu

I64, I32, U32, P = C.c_int64, C.c_int32, C.c_uint32, C.c_void_p
OPT_NO_SERVER, OPT_DEVICE_MAIL, OPT_STRIDE = 9, 10, 11


def plain_step(ops, dt, out, a, b, q):
    if len(out):
        assert ops.lib.vft_average_chain(ops.ctx, I32(len(out)), ptr(out), ptr(a), ptr(b)) == 0
    if q is None:
        return None
    pi = np.array([q[0], q[0], q[0], q[1], q[1], q[2]], np.int64)
    pj = np.array([q[1], q[2], q[3], q[2], q[3], q[3]], np.int64)
    d, _ = ops.profileDist(pi, pj)
    return np.asarray(d, dt)


def random_steps(rng, free, n_steps, n_leaves=48, long_every=0):
    """steps as the walks produce them: a few averages, most reading the one before, some an older output, some rewriting a node
    the step read or wrote earlier; the quartet takes two of the step's outputs, a leaf and any node"""
    scratch = list(range(free, free + 24))
    steps = []
    for t in range(n_steps):
        n = int(rng.integers(0, 9))
        if long_every and t % long_every == long_every - 1:
            n = int(rng.integers(20, 60))   # more averages than one command holds
        out, a, b = [], [], []
        for k in range(n):
            o = scratch[int(rng.integers(0, len(scratch)))] if rng.random() < 0.8 else int(rng.integers(n_leaves, free))
            x = out[-1] if out and rng.random() < 0.6 else int(rng.integers(0, free + 24))
            y = out[int(rng.integers(0, len(out)))] if out and rng.random() < 0.3 else int(rng.integers(0, free + 24))
            out.append(o)
            a.append(x)
            b.append(y)
        q = [int(rng.integers(0, n_leaves)), int(rng.integers(n_leaves, free)), int(rng.integers(0, free + 24)), int(rng.integers(0, free + 24))]
        if n >= 2:
            q[2], q[3] = out[-1], out[0]
        if t % 7 == 3:
            q[1] = int(rng.integers(0, n_leaves))   # a leaf x leaf pair
        if q[2] == q[3]:
            q[3] = q[0]
        steps.append((np.array(out, np.int64), np.array(a, np.int64), np.array(b, np.int64), np.array(q, np.int64)))
    return steps


def init_scratch(ops_list, free):
    """the scratch rows the steps read before writing must hold the same profile in both contexts"""
    out = np.arange(free, free + 24, dtype=np.int64)
    a = np.arange(24, dtype=np.int64)
    b = free - 1 - a
    for ops in ops_list:
        assert ops.lib.vft_average_chain(ops.ctx, I32(24), ptr(out), ptr(a), ptr(b)) == 0


@pytest.mark.parametrize("stride", [8, 1])
@pytest.mark.parametrize("dt,nc,L", [(np.float32, 4, 137), (np.float32, 4, 1000), (np.float64, 4, 200), (np.float32, 20, 90), (np.float64, 20, 300),
                                      (np.float32, 4, 3500), (np.float64, 20, 1700)])   # (the last two: several slices per wavefront, several trips of the pair loop, more than 48 KB of staging)
def test_server_steps_equal_plain_calls(dt, nc, L, stride):
    s_ops, rng, free = make_state(dt, nc, L=L, seed=21)
    p_ops, _, _ = make_state(dt, nc, L=L, seed=21)
    init_scratch([s_ops, p_ops], free)
    steps = random_steps(rng, free, 60, long_every=11)
    assert s_ops.lib.vft_debug_option(s_ops.ctx, I32(OPT_STRIDE), I64(stride)) == 0
    assert s_ops.lib.vft_walk_server_start(s_ops.ctx) == 0, s_ops.lib.vft_last_error(s_ops.ctx)
    for t, (out, a, b, q) in enumerate(steps):
        d1 = np.zeros(6, dt)
        assert s_ops.lib.vft_walk_step(s_ops.ctx, I32(len(out)), ptr(out), ptr(a), ptr(b), ptr(q), ptr(d1)) == 0, s_ops.lib.vft_last_error(s_ops.ctx)
        d2 = plain_step(p_ops, dt, out, a, b, q)
        assert np.array_equal(d1.view(np.uint8), d2.view(np.uint8)), (t, d1, d2)
    assert s_ops.lib.vft_walk_server_stop(s_ops.ctx) == 0, s_ops.lib.vft_last_error(s_ops.ctx)
    for x in range(48, free + 24):
        assert same(s_ops.profile_download(x), p_ops.profile_download(x)), x
    s_ops.close()
    p_ops.close()


@pytest.mark.parametrize("device_mail", [0, 1])
def test_two_steps_in_flight_and_averages_alone(device_mail):
    """vft_walk_submit / vft_walk_collect: the forced first step of an SPR chain does not gate the second (two tickets in flight);
    averages without distances (flushAverages); a restart of the server on the same context; the mailbox in device memory"""
    dt, nc = np.float32, 4
    s_ops, rng, free = make_state(dt, nc, L=200, seed=33)
    p_ops, _, _ = make_state(dt, nc, L=200, seed=33)
    init_scratch([s_ops, p_ops], free)
    assert s_ops.lib.vft_debug_option(s_ops.ctx, I32(OPT_DEVICE_MAIL), I64(device_mail)) == 0
    for round_ in range(2):
        steps = random_steps(rng, free, 40)
        assert s_ops.lib.vft_walk_server_start(s_ops.ctx) == 0, s_ops.lib.vft_last_error(s_ops.ctx)
        t = 0
        while t + 2 < len(steps):
            tickets = []
            for out, a, b, q in steps[t:t + 2]:
                tk = U32(0)
                assert s_ops.lib.vft_walk_submit(s_ops.ctx, I32(len(out)), ptr(out), ptr(a), ptr(b), ptr(q), C.byref(tk)) == 0, s_ops.lib.vft_last_error(s_ops.ctx)
                tickets.append(tk)
            out, a, b, _ = steps[t + 2]   # averages alone behind them
            tk = U32(0)
            assert s_ops.lib.vft_walk_submit(s_ops.ctx, I32(len(out)), ptr(out), ptr(a), ptr(b), None, C.byref(tk)) == 0
            for k in range(2):
                d1 = np.zeros(6, dt)
                assert s_ops.lib.vft_walk_collect(s_ops.ctx, tickets[k], ptr(d1)) == 0, s_ops.lib.vft_last_error(s_ops.ctx)
                o2, a2, b2, q2 = steps[t + k]
                d2 = plain_step(p_ops, dt, o2, a2, b2, q2)
                assert np.array_equal(d1.view(np.uint8), d2.view(np.uint8)), (round_, t + k, d1, d2)
            assert s_ops.lib.vft_walk_collect(s_ops.ctx, tk, None) == 0
            plain_step(p_ops, dt, out, a, b, None)
            t += 3
        if round_ == 0:
            assert s_ops.lib.vft_walk_server_stop(s_ops.ctx) == 0, s_ops.lib.vft_last_error(s_ops.ctx)
        # round 1: no explicit stop - the download below launches on the context's stream, which retires the server first
        for x in range(48, free + 24):
            assert same(s_ops.profile_download(x), p_ops.profile_download(x)), (round_, x)
    s_ops.close()
    p_ops.close()


def test_server_refuses_without_rows_and_when_switched_off():
    from veryfasttree_amd import HipProfileOps, synth
    codes = synth.random_descent_codes(20, 60, 4, 0.1, 0.02, seed=3)
    ops = HipProfileOps(20, 60, 4, np.float32, max_nodes=80)
    ops.upload_leaves(codes)
    assert ops.lib.vft_walk_server_start(ops.ctx) == 3   # VFT_ERR_STATE: profiles are not rows yet
    assert ops.lib.vft_set_profile_rows(ops.ctx, I32(1)) == 0
    assert ops.lib.vft_debug_option(ops.ctx, I32(OPT_NO_SERVER), I64(1)) == 0
    assert ops.lib.vft_walk_server_start(ops.ctx) == 3
    tk = U32(0)
    out = np.array([25], np.int64)
    assert ops.lib.vft_walk_submit(ops.ctx, I32(1), ptr(out), ptr(out), ptr(out), None, C.byref(tk)) == 3
    ops.close()
