"""GPU (-m gpu): runtime behaviour of the library around the kernels -- stream capture into a graph, concurrent callers on
their own streams.  (The reference op launches on the current stream without synchronising and keeps no state,
SURVEY.md section 8b; this library keeps a little -- the locality monitor, LDS-limit bookkeeping -- and must stay usable
the same way.)"""
import threading

import numpy as np
import pytest
import torch

from richsem_amd import _lib, workload as W
from richsem_amd.capture import quiet_gc
from richsem_amd import MultiScaleDeformableAttention as MSDA

pytestmark = pytest.mark.gpu


def _run(t):
    out = MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
    grads = MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
    return out, grads


def _close(a, b, tol):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30)) < tol


@pytest.mark.parametrize("which", ["E", "Dd"])   # window kernels + locality monitor / level-sum kernel + direct kernels
def test_calls_can_be_captured_into_a_graph(which):
    call = W.shrunk({"E": W.call_E, "Dd": W.call_Dd}[which](2), 2)
    t = W.make_inputs(call, "init", seed=3, device="cuda")
    ref_out, ref_g = _run(t)            # also warms up: host mirrors cached, LDS limits granted, first monitor probes
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            _run(t)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with quiet_gc(), torch.cuda.graph(graph):
        out, grads = _run(t)
    out.zero_()
    for x in grads:
        x.fill_(float("nan"))
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)
    assert _close(grads[0], ref_g[0], 2e-4)
    assert _close(grads[1], ref_g[1], 2e-4) and _close(grads[2], ref_g[2], 2e-4)


def test_concurrent_callers_on_their_own_streams():
    call = W.shrunk(W.call_E(2), 2)
    t = W.make_inputs(call, "init", seed=4, device="cuda")
    ref_out, ref_g = _run(t)
    torch.cuda.synchronize()
    results, errors = [], []

    def worker():
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(40):
                    out, grads = _run(t)
                st.synchronize()
            results.append((torch.equal(out, ref_out), _close(grads[0], ref_g[0], 2e-4), _close(grads[1], ref_g[1], 2e-4)))
        except Exception as e:   # noqa: BLE001 -- reported below, a thread must not swallow it
            errors.append(repr(e))

    threads = [threading.Thread(target=worker) for _ in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert results == [(True, True, True)] * 3


def test_routed_backward_replays_from_a_graph():
    """The routed backward keeps device state between calls (per-bin (records, runs) counters that the tile kernel zeroes when it has
    consumed a bin, run tables and a record pool per stream): captured on a stream that has run it before, the graph must reproduce the
    eager result on every replay, also with a call of another geometry in between."""
    call = W.shrunk(W.call_E(2), 2)
    t = W.make_inputs(call, "uniform", seed=11, device="cuda")
    other = W.make_inputs(W.shrunk(W.call_E(2), 4), "sigma4", seed=12, device="cuda")
    _lib.set_option("bwd_variant", 4)
    try:
        ref = MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):       # the workspace of this stream is allocated here, outside the capture
            for _ in range(2):
                MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with quiet_gc(), torch.cuda.graph(graph, stream=side):
            grads = MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
        for rep in range(3):
            for x in grads:
                x.fill_(float("nan"))
            graph.replay()
            torch.cuda.synchronize()
            for a, b in zip(grads, ref):
                assert torch.isfinite(a).all() and _close(a, b, 2e-4), rep
            with torch.cuda.stream(side):   # another geometry on the same stream between replays
                MSDA.ms_deform_attn_backward(other["value"], other["shapes"], other["lsi"], other["loc"], other["aw"],
                                             other["grad_out"], 64)
            torch.cuda.synchronize()
    finally:
        _lib.set_option("bwd_variant", 0)


def test_routed_backward_state_survives_changing_shapes():
    """calls of three different geometries (different bin counts, run-table strides and pool stretches) in turn, twice: every one must
    match the oracle -- a counter a call left non-zero, or a run table read at the wrong stride, would show in the next call"""
    from oracle import msda_oracle as O
    _lib.set_option("bwd_variant", 4)
    try:
        calls = [W.shrunk(W.call_E(2), 4), W.shrunk(W.call_E(1), 2), W.shrunk(W.call_E(2), 3)]
        sets = [W.make_inputs(c, mode, seed=20 + i, device="cuda") for i, (c, mode) in enumerate(zip(calls, ("uniform", "init", "sigma4")))]
        refs = []
        for t in sets:
            z = {k: v.cpu().numpy() for k, v in t.items() if k in ("value", "shapes", "lsi", "loc", "aw", "grad_out")}
            refs.append(O.backward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"], z["grad_out"]))
        for rep in range(2):
            for t, ref in zip(sets, refs):
                got = MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
                for a, b in zip(got, ref):
                    assert _close(a, torch.from_numpy(b).cuda(), 2e-4), rep
    finally:
        _lib.set_option("bwd_variant", 0)


def test_profile_filter_brackets_only_the_calls_it_names():
    """Library option ``profile_filter`` (bench.py: only the dominant kernel is bracketed with events inside the timed region): 0 = every call,
    (kind + 1) * 16 + variant = only those; results do not depend on it."""
    enc, dec = W.shrunk(W.call_E(2), 4), W.shrunk(W.call_Dd(2), 4)
    te, td = (W.make_inputs(c, "init", seed=9, device="cuda") for c in (enc, dec))
    ref = _run(te)
    torch.cuda.synchronize()

    def records(flt):
        _lib.set_option("profile_filter", flt)
        _lib.profile_enable(16)
        out = _run(te)
        _run(td)
        torch.cuda.synchronize()
        recs = [(r["kind"], r["variant"]) for r in _lib.profile_collect()]
        _lib.profile_enable(0)
        return out, recs
    try:
        _lib.set_option("locality_monitor", 0)
        _lib.set_option("fwd_variant", 2)
        out, everything = records(0)
        assert sorted(everything) == sorted([("fwd", 2), ("bwd", 4), ("fwd", 3), ("bwd", 1)]), everything
        out, only = records((1 + 1) * 16 + 4)
        assert only == [("bwd", 4)], only
        assert torch.equal(out[0], ref[0])
        out, only = records((0 + 1) * 16 + 3)
        assert only == [("fwd", 3)], only
    finally:
        _lib.set_option("profile_filter", 0)
        _lib.set_option("fwd_variant", 0)
        _lib.set_option("locality_monitor", 1)
