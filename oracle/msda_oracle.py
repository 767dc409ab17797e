"""ctypes front-end of the CPU oracle (oracle/msda_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this
module; nothing under richsem_amd/ does.  See the header of msda_oracle.c for the reference
lines the restatement follows and for how it is pinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (a few seconds).  Idempotent."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "clean", "all"])
    return _SO


def build_native():
    """bench.py's cpu_baseline leg only: rebuild the oracle ON THIS BOX with -march=native (the committed recipe's
    portable -mavx2 build travels with the snapshot; the host CPU of the GPU box differs from the build container's)
    and switch to it.  Returns False -- and keeps the portable build -- when there is no compiler here."""
    global _lib
    src = os.path.join(_HERE, "msda_oracle.c")
    so = os.path.join(_HERE, "libmsda_oracle_native.so")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-fopenmp", "-std=c11",
                               "-shared", "-o", so, src, "-lm"], stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return False
    _lib = None
    lib(so)
    return True


def lib(path=None):
    global _lib
    if _lib is None:
        if path is None:
            build()
        L = ctypes.CDLL(path or _SO)
        vp, ip = ctypes.c_void_p, ctypes.c_int
        for sfx in ("f32", "f64"):
            f = getattr(L, "msda_oracle_forward_" + sfx)
            f.argtypes = [vp, vp, vp, vp, vp] + [ip] * 7 + [vp]
            f.restype = None
            g = getattr(L, "msda_oracle_backward_" + sfx)
            g.argtypes = [vp, vp, vp, vp, vp, vp] + [ip] * 7 + [vp, vp, vp]
            g.restype = None
        L.msda_oracle_set_threads.argtypes = [ip]
        L.msda_oracle_max_threads.restype = ip
        _lib = L
    return _lib


def set_threads(n):
    lib().msda_oracle_set_threads(int(n))


def max_threads():
    return int(lib().msda_oracle_max_threads())


def _np(a, dtype):
    if hasattr(a, "detach"):  # torch tensor
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def _dims(value, shapes, loc):
    N, S, M, D = value.shape
    L = shapes.shape[0]
    Lq, P = loc.shape[1], loc.shape[4]
    assert loc.shape == (N, Lq, M, L, P, 2), loc.shape
    return N, S, M, D, L, Lq, P


def _sfx(dtype):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]


def forward(value, shapes, lsi, loc, aw, dtype=None):
    """out (N, Lq, M*D) as numpy; same argument meaning as ms_deform_attn_forward."""
    dtype = np.dtype(dtype or _np(value, None).dtype)
    value, loc, aw = _np(value, dtype), _np(loc, dtype), _np(aw, dtype)
    shapes, lsi = _np(shapes, np.int64), _np(lsi, np.int64)
    N, S, M, D, L, Lq, P = _dims(value, shapes, loc)
    assert aw.shape == (N, Lq, M, L, P) and int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    out = np.empty((N, Lq, M * D), dtype=dtype)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    getattr(lib(), "msda_oracle_forward_" + _sfx(dtype))(
        p(value), p(shapes), p(lsi), p(loc), p(aw), N, S, M, D, L, Lq, P, p(out))
    return out


def backward(value, shapes, lsi, loc, aw, grad_out, dtype=None):
    """(grad_value, grad_loc, grad_aw) as numpy; meaning as ms_deform_attn_backward."""
    dtype = np.dtype(dtype or _np(value, None).dtype)
    value, loc, aw, grad_out = (_np(a, dtype) for a in (value, loc, aw, grad_out))
    shapes, lsi = _np(shapes, np.int64), _np(lsi, np.int64)
    N, S, M, D, L, Lq, P = _dims(value, shapes, loc)
    assert grad_out.size == N * Lq * M * D
    gv, gl, ga = np.empty_like(value), np.empty_like(loc), np.empty_like(aw)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    getattr(lib(), "msda_oracle_backward_" + _sfx(dtype))(
        p(value), p(shapes), p(lsi), p(loc), p(aw), p(grad_out), N, S, M, D, L, Lq, P,
        p(gv), p(gl), p(ga))
    return gv, gl, ga
