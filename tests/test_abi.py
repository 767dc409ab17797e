"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/richsem_msda.h declares,
and rejects bad arguments on the host (no kernel is launched in this file)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from richsem_amd import _build, _lib

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "richsem_msda.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msda_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    _build.build()
    return _lib.load()


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.lib_path()], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    for sym in declared_symbols():
        assert sym in exported, sym
        assert getattr(lib, sym) is not None


def test_library_contains_gfx950_code_object():
    blob = open(_lib.lib_path(), "rb").read()
    assert b"gfx950" in blob and b"fwd_direct_kernel" in blob and b"bwd_direct_kernel" in blob


def test_abi_version(lib):
    assert lib.msda_abi_version() == _lib.ABI_VERSION
    m = re.search(r"#define RICHSEM_MSDA_ABI_VERSION (\d+)", open(HEADER).read())
    assert int(m.group(1)) == _lib.ABI_VERSION


def test_options_roundtrip(lib):
    for key in ("fwd_variant", "bwd_variant", "bwd_split", "profile_filter"):
        old = _lib.get_option(key)
        _lib.set_option(key, 1)
        assert _lib.get_option(key) == 1
        _lib.set_option(key, old)
    with pytest.raises(RuntimeError, match="MSDA_ERR_BAD_OPTION"):
        _lib.set_option("profile_filter", 48)
    with pytest.raises(RuntimeError, match="MSDA_ERR_BAD_OPTION"):
        _lib.set_option("fwd_variant", 4)
    assert _lib.get_option("locality_monitor") == 1          # on by default
    _lib.set_option("locality_monitor", 0)
    assert _lib.get_option("locality_monitor") == 0
    _lib.set_option("locality_monitor", 1)
    assert _lib.get_option("locality_share_ppm") == -1       # nothing measured (and nothing can be, without a GPU)
    with pytest.raises(RuntimeError, match="MSDA_ERR_BAD_OPTION"):
        _lib.set_option("locality_share_ppm", 5)             # read-only
    with pytest.raises(RuntimeError, match="MSDA_ERR_BAD_OPTION"):
        _lib.set_option("no_such_option", 1)
    with pytest.raises(RuntimeError, match="MSDA_ERR_BAD_OPTION"):
        _lib.set_option("fwd_variant", 99)


def _call_forward(lib, dims, shapes, lsi, im2col_step=64, null=None, ptr=0x1000):
    """Forward entry with fake (never dereferenced) device pointers: every check below fails on the host,
    before anything touches a device."""
    N, S, M, D, L, Lq, P = dims
    sh = np.asarray(shapes, dtype=np.int64)
    ls = np.asarray(lsi, dtype=np.int64)
    p = [ptr] * 5
    if null is not None:
        p[null] = None
    return lib.msda_forward_f32(p[0], p[1], p[2], p[3], p[4], N, S, M, D, L, Lq, P, im2col_step, ptr,
                                sh.ctypes.data, ls.ctypes.data, None)


def test_argument_errors_are_reported_not_printed(lib):
    good = (2, 30, 2, 4, 2, 5, 2)
    shapes, lsi = [[6, 4], [3, 2]], [0, 24]
    assert _call_forward(lib, good, shapes, lsi, null=0) == -1                      # MSDA_ERR_NULL_POINTER
    assert _call_forward(lib, (2, 30, 2, 0, 2, 5, 2), shapes, lsi) == -2            # non-positive D
    assert _call_forward(lib, (2, 31, 2, 4, 2, 5, 2), shapes, lsi) == -2            # sum H*W != S
    assert "!= S" in _lib.last_error()
    assert _call_forward(lib, good, shapes, [0, 25]) == -2                          # level runs past S
    assert _call_forward(lib, good, [[6, 4], [0, 2]], lsi) == -2                    # empty level
    assert _call_forward(lib, (3, 30, 2, 4, 2, 5, 2), shapes, lsi, im2col_step=2) == -3   # 3 % min(3,2) != 0
    assert "must divide im2col_step" in _lib.last_error()
    assert _call_forward(lib, good, shapes, lsi, im2col_step=0) == -3
    assert _call_forward(lib, (4, 1 << 20, 8, 64, 1, 5, 2), [[1024, 1024]], [0]) == -4   # >= 2^31 elements
    assert _call_forward(lib, good, shapes, lsi, ptr=0x1002) == -5                  # misaligned pointers
    with pytest.raises(RuntimeError, match="MSDA_ERR_MISALIGNED"):
        _lib.check(-5)


def test_host_variants_are_declared_and_not_implemented_as_in_the_reference(lib):
    """msda_forward_cpu / msda_backward_cpu (SURVEY.md section 8b "host (_cpu) variants of both"): the reference's bodies are
    AT_ERROR("Not implement on cpu") (src/cpu/ms_deform_attn_cpu.cpp:17-41); these return MSDA_ERR_NOT_ON_CPU with that text, touch no
    argument and launch nothing (called here with null pointers, on a machine without a GPU)."""
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.msda_forward_cpu.argtypes = [vp] * 5 + [ci] * 8 + [vp]
    lib.msda_forward_cpu.restype = ci
    lib.msda_backward_cpu.argtypes = [vp] * 6 + [ci] * 8 + [vp] * 3
    lib.msda_backward_cpu.restype = ci
    assert lib.msda_forward_cpu(*([None] * 5), 1, 1, 1, 1, 1, 1, 1, 64, None) == -8
    assert _lib.last_error() == "Not implement on cpu"
    assert lib.msda_backward_cpu(*([None] * 6), 1, 1, 1, 1, 1, 1, 1, 64, None, None, None) == -8
    assert _lib.last_error() == "Not implement on cpu"


def test_im2col_step_contract_matches_reference(lib):
    """reference ms_deform_attn_cuda.cu:48-52: step' = min(N, step); N % step' must be 0."""
    shapes, lsi = [[6, 4], [3, 2]], [0, 24]
    for N, step in [(6, 4), (5, 2), (7, 3), (64 * 3 + 1, 64)]:
        assert _call_forward(lib, (N, 30, 2, 4, 2, 5, 2), shapes, lsi, im2col_step=step) == -3
        assert f"batch({N}) must divide im2col_step({min(N, step)})" in _lib.last_error()


def test_error_notes_of_the_other_translation_units_name_the_call():
    """A failed argument check in the conv / lin256 / ffn / cls / attn / rows entry points leaves "<entry point>: <class of error>" in
    msda_last_error() (no GPU needed: the checks come before any launch)."""
    from richsem_amd import _lib
    L = _lib.load()
    n = ctypes.c_int64(0)
    rc = L.msda_conv_wgrad_workspace_bytes(1, 8, 8, 100, 128, 1, 1, 1, 0, ctypes.byref(n))        # C_in not a multiple of 128
    assert rc != 0
    with pytest.raises(RuntimeError, match="msda_conv_wgrad_workspace_bytes.*dimension"):
        _lib.check(rc)
    rc = L.msda_sine_embed_bf16(None, 4, 1, 4, 128, 10000.0, None, None)
    with pytest.raises(RuntimeError, match="msda_sine_embed_bf16.*null pointer"):
        _lib.check(rc)


def test_round4_entry_points_check_their_arguments_on_the_host():
    """ABI v7's additions (convolution rings / fused input gradient / grouped weight gradients, box refinement with the reference's gradient,
    the criterion's pair kernels): null pointers, bad dimensions, misalignment and bad option values are refused before any launch"""
    L = _lib.load()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    for bad in (2, 5, 7, -2):
        assert L.msda_conv_set_ring(bad) != 0
    with pytest.raises(RuntimeError, match="msda_conv_set_ring.*BAD_OPTION"):
        _lib.check(L.msda_conv_set_ring(5))
    for ok in (-1, 3, 4, 6, 0):
        assert L.msda_conv_set_ring(ok) == 0
    assert L.msda_conv_set_wgrad_ring(2) != 0 and L.msda_conv_set_wgrad_ring(1) == 0
    # fused input gradient: nulls, C_out not a multiple of 32, an output size that does not match, a misaligned `add`
    args = lambda dy, w, dx, add=None, mask=None, Cout=64, Ho=4: (dy, w, 1, Ho, 4, Cout, 64, 1, 1, 1, 0, 4, 4, add, mask, dx, None, None)
    assert L.msda_conv_dgrad_fused_bf16(*args(None, p16, p16)) == -1
    assert L.msda_conv_dgrad_fused_bf16(*args(p16, p16, p16, Cout=48)) == -2
    assert L.msda_conv_dgrad_fused_bf16(*args(p16, p16, p16, Ho=5)) == -2
    with pytest.raises(RuntimeError, match="msda_conv_dgrad_fused_bf16.*align"):
        _lib.check(L.msda_conv_dgrad_fused_bf16(*args(p16, p16, p16, add=p16 + 2)))
    # grouped weight gradients: no problems, too many, a channel count the kernel does not take, a missing workspace
    n = ctypes.c_int64(0)
    arr = (_lib.WgradProblem * 9)()
    for j in range(9):
        arr[j] = _lib.WgradProblem(p16, p16, p16, None, None, 1, 64, 64, 128, 128, 1, 1, 1, 0)
    assert L.msda_conv_wgrad_group_workspace_bytes(arr, 0, ctypes.byref(n)) == -2
    assert L.msda_conv_wgrad_group_workspace_bytes(arr, 9, ctypes.byref(n)) == -2
    assert L.msda_conv_wgrad_group_workspace_bytes(arr, 3, None) == -1
    assert L.msda_conv_wgrad_group_workspace_bytes(arr, 3, ctypes.byref(n)) == 0 and n.value > 0 and n.value % 4 == 0
    assert L.msda_conv_wgrad_group_bf16(arr, 3, None, None) == -1                     # the split needs its workspace
    arr[1].Cin = 100
    with pytest.raises(RuntimeError, match="msda_conv_wgrad_group_workspace_bytes.*dimension"):
        _lib.check(L.msda_conv_wgrad_group_workspace_bytes(arr, 3, ctypes.byref(n)))
    # box refinement / pair losses
    assert L.msda_box_refine_backward_ref(p, p, 8, p, 0, None, 1e-3, p, None) == -1        # grad_ref without ref
    assert L.msda_box_refine_backward_ref(p, p, 0, p, 0, p, 1e-3, p, None) == -2
    assert L.msda_box_pair_loss_f32(p16, p16, p, 0, 5.0, 2.0, p, p16, None) == -2
    assert L.msda_box_pair_loss_f32(p16, None, p, 4, 5.0, 2.0, p, p16, None) == -1
    with pytest.raises(RuntimeError, match="msda_box_pair_loss_f32.*align"):
        _lib.check(L.msda_box_pair_loss_f32(p16 + 4, p16, p, 4, 5.0, 2.0, p, p16, None))
    assert L.msda_focal_pos_sum_f32(p, p, 0, 0.25, p, p, None) == -2
    assert L.msda_focal_pos_sum_f32(None, p, 4, 0.25, p, p, None) == -1
