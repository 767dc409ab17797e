"""richsem_amd -- MI355X (gfx950) native multi-scale deformable attention for RichSem.

Scope: the one data-parallel hot path of the reference (MengLcool/RichSem), i.e. the operator
behind ``models/richsem/ops`` -- hand-written HIP kernels behind a C ABI
(include/richsem_msda.h, richsem_amd/csrc), a drop-in for the reference's compiled module
``MultiScaleDeformableAttention`` and host-side mirrors of ``MSDeformAttnFunction`` /
``MSDeformAttn`` with the reference's names, arguments and state-dict keys.

There is no CPU / PyTorch fallback: a missing HIP library raises.
"""
import sys

__version__ = "0.1.0"


def install_dropin():
    """Register the drop-in under the reference's extension name, so that the reference's own
    ``import MultiScaleDeformableAttention as MSDA`` (ops/functions/ms_deform_attn_func.py:18) resolves to it."""
    from . import MultiScaleDeformableAttention as shim
    sys.modules.setdefault("MultiScaleDeformableAttention", shim)
    return shim
