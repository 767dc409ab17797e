"""Top-level alias so that `import MultiScaleDeformableAttention as MSDA` (reference
models/richsem/ops/functions/ms_deform_attn_func.py:18) works unchanged when this repository
is on sys.path.  The implementation is richsem_amd/MultiScaleDeformableAttention.py."""
from richsem_amd.MultiScaleDeformableAttention import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
