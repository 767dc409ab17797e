from .ms_deform_attn_func import MSDeformAttnFunction  # noqa: F401
from .fused import MaskRows, MSDeformAttnFusedFunction  # noqa: F401
from .ffn import AddLayerNormFunction, FFNSmallFunction, FusedFFNFunction, ffn_forward_bf16, pack_w2_bf16  # noqa: F401
