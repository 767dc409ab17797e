from .ms_deform_attn_func import MSDeformAttnFunction  # noqa: F401
from .fused import MaskRows, MSDeformAttnFusedFunction  # noqa: F401
