from .ms_deform_attn import MSDeformAttn  # noqa: F401
from .ffn import FFN  # noqa: F401
