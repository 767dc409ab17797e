from .ms_deform_attn import MSDeformAttn  # noqa: F401
from .encoder_layer import DeformableTransformerEncoderLayer, get_reference_points  # noqa: F401
from .decoder_layer import DeformableTransformerDecoderLayer  # noqa: F401
from .attnpool import AttentionPool2d, clip_box_targets  # noqa: F401
from .decoder import TransformerDecoder, MLP, gen_sineembed_for_position, inverse_sigmoid, refine_boxes, sine_embed_bf16  # noqa: F401
