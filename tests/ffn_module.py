"""Test helper (moved out of the product in round 5: only tests/test_gpu_ffn.py uses it -- the layers call the fused block directly).
The feed-forward block of the deformable transformer layers, mirror of the reference's ``forward_ffn``
(models/richsem/deformable_transformer.py:840-866 encoder layer, :907-944 decoder layer):

    src = norm(src + dropout(linear2(dropout(activation(linear1(src))))))

Same sub-module names as the reference layers use for it (``linear1``, ``linear2``, ``norm``; the layers call the norm
``norm2`` / ``norm3`` -- load with ``strict=False`` or rename), so its parameters load from a reference checkpoint.  When the
input is a CUDA bfloat16 tensor with d_model = 256, the activation is relu and no dropout is active, the whole block is ONE
MFMA kernel (richsem_amd/csrc/ffn_mfma.hip); otherwise it is the reference's op-by-op sequence.
"""
import torch
import torch.nn.functional as F
from torch import nn

from richsem_amd.functions.ffn import FUSED_FFN_MIN_TOKENS, FusedFFNFunction


class FFN(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu"):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm = nn.LayerNorm(d_model)
        self.dropout_p = dropout
        self.activation = activation
        self.fused = True
        self.fused_min_tokens = FUSED_FFN_MIN_TOKENS   # below it the op sequence is faster (functions/ffn.py)

    def _can_fuse(self, src):
        drop = self.dropout_p > 0 and self.training
        return (self.fused and src.is_cuda and src.dtype == torch.bfloat16 and self.activation == "relu" and not drop
                and src.shape[-1] == 256 and self.linear1.out_features % 32 == 0 and self.linear1.out_features <= 4096
                and src.numel() // src.shape[-1] >= self.fused_min_tokens)

    def forward(self, src):
        if self._can_fuse(src):
            return FusedFFNFunction.apply(src, self.linear1.weight.to(torch.bfloat16), self.linear1.bias.float(),
                                          self.linear2.weight.to(torch.bfloat16), self.linear2.bias.float(),
                                          self.norm.weight.float(), self.norm.bias.float(), self.norm.eps)
        # op-by-op sequence (fp32, or bf16 activations with fp32 master parameters: the parameters are cast to the input's type)
        act = {"relu": F.relu, "gelu": F.gelu}[self.activation]
        dt = src.dtype
        h = F.dropout(act(F.linear(src, self.linear1.weight.to(dt), self.linear1.bias.to(dt))), self.dropout_p, self.training)
        src = src + F.dropout(F.linear(h, self.linear2.weight.to(dt), self.linear2.bias.to(dt)), self.dropout_p, self.training)
        return F.layer_norm(src, (src.shape[-1],), self.norm.weight.to(dt), self.norm.bias.to(dt), self.norm.eps)
