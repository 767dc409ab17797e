"""Two-stage query selection without the class logits (second half).

Reference: models/richsem/deformable_transformer.py:368-372 -- ``enc_out_class_embed(output_memory)`` ((bs, sum(HW), classes) logits:
215 MB at the training shape) of which only ``.max(-1)[0]`` feeds ``torch.topk(..., num_queries, dim=1)[1]`` -- with the CLIP-text
classifier of models/richsem/richsem.py:176-184 in its shipped configuration (``dino_visual_proj`` = bias-free ``nn.Linear(256, 1024)``,
frozen ``logit_scale``).  :class:`ClassScorer` collapses projection, normalisation, text product and row maximum into one MFMA kernel
over the 256-wide memory (csrc/cls_mfma.hip; C ABI ``msda_cls_pack`` / ``msda_cls_max_scores``) and hands the scores to the library's
top-k kernel (``msda_topk_f32``).  Forward only: the selection carries no gradient (the reference gathers with the indices).
"""
import ctypes

import torch

from . import _lib
from .dn import topk_indices


def _stream(dev):
    return _lib.raw_stream(dev)


class ClassScorer:
    """``prepare`` once per weight update, ``max_logits`` / ``topk_proposals`` per forward.

    proj_weight (proj_dim, 256) = ``class_embed.dino_visual_proj.weight``; text_embed (classes, proj_dim) = ``class_embed.text_embed``
    (un-normalised, as stored); logit_scale = the CLIP parameter (log of the scale).  ``parts`` = 2: weights as bf16 hi + lo parts
    (fp32-level scores from fp32 memory); 1: plain bf16 product."""

    def __init__(self, parts=2):
        assert parts in (1, 2)
        self.parts = parts
        self.packed = None
        self.classes = 0
        self.scale = 1.0

    @torch.no_grad()
    def prepare(self, proj_weight, text_embed, logit_scale):
        if not proj_weight.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        assert proj_weight.dim() == 2 and proj_weight.shape[1] == 256, "the encoder memory must be 256 wide"
        assert text_embed.dim() == 2 and text_embed.shape[1] == proj_weight.shape[0]
        dev = proj_weight.device
        wp = proj_weight.detach().double()
        te = text_embed.detach().to(dev).double()
        te = te / te.norm(dim=-1, keepdim=True)                       # richsem.py:180
        # formed in fp64 and rounded once (two small products per weight update): the kernel then splits them into bf16 parts
        G = (te @ wp).float().contiguous()                            # (classes, 256): t^_c . (Wp x) = (G x)_c
        A = (wp.t() @ wp).float().contiguous()                        # (256, 256):  |Wp x|^2 = x . (A x)
        self.classes = G.shape[0]
        self.scale = float(torch.as_tensor(logit_scale).detach().double().exp())      # richsem.py:181
        L = _lib.load()
        n = ctypes.c_int64(0)
        _lib.check(L.msda_cls_packed_elems(self.classes, ctypes.byref(n)))
        self.packed = torch.empty(n.value, dtype=torch.int16, device=dev)
        with _lib.on_device(dev):
            _lib.check(L.msda_cls_pack(G.data_ptr(), self.classes, A.data_ptr(), 256, self.packed.data_ptr(), _stream(dev)))
        return self

    @torch.no_grad()
    def max_logits(self, memory):
        """memory (..., 256) float32 or bfloat16 on the GPU -> (...) float32: ``enc_out_class_embed(memory).max(-1)[0]``"""
        if self.packed is None:
            raise RuntimeError("ClassScorer.prepare has not been called")
        if not memory.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        if memory.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError(f"ClassScorer: float32 / bfloat16 memory, got {memory.dtype}")
        assert memory.shape[-1] == 256
        x = memory.detach().contiguous()
        tokens = x.numel() // 256
        scores = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        if tokens == 0:
            return scores
        with _lib.on_device(x.device):
            _lib.check(_lib.load().msda_cls_max_scores(x.data_ptr(), int(x.dtype == torch.bfloat16), self.packed.data_ptr(), tokens, 256,
                                                       self.classes, self.scale, self.parts, scores.data_ptr(), _stream(x.device)))
        return scores

    @torch.no_grad()
    def topk_proposals(self, output_memory, k):
        """deformable_transformer.py:368-372: (bs, sum(HW), 256) -> (bs, k) int64 indices of the k largest row maxima"""
        assert output_memory.dim() == 3
        return topk_indices(self.max_logits(output_memory), k)
