"""Host-side mirror of the integer part of the reference's ``prepare_for_cdn`` (models/richsem/dn_components.py:11-193): the
denoising-group arithmetic on the host (plain Python ints, as the reference) and the index / mask tensors on the device
(kernels richsem_amd/csrc/msda_dn.h, C ABI ``msda_dn_indices_i64`` / ``msda_dn_attn_mask_u8``).  SURVEY.md section 8 row a12:
int64 / bool work, bit-exact.  The noisy labels / boxes and the embeddings around it stay with the caller (they are random
floating-point work, not part of this row).
"""
import torch

from . import _lib


def dn_group_count(dn_number, known_num, add_gt=False):
    """dn_components.py:27-41: denoising groups from the configured dn_number and the per-image ground-truth counts."""
    dn_number = dn_number * 2
    mx = int(max(known_num)) if len(known_num) else 0
    if mx == 0:
        dn_number = 1
    elif dn_number >= 100:
        dn_number = dn_number // (mx * 2)
    elif dn_number < 1:
        dn_number = 1
    if dn_number == 0:
        dn_number = 1
    if add_gt:
        dn_number += 1
    return dn_number


def prepare_dn_layout(known_num, dn_number, num_queries, use_cdn=True, add_gt=False, device="cuda"):
    """known_num: ground-truth boxes per image; dn_number as configured (before the reference's scaling).  Returns a dict with
    the reference's names: ``known_bid``, ``map_known_indice`` (int64), ``attn_mask`` (bool (tgt, tgt)), ``positive_idx`` /
    ``negative_idx`` (int64), ``pad_size``, ``num_dn_group``, ``single_pad``, ``group_pad``."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("Not implemented on the CPU")
    known_num = [int(k) for k in known_num]
    batch = len(known_num)
    groups = dn_group_count(dn_number, known_num, add_gt)
    total = sum(known_num)
    single_pad = int(max(known_num)) if batch else 0
    pad_size = single_pad * 2 * groups
    lib = _lib.load()
    stream = _lib.raw_stream(dev)
    cum = torch.tensor([0] + list(torch.tensor(known_num, dtype=torch.int64).cumsum(0).tolist()) if batch else [0], dtype=torch.int64,
                       device=dev)
    n = total * 2 * groups
    known_bid = torch.empty(n, dtype=torch.int64, device=dev)
    map_known_indice = torch.empty(n, dtype=torch.int64, device=dev)
    with _lib.on_device(dev):
        if batch and n:
            _lib.check(lib.msda_dn_indices_i64(cum.data_ptr(), batch, total, 2 * groups, single_pad, known_bid.data_ptr(),
                                               map_known_indice.data_ptr(), stream))
        # dn_components.py:58-61: positive_idx = arange(total) + 2 * total * group, negative_idx = positive_idx + total
        grp = torch.arange(groups, dtype=torch.int64, device=dev)[:, None]
        positive_idx = (torch.arange(total, dtype=torch.int64, device=dev)[None, :] + grp * (total * 2)).flatten()
        negative_idx = positive_idx + total
        if use_cdn:
            group_pad = single_pad * 2
        else:   # dn_components.py:144-151: the negative halves are dropped; positive_idx now selects the padded slots of the
            # positive halves (single_pad per group), the caller applies it to the padded query tensors
            positive_idx = (torch.arange(single_pad, dtype=torch.int64, device=dev)[None, :] + grp * (single_pad * 2)).flatten()
            pad_size = pad_size // 2
            group_pad = single_pad
        tgt = pad_size + num_queries
        mask = torch.empty((tgt, tgt), dtype=torch.uint8, device=dev)
        _lib.check(lib.msda_dn_attn_mask_u8(mask.data_ptr(), tgt, pad_size, group_pad, stream))
    return {"known_bid": known_bid, "map_known_indice": map_known_indice, "attn_mask": mask.view(torch.bool),
            "positive_idx": positive_idx, "negative_idx": negative_idx, "pad_size": pad_size, "num_dn_group": groups,
            "single_pad": single_pad, "group_pad": group_pad}


def topk_indices(scores, k, return_values=False):
    """``torch.topk(scores, k, dim=1)[1]`` for a (rows, n) float32 tensor on the GPU (deformable_transformer.py:370-372: the 900
    encoder proposals with the largest class score), one workgroup per row: indices in descending order of the score, equal scores
    lowest index first.  Rows too long for the kernel (n > 36864) or k > 1024 go to ``torch.topk``."""
    if not scores.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    assert scores.dim() == 2 and scores.dtype == torch.float32
    rows, n = scores.shape
    if n > 36864 or k > 1024:
        v, i = torch.topk(scores, k, dim=1)
        return (i, v) if return_values else i
    s = scores.contiguous()
    idx = torch.empty((rows, k), dtype=torch.int64, device=s.device)
    val = torch.empty((rows, k), dtype=torch.float32, device=s.device) if return_values else None
    with _lib.on_device(s.device):
        _lib.check(_lib.load().msda_topk_f32(s.data_ptr(), rows, n, k, idx.data_ptr(), val.data_ptr() if val is not None else None,
                                             _lib.raw_stream(s.device)))
    return (idx, val) if return_values else idx
