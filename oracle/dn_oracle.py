"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the integer part of the reference's ``prepare_for_cdn``
(/root/reference/models/richsem/dn_components.py), statement by statement, for the parity tests of richsem_amd/dn.py.

PARITY UNPINNED: the reference function cannot be run here (its module imports torchvision, which this image lacks, and it
calls ``.cuda()`` unconditionally), and the reference ships no test or fixture for it; this file follows the source text only.
"""
import numpy as np


def prepare_for_cdn_indices(known_num, dn_number, num_queries, use_cdn=True, add_gt=False):
    # dn_components.py:27-41
    dn_number = dn_number * 2
    batch_size = len(known_num)
    if int(max(known_num)) == 0:
        dn_number = 1
    else:
        if dn_number >= 100:
            dn_number = dn_number // (int(max(known_num) * 2))
        elif dn_number < 1:
            dn_number = 1
    if dn_number == 0:
        dn_number = 1
    if add_gt:
        dn_number += 1
    # :44, :50-53  batch_idx = cat(full_like(labels, i)); known_bid = batch_idx.repeat(2 * dn_number, 1).view(-1)
    batch_idx = np.concatenate([np.full(n, i, dtype=np.int64) for i, n in enumerate(known_num)]) if batch_size else np.zeros(0, np.int64)
    known_bid = np.tile(batch_idx, (2 * dn_number, 1)).reshape(-1)
    n_boxes = int(sum(known_num))
    # :55-61
    single_pad = int(max(known_num))
    pad_size = int(single_pad * 2 * dn_number)
    positive_idx = np.tile(np.arange(n_boxes, dtype=np.int64)[None, :], (dn_number, 1))
    positive_idx = positive_idx + (np.arange(dn_number, dtype=np.int64) * n_boxes * 2)[:, None]
    positive_idx = positive_idx.flatten()
    negative_idx = positive_idx + n_boxes
    # :135-139
    map_known_indice = np.zeros(0, np.int64)
    if len(known_num):
        m = np.concatenate([np.arange(num, dtype=np.int64) for num in known_num])
        map_known_indice = np.concatenate([m + single_pad * i for i in range(2 * dn_number)]).astype(np.int64)
    # :144-153
    if not use_cdn:
        pad_size = pad_size // 2
        group_pad = single_pad
    else:
        group_pad = single_pad * 2
    # :155-176
    tgt_size = pad_size + num_queries
    attn_mask = np.zeros((tgt_size, tgt_size), dtype=bool)
    attn_mask[pad_size:, :pad_size] = True
    for i in range(dn_number):
        if i == 0:
            attn_mask[group_pad * i:group_pad * (i + 1), group_pad * (i + 1):pad_size] = True
        if i == dn_number - 1:
            attn_mask[group_pad * i:group_pad * (i + 1), :group_pad * i] = True
        else:
            attn_mask[group_pad * i:group_pad * (i + 1), group_pad * (i + 1):pad_size] = True
            attn_mask[group_pad * i:group_pad * (i + 1), :group_pad * i] = True
    return {"known_bid": known_bid, "map_known_indice": map_known_indice, "attn_mask": attn_mask, "positive_idx": positive_idx,
            "negative_idx": negative_idx, "pad_size": pad_size, "num_dn_group": dn_number, "single_pad": single_pad,
            "group_pad": group_pad}
