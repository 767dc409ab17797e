"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the reference's AttentionPool2d, never imported by the product.

Follows clip/model.py:69-91: tokens = [mean; features] + positional embedding (:70-72), then torch's multi-head attention with
separate projection weights for the single query token (:73-90): q = (x_0 Wq^T + bq) / sqrt(head_dim), k = x Wk^T + bk,
v = x Wv^T + bv per head, softmax over the HW + 1 tokens, concatenated heads through c_proj.  Written the plain way (every token is
projected), NOT the restructured way the product uses.  Pinned by tests/golden/attnpool_*.npz, which
tests/golden/make_golden_attnpool.py generated from the reference class itself (tests/test_oracle_attnpool.py)."""
import numpy as np


def attnpool(x, params, num_heads):
    """x (K, C, H, W); params: dict with positional_embedding, {q,k,v,c}_proj.weight / .bias -> (K, output_dim)"""
    K, C = x.shape[:2]
    t = x.reshape(K, C, -1).transpose(2, 0, 1)                        # (HW, K, C)   model.py:70
    t = np.concatenate([t.mean(axis=0, keepdims=True), t], axis=0)     # :71
    t = t + params["positional_embedding"][:, None, :]                 # :72
    hd = C // num_heads
    q = (t[:1] @ params["q_proj.weight"].T + params["q_proj.bias"]) * (hd ** -0.5)
    k = t @ params["k_proj.weight"].T + params["k_proj.bias"]
    v = t @ params["v_proj.weight"].T + params["v_proj.bias"]
    T1 = t.shape[0]
    qh = q.reshape(1, K, num_heads, hd)
    kh = k.reshape(T1, K, num_heads, hd)
    vh = v.reshape(T1, K, num_heads, hd)
    s = np.einsum("qkhd,tkhd->kht", qh, kh)
    s = s - s.max(axis=-1, keepdims=True)
    a = np.exp(s)
    a = a / a.sum(axis=-1, keepdims=True)
    o = np.einsum("kht,tkhd->khd", a, vh).reshape(K, C)
    return o @ params["c_proj.weight"].T + params["c_proj.bias"]


def load_case(path):
    z = np.load(path)
    params = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
    return z["x"], params, int(z["num_heads"]), z["y"]
