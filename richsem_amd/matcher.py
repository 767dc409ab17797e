"""Hungarian matching with the cost blocks formed on the GPU (criterion plumbing).

Mirror of the reference's ``HungarianMatcher`` (models/richsem/matcher.py:8-78; same constructor, same ``forward(outputs,
targets)`` result: a list over images of (query indices, target indices) as int64 tensors).  What changes is the plumbing the
survey names: the reference forms the cost of every query against every target of the whole batch with a dozen element-wise
kernels, keeps the diagonal blocks only, and brings each matrix to the host with a synchronising ``.cpu()`` -- seven times a step
(richsem.py:1136, :1204, :1255).  Here one kernel (``msda_matcher_cost_*``, csrc/msda_matcher.h) writes the diagonal blocks only,
and :func:`match_many` lays the blocks of ALL decoder outputs of a step into one buffer that reaches the host with ONE
asynchronous copy and ONE event wait.  The assignment itself stays scipy's ``linear_sum_assignment`` on the host, as
``north_star`` asks (criterion in host Python).

:func:`global_num_boxes` / :class:`AsyncLossLog` remove the two remaining per-step device round trips of the criterion and the
engine (richsem.py:1143-1147, engine.py:84-91): the box count is known on the host and reduced there, the reduced loss dictionary
for the log is read one step late.
"""
import ctypes

import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from . import _lib


def _stream(dev):
    return _lib.raw_stream(dev)


class CostPlan:
    """Host-side bookkeeping of one batch of targets: per-image counts, their prefix on host and device, concatenated labels /
    boxes (matcher.py:55-57)."""

    def __init__(self, targets, device, dtype):
        self.sizes = [int(v["boxes"].shape[0]) for v in targets]
        self.offsets = [0]
        for s in self.sizes:
            self.offsets.append(self.offsets[-1] + s)
        self.total = self.offsets[-1]
        self.offsets_dev = torch.tensor(self.offsets, dtype=torch.int64).to(device, non_blocking=True)
        if self.total:
            self.tgt_ids = torch.cat([v["labels"] for v in targets]).to(device=device, dtype=torch.int64).contiguous()
            self.tgt_boxes = torch.cat([v["boxes"] for v in targets]).to(device=device, dtype=dtype).contiguous()
        else:
            self.tgt_ids = torch.zeros(0, dtype=torch.int64, device=device)
            self.tgt_boxes = torch.zeros(0, 4, dtype=dtype, device=device)


def cost_blocks(pred_logits, pred_boxes, plan, cost_class, cost_bbox, cost_giou, focal_alpha, out=None):
    """Diagonal cost blocks of one decoder output on the device: a flat tensor of nq * plan.total elements, image b's (nq x T_b)
    block starting at nq * plan.offsets[b].  ``out``: slice of a larger buffer to write into."""
    if not pred_logits.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    dt = pred_logits.dtype
    if dt not in (torch.float32, torch.float64):
        raise RuntimeError(f"matcher cost: float32 / float64 logits, got {dt}")
    bs, nq, C = pred_logits.shape
    assert pred_boxes.shape == (bs, nq, 4) and len(plan.sizes) == bs
    logits, boxes = pred_logits.detach().contiguous(), pred_boxes.detach().to(dt).contiguous()
    n = nq * plan.total
    if out is None:
        out = torch.empty(n, dtype=dt, device=logits.device)
    assert out.numel() == n and out.dtype == dt and out.is_contiguous()
    if n:
        fn = getattr(_lib.load(), "msda_matcher_cost_" + ("f32" if dt == torch.float32 else "f64"))
        with _lib.on_device(logits.device):
            _lib.check(fn(logits.data_ptr(), boxes.data_ptr(), plan.tgt_ids.data_ptr(), plan.tgt_boxes.data_ptr(),
                          plan.offsets_dev.data_ptr(), bs, nq, C, plan.total, float(cost_class), float(cost_bbox), float(cost_giou),
                          float(focal_alpha), out.data_ptr(), _stream(logits.device)))
    return out


def _assign(host, nq, plan):
    res = []
    for b, tb in enumerate(plan.sizes):
        block = host[nq * plan.offsets[b]: nq * plan.offsets[b + 1]].view(nq, tb)
        i, j = linear_sum_assignment(block.numpy())
        res.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
    return res


class HungarianMatcher(nn.Module):
    """Same signature and result as the reference's class (matcher.py:8-78)."""

    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, focal_alpha=0.25):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self.focal_alpha = focal_alpha
        self._pinned = None

    def _host_buffer(self, n, dtype):
        if self._pinned is None or self._pinned.numel() < n or self._pinned.dtype != dtype:
            self._pinned = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
        return self._pinned[:n]

    @torch.no_grad()
    def match_many(self, outputs_list, targets):
        """Match several decoder outputs (each a dict with "pred_logits" (bs, nq_o, C) and "pred_boxes" (bs, nq_o, 4)) against the
        same targets: one cost kernel per output into one device buffer, one copy to the host, one wait.  Returns a list (per
        output) of the reference's per-image index pairs."""
        return self.match_many_end(self.match_many_begin(outputs_list, targets))

    @torch.no_grad()
    def match_many_begin(self, outputs_list, targets):
        """the device half of :meth:`match_many`: cost kernels, the copy to pinned host memory and an event behind it are ENQUEUED; nothing
        waits.  Work enqueued after this call (a frozen teacher's forward, say) runs on the GPU while :meth:`match_many_end` waits for the
        copy and solves the assignments on the host."""
        first = outputs_list[0]["pred_logits"]
        plan = CostPlan(targets, first.device, first.dtype)
        nqs = [o["pred_logits"].shape[1] for o in outputs_list]
        total = sum(nq * plan.total for nq in nqs)
        dev_buf = torch.empty(total, dtype=first.dtype, device=first.device)
        at = 0
        for o, nq in zip(outputs_list, nqs):
            cost_blocks(o["pred_logits"], o["pred_boxes"], plan, self.cost_class, self.cost_bbox, self.cost_giou, self.focal_alpha,
                        out=dev_buf[at: at + nq * plan.total])
            at += nq * plan.total
        host = self._host_buffer(total, first.dtype)
        host.copy_(dev_buf, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(first.device))
        return plan, nqs, host, done, dev_buf

    @torch.no_grad()
    def match_many_end(self, pending):
        plan, nqs, host, done, _keep = pending
        done.synchronize()          # the one wait of the step's matching (the reference: one .cpu() per output)
        res, at = [], 0
        for nq in nqs:
            res.append(_assign(host[at: at + nq * plan.total], nq, plan))
            at += nq * plan.total
        return res

    @torch.no_grad()
    def forward(self, outputs, targets):
        return self.match_many([outputs], targets)[0]


def global_num_boxes(indices, world_size=1, group=None):
    """``num_boxes`` of the criterion (richsem.py:1143-1147: number of matched targets summed over the ranks, divided by the world
    size, clamped to >= 1) without the device round trip: the count is a host integer already, so it is reduced as a host tensor
    (a gloo group) -- or not at all on one rank -- instead of a CUDA tensor + ``.item()``."""
    n = float(sum(len(t[1]) for t in indices))
    if world_size > 1:
        import torch.distributed as dist
        t = torch.tensor([n], dtype=torch.float32)
        dist.all_reduce(t, group=group)     # `group` must be a CPU (gloo) group: the tensor lives on the host
        n = float(t.item())
    return max(n / world_size, 1.0)


class AsyncLossLog:
    """The engine's logging reduction (engine.py:84-91 with util/misc.py:139-164) without its per-step ``.item()``: the loss
    dictionary is stacked in sorted key order (as reduce_dict does), all-reduced asynchronously, copied to pinned memory, and
    READ ONE STEP LATE -- ``push`` returns the previous step's reduced dictionary (None on the first call), ``flush`` the last."""

    def __init__(self, world_size=1, group=None, average=True):
        self.world_size, self.group, self.average = world_size, group, average
        self._pending = None

    def _finish(self):
        if self._pending is None:
            return None
        names, host, event = self._pending
        if event is not None:
            event.synchronize()
        self._pending = None
        return {k: float(v) for k, v in zip(names, host.tolist())}

    @torch.no_grad()
    def push(self, loss_dict):
        prev = self._finish()
        names = sorted(loss_dict.keys())
        values = torch.stack([loss_dict[k].detach().float().reshape(()) for k in names], dim=0)
        if self.world_size > 1:
            import torch.distributed as dist
            # on RCCL the wait below is a stream dependency, not a host block; on a host (gloo) group it blocks, as it must
            dist.all_reduce(values, group=self.group, async_op=True).wait()
            if self.average:
                values = values / self.world_size
        if values.is_cuda:
            host = torch.empty(values.shape, dtype=values.dtype, pin_memory=True)
            host.copy_(values, non_blocking=True)
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(values.device))
        else:
            host, event = values, None
        self._pending = (names, host, event)
        return prev

    def flush(self):
        return self._finish()


class FocalNegativeSum(torch.autograd.Function):
    """``sum_rows w[row] * sum_c (1 - alpha) * sigmoid(x)^2 * softplus(x)``: the sigmoid focal loss (reference richsem.py:1124-1160 through
    ``sigmoid_focal_loss``) of a logit tensor as if every entry were negative, with one weight per row (query) -- the criterion adds what
    the few positive entries contribute instead.  One kernel forward, one backward (``msda_focal_neg_sum_f32 / msda_focal_neg_grad_f32``,
    csrc/rows_api.hip) where PyTorch's ops make a dozen passes over the (6, N, queries, classes) logits.
    ``apply(logits (..., C) float32, row_weight (...) float32, alpha)`` -> 0-dim float32."""

    @staticmethod
    def forward(ctx, logits, row_weight, alpha):
        if not logits.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        assert logits.dtype == torch.float32 and row_weight.dtype == torch.float32 and row_weight.numel() * logits.shape[-1] == logits.numel()
        x, w = logits.contiguous(), row_weight.contiguous()
        rows, C = w.numel(), x.shape[-1]
        partial = torch.empty(4096, dtype=torch.float64, device=x.device)
        n = ctypes.c_int(0)
        with _lib.on_device(x.device):
            _lib.check(_lib.load().msda_focal_neg_sum_f32(x.data_ptr(), w.data_ptr(), rows, C, float(alpha), partial.data_ptr(), 4096,
                                                          ctypes.byref(n), _lib.raw_stream(x.device)))
        ctx.save_for_backward(x, w)
        ctx.alpha = float(alpha)
        return partial[:n.value].sum().float()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = torch.empty_like(x)
        gs = g.reshape(1).float().contiguous()
        with _lib.on_device(x.device):
            _lib.check(_lib.load().msda_focal_neg_grad_f32(x.data_ptr(), w.data_ptr(), w.numel(), x.shape[-1], ctx.alpha, gs.data_ptr(),
                                                           gx.data_ptr(), _lib.raw_stream(x.device)))
        return gx, None, None


class _PairSum(torch.autograd.Function):
    """a weighted per-pair loss as one kernel: the forward computes the sum AND its gradient w.r.t. the predictions; the backward scales"""

    @staticmethod
    def _launch(ctx, fn, pred, *args):
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred)
        with _lib.on_device(pred.device):
            _lib.check(fn(pred.data_ptr(), *args, loss.data_ptr(), grad.data_ptr(), _lib.raw_stream(pred.device)))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return (grad * g,) + (None,) * (ctx.n_inputs - 1)


class BoxPairLoss(_PairSum):
    """``sum_k w[k] * (c_l1 * |p_k - t_k|_1 + c_giou * (1 - GIoU(p_k, t_k)))`` over K matched (cx, cy, w, h) pairs (SetCriterion.loss_boxes,
    richsem.py:1162-1188) as one kernel (``msda_box_pair_loss_f32``): ``apply(pred (K, 4), target (K, 4), weight (K), c_l1, c_giou)`` -> 0-dim;
    gradient for ``pred``.  PyTorch's op sequence is ~35 launches forward and ~70 backward on a few thousand pairs."""

    @staticmethod
    def forward(ctx, pred, target, weight, c_l1, c_giou):
        if not pred.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        assert pred.dtype == torch.float32 and pred.dim() == 2 and pred.shape[1] == 4 and target.shape == pred.shape and weight.numel() == pred.shape[0]
        p, t, w = pred.contiguous(), target.detach().float().contiguous(), weight.detach().float().contiguous()
        ctx.n_inputs = 5
        if p.shape[0] == 0:
            ctx.save_for_backward(torch.zeros_like(p))
            return p.sum()
        return _PairSum._launch(ctx, _lib.load().msda_box_pair_loss_f32, p, t.data_ptr(), w.data_ptr(), p.shape[0], float(c_l1), float(c_giou))


class FocalPositiveSum(_PairSum):
    """``sum_k w[k] * (alpha (1 - q_k)^2 softplus(-x_k) - (1 - alpha) q_k^2 softplus(x_k))``, q = sigmoid(x): what the positive entries of a
    sigmoid focal loss contribute instead of the all-negative term :class:`FocalNegativeSum` counted for them, as one kernel
    (``msda_focal_pos_sum_f32``): ``apply(x (K), weight (K), alpha)`` -> 0-dim; gradient for ``x``."""

    @staticmethod
    def forward(ctx, x, weight, alpha):
        if not x.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        assert x.dtype == torch.float32 and x.dim() == 1 and weight.numel() == x.numel()
        xc, w = x.contiguous(), weight.detach().float().contiguous()
        ctx.n_inputs = 3
        if xc.numel() == 0:
            ctx.save_for_backward(torch.zeros_like(xc))
            return xc.sum()
        return _PairSum._launch(ctx, _lib.load().msda_focal_pos_sum_f32, xc, w.data_ptr(), xc.numel(), float(alpha))
