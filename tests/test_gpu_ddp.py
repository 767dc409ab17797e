"""GPU (-m gpu): the operator under ``DistributedDataParallel`` (reference main.py:204-206 wraps the whole model in DDP,
backend nccl = RCCL on ROCm).  World size 1 on the one GPU of the test box: what is exercised is the machinery around
the op, not the wire -- backward runs on the autograd engine's thread, DDP's bucket hooks fire between the op's kernels,
and the (self-)all-reduce is enqueued on RCCL's stream while the op's HIP kernels run on the compute stream.  Gradients
must equal the un-wrapped module's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from richsem_amd import workload as W
from richsem_amd.modules import MSDeformAttn

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture
def nccl_world1():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_module_under_ddp_matches_unwrapped(nccl_world1):
    torch.manual_seed(0)
    call = W.shrunk(W.call_E(2), 4)
    shapes, lsi = W.level_tensors(call, "cuda")
    C = call.M * call.D
    plain = MSDeformAttn(C, call.L, call.M, call.P).cuda()
    with torch.no_grad():   # offsets / weights that depend on the query, so that every parameter gets a gradient
        plain.sampling_offsets.weight.normal_(0, 0.02)
        plain.attention_weights.weight.normal_(0, 0.1)
    wrapped = torch.nn.parallel.DistributedDataParallel(
        MSDeformAttn(C, call.L, call.M, call.P).cuda(), device_ids=[0], bucket_cap_mb=1)   # several buckets
    wrapped.module.load_state_dict(plain.state_dict())
    query = torch.randn(call.N, call.Lq, C, device="cuda")
    src = torch.randn(call.N, call.S, C, device="cuda")
    ref = W.encoder_reference_points(call).cuda()[None, :, None, :].expand(call.N, call.Lq, call.L, 2).contiguous()
    grad = torch.randn(call.N, call.Lq, C, device="cuda")

    def run(mod):
        q, s = query.clone().requires_grad_(True), src.clone().requires_grad_(True)
        out = mod(q, ref, s, shapes, lsi, None)
        out.backward(grad)
        torch.cuda.synchronize()
        return out.detach(), q.grad, s.grad

    o1, q1, s1 = run(plain)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):   # unrelated traffic on another stream while DDP's backward runs
        junk = torch.randn(4096, 4096, device="cuda")
        for _ in range(8):
            junk = junk @ junk * 1e-3
    o2, q2, s2 = run(wrapped)
    side.synchronize()
    # the window kernels' sums depend on the arrival order of LDS / global atomics at rounding level
    tol = dict(rtol=1e-4, atol=1e-5)
    assert torch.allclose(o1, o2, **tol) and torch.allclose(q1, q2, **tol) and torch.allclose(s1, s2, **tol)
    for (n1, p1), (n2, p2) in zip(plain.named_parameters(), wrapped.module.named_parameters()):
        assert n1 == n2 and p1.grad is not None and p2.grad is not None
        scale = float(p1.grad.abs().max()) + 1e-12
        assert float((p1.grad - p2.grad).abs().max()) / scale < 1e-4, n1
    assert np.isfinite(float(o2.abs().sum()))
