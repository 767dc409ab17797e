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


# ---- two ranks, the REAL kernels, sharded by bench.shard_batch (both ranks share the box's one GPU; the process group is gloo
# because RCCL refuses two ranks on one device -- what is exercised is the sharding, the per-rank streams and the timing
# reduction of bench.py around the HIP kernels) -------------------------------------------------------------------------------
def _rank_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from richsem_amd import MultiScaleDeformableAttention as MSDA
    call = W.shrunk(W.call_E(2 * world), 4)
    full = W.make_inputs(call, "sigma4", seed=77)
    t = {k: v.cuda() for k, v in bench.shard_batch(full, rank, world).items()}
    out = MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), out=out.cpu().numpy(), gv=gv.cpu().numpy(), gl=gl.cpu().numpy(),
             ga=ga.cpu().numpy())
    assert abs(bench.reduce_elapsed(0.010 * (rank + 1), dist) - 0.010 * world) < 1e-9
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_run_the_hip_kernels_on_their_shards(tmp_path):
    import torch.multiprocessing as mp
    from richsem_amd import MultiScaleDeformableAttention as MSDA
    world = 2
    mp.spawn(_rank_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    call = W.shrunk(W.call_E(2 * world), 4)
    t = {k: v.cuda() for k, v in W.make_inputs(call, "sigma4", seed=77).items()}
    out = MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    for name, ref in (("out", out), ("gv", gv), ("gl", gl), ("ga", ga)):
        got = np.concatenate([p[name] for p in parts], 0)
        ref = ref.cpu().numpy()
        # images are independent; the kernels chosen for a batch of 2 and of 4 may differ, so sums agree at rounding level
        assert np.abs(got - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-12), name


def test_composed_step_under_ddp_with_optimizer(nccl_world1):
    """bench.py's `full_step_ddp` protocol (bench_step.run_ddp) on the small composed step: DistributedDataParallel over RCCL (world 1 here: a
    self-all-reduce on RCCL's stream beside the step's kernels), AdamW on every trained parameter -- every trained parameter must get a
    gradient (DDP is built without find_unused_parameters), and the bf16 caches of the layers must follow the optimizer's updates"""
    import bench_step
    dev = torch.device("cuda", 0)
    holder = {}

    def make():
        m = bench_step.Step(n_img=2, height=256, width=320, boxes_per_image=5, seed=3, dev=dev)
        holder["m"] = m
        holder["w0"] = m.encoder[0].linear1.weight.detach().clone()

        def batch():
            b = bench_step.Step.batch(m, seed=0)
            m.prepare(b[1], b[2])
            return b
        m.batch = batch
        return m

    res = bench_step.run_ddp(2, dev, dist, steps=2, warmup=1, optimizer=True, make_model=make)
    assert res["world"] == 1 and res["ms"] > 0 and res["ms_no_collective"] > 0
    assert res["parameters_without_gradient"] == [], res["parameters_without_gradient"]
    assert np.isfinite(res["loss"])
    assert not torch.equal(holder["m"].encoder[0].linear1.weight.detach(), holder["w0"])      # AdamW moved the weights
