"""CPU, world_size 2 over gloo: the multi-GPU leg of bench.py shards the path by image with no data-path collective
(SURVEY.md section 8e); only the timing is reduced (max over ranks) behind a barrier.  Here two processes run that
protocol with the CPU oracle standing in for the device kernels (the oracle is test infrastructure -- this file is
a test), and check that the sharded results equal the unsharded ones."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import msda_oracle as O
    from richsem_amd import workload as W
    import bench

    call = W.shrunk(W.call_E(2 * world), 8)                 # the global batch: 2 images per rank
    full = W.make_inputs(call, "init", seed=123)
    shard = bench.shard_batch(full, rank, world)            # this rank's images
    assert shard["value"].shape[0] == 2
    z = {k: v.numpy() for k, v in shard.items()}
    out = O.forward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"])
    gv, gl, ga = O.backward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"], z["grad_out"])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), out=out, gv=gv, gl=gl, ga=ga)
    # timing protocol of bench.py: barrier, local elapsed, MAX over ranks, images summed over ranks
    elapsed = bench.reduce_elapsed(0.010 * (rank + 1), dist)
    assert abs(elapsed - 0.010 * world) < 1e-9
    total = bench.total_images(2, world)
    assert total == 2 * world
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_unsharded(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import msda_oracle as O
    from richsem_amd import workload as W
    call = W.shrunk(W.call_E(2 * world), 8)
    full = {k: v.numpy() for k, v in W.make_inputs(call, "init", seed=123).items()}
    out = O.forward(full["value"], full["shapes"], full["lsi"], full["loc"], full["aw"])
    gv, gl, ga = O.backward(full["value"], full["shapes"], full["lsi"], full["loc"], full["aw"], full["grad_out"])
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    # images are independent: concatenating the ranks' results along the batch axis reproduces the full batch exactly
    assert np.array_equal(np.concatenate([p["out"] for p in parts], 0), out)
    assert np.array_equal(np.concatenate([p["gv"] for p in parts], 0), gv)
    assert np.array_equal(np.concatenate([p["gl"] for p in parts], 0), gl)
    assert np.array_equal(np.concatenate([p["ga"] for p in parts], 0), ga)


def test_shard_batch_partitions_images():
    sys.path.insert(0, ROOT)
    import bench
    t = dict(value=torch.arange(8.).view(8, 1, 1, 1), loc=torch.zeros(8, 1, 1, 1, 1, 2), aw=torch.zeros(8, 1, 1, 1, 1),
             grad_out=torch.zeros(8, 1, 1), shapes=torch.tensor([[1, 1]]), lsi=torch.tensor([0]))
    seen = []
    for r in range(4):
        s = bench.shard_batch(t, r, 4)
        assert s["shapes"] is t["shapes"] and s["value"].is_contiguous()
        seen += s["value"].flatten().tolist()
    assert seen == list(range(8))
    with pytest.raises(ValueError):
        bench.shard_batch(t, 0, 3)


def _allreduce_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    n = 1000
    flat = torch.full((n,), float(rank + 1))
    buckets = bench.grad_buckets(n, 4 * 128)            # 128-element buckets
    fire = bench.bucket_schedule(len(buckets), 12)
    works, nb = [], 0
    for k in range(1, 13):                               # the 12 backward calls of a step
        while nb < len(buckets) and fire[nb] <= k:
            s, e = buckets[nb]
            works.append(dist.all_reduce(flat[s:e], async_op=True))
            nb += 1
    assert nb == len(buckets)
    for w in works:
        w.wait()
    assert torch.equal(flat, torch.full((n,), float(sum(range(1, world + 1)))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_bucketed_gradient_allreduce():
    """bench.py --gpus N issues the reference's DDP gradient all-reduce bucket by bucket between the backward calls
    (reference main.py:204-206); here the same schedule over gloo, world size 2."""
    mp.spawn(_allreduce_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_bucket_plan_covers_the_gradient_once():
    sys.path.insert(0, ROOT)
    import bench
    from richsem_amd import workload as W
    b = bench.grad_buckets(W.GRAD_ALLREDUCE_ELEMS, W.DDP_BUCKET_BYTES)
    assert b[0][0] == 0 and b[-1][1] == W.GRAD_ALLREDUCE_ELEMS and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert len(b) == 8 and 4 * (b[0][1] - b[0][0]) == 25 * 1024 * 1024
    sched = bench.bucket_schedule(len(b), 12)
    assert sched == sorted(sched) and sched[-1] == 12 and sched[0] >= 1


def test_self_launch_builds_a_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts its ranks itself (as children, before any GPU call)."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        class R:
            returncode = 0
        return R()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


# ---- the data-parallel training-step protocol of bench.py's `full_step_ddp` (bench_step.run_ddp) on two gloo ranks -----------------------
class _TinyStep(torch.nn.Module):
    """stand-in for bench_step.Step on the CPU (the Step's rows have no CPU path): what run_ddp needs of a module -- ``batch()`` -> forward
    arguments, a scalar loss out of ``forward`` -- with rank-dependent data, so that equal parameters after the steps prove the all-reduce"""

    def __init__(self, rank):
        super().__init__()
        torch.manual_seed(5)                      # equal initial weights on both ranks
        self.net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
        self.rank = rank

    def batch(self):
        g = torch.Generator().manual_seed(100 + self.rank)
        return (torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g))

    def forward(self, x, y):
        return ((self.net(x) - y) ** 2).mean()


def _ddp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench_step
    holder = {}

    def make():
        holder["m"] = _TinyStep(rank)
        return holder["m"]

    res = bench_step.run_ddp(8, "cpu", dist, steps=3, warmup=1, optimizer=True, make_model=make)
    assert res["world"] == world and res["ms"] > 0 and res["ms_no_collective"] > 0 and res["parameters_without_gradient"] == []
    assert abs(res["img_per_s"] - world * 8 / (res["ms"] * 1e-3)) < 0.05 * res["img_per_s"]      # (`ms` is rounded to 10 us)
    torch.save({"res": res, "w": holder["m"].net[0].weight.detach().clone()}, os.path.join(out_dir, f"ddp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_step_protocol_on_two_gloo_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_ddp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), f"ddp{r}.pt"), weights_only=False) for r in range(world))
    assert a["res"]["ms"] == b["res"]["ms"] and a["res"]["ms_no_collective"] == b["res"]["ms_no_collective"]      # MAX over ranks, on every rank
    # the timed DDP steps (with all-reduce) moved both ranks' weights together; the no_sync steps that follow let them drift by their own
    # data -- so the weights differ now, but only by those last (1 + 3) un-synchronised AdamW steps of lr 1e-4
    assert float((a["w"] - b["w"]).abs().max()) < 4 * 1e-4 * 1.5
