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
