#!/usr/bin/env python3
"""bench.py -- RichSem training hot path (multi-scale deformable attention) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--loc init|uniform]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over one synthetic batch of BASELINE.json configs[1]
("RichSem R50 4-scale LVIS, bs=2/GPU, 1xMI355X"): per GPU N=2 images of 1333x800 (padded 800x1344, S=22323),
the 12 MSDeformAttn forward and 12 backward calls of one training step -- 6 encoder calls E (Lq = 22323) and
6 decoder calls Dd (Lq = 1092), M=8, D=32, L=P=4, fp32 (the reference op is fp32/fp64 only) -- issued through the
drop-in module `MultiScaleDeformableAttention` -> C ABI -> gfx950 HIP kernels.  Inputs are resident in HBM before
the timed region.  value = images/s of THIS PATH (images per step / step time), aggregated over all GPUs; the
backbone, the GEMMs and the criterion of a full training step are not part of the path and not in the number.

Multi-GPU: the path shards by image (data parallel): every rank owns its own 2 images, no
data-path collective (the op has no parameters); weak scaling.  Timing = barrier + synchronize on both sides, max
over ranks.

Also on the JSON line:
  roofline      dominant kernel of the step: achieved = algorithmic bytes per launch / average
                launch duration, measured live by HIP events recorded by the library around that kernel on the
                stream it is launched on, inside the timed region; peak = HBM3E 8 TB/s.
  cpu_baseline  the CPU oracle (oracle/msda_oracle.c, OpenMP) timed on this box's host cores on a bounded sample:
                one E and one Dd forward+backward, scaled to the 6+6 calls of a step (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--loc", choices=["init", "uniform"], default="init",
                    help="sampling-location distribution (SURVEY.md section 8d); init = realistic local pattern")
    ap.add_argument("--images-per-gpu", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fwd-variant", type=int, default=0)
    ap.add_argument("--bwd-variant", type=int, default=0)
    return ap.parse_args()


def shard_batch(tensors, rank, world):
    """Data-parallel sharding of one call's tensors: rank r owns images [r*n, (r+1)*n) of the global batch; the
    level geometry is shared.  No data-path collective: images are independent."""
    n_total = tensors["value"].shape[0]
    if n_total % world:
        raise ValueError(f"global batch {n_total} is not divisible by world size {world}")
    n = n_total // world
    out = {}
    for k, v in tensors.items():
        out[k] = v if k in ("shapes", "lsi") else v[rank * n:(rank + 1) * n].contiguous()
    return out


def reduce_elapsed(elapsed, dist):
    """Step time of the job = the slowest rank's (MAX all-reduce); `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed
    import torch as _t
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = _t.tensor([elapsed], dtype=_t.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def total_images(images_per_gpu, world):
    return images_per_gpu * world


def measured_traffic(hip_kernels):
    """HBM bytes per launch of the given HIP kernels from the latest committed PMC summary (profiles/*_traffic.json,
    made by profiles/summarize.py from separate rocprofv3 --pmc passes), or None.  The counters cannot be read from
    inside this process; the number is attached so that it sits next to the algorithmic bytes it is compared with."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=os.path.getmtime)
    if not files:
        return None
    try:
        table = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
    total, found = 0, 0
    for want in hip_kernels.split(" + "):
        want = want.strip().rstrip(">")   # "tiled_gather_kernel<false" matches "...tiled_gather_kernel<false, true, 16>"
        hits = [v for k, v in table.items() if want in k and v["hbm_bytes_est"] > 0]
        if hits:
            total += max(h["hbm_bytes_est"] for h in hits)   # E-sized launch of that kernel
            found += 1
    return total if found else None


def cpu_baseline(calls, loc_mode, n_images):
    """Oracle timed on the host: one forward+backward of each distinct call, scaled by its repetitions."""
    from oracle import msda_oracle as O
    from richsem_amd import workload as W
    cores = min(len(os.sched_getaffinity(0)), O.max_threads(), 64)
    O.set_threads(cores)
    step_s = 0.0
    parts = []
    for call, reps in calls:
        t = W.make_inputs(call, loc_mode, seed=0)
        z = {k: v.numpy() for k, v in t.items()}
        best = float("inf")
        for _ in range(2):   # first pass warms the page cache / OpenMP pool
            t0 = time.perf_counter()
            O.forward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"])
            O.backward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"], z["grad_out"])
            best = min(best, time.perf_counter() - t0)
        step_s += reps * best
        parts.append(f"{call.name} fwd+bwd {best * 1e3:.0f} ms x{reps}")
    O.set_threads(1)
    return {"value": round(n_images / step_s, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "oracle/msda_oracle.c (OpenMP), min of 2 runs of one fwd+bwd per distinct call, scaled to "
                      "the step: " + ", ".join(parts)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the MSDeformAttn path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from richsem_amd import _lib, workload as W
    from richsem_amd import MultiScaleDeformableAttention as MSDA
    _lib.load()
    _lib.set_option("fwd_variant", args.fwd_variant)
    _lib.set_option("bwd_variant", args.bwd_variant)

    n_img = args.images_per_gpu
    calls = W.training_step_calls(n_img)
    # inputs resident in HBM before the timed region; every rank has its own images (seed by rank)
    data = [(c, reps, W.make_inputs(c, args.loc, seed=1000 * rank + i, device=dev)) for i, (c, reps) in enumerate(calls)]

    def step():
        for c, reps, t in data:
            for _ in range(reps):
                MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
        for c, reps, t in reversed(data):
            for _ in range(reps):
                MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    calls_per_step = 2 * sum(r for _, r in calls)
    for _ in range(args.warmup):
        step()
    fence()
    _lib.profile_enable(calls_per_step * args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    records = _lib.profile_collect()
    _lib.profile_enable(0)

    elapsed = reduce_elapsed(elapsed, dist)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        n_total = total_images(n_img, world)
        # per-kernel statistics from the library's event log (this rank)
        by = {}
        for r in records:
            by.setdefault((r["kind"], r["Lq"], r["variant"]), []).append(r["kernel_ms"])
        kernels = []
        for (kind, Lq, variant), ms in sorted(by.items()):
            call = next(c for c, _ in calls if c.Lq == Lq)
            nbytes = call.bytes_bwd() if kind == "bwd" else call.bytes_fwd()
            avg = sum(ms) / len(ms)
            scatter = {0: "tiled_scatter_kernel", 1: "tiled_scatter_bfp_kernel", 2: "tiled_scatter_sorted_kernel"}[
                _lib.get_option("tile_accum")]
            hip = {("fwd", 1): "fwd_direct_kernel", ("bwd", 1): "bwd_levelsum_kernel + bwd_direct_kernel", ("fwd", 2): "tiled_gather_kernel<false>",
                   ("bwd", 2): scatter + " + tiled_gather_kernel<true>"}[(kind, variant)]
            kernels.append({"kernel": f"msda_{kind}_{'direct' if variant == 1 else 'tiled'}[{call.name}]", "hip_kernels": hip,
                            "launches": len(ms), "avg_us": round(avg * 1e3, 2), "total_ms": round(sum(ms), 3),
                            "alg_bytes": nbytes, "GBps": round(nbytes / (avg * 1e-3) / 1e9, 1)})
        dom = max(kernels, key=lambda k: k["total_ms"])
        roofline = {"bound": "hbm", "kernel": dom["kernel"], "hip_kernels": dom["hip_kernels"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4), "traffic": measured_traffic(dom["hip_kernels"]),
                    "alg_bytes_per_launch": dom["alg_bytes"], "avg_launch_us": dom["avg_us"]}
        line = {
            "metric": "training images/sec, RichSem R50 4-scale 1333x800 (MSDeformAttn hot path: 12 fwd + 12 bwd "
                      "calls per step)",
            "value": round(n_total / (elapsed / args.steps), 3), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: R50 4-scale 800x1344 (S=22323), bs={n_img}/GPU, M=8 D=32 L=P=4; "
                                   f"per step 6x E(Lq=22323) + 6x Dd(Lq=1092), forward and backward; loc-{args.loc}",
                       "loc": args.loc, "images_per_gpu": n_img, "parallelism": f"dp{world} (replicas, no collective)"},
            "roofline": roofline,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(calls, args.loc, n_img)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
