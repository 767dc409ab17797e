#!/usr/bin/env python3
"""bench.py -- RichSem training hot path (multi-scale deformable attention) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 launches its own N ranks (one process per GPU, `python -m torch.distributed.run`, started
before this process touches the GPU); under an external launcher (WORLD_SIZE set) it is one of the ranks.

One STEP = one pass of the hot path over one synthetic batch of BASELINE.json configs[1]
("RichSem R50 4-scale LVIS, bs=2/GPU, 1xMI355X"): per GPU N=2 images of 1333x800 (padded 800x1344, S=22323),
the 12 MSDeformAttn forward and 12 backward calls of one training step -- 6 encoder calls E (Lq = 22323) and
6 decoder calls Dd (Lq = 1092), M=8, D=32, L=P=4, fp32 (the reference op is fp32/fp64 only) -- issued through the
drop-in module `MultiScaleDeformableAttention` -> C ABI -> gfx950 HIP kernels.  Every one of the 6 + 6 layers has
its OWN value / sampling_loc / attn_weight / grad_out tensors (as in the network: each layer projects its own value
and predicts its own offsets, reference deformable_transformer.py:870,1017), so a step streams ~1 GB of distinct
inputs instead of re-reading one cache-resident set.  Inputs are resident in HBM before the timed region.
value = images/s of THIS PATH (images per step / step time), aggregated over all GPUs; the backbone, the GEMMs and
the criterion of a full training step are not part of the path and not in the number.

Three sampling-location distributions are timed in the same run (SURVEY.md section 8d asks for the local and the
uniform one; sigma4 sits between): "init" = reference points + the module's initial offsets + N(0, 1 px) -- the
top-level `value`; "sigma4" = the same with N(0, 4 px); "uniform" = U[0,1)^2.  `value_sigma4`, `value_uniform` and
`distributions` carry the other two with their own rooflines.

Multi-GPU: the path shards by image (data parallel): every rank owns its own 2 images.  The one
collective of the reference's training step (DDP gradient all-reduce, reference main.py:204-206: ~47.6 M trainable
parameters = 190 MB fp32 in 25 MB buckets) is issued on RCCL, bucket by bucket between the backward calls, so that it
overlaps with them as DDP's does; `ms_per_step_no_collective` is the same step without it.  Weak scaling.
Timing = barrier + synchronize on both sides, max over ranks.

Also on the JSON line:
  roofline      dominant kernel of the step: achieved = algorithmic bytes per launch / average
                launch duration, measured live by HIP events recorded by the library around that call's kernels on the
                stream they are launched on, inside the timed region; peak = HBM3E 8 TB/s.
  cpu_baseline  the CPU oracle (oracle/msda_oracle.c, OpenMP) timed on this box's host cores on a bounded sample:
                one E and one Dd forward+backward, scaled to the 6+6 calls of a step (rank 0, N=1 only); `grid_sample` inside it =
                the reference's own CPU path (F.grid_sample formulation, oracle/msda_torch_oracle.py) timed the same way.
  full_step     ONE composed training step on the library's rows (bench_step.py: backbone ... criterion, forward + backward):
                eager ms / img/s, per-section times, and the same step replayed as a HIP graph with GPU ms per row.  Beside
                `value`, never it (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# what the library's profile records call "variant" -> the HIP kernels behind a call
HIP_KERNELS = {
    ("fwd", 1): "fwd_direct_kernel",
    ("bwd", 1): "bwd_levelsum_kernel + bwd_split_kernel",      # (small calls; bwd_direct_kernel instead of the split kernel for large ones)
    ("fwd", 2): "tiled_gather_kernel",
    ("fwd", 3): "fwd_split_kernel",
    ("bwd", 4): "rps_route_kernel + rps_tile_kernel",
    ("bwd", 5): "bwd_band_kernel",
    ("fwd", 5): "fwd_direct_prep_kernel",
    ("fwd", 6): "tiled_gather_kernel (raw projection in)",
}
VARIANT_NAMES = {1: "direct", 2: "tiled", 3: "split", 4: "routed", 5: "band", 6: "tiled_prep"}
PROFILE_DOMINANT = (1 + 1) * 16 + 4   # library option "profile_filter": backward calls of variant 4 (the routed kernels)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--loc", default="init,sigma4,uniform",
                    help="comma-separated sampling-location distributions to time (the first is the headline)")
    ap.add_argument("--images-per-gpu", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-collective", action="store_true", help="N > 1: leave the gradient all-reduce out")
    ap.add_argument("--no-bf16", action="store_true", help="skip the extra bf16 pass")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed rehearsal of the first measurement")
    ap.add_argument("--no-em", action="store_true", help="skip the second encoder shape (1280 x 1280 mosaic batches, S = 34000)")
    ap.add_argument("--no-ffn", action="store_true", help="skip the feed-forward (MFMA) row beside the path")
    ap.add_argument("--no-graph", action="store_true", help="skip the graph-replay variant of the step")
    ap.add_argument("--no-full-step", action="store_true", help="skip the composed training step (bench_step.py) beside the path")
    ap.add_argument("--fwd-variant", type=int, default=0)
    ap.add_argument("--bwd-variant", type=int, default=0)
    return ap.parse_args(argv)


# ---- host logic shared with the CPU tests ----------------------------------------------------------------------------
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


def grad_buckets(n_elems, bucket_bytes, elem_bytes=4):
    """[(start, stop)] element ranges of the gradient buckets DDP would all-reduce (reference main.py:204-206)."""
    per = max(1, bucket_bytes // elem_bytes)
    return [(s, min(s + per, n_elems)) for s in range(0, n_elems, per)]


def bucket_schedule(n_buckets, n_backward_calls):
    """After which backward call (1-based) each bucket is launched: spread evenly over the backward pass, the last
    bucket after the last call (like DDP, whose last bucket is the one nothing is left to hide)."""
    return [-(-(i + 1) * n_backward_calls // n_buckets) for i in range(n_buckets)]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """--gpus N without a launcher: start the N ranks as children of this (still GPU-free) process and relay rank 0's
    line.  Never re-executes a process that has touched the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def measured_traffic(hip_kernels, dtype, table_key="kernels"):
    """HBM bytes per launch of the given HIP kernels IN THE GIVEN VALUE DTYPE ("f32" | "bf16") from the latest committed PMC summary
    (profiles/*_traffic.json, made by profiles/summarize.py from separate rocprofv3 --pmc passes), or None.  The counters cannot be
    read from inside this process; the number is attached so that it sits next to the algorithmic bytes it is compared with.  A
    kernel instantiated per value type carries the type as its last template argument ("..., float>" / "..., __hip_bfloat16>"); the
    route kernels are not typed and match either.  ``table_key``: "kernels" (the headline shape E) or "kernels_Em" (the mosaic shape)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        table = json.load(open(files[-1])).get(table_key) or {}
    except Exception:
        return None
    total, found = 0, 0
    for want in hip_kernels.split(" + "):
        want = want.strip().rstrip(">")   # "tiled_gather_kernel<false" matches "...tiled_gather_kernel<false, true, 16>"
        # (a kernel instantiated for bf16 storage carries __hip_bfloat16 among its template arguments; kernels without a storage type -- the route
        # pass -- match either)
        def typed_ok(name):
            return ("<" not in name or "rps_route" in name) or (("__hip_bfloat16" in name) == (dtype == "bf16"))
        hits = [v for k, v in table.items() if want in k and v["hbm_bytes_est"] > 0 and typed_ok(k.split("|")[0])]
        if hits:
            total += max(h["hbm_bytes_est"] for h in hits)   # E-sized launch of that kernel
            found += 1
    return total if found else None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (a
    GPU box hands each job a share of its host cores; more OpenMP threads than that only fight each other)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(calls, n_images):
    """Oracle timed on the host: forward+backward of each distinct call (loc-init, layer 0), min of 5 runs at all
    cores and min of 2 at one thread, scaled by the call's repetitions per step."""
    from oracle import msda_oracle as O
    from richsem_amd import workload as W
    native = O.build_native()   # -march=native build made on THIS box (falls back to the portable -mavx2 one)
    cores = min(cpu_share(), O.max_threads())
    # the fastest thread count for this box among a few candidates (one quick E forward+backward each)
    probe = W.make_inputs(calls[0][0], "init", seed=0)
    pz = {k: v.numpy() for k, v in probe.items()}
    best_t, best_dt = cores, float("inf")
    for cand in sorted({cores, min(cores, 64), min(cores, 32), min(cores, 16)}):
        O.set_threads(cand)
        dts = []
        for _ in range(2):
            t0 = time.perf_counter()
            O.forward(pz["value"], pz["shapes"], pz["lsi"], pz["loc"], pz["aw"])
            O.backward(pz["value"], pz["shapes"], pz["lsi"], pz["loc"], pz["aw"], pz["grad_out"])
            dts.append(time.perf_counter() - t0)
        if min(dts) < best_dt:
            best_t, best_dt = cand, min(dts)
    cores = best_t
    out = {}
    parts = []
    for label, threads, reps_timed in (("all", cores, 5), ("one", 1, 2)):
        O.set_threads(threads)
        step_s = 0.0
        for call, reps in calls:
            t = W.make_inputs(call, "init", seed=0)
            z = {k: v.numpy() for k, v in t.items()}
            best = float("inf")
            for i in range(reps_timed + (1 if label == "all" else 0)):   # the first all-core pass warms caches / the OpenMP pool
                t0 = time.perf_counter()
                O.forward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"])
                O.backward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"], z["grad_out"])
                dt = time.perf_counter() - t0
                if label != "all" or i > 0:
                    best = min(best, dt)
            step_s += reps * best
            parts.append(f"{call.name} fwd+bwd {best * 1e3:.0f} ms x{reps} @{threads}t")
        out[label] = n_images / step_s
    O.set_threads(1)
    gs = grid_sample_baseline(calls, n_images, cores)
    return {"value": round(out["all"], 4), "unit": "img/s", "cores": cores, "kind": "port", "grid_sample": gs,
            "value_1thread": round(out["one"], 5), "cpu_model": cpu_model(),
            "build": "gcc -O3 -march=native -ffp-contract=off -fopenmp (built on this box)" if native else
                     "gcc -O3 -mavx2 -ffp-contract=off -fopenmp (portable build; no compiler on this box)",
            "sample": "oracle/msda_oracle.c (OpenMP), loc-init, one fwd+bwd per distinct call scaled to the 6+6 "
                      "calls of a step; min of 5 runs after a warm-up at all cores, min of 2 at 1 thread: "
                      + ", ".join(parts)}


def grid_sample_baseline(calls, n_images, cores):
    """The reference's OWN CPU path for the op -- ``ms_deform_attn_core_pytorch`` (functions/ms_deform_attn_func.py:41-61: one
    F.grid_sample per level + weighted sum), restated in oracle/msda_torch_oracle.py -- forward + autograd backward on the host's cores:
    what a user of the reference without its CUDA extension gets.  One run per distinct call after a warm-up, scaled like the oracle."""
    import torch
    from oracle import msda_torch_oracle as T
    from richsem_amd import workload as W
    old = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        step_s, parts = 0.0, []
        for call, reps in calls:
            t = W.make_inputs(call, "init", seed=0)
            args = (t["value"], t["shapes"], t["loc"], t["aw"], t["grad_out"])
            best = float("inf")
            for i in range(3 if call.Lq < 5000 else 2):      # the first pass is the warm-up
                t0 = time.perf_counter()
                T.forward_backward(*args)
                dt = time.perf_counter() - t0
                if i > 0:
                    best = min(best, dt)
            step_s += reps * best
            parts.append(f"{call.name} fwd+bwd {best * 1e3:.0f} ms x{reps}")
        return {"value": round(n_images / step_s, 4), "unit": "img/s", "cores": cores, "kind": "reference CPU path restated "
                "(torch F.grid_sample + autograd, oracle/msda_torch_oracle.py)", "sample": ", ".join(parts)}
    except Exception as e:      # noqa: BLE001 -- an extra leg must never take the benchmark line down
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    finally:
        torch.set_num_threads(old)


# ---- the measured loop -----------------------------------------------------------------------------------------------
def ffn_row(n_img, dev):
    """The first MFMA row beside the path (NOT part of the timed step or of `value`): the
    encoder layer's feed-forward block on n_img x 22323 tokens as the library's one-kernel bf16 forward, timed with events on the
    current stream, against the same block as PyTorch bf16 ops.  Roofline: dense bf16 MFMA peak of the guide (2.5 PFLOP/s)."""
    import torch
    import torch.nn.functional as F
    from richsem_amd import workload as W
    from richsem_amd.functions import ffn_forward_bf16, pack_w2_bf16
    T, D, Fh = n_img * W.call_E(n_img).S, 256, 2048
    g = torch.Generator(device=dev).manual_seed(7)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    x, w1, w2 = r(T, D).to(torch.bfloat16), (r(Fh, D) * D ** -0.5).to(torch.bfloat16), (r(D, Fh) * Fh ** -0.5).to(torch.bfloat16)
    b1, b2, gw, gb = 0.1 * r(Fh), 0.1 * r(D), 1 + 0.1 * r(D), 0.1 * r(D)
    w2p = pack_w2_bf16(w2)

    def fused():
        return ffn_forward_bf16(x, w1, b1, w2p, b2, gw, gb)

    def ops():
        h = torch.relu(F.linear(x, w1, b1.to(torch.bfloat16)))
        return F.layer_norm(x + F.linear(h, w2, b2.to(torch.bfloat16)), (D,), gw.to(torch.bfloat16), gb.to(torch.bfloat16))

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    flop = 4.0 * T * D * Fh
    t_f, t_o = timeit(fused), timeit(ops)
    # training: forward + backward through FusedFFNFunction (LayerNorm gradient kernel, K = 256 products on lin256_kernel, weight gradients
    # on conv_wgrad_kernel, dx by the library) against autograd through the same block as PyTorch bf16 ops
    from richsem_amd.functions import FusedFFNFunction
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    xa, w1a, w2a, b1a, b2a, gwa, gba = (leaf(t) for t in (x, w1, w2, b1, b2, gw, gb))
    go = torch.randn(T, D, device=dev, generator=g).to(torch.bfloat16)

    def train_fused():
        for p in (xa, w1a, w2a, b1a, b2a, gwa, gba):
            p.grad = None
        FusedFFNFunction.apply(xa, w1a, b1a, w2a, b2a, gwa, gba, 1e-5).backward(go)

    def train_ops():
        for p in (xa, w1a, w2a, b1a, b2a, gwa, gba):
            p.grad = None
        h = torch.relu(F.linear(xa, w1a, b1a.to(torch.bfloat16)))
        F.layer_norm(xa + F.linear(h, w2a, b2a.to(torch.bfloat16)), (D,), gwa.to(torch.bfloat16), gba.to(torch.bfloat16)).backward(go)

    t_tf, t_to = timeit(train_fused, 10), timeit(train_ops, 10)
    return {"what": "encoder feed-forward block forward (256 -> 2048 -> 256, relu, residual, LayerNorm), bf16, one HIP kernel; "
                    "outside the timed step", "hip_kernel": "ffn_fwd_kernel", "tokens": T, "flop": flop,
            "avg_launch_us": round(t_f * 1e6, 1),
            "train_forward_backward_us": round(t_tf * 1e6, 1), "pytorch_bf16_autograd_forward_backward_us": round(t_to * 1e6, 1),
            "roofline": {"bound": "mfma", "achieved": round(flop / t_f / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flop / t_f / 1e12 / 2500.0, 4)},
            "pytorch_bf16_ops_us": round(t_o * 1e6, 1)}


def _time_events(fn, reps):
    import torch
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def cls_row(n_img, dev):
    """Two-stage selection score (second half; NOT part of `value`): row maxima of the class logits of
    n_img x 22323 encoder tokens against 1204 text embeddings as ONE MFMA kernel on bf16 hi/lo split operands (fp32-level result), against
    the reference's op sequence (project, normalise, text product, max) as PyTorch fp32 ops.  `flop` counts the MFMA work issued."""
    import torch
    from richsem_amd import workload as W
    from richsem_amd.two_stage import ClassScorer
    T, C, P = n_img * W.call_E(n_img).S, 1204, 1024
    g = torch.Generator(device=dev).manual_seed(11)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    mem, wp, text, ls = r(T, 256), r(P, 256) * P ** -0.5, r(C, P), torch.tensor(2.659)
    sc = ClassScorer(2).prepare(wp, text, ls)

    def ops():
        f = mem @ wp.t()
        f = f / f.norm(dim=-1, keepdim=True)
        tt = text / text.norm(dim=-1, keepdim=True)
        return (ls.exp().to(dev) * (f @ tt.t())).max(-1)[0]

    want = ops()
    got = sc.max_logits(mem)
    err = float((got - want).abs().max() / want.abs().max())
    flop = 2.0 * T * 256 * (16 * ((C + 15) // 16) + 256) * 3
    t_f, t_o = _time_events(lambda: sc.max_logits(mem), 20), _time_events(ops, 5)
    return {"what": "two-stage query selection score: max over 1204 class logits per encoder token, projection + normalisation + text "
                    "product collapsed onto the 256-wide memory, bf16 hi/lo split operands (three MFMAs per tile), one HIP kernel; "
                    "outside the timed step", "hip_kernel": "cls_score_kernel<true, 2>", "tokens": T, "flop": flop,
            "avg_launch_us": round(t_f * 1e6, 1), "max_rel_diff_vs_pytorch_fp32": err,
            "roofline": {"bound": "mfma", "achieved": round(flop / t_f / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flop / t_f / 1e12 / 2500.0, 4)},
            "pytorch_fp32_ops_us": round(t_o * 1e6, 1)}


def backbone_row(n_img, dev):
    """ResNet-50 (frozen BatchNorm) forward at the training shape n_img x 3 x 800 x 1344 on the MFMA convolution kernel (SURVEY.md
    section 8a row a10, forward; NOT part of `value`): 53 convolutions with affine / ReLU / residual in their epilogues, NHWC bf16,
    against the same network as PyTorch convolutions (MIOpen) in channels-last bf16.  Synthetic weights."""
    import torch
    import torch.nn.functional as F
    from richsem_amd import workload as W
    from richsem_amd.backbone import ResNet50Frozen
    from richsem_amd.conv import ConvAffine
    sd = W.resnet50_state_dict()
    net = ResNet50Frozen(sd, device=dev)
    x = torch.randn(n_img, 3, 800, 1344, device=dev, generator=torch.Generator(device=dev).manual_seed(13))
    ConvAffine.flop_counter = [0.0]
    net(x)
    flop = ConvAffine.flop_counter[0]
    ConvAffine.flop_counter = None

    P = {}
    for k, v in sd.items():
        if v.dim() == 4:
            P[k] = v.to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for k in sd:
        if k.endswith("running_var"):
            p = k[: -len(".running_var")]
            scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
            P[p + ".s"] = scale.to(dev, torch.bfloat16).reshape(1, -1, 1, 1)
            P[p + ".b"] = (sd[p + ".bias"] - sd[p + ".running_mean"] * scale).to(dev, torch.bfloat16).reshape(1, -1, 1, 1)

    def ops():
        y = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = torch.relu(F.conv2d(y, P["conv1.weight"], stride=2, padding=3) * P["bn1.s"] + P["bn1.b"])
        y = F.max_pool2d(y, 3, 2, 1)
        for li in range(1, 5):
            b = 0
            while f"layer{li}.{b}.conv1.weight" in P:
                p, s = f"layer{li}.{b}.", 2 if (li > 1 and b == 0) else 1
                o = torch.relu(F.conv2d(y, P[p + "conv1.weight"]) * P[p + "bn1.s"] + P[p + "bn1.b"])
                o = torch.relu(F.conv2d(o, P[p + "conv2.weight"], stride=s, padding=1) * P[p + "bn2.s"] + P[p + "bn2.b"])
                o = F.conv2d(o, P[p + "conv3.weight"]) * P[p + "bn3.s"] + P[p + "bn3.b"]
                idt = (F.conv2d(y, P[p + "downsample.0.weight"], stride=s) * P[p + "downsample.1.s"] + P[p + "downsample.1.b"]) if b == 0 else y
                y = torch.relu(o + idt)
                b += 1
        return y

    t_f, t_o = _time_events(lambda: net(x), 5), _time_events(ops, 3)
    # training form: layer2-4 trained (backbone.py:65-67), forward + input gradients + weight gradients on the library's kernels
    from richsem_amd.backbone import ResNet50
    tnet = ResNet50().to(dev)
    tnet.load_state_dict(sd)
    gs = [torch.randn_like(o) for o in tnet(x)]

    def train_step():
        for p in tnet.parameters():
            p.grad = None
        torch.autograd.backward(tnet(x), gs)

    t_t = _time_events(train_step, 3)
    return {"what": "ResNet-50 + FrozenBatchNorm2d (53 convolutions, affine / relu / residual in the epilogues), NHWC bf16, HIP implicit-GEMM "
                    "kernels; outside the timed step", "hip_kernel": "conv_fwd_kernel<CO_TILES, PT, MODE> (+ conv_wgrad_kernel in training)",
            "input": [n_img, 3, 800, 1344], "flop": flop, "ms": round(t_f * 1e3, 3),
            "train_forward_backward_ms": round(t_t * 1e3, 3),
            "roofline": {"bound": "mfma", "achieved": round(flop / t_f / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flop / t_f / 1e12 / 2500.0, 4)},
            "pytorch_bf16_channels_last_ms": round(t_o * 1e3, 3)}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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

    modes = [m for m in args.loc.split(",") if m]
    for m in modes:
        if m not in W.LOC_MODES:
            raise SystemExit(f"unknown --loc {m}; choose from {W.LOC_MODES}")
    n_img = args.images_per_gpu
    calls = W.training_step_calls(n_img)
    # inputs resident in HBM before the timed region: one tensor set PER LAYER, every rank its own images (seed by rank)
    layers = []   # (call, dict of tensors, {mode: loc})
    for ci, (c, reps) in enumerate(calls):
        for layer in range(reps):
            seed = 100000 * rank + 1000 * ci + layer
            t = W.make_inputs(c, "init", seed=seed, device=dev)
            locs = {m: (t["loc"] if m == "init" else W.make_loc(c, m, seed=seed, device=dev)) for m in modes}
            layers.append((c, t, locs))
    calls_per_step = 2 * len(layers)

    collective = dist is not None and not args.no_collective
    works = []
    if collective:
        flat_grads = torch.zeros(W.GRAD_ALLREDUCE_ELEMS, dtype=torch.float32, device=dev)
        buckets = grad_buckets(W.GRAD_ALLREDUCE_ELEMS, W.DDP_BUCKET_BYTES)
        fire_after = bucket_schedule(len(buckets), len(layers))

    def step(mode, with_collective, bf16=False, layers_=None):
        vk, gk = ("value_bf16", "grad_out_bf16") if bf16 else ("value", "grad_out")
        layers_ = layers if layers_ is None else layers_
        for c, t, locs in layers_:
            MSDA.ms_deform_attn_forward(t[vk], t["shapes"], t["lsi"], locs[mode], t["aw"], 64)
        nb = 0
        for k, (c, t, locs) in enumerate(reversed(layers_), 1):
            MSDA.ms_deform_attn_backward(t[vk], t["shapes"], t["lsi"], locs[mode], t["aw"], t[gk], 64)
            while with_collective and nb < len(buckets) and fire_after[nb] <= k:
                s, e = buckets[nb]
                works.append(dist.all_reduce(flat_grads[s:e], async_op=True))   # RCCL, its own stream: overlaps
                nb += 1
        for w in works:   # the optimizer step needs every bucket: the step ends when they are all in
            w.wait()
        works.clear()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mode, with_collective, profile, bf16=False, layers_=None):
        _lib.set_option("locality_monitor", _lib.get_option("locality_monitor"))   # forget the previous distribution
        for _ in range(max(args.warmup, 3)):   # (the locality monitor settles within the first two steps)
            step(mode, with_collective, bf16, layers_)
        fence()
        n_calls = (calls_per_step if layers_ is None else 2 * len(layers_)) * args.steps
        if profile:
            # Inside the timed region the library brackets ONLY the routed backward -- the dominant kernel, whose live duration the
            # `roofline` object is about -- with its event pair: an event pair is two more packets on the stream, and bracketing all 24
            # calls of a step cost 0.19-0.21 ms of its 3.0 (MI355X, tools/event_cost.py: 2.99-3.01 ms with, 2.80-2.81 ms without).  The
            # other kernels' durations come from a second, UNTIMED pass of the same steps below.
            _lib.set_option("profile_filter", PROFILE_DOMINANT)
            _lib.profile_enable(n_calls)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(mode, with_collective, bf16, layers_)
        fence()
        elapsed = time.perf_counter() - t0
        records = []
        if profile:
            records = [dict(r, measured="timed loop") for r in _lib.profile_collect()]
            _lib.profile_enable(0)
            _lib.set_option("profile_filter", 0)
            _lib.profile_enable(n_calls)
            for _ in range(args.steps):
                step(mode, False, bf16, layers_)
            fence()
            rest = [dict(r, measured="untimed pass") for r in _lib.profile_collect()]
            _lib.profile_enable(0)
            dominant = {(r["kind"], r["variant"]) for r in records}
            records += [r for r in rest if (r["kind"], r["variant"]) not in dominant]   # (all of them if the routed backward did not run)
        return reduce_elapsed(elapsed, dist), records

    def graph_replay(mode):
        """The same step captured once into a HIP graph and replayed (what a caller that captures its training step gets: no
        launch gaps between the ~40 kernels of a step).  Reported beside `value`, never as it; no collective inside.  The library
        allocates nothing during capture, so the stream is warmed up first; None if capture is not possible here."""
        try:
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                for _ in range(3):
                    step(mode, False)
            torch.cuda.synchronize()
            from richsem_amd.capture import quiet_gc
            graph = torch.cuda.CUDAGraph()
            with quiet_gc(), torch.cuda.graph(graph, stream=side):
                step(mode, False)
            for _ in range(2):
                graph.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.steps):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / args.steps
        except Exception as e:   # noqa: BLE001 -- a diagnostic extra must never take the benchmark line down
            print(f"[bench] graph replay skipped: {e}", file=sys.stderr)
            return None

    # Untimed rehearsal of the first measurement: on a freshly leased box the FIRST timed loop of the process ran 0.4-0.5 ms per step
    # slower than every later one -- BENCH_r03 / r04: init 3.57 / 3.61 ms against sigma4 3.49 / 3.44 ms and graph replay 3.02 ms on the
    # same box, although init's kernels are the faster ones (3.05 against 3.31 ms of kernel time per step); boxes that had run anything
    # before do not show it.  Whatever settles during that first loop (clocks, first use of the event pool, the host's caches) is not the
    # path's steady state, so the same loop runs once before it is timed.  W warm-up steps then K timed steps still follow, as asked.
    settle_steps = 0
    if not args.no_settle:
        timed(modes[0], collective, True)
        settle_steps = max(args.warmup, 3) + args.steps

    def host_issue_us(mode):
        """host time to ISSUE one call (python shim + ctypes + the library's planning + hipLaunchKernel), no waiting: median over 5 steps"""
        per = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(mode, False)
            per.append((time.perf_counter() - t0) / calls_per_step * 1e6)
            torch.cuda.synchronize()
        return sorted(per)[len(per) // 2]

    results = {}
    for mode in modes:
        elapsed, records = timed(mode, collective, True)
        res = {"elapsed": elapsed, "records": records}
        if collective:
            res["elapsed_nc"], _ = timed(mode, False, False)
        results[mode] = res
    graph_ms = graph_replay(modes[0]) if (world == 1 and not args.no_graph) else None
    host_us = host_issue_us(modes[0])
    # The second encoder shape of BASELINE.json's configs[2] / [3]: every other step there is a 1280 x 1280 ImageNet-LVIS mosaic batch
    # (reference main.py:53-71, datasets/transforms.py:356-357,437-445) -> S = Lq = 34000 (SURVEY.md section 8d "Em").  Its six encoder
    # forward + backward calls, each layer on its own tensors, timed like the headline step; beside `value`, never it (rank 0's GPU only:
    # it is a per-kernel measurement, no collective).
    if world == 1 and not args.no_em:
        em = W.call_Em(n_img)
        em_layers = []
        for layer in range(6):
            t = W.make_inputs(em, "init", seed=7000 + layer, device=dev)
            em_layers.append((em, t, {m: (t["loc"] if m == "init" else W.make_loc(em, m, seed=7000 + layer, device=dev))
                                      for m in ("init", "uniform")}))
        for m in ("init", "uniform"):
            elapsed, records = timed(m, False, True, layers_=em_layers)
            results["Em/" + m] = {"elapsed": elapsed, "records": records}
        del em_layers
        torch.cuda.empty_cache()
    if not args.no_bf16:   # the same step with bf16 value / out / grad tensors (library entry points msda_*_bf16), headline distribution
        for c, t, locs in layers:
            t["value_bf16"], t["grad_out_bf16"] = t["value"].to(torch.bfloat16), t["grad_out"].to(torch.bfloat16)
        elapsed, records = timed(modes[0], collective, True, bf16=True)
        results["bf16"] = {"elapsed": elapsed, "records": records}

    # N > 1: the composed step as a data-parallel TRAINING step (DistributedDataParallel over RCCL + AdamW), by every rank; beside
    # `value`, never it (what the scaling curve of `value` hides behind 3 ms of operator kernels has 27 ms of step to hide in here)
    full_step_ddp = None
    if world > 1 and not args.no_full_step:
        # (no try / except here: a rank that swallowed its own failure would leave the others inside DDP's all-reduces until the driver's
        # timeout; an exception ends this rank, and torch.distributed.run tears the job down with a non-zero exit)
        import bench_step
        full_step_ddp = bench_step.run_ddp(n_img, dev, dist, steps=5, warmup=3)
    if rank == 0:
        n_total = total_images(n_img, world)

        def summarise(mode):
            res = results[mode]
            by = {}
            how = {}
            for r in res["records"]:
                by.setdefault((r["kind"], r["Lq"], r["variant"]), []).append(r["kernel_ms"])
                how[(r["kind"], r["Lq"], r["variant"])] = r.get("measured", "timed loop")
            kernels = []
            for (kind, Lq, variant), ms in sorted(by.items()):
                call = next(c for c in [c for c, _ in calls] + [W.call_Em(n_img)] if c.Lq == Lq)
                e_v = 2 if mode == "bf16" else 4   # bytes per value / out / grad element
                nbytes = call.bytes_bwd(e_v) if kind == "bwd" else call.bytes_fwd(e_v)
                avg = sum(ms) / len(ms)
                kernels.append({"kernel": f"msda_{kind}_{VARIANT_NAMES.get(variant, variant)}[{call.name}]",
                                "hip_kernels": HIP_KERNELS.get((kind, variant), "?"), "launches": len(ms),
                                "avg_us": round(avg * 1e3, 2), "total_ms": round(sum(ms), 3), "alg_bytes": nbytes,
                                "GBps": round(nbytes / (avg * 1e-3) / 1e9, 1), "measured": how[(kind, Lq, variant)]})
            dom = max(kernels, key=lambda k: k["total_ms"])
            roofline = {"bound": "hbm", "kernel": dom["kernel"], "hip_kernels": dom["hip_kernels"],
                        "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4),
                        "traffic": measured_traffic(dom["hip_kernels"], "bf16" if mode == "bf16" else "f32", "kernels_Em" if mode.startswith("Em/") else "kernels"),
                        "alg_bytes_per_launch": dom["alg_bytes"], "avg_launch_us": dom["avg_us"]}
            out = {"value": round(n_total / (res["elapsed"] / args.steps), 3),
                   "ms_per_step": round(res["elapsed"] / args.steps * 1e3, 4), "roofline": roofline, "kernels": kernels}
            if "elapsed_nc" in res:
                out["ms_per_step_no_collective"] = round(res["elapsed_nc"] / args.steps * 1e3, 4)
            return out

        summ = {m: summarise(m) for m in modes}
        head = summ[modes[0]]
        par = f"dp{world}" + (" (gradient all-reduce 190 MB fp32 in 25 MB buckets on RCCL, overlapped with backward)"
                              if collective else " (replicas, no collective)")
        line = {
            "metric": "training images/sec, RichSem R50 4-scale 1333x800 (MSDeformAttn hot path: 12 fwd + 12 bwd "
                      "calls per step)",
            "value": head["value"], "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: R50 4-scale 800x1344 (S=22323), bs={n_img}/GPU, M=8 D=32 L=P=4; "
                                   f"per step 6x E(Lq=22323) + 6x Dd(Lq=1092), forward and backward, 12 distinct "
                                   f"per-layer tensor sets; loc-{modes[0]}",
                       "loc": modes[0], "images_per_gpu": n_img, "parallelism": par},
            "roofline": head["roofline"],
            "kernels": head["kernels"],
            "host_us_per_call": round(host_us, 2),
            "profiling": "inside the timed region the library brackets only the routed backward (the dominant kernel: `roofline`) with HIP "
                         "events on the launch stream; every other kernel's duration comes from an untimed pass of the same steps afterwards "
                         "(`kernels[].measured`) -- bracketing all 24 calls of a step cost the step 0.2 ms of event packets; a bracket includes 1-2 us of its own "
                         "packets, so the brackets' sum (`kernel_ms_per_step`) can come out above the step they were not all part of",
            "kernel_ms_per_step": round(sum(k["total_ms"] for k in head["kernels"]) / args.steps, 4),
            "settle": {"untimed_steps_before_warmup": settle_steps,
                       "why": "one untimed rehearsal of the first measurement loop: the first timed loop of a process on a freshly leased box "
                              "ran 0.4-0.5 ms per step slower than later ones with the same kernel times (BENCH_r03 / r04)"},
        }
        if "ms_per_step_no_collective" in head:
            line["ms_per_step_no_collective"] = head["ms_per_step_no_collective"]
        for m in modes[1:]:
            line["value_" + m] = summ[m]["value"]
        line["distributions"] = {m: {k: v for k, v in summ[m].items() if k != "kernels" or m != modes[0]} for m in modes}
        if "bf16" in results:
            b = summarise("bf16")
            line["bf16"] = {"dtype": "bf16 value/out/grad, f32 locations/weights and accumulation", "loc": modes[0],
                            "value": b["value"], "ms_per_step": b["ms_per_step"], "roofline": b["roofline"],
                            "kernels": b["kernels"]}
        if "Em/init" in results:
            wl = {}
            for m in ("init", "uniform"):
                e = summarise("Em/" + m)
                wl[m] = {"ms_per_6_layers_fwd_bwd": e["ms_per_step"], "kernels": e["kernels"], "roofline": e["roofline"]}
            line["workloads"] = {"Em": {
                "what": "the second encoder shape: 1280 x 1280 ImageNet-LVIS mosaic batches of configs[2] / [3] (every other step there), "
                        f"bs={n_img}/GPU, S = Lq = 34000 (160^2, 80^2, 40^2, 20^2), M=8 D=32 L=P=4, fp32; six encoder forward + backward "
                        "calls on six tensor sets; algorithmic bytes per call 243.7 MB forward / 417.8 MB backward (SURVEY.md section 8d); "
                        "beside `value`, never it",
                "loc": wl}}
        if graph_ms is not None:
            line["graph_replay"] = {"what": "the same step (no collective) captured into one HIP graph and replayed; not `value`",
                                    "ms_per_step": round(graph_ms, 4), "img_per_s": round(n_total / (graph_ms * 1e-3), 3)}
        if not args.no_ffn and world == 1:      # (the rows beside the path are single-GPU measurements, like the CPU baseline)
            line["mfma_row"] = ffn_row(n_img, dev)
            line["mfma_rows"] = {"two_stage_class_score": cls_row(n_img, dev), "resnet50_forward": backbone_row(n_img, dev)}
        if world == 1 and not args.no_full_step:      # the composed training step on the library's rows; beside `value`, never it
            layers.clear()                             # (the path's 12 resident tensor sets are not needed any more)
            torch.cuda.empty_cache()
            try:
                import bench_step
                line["full_step"] = bench_step.run(n_img, dev, steps=5, warmup=3)
            except Exception as e:      # noqa: BLE001
                line["full_step"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if full_step_ddp is not None:
            line["full_step_ddp"] = full_step_ddp
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(calls, n_img)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
