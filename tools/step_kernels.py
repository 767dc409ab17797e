#!/usr/bin/env python3
"""rocprofv3 kernel stats of `bench_step.py --no-graph --steps K --warmup W` -> profiles/<tag>_step_kernels.md: the per-kernel table of
the composed training step (per step = totals / (K + W)).

    python tools/step_kernels.py gpurun_out/r03/step_trace 13 r03
"""
import csv
import glob
import os
import sys

GROUPS = (("library: msda (deformable attention)", ("msda::",)),
          ("library: conv (forward / dgrad / wgrad / pool / groupnorm)", ("conv_", "pool_nhwc", "groupnorm")),
          ("library: linear / ffn / attention / layernorm", ("lin256", "ffn_", "attn_", "add_layernorm", "pack_w2")),
          ("library: class score / top-k / roi / attnpool / matcher / dn", ("cls_", "topk_", "roi_", "attnpool", "matcher_", "dn_", "mask_rows")),
          ("hipBLASLt / rocBLAS GEMM", ("Cijk", "rocblas")),
          ("runtime copy / fill", ("__amd_rocclr",)),
          ("PyTorch element-wise / reduce / sort / index kernels", ("at::native", "rocprim", "at_cuda_detail", "at::cuda", "indexing_", "softmax_",
                                                                   "reduction_prod")))


def group_of(name):
    for g, keys in GROUPS:
        if any(k in name for k in keys):
            return g
    return "other"


def main():
    d, steps, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    f = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))[0]
    rows = [(r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))]
    total = sum(r[3] for r in rows)
    out = [f"# composed training step: per-kernel table ({tag})", "",
           f"`rocprofv3 --kernel-trace --stats -- python3 bench_step.py --no-graph --steps 10 --warmup 3` on one MI355X; {steps} eager steps "
           "in the trace, figures PER STEP (totals / steps).  N = 2 images of 800 x 1333, bf16 activations; see bench_step.py for what a step is.", "",
           f"kernel time per step: **{total / steps:.2f} ms** in **{sum(r[1] for r in rows) / steps:.0f} launches**", "",
           "## by group", "", "| group | ms / step | launches / step | % |", "|---|---|---|---|"]
    acc = {}
    for name, calls, avg, tot in rows:
        a = acc.setdefault(group_of(name), [0.0, 0])
        a[0] += tot
        a[1] += calls
    for g, (tot, calls) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        out.append(f"| {g} | {tot / steps:.2f} | {calls / steps:.0f} | {100 * tot / total:.1f} |")
    out += ["", "## kernels (top 60 by time)", "", "| kernel | launches / step | avg µs | ms / step | % |", "|---|---|---|---|---|"]
    for name, calls, avg, tot in sorted(rows, key=lambda r: -r[3])[:60]:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        out.append(f"| `{short[:110]}` | {calls / steps:.1f} | {avg:.1f} | {tot / steps:.3f} | {100 * tot / total:.1f} |")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"{tag}_step_kernels.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
