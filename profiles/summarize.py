#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs collected by tools/collect_profiles.sh into the committed summary profiles/<tag>_summary.md.

    python profiles/summarize.py gpurun_out/r01 r01
"""
import collections
import csv
import glob
import json
import os
import sys


def kernel_stats(d, sub="trace"):
    f = glob.glob(os.path.join(d, sub, "*", "*_kernel_stats.csv"))
    rows = list(csv.DictReader(open(f[0]))) if f else []
    return [(r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]))
            for r in rows]


def per_grid(d):
    """Average duration per (kernel, grid size): separates the E and Dd launches of the direct kernels."""
    f = glob.glob(os.path.join(d, "trace", "*", "*_kernel_trace.csv"))
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            name = r["Kernel_Name"].split("(")[0]
            grid = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
            acc[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc


def pmc(d, sub, counter):
    f = glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv"))
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter:
                acc[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


OURS = ("msda::", "(anonymous namespace)::", "tiled_", "rps_", "fwd_direct", "bwd_", "ffn_fwd", "cls_", "conv_fwd", "conv_wgrad", "conv_pack",
        "prep_", "mask_rows", "topk_rows", "roi_align", "matcher_cost", "attnpool", "dn_")


def ours(name):
    """kernels of this library (the bench's comparison rows also launch PyTorch / MIOpen / hipBLASLt kernels: summarised as one line)"""
    return any(t in name for t in OURS) and "at::native" not in name and "ck::" not in name


def main():
    d, tag = sys.argv[1], sys.argv[2]
    if not os.path.isfile(os.path.join(d, "bench.json")):      # (never overwrite the committed summaries with an empty one)
        sys.exit(f"{d}: no bench.json -- was the collection run?")
    out = [f"# rocprofv3 summary {tag}", ""]
    try:
        line = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
        out += ["## bench.py line (un-profiled run)", "", "```json", json.dumps(line, indent=1), "```", ""]
    except Exception as e:  # pragma: no cover
        out += [f"(no bench line: {e})", ""]
    out += ["## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline`", "",
            "| kernel | calls | avg µs | total ms | % |", "|---|---|---|---|---|"]
    other_calls, other_ms, other_pct = 0, 0.0, 0.0
    for name, calls, avg, tot, pct in kernel_stats(d):
        if ours(name):
            out.append(f"| `{name[:90]}` | {calls} | {avg:.1f} | {tot:.2f} | {pct:.1f} |")
        else:
            other_calls, other_ms, other_pct = other_calls + calls, other_ms + tot, other_pct + pct
    out.append(f"| (PyTorch / MIOpen / hipBLASLt / RCCL / memset kernels of the comparison rows and the harness) | {other_calls} | | {other_ms:.2f} | {other_pct:.1f} |")
    init_rows = [r for r in kernel_stats(d, "trace_init") if ours(r[0])]
    if init_rows:
        out += ["", "### the headline distribution alone: `... bench.py --steps 10 --warmup 3 --loc init --no-bf16 --no-em --no-settle --no-cpu-baseline --no-full-step`", "",
                "(the trace above mixes the init / sigma4 / uniform distributions, bf16 storage and the mosaic shape in one row per kernel; `roofline.avg_launch_us` of the",
                "bench line is the routed backward at the init pattern: `rps_route_kernel` + `rps_tile_kernel` of THIS table, plus the 1-2 us of its event pair)", "",
                "| kernel | calls | avg µs | total ms | % |", "|---|---|---|---|---|"]
        for name, calls, avg, tot, pct in init_rows:
            out.append(f"| `{name[:90]}` | {calls} | {avg:.1f} | {tot:.2f} | {pct:.1f} |")
    out += ["", "### per (kernel, grid): E and Dd launches of the same kernel separated", "",
            "| kernel | grid (threads) | launches | avg µs |", "|---|---|---|---|"]
    for (name, grid), v in sorted(per_grid(d).items()):
        if not ours(name):
            continue
        out.append(f"| `{name[:70]}` | {'x'.join(g for g in grid if g)} | {len(v)} | {sum(v) / len(v):.1f} |")
    fetch, write = pmc(d, "pmc_fetch", "FETCH_SIZE"), pmc(d, "pmc_write", "WRITE_SIZE")
    out += ["", "## HBM traffic per launch (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)", "",
            "FETCH_SIZE / WRITE_SIZE are in KiB. Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950",
            "reports half the bytes of a wide coalesced stream, so the read side is doubled below; WRITE_SIZE is exact for",
            "16-B-per-lane stores and float atomics. Other access widths are uncalibrated, so treat the sum as an estimate.", "",
            "| kernel | grid | FETCH_SIZE KiB | WRITE_SIZE KiB | est. HBM MB = (2·fetch + write)·1024/1e6 |", "|---|---|---|---|---|"]
    for key in sorted(set(fetch) | set(write)):
        if not ours(key[0]):
            continue
        f = sum(fetch.get(key, [0])) / max(1, len(fetch.get(key, [0])))
        w = sum(write.get(key, [0])) / max(1, len(write.get(key, [0])))
        out.append(f"| `{key[0][:60]}` | {key[1]} | {f:.0f} | {w:.0f} | {(2 * f + w) * 1024 / 1e6:.1f} |")
    # machine-readable traffic table for bench.py's roofline.traffic (bytes per launch, corrected as above)
    traffic = {}
    for key in sorted(set(fetch) | set(write)):
        if not ours(key[0]):
            continue
        f = sum(fetch.get(key, [0])) / max(1, len(fetch.get(key, [0])))
        w = sum(write.get(key, [0])) / max(1, len(write.get(key, [0])))
        traffic[f"{key[0]}|{key[1]}"] = {"fetch_size_kib": f, "write_size_kib": w, "hbm_bytes_est": int((2 * f + w) * 1024)}
    # the mosaic shape Em (S = Lq = 34000), counted in passes of its own (tools/time_calls.py --calls Em): the persistent kernels' grid does not tell the shape
    fetch_em, write_em = pmc(d, "pmc_fetch_em", "FETCH_SIZE"), pmc(d, "pmc_write_em", "WRITE_SIZE")
    traffic_em = {}
    if fetch_em or write_em:
        out += ["", "### the mosaic shape Em (1280 x 1280: S = Lq = 34000), `tools/time_calls.py --calls Em --fwd 2 --bwd 4` under the same two counters", "",
                "| kernel | grid | FETCH_SIZE KiB | WRITE_SIZE KiB | est. HBM MB |", "|---|---|---|---|---|"]
        for key in sorted(set(fetch_em) | set(write_em)):
            if not ours(key[0]):
                continue
            f = sum(fetch_em.get(key, [0])) / max(1, len(fetch_em.get(key, [0])))
            w = sum(write_em.get(key, [0])) / max(1, len(write_em.get(key, [0])))
            out.append(f"| `{key[0][:60]}` | {key[1]} | {f:.0f} | {w:.0f} | {(2 * f + w) * 1024 / 1e6:.1f} |")
            traffic_em[f"{key[0]}|{key[1]}"] = {"fetch_size_kib": f, "write_size_kib": w, "hbm_bytes_est": int((2 * f + w) * 1024)}
    json.dump({"tag": tag, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; read side doubled "
               "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)", "kernels": traffic, "kernels_Em": traffic_em},
              open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_traffic.json"), "w"), indent=1)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_summary.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
