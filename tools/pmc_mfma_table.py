#!/usr/bin/env python3
"""Matrix-pipe / LDS-pipe busy shares per (kernel, grid) from a tools/pmc_kernel.sh output directory:
    python tools/pmc_mfma_table.py gpurun_out/pmc_<tag> [kernel substring]
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); LDS busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256 CUs)."""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sub in r["Kernel_Name"]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[(name[:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | grid (threads) | launches | MFMA busy | LDS busy | LDS bank-conflict share of LDS cycles |")
print("|---|---|---|---|---|---|")
for (name, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = lambda k: sum(d[k]) / max(len(d[k]), 1) if k in d else 0.0
    gui = m("GRBM_GUI_ACTIVE") / 8.0
    if gui <= 0:
        continue
    lds = m("SQ_LDS_IDX_ACTIVE")
    print(f"| `{name}` | {grid} | {len(d['GRBM_GUI_ACTIVE'])} | {m('SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024):.3f} | {lds / (gui * 256):.3f} | "
          f"{(m('SQ_LDS_BANK_CONFLICT') / lds if lds else 0):.2f} |")
