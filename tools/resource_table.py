#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    hipcc <flags of richsem_amd/_build.py> -Rpass-analysis=kernel-resource-usage -o /dev/null csrc/msda_api.hip 2> usage.txt
    python tools/resource_table.py usage.txt > profiles/r02_resources.md
"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main(path):
    rows, cur = [], None
    for line in open(path):
        m = re.search(r"remark: (.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B (static) | occupancy (waves/SIMD) |")
    print("|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
        n = re.sub(r"\(.*$", "", n).replace("msda::", "")
        print(f"| `{n[:110]}` | {r.get('VGPRs', '?')} | {r.get('AGPRs', '?')} | {r.get('SGPRs', '?')} | {r.get('ScratchSize [bytes/lane]', '?')} "
              f"| {r.get('LDS Size [bytes/block]', '?')} | {r.get('Occupancy [waves/SIMD]', '?')} |")


if __name__ == "__main__":
    main(sys.argv[1])
