#!/bin/bash
# A/B timing of two builds of the library inside ONE gpurun call (run-to-run drift between boxes is several per cent):
#   tools/ab_probe.sh <libA.so> <libB.so> <rounds> <kernel_probe args...>
# The two files are copied over richsem_amd/lib/librichsem_msda.so in turn; the last one stays in place.
set -e
A=$1; B=$2; R=$3; shift 3
LIB=richsem_amd/lib/librichsem_msda.so
for i in $(seq $R); do
  cp $A $LIB; echo -n "A: "; python tools/kernel_probe.py "$@" 2>/dev/null | tail -1
  cp $B $LIB; echo -n "B: "; python tools/kernel_probe.py "$@" 2>/dev/null | tail -1
done
