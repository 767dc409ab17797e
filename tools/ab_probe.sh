#!/bin/bash
# A/B timing of two builds of the library inside ONE gpurun call (run-to-run drift between boxes is several per cent):
#   tools/ab_probe.sh <libA.so> <libB.so> <rounds> <kernel_probe args...>
# Each build is loaded from where it lies through RICHSEM_MSDA_LIB (richsem_amd/_lib.py); the product library is never overwritten.
set -e
A=$1; B=$2; R=$3; shift 3
for i in $(seq $R); do
  echo -n "A: "; RICHSEM_MSDA_LIB=$A python tools/kernel_probe.py "$@" 2>/dev/null | tail -1
  echo -n "B: "; RICHSEM_MSDA_LIB=$B python tools/kernel_probe.py "$@" 2>/dev/null | tail -1
done
