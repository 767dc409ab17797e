#!/bin/bash
# Diagnostic builds of the library for tools/r04_ring_ablate.sh and tools/r04_wgrad_ablate.sh (results WRONG by construction), made in the
# container (hipcc cross-compiles gfx950; the libraries travel to the GPU box with the snapshot: build_ablate/ is git-ignored, not
# gpurun-ignored); the GPU-side scripts load them through RICHSEM_MSDA_LIB.  The product library is rebuilt at the end (the flag stamp of
# richsem_amd/_build.py would rebuild it anyway: a library built with other flags counts as stale).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ablate
build() { RICHSEM_HIPCC_EXTRA="$1" python -c "from richsem_amd import _build; _build.build(verbose=False, force=True)"; cp richsem_amd/lib/librichsem_msda.so "build_ablate/$2"; }
for v in 0 1 2 3; do build "-DCONV_RING_ABLATE=$v" lib$v.so; done                 # conv_ring_kernel: 1 = no weight requests, 2 = no activation requests
for v in 0 1 2 5 9 13 17; do build "-DWGRAD_RING_ABLATE=$v" wlib$v.so; done       # wgrad_block_ring: see csrc/conv_wgrad.hip
python -c "from richsem_amd import _build; _build.build(verbose=False, force=True)"
