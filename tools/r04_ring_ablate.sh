#!/bin/bash
# round 4: what bounds conv_ring_kernel?  Three diagnostic builds (results wrong): without the weight requests, without the activation
# requests, without both -- built HERE (no hipcc run on the GPU box is needed: the libraries travel), timed on the deep layers' shapes.
R=$GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  export RICHSEM_MSDA_LIB=$R/build_ablate/lib$v.so      # (loaded from where it was built: the product library is never overwritten)
  echo "== CONV_RING_ABLATE=$v"
  timeout -k 10 300 python3 $R/tools/time_conv.py --ring --reps 50 --only "l3 3x3 256,l3 1x1 1024-256,l4 3x3 512,l4 1x1 2048-512" 2>&1 | grep -v amdgpu
done
