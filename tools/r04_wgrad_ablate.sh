#!/bin/bash
# round 4: what bounds the ring form of the weight-gradient kernel?  Diagnostic builds (results wrong), built in the container:
# WGRAD_RING_ABLATE 1 = no requests inside the loop, 2 = no products (requests only)
R=$GRAFT_REPO_ROOT
for v in 0 1 2 5 9 17 13; do
  export RICHSEM_MSDA_LIB=$R/build_ablate/wlib$v.so      # (loaded from where it was built: the product library is never overwritten)
  echo "== WGRAD_RING_ABLATE=$v"
  timeout -k 10 300 python3 $R/tools/r04_wgrad_ring.py 2>&1 | grep -v amdgpu | grep -E "44646|block"
done
