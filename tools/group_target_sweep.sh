#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for T in 256 384 512 768 1024; do
  out=$(KLAB_GEMM_GROUP_TARGET=$T timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
  echo "target $T: $out"
done; done
