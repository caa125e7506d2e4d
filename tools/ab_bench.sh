#!/bin/bash
# same-box A/B of bench.py: `bash tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [rounds] [extra bench args]` -- interleaved runs, ms/step of each
A="$1"; B="$2"; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq 1 $R); do
  for side in A B; do
    if [ $side = A ]; then E="$A"; else E="$B"; fi
    out=$(env $E timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
    echo "$side [$E] $out"
  done
done
