#!/bin/bash
# One gpurun call: the GPU suite, a bench line, and (optionally) a kernel + HIP-API trace of a few steps.
#   gpurun --timeout 1100 -- 'bash tools/gpu_round.sh <tag> [suite|nosuite] [trace|notrace] [pytest -k expr]'
# A step that dies on a signal or a timeout stops the script (no further GPU step in the same call).
TAG=${1:-x}; SUITE=${2:-suite}; TRACE=${3:-notrace}; KEXPR=${4:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
if [ "$SUITE" = "suite" ]; then
  if [ -n "$KEXPR" ]; then
    timeout -k 10 900 python -m pytest tests -m gpu -q -k "$KEXPR" -s > $OUT/${TAG}_suite.log 2>&1; rc=$?
  else
    timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_suite.log 2>&1; rc=$?
  fi
  tail -n 25 $OUT/${TAG}_suite.log
  echo "[gpu_round] pytest rc=$rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
fi
timeout -k 10 400 python bench.py --steps 30 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; rc=$?
echo "[gpu_round] bench rc=$rc"; cat $OUT/${TAG}_bench.json | cut -c1-400
if [ $rc -ne 0 ]; then tail -n 20 $OUT/${TAG}_bench.err; exit $rc; fi
if [ "$TRACE" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err; rc=$?
  echo "[gpu_round] trace rc=$rc"
  ls -la $OUT/${TAG}_trace/* | head
fi
