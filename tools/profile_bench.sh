#!/bin/bash
# Profiles of `bench.py` (default workload) on the GPU box, per MI355X_MICROARCH.md's rocprofv3 recipe: one kernel-trace pass
# and separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing beside counters).  Run through gpurun:
#   gpurun -- 'bash tools/profile_bench.sh r02'
# Raw output -> gpurun_out/prof_<tag>/ ; summarise afterwards with tools/summarize_profiles.py into profiles/.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 3 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py $ARGS > $OUT/write_bench.json 2> $OUT/write.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o m -- python3 $ROOT/bench.py $ARGS > $OUT/mfma_bench.json 2> $OUT/mfma.err
ls -R $OUT | head -40
