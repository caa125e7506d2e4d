#!/bin/bash
# round-3 record: the headline line, the secondary workloads, and the rocprofv3 passes behind roofline.traffic
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 300 python bench.py --steps 30 --warmup 5 > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err || exit 1
tail -c 600 $OUT/r03_bench_default.json; echo
for wl in cfg3 spanmask cfg5; do
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r03_bench_$wl.json 2> $OUT/r03_bench_$wl.err || exit 1
  python3 -c "import json; d=json.load(open('$OUT/r03_bench_$wl.json')); print('$wl', d['ms_per_step'], d['value'], d['config'].get('step_mfma_frac'))"
done
timeout -k 10 300 python bench.py --workload cfg5 --dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r03_bench_cfg5_fp8.json 2> $OUT/r03_bench_cfg5_fp8.err || exit 1
python3 -c "import json; d=json.load(open('$OUT/r03_bench_cfg5_fp8.json')); print('cfg5 fp8', d['ms_per_step'], d['value'])"
bash tools/profile_bench.sh r03 > $OUT/r03_profile.log 2>&1
tail -n 5 $OUT/r03_profile.log
