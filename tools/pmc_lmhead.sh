#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and issue/stall counters of the LM-head logits kernels, standalone
# (tools/gemm_bench.py "lmhead fwd"):  gpurun -- 'bash tools/pmc_lmhead.sh'   ->  gpurun_out/pmc_lmhead/<variant>_<pass>/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lmhead
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # variant-name pass-name counters...
  local v=$1 n=$2; shift 2
  timeout -k 10 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${v}_$n -o p -- python3 $ROOT/tools/gemm_bench.py "lmhead fwd" > /dev/null 2>&1 || exit 1
}
for v in areg rot tiled; do
  unset KLAB_LMHEAD_AREG KLAB_LMHEAD_ROT
  [ $v = rot ] && export KLAB_LMHEAD_ROT=1
  [ $v = tiled ] && export KLAB_LMHEAD_AREG=0
  run $v fetch FETCH_SIZE
  run $v write WRITE_SIZE
  run $v sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run $v mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
done
ls $OUT
