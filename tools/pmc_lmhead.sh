#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3n_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export KLAB_LMHEAD_AREG=$v
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a$v -o p -- python3 $ROOT/tools/gemm_bench.py "lmhead fwd" > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/b$v -o p -- python3 $ROOT/tools/gemm_bench.py "lmhead fwd" > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/c$v -o p -- python3 $ROOT/tools/gemm_bench.py "lmhead fwd" > /dev/null 2>&1 || exit 1
done
ls $OUT/*
