#!/bin/bash
# Same-box A/B of every switch added in round 3 (bench.py default workload, 30 steps, interleaved, 2 rounds each): one file for profiles/.
#   gpurun --timeout 1100 -- 'bash tools/r03_ab_all.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_ab_switches.txt
cd $ROOT
: > $OUT
ab() { echo "## $3" >> $OUT; bash tools/ab_bench.sh "$1" "$2" 2 >> $OUT 2>&1; tail -n 4 $OUT; }
ab "KLAB_T5_ATTN_FUSED=0" "KLAB_T5_ATTN_FUSED=1" "T5 attention sub-layer: rms-norm + projection + attention in one launch (B = default)"
ab "KLAB_SWIN_FUSED_EMBED=0" "KLAB_SWIN_FUSED_EMBED=1" "Swin patch embedding: conv + LayerNorm from the pixels in one launch"
ab "KLAB_SWIN_FUSED_LIN_LN=0" "KLAB_SWIN_FUSED_LIN_LN=1" "Swin C = 256 stage: Linear + LayerNorm + residual in one launch"
ab "KLAB_GEMM_P8=0" "KLAB_GEMM_P8=1" "256 x 256 tile GEMM (mm8p) inside its envelope (LM-head dgrad) against the 128-wide kernels"
ab "KLAB_GEMM_TILE_MIN=240" "KLAB_GEMM_TILE_MIN=224" "tile threshold: 64 x 64 against 128 x 64 tiles for the encoder's M = 3712"
ab "KLAB_DIAG_WGRAD=main" "KLAB_DIAG_WGRAD=0" "grouped weight gradients on the main stream against the side stream"
ab "KLAB_DIAG_WGRAD=skip" "KLAB_DIAG_WGRAD=0" "grouped weight gradients skipped altogether (diagnostic: wrong gradients) against the side stream"
ab "KLAB_SIDE_CUS=64" "KLAB_SIDE_CUS=0" "side stream confined to 64 CUs by a CU mask against the unmasked side stream"
ab "KLAB_WGRAD_P8=1" "KLAB_WGRAD_P8=0" "grouped weight gradients on 256 x 256 tiles against 128 x 128"
ab "KLAB_DDP_WIRE_DTYPE=bf16" "KLAB_DDP_WIRE_DTYPE=fp32" "(N = 1: no collective runs; the knob must cost nothing)"
