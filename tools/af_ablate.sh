cd /tmp && export TMPDIR=/tmp
for a in 0 7; do
  KLAB_AF_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_abl$a -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "ablate=$a"; grep -E "fused_fwd" $GRAFT_REPO_ROOT/gpurun_out/r3_abl$a/t_kernel_stats.csv | cut -d, -f1,4 | cut -c1-90
done
