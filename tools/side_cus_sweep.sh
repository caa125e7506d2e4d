for n in 0 64 128 192; do
  out=$(KLAB_SIDE_CUS=$n timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "KLAB_SIDE_CUS=$n $out"
done
for n in 0 128; do
  out=$(KLAB_SIDE_CUS=$n timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "KLAB_SIDE_CUS=$n $out"
done
