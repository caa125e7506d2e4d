for v in none skip main none skip; do
  out=$(KLAB_DIAG_WGRAD=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "KLAB_DIAG_WGRAD=$v $out"
done
