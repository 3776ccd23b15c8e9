for i in 1 2 3; do
for v in new old; do
  if [ $v != new ]; then export CF_LIB_PATH=$PWD/clusterfusion_amd/libexp_$v.so; else unset CF_LIB_PATH; fi
  for S in 4096 3072; do
  python bench.py --no-cpu-baseline --no-configs --seq $S --steps 100 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $S, r['ms_per_step']*1000/32, r['roofline']['frac'])"
  done
done
done
