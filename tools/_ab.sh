for i in 1 2 3; do
for v in new old; do
  if [ $v != new ]; then export CF_LIB_PATH=$PWD/clusterfusion_amd/libexp_$v.so; else unset CF_LIB_PATH; fi
  python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', [round(c['us_per_call'],2) for c in r['configs']])"
done
done
