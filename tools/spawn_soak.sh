#!/bin/bash
# N consecutive runs of the self-spawned multi-rank entry (bench.py --gpus 1 --spawn, CF_BENCH_TP=8 CF_BENCH_FORCE_DIST=1: the
# rendezvous, RCCL init, the three legs, teardown) -- every rc, and the stderr tail of every failure, go to $2.
N=${1:-50}; LOG=${2:-gpurun_out/spawn_soak.log}
mkdir -p "$(dirname "$LOG")"; : > "$LOG"
fail=0
for i in $(seq 1 "$N"); do
  t0=$(date +%s.%N)
  CF_BENCH_FORCE_DIST=1 CF_BENCH_TP=8 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --gpus 1 --spawn --steps 3 --warmup 1 \
      --no-cpu-baseline --no-configs > /tmp/soak_out.txt 2> /tmp/soak_err.txt
  rc=$?
  t1=$(date +%s.%N)
  ok=$(tail -1 /tmp/soak_out.txt | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tp_parity']['ok'], r['oneshot']['status'], r['inkernel_publish']['status'])" 2>/dev/null)
  printf "run %02d rc %d %ss legs: %s\n" "$i" "$rc" "$(python -c "print(round($t1 - $t0, 1))")" "$ok" >> "$LOG"
  if [ "$rc" != 0 ]; then fail=$((fail+1)); echo "---- stderr tail of run $i" >> "$LOG"; grep -v "^frame #" /tmp/soak_err.txt | tail -150 >> "$LOG"; fi
done
echo "runs $N failures $fail" >> "$LOG"
tail -1 "$LOG"
