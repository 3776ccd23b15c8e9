mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
for cfg in "4096 0" "8192 0 gqa" "1024 0 b2" "1024 0 b4" "4096 0 tp8"; do
  tag=$(echo $cfg | tr ' ' '_')
  CF_TL_ACCT=1 CF_TL_GRAPH=1 timeout 200 python tools/fused_timeline.py $cfg > $O/acct_$tag.txt 2>&1
done
for S in 512 1024 2048 4096; do
  (timeout 400 python tools/batch_bench.py $S 16,17,20,24,26,28,30,32; CF_FLAGS=32 timeout 400 python tools/batch_bench.py $S 16,17,20,24,26,28,30,32) 2>/dev/null > $O/route_$S.jsonl
done
python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/gputests_2.txt
cat $O/gputests_2.txt; grep -h "accounting\|leaders\|others\|LAST\|NON-leader\|by XCD" $O/acct_*.txt
