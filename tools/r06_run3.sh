mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "gqa_small_batch or fuzz_gqa_small or batch_sizes_mfma or mid_batch or fuzz_paged_batch" 2>&1 | tail -15 > $O/gputests_3.txt
cat $O/gputests_3.txt
tools/ubench/launch_gap2 > $O/launch_gap2_b.txt 2>&1
for cfg in "4096 0" "8192 0 gqa" "1024 0 b2" "1024 0 b4" "4096 0 tp8"; do
  tag=$(echo $cfg | tr ' ' '_')
  CF_TL_LAYERS=32 CF_TL_ACCT=1 CF_TL_GRAPH=1 timeout 300 python tools/fused_timeline.py $cfg > $O/acct32_$tag.txt 2>&1
done
grep -h "accounting" $O/acct32_*.txt; head -3 $O/launch_gap2_b.txt
