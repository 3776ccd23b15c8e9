mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
(python tools/gqa_batch_bench.py 1024; python tools/gqa_batch_bench.py 4096; python tools/gqa_batch_bench.py 8192) 2>/dev/null | grep '^{' > $O/gqa_batch.jsonl; cat $O/gqa_batch.jsonl
CF_TL_LAYERS=32 CF_TL_ACCT=1 CF_TL_GRAPH=1 timeout 300 python tools/fused_timeline.py 4096 0 tp8 2>/dev/null | grep -v Warning > $O/tl_tp8_default.txt
head -16 $O/tl_tp8_default.txt
