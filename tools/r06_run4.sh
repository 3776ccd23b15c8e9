mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
python tools/parity_quick.py 32,8 2>&1 | tail -4 > $O/lsplit_parity.txt; cat $O/lsplit_parity.txt
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "gqa and not small_batch" 2>&1 | tail -4 >> $O/lsplit_parity.txt; tail -4 $O/lsplit_parity.txt
(for S in 8192 4096 2048; do bash tools/ab_libs.sh "32 8 $S" tools/ab/lsplit0.so tools/ab/lsplit1.so; done) > $O/lsplit_ab.txt 2>&1; cat $O/lsplit_ab.txt
CF_TL_LAYERS=32 CF_TL_ACCT=1 CF_TL_GRAPH=1 timeout 300 python tools/fused_timeline.py 8192 0 gqa > $O/acct32_gqa_lsplit.txt 2>&1; grep -A4 "^accounting\|boundary" $O/acct32_gqa_lsplit.txt | head -30
