mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
python tools/parity_quick.py 32,8 2>&1 | tail -2 > $O/lpre_parity.txt; cat $O/lpre_parity.txt
(for S in 8192 4096 2048; do bash tools/ab_libs.sh "32 8 $S" tools/ab/lpre0.so tools/ab/lpre1.so; done) > $O/lpre_ab.txt 2>&1; cat $O/lpre_ab.txt
CF_TL_LAYERS=32 CF_TL_ACCT=1 CF_TL_GRAPH=1 timeout 300 python tools/fused_timeline.py 8192 0 gqa > $O/acct32_gqa_lpre.txt 2>&1; grep -A4 "^accounting\|boundary" $O/acct32_gqa_lpre.txt | head -30
