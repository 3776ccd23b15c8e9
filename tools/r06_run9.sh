mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
CF_DEBUG_FLAGS=8192 python tools/parity_quick.py 4,4 2>&1 | tail -14
for r in 1 2 3; do for S in 4096 1024 8192; do for f in 0 8192; do echo -n "flags $f: "; CF_DEBUG_FLAGS=$f python tools/shard_ab.py 4 4 $S 2>&1 | tail -1; done; done; done | tee $O/s32_ab.txt
CF_TL_LAYERS=32 CF_TL_ACCT=1 CF_TL_GRAPH=1 timeout 300 python tools/fused_timeline.py 4096 8192 tp8 2>/dev/null | grep -v Warning > $O/tl_tp8_s32.txt; head -12 $O/tl_tp8_s32.txt; grep accounting $O/tl_tp8_s32.txt
