#!/bin/bash
# Round-end evidence run on the GPU box: `gpurun -- bash tools/collect_profiles.sh r02`.  Everything lands under
# gpurun_out/<tag>/; tools/rocprof_summary.py turns the .db files into the tables committed under profiles/.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for s in 512 1024 2048 4096 8192 16384 32768 65536; do timeout 200 python bench.py --no-cpu-baseline --no-configs --seq $s --steps 20 2>/dev/null | tail -1 >> $O/seq.jsonl; done
CF_TL_GRAPH=1 timeout 120 python tools/fused_timeline.py 4096 0 > $O/timeline_headline.txt 2>&1
CF_TL_GRAPH=1 timeout 120 python tools/fused_timeline.py 8192 0 gqa > $O/timeline_gqa.txt 2>&1
CF_TL_GRAPH=1 timeout 120 python tools/fused_timeline.py 4096 0 tp8 > $O/timeline_tp8.txt 2>&1
CF_TL_GRAPH=1 timeout 120 python tools/fused_timeline.py 1024 0 b2 > $O/timeline_b2.txt 2>&1
CF_TL_GRAPH=1 timeout 120 python tools/fused_timeline.py 1024 0 b4 > $O/timeline_b4.txt 2>&1
CF_TL_GRAPH=1 timeout 120 python tools/fused_timeline.py 1024 0 b8 > $O/timeline_b8.txt 2>&1
CF_TL_GRAPH=1 timeout 120 python tools/fused_timeline.py 1024 0 b16 > $O/timeline_b16.txt 2>&1
(timeout 400 python tools/batch_bench.py 1024 1,2,3,4,5,8,12,16,17,24,32; CF_FLAGS=32 timeout 300 python tools/batch_bench.py 1024 2,4,5,8,12,16,17,24,32; CF_NL=8 timeout 300 python tools/batch_bench.py 1024 33,48,64,96,128; timeout 300 python tools/batch_bench.py 4096 2,4,8,16,32; CF_FLAGS=32 timeout 300 python tools/batch_bench.py 4096 2,4,8,16,32) > $O/batch.jsonl 2>/dev/null
timeout 300 python tools/decode_bench.py 4000 64 > $O/decode_model.jsonl 2>/dev/null; timeout 300 python tools/decode_bench.py 1024 64 >> $O/decode_model.jsonl 2>/dev/null; timeout 300 python tools/decode_bench.py 8000 32 llama3 >> $O/decode_model.jsonl 2>/dev/null
(timeout 200 python tools/mla_bench.py; timeout 200 python tools/mla_bench.py) 2>/dev/null | grep '^{' > $O/mla.jsonl
timeout 200 python tools/mla_timeline.py 2>/dev/null | grep -v amdgpu.ids > $O/mla_timeline.txt
cd /tmp && export TMPDIR=/tmp
# (the persistent kernels choose their length arm on the device: ONE kernel name serves every S, so the headline trace holds the
#  headline workload only; the other configs' kernels get their own trace)
mkdir -p $O/kt && timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-configs > $O/kt/log.txt 2>&1
mkdir -p $O/ktc && timeout 900 rocprofv3 --kernel-trace --stats -d $O/ktc -o ktc -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/ktc/log.txt 2>&1
mkdir -p $O/pf && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o pf -- python $R/bench.py --steps 3 --warmup 1 --layers 8 --no-graph --no-cpu-baseline --no-configs > $O/pf/log.txt 2>&1
mkdir -p $O/pw && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o pw -- python $R/bench.py --steps 3 --warmup 1 --layers 8 --no-graph --no-cpu-baseline --no-configs > $O/pw/log.txt 2>&1
for bs in 64 128; do mkdir -p $O/ktb$bs && CF_NL=8 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktb$bs -o ktb -- python $R/tools/batch_bench.py 1024 $bs > $O/ktb$bs/log.txt 2>&1; done
cd $R
for d in kt ktc pf pw ktb64 ktb128; do python tools/rocprof_summary.py $O/$d --filter cf > $O/${d}_summary.md 2>&1; done
find $O -name "*.db" -size +20M -delete
ls -la $O
