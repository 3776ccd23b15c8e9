#!/bin/bash
# HBM read traffic (PMC FETCH_SIZE, its own pass per workload -- never combined with the trace domains gpurun refuses) of the
# persistent kernels OTHER than the headline's: `gpurun -- bash tools/pmc_other.sh r04`, then
# `python tools/pmc_other_md.py r04` writes profiles/<tag>_pmc_other.md (traffic vs algorithmic bytes per launch).
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG/pmc_other
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name, command...
    local name=$1; shift
    mkdir -p $O/$name
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$name -o p -- "$@" > $O/$name/log.txt 2>&1
    python $R/tools/rocprof_summary.py $O/$name --filter cf > $O/${name}.md 2>&1
    find $O/$name -name "*.db" -delete
}
run gqa_32_8_s8192   python $R/tools/shard_ab.py 32 8 8192
run tp8_4_4_s4096    python $R/tools/shard_ab.py 4 4 4096
run tp8_gqa_4_1_s8192 python $R/tools/shard_ab.py 4 1 8192
run plain_32_32_s1024 python $R/tools/shard_ab.py 32 32 1024
CF_NL=8 run batch2_s1024  python $R/tools/batch_bench.py 1024 2
CF_NL=8 run batch8_s1024  python $R/tools/batch_bench.py 1024 8
CF_NL=8 run batch16_s1024 python $R/tools/batch_bench.py 1024 16
CF_NL=8 run batch32_s1024 python $R/tools/batch_bench.py 1024 32
ls -la $O
