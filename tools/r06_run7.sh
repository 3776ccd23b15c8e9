mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
timeout 400 python tools/soak.py 150 > $O/soak.txt 2>&1; tail -6 $O/soak.txt
timeout 700 python tools/fuzz_extended.py 150 20000 2>/dev/null | grep '^{' > $O/fuzz.jsonl; cut -c1-300 $O/fuzz.jsonl
