mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/gputests_final.txt; cat $O/gputests_final.txt
timeout 600 python tools/soak.py 300 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
timeout 900 python tools/fuzz_extended.py 200 30000 2>/dev/null | grep '^{' > $O/fuzz.jsonl; cut -c1-200 $O/fuzz.jsonl
