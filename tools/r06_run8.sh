mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06
CF_LIB_PATH=$PWD/tools/ab/skew2.so python tools/parity_quick.py 32,32 2>&1 | tail -2
(for S in 1024 128 2048 3072; do bash tools/ab_libs.sh "32 32 $S" tools/ab/skew0.so tools/ab/skew1.so tools/ab/skew2.so tools/ab/skew3.so; done) > $O/skew_ab.txt 2>&1; cat $O/skew_ab.txt
