set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "growing or stale or page_table_longer or reaches_the_straight or plain_vs_reference or fused_kernel_ragged or other_geometries or fused_kernel_paged or gqa" 2>&1 | tail -15 > gpurun_out/r03a/tests.log
cat gpurun_out/r03a/tests.log
for s in 4096 1024 2048 4100 512; do
  timeout 600 python tools/ab_bench.py clusterfusion_amd/libclusterfusion_hip.so clusterfusion_amd/libexp_r2.so 3 $s 2>&1 | tail -2 | tee -a gpurun_out/r03a/ab.log
done
