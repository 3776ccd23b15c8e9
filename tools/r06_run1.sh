mkdir -p gpurun_out/r06; O=$PWD/gpurun_out/r06; R=$PWD
(tools/ubench/hbm_map stride; tools/ubench/hbm_map class 8192; tools/ubench/hbm_map classes 8192; tools/ubench/hbm_map class 2048; tools/ubench/hbm_map classes 2048; tools/ubench/hbm_map deal) > $O/hbm_map.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in "deal 0" "deal 1" "class 8192 0" "class 8192 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  mkdir -p $O/pmc_$tag
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_REQ TCC_TAG_STALL TCC_BUBBLE --kernel-trace -d $O/pmc_$tag -o p -- $R/tools/ubench/hbm_map one $cfg > $O/pmc_$tag/log.txt 2>&1
done
cd $R
tools/spawn_soak.sh 60 gpurun_out/r06/spawn_soak_d.log
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/gputests_1.txt
cat $O/gputests_1.txt; cat $O/hbm_map.txt; ls -la $O/pmc_*/
