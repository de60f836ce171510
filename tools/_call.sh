mkdir -p gpurun_out/r4c
tools/ab_time.sh gpurun_out/r4c/times.txt; cat gpurun_out/r4c/times.txt
for v in aofar udiv both; do echo $v; RAYMARCH_LIB=libraymarch_hip_ab_$v.so timeout 900 python -m pytest tests/test_gpu_device_contract.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_dark_pairs.py -x -q -m gpu 2>&1 | tail -2; done > gpurun_out/r4c/parity.txt 2>&1; cat gpurun_out/r4c/parity.txt
RAYMARCH_LIB=libraymarch_hip_ab_stats.so timeout 900 python tools/wave_stats.py > gpurun_out/r4c/stats_aofar.txt 2>&1; head -30 gpurun_out/r4c/stats_aofar.txt
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "WRITE_SIZE" "FETCH_SIZE"; do
  rm -rf /tmp/pq; RAYMARCH_SPLIT=1 timeout 300 rocprofv3 --pmc $SET -d /tmp/pq -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 --frames-in-flight 1 > /tmp/pq.log 2>&1
  for k in march_frame light_frame; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pq -name "*_results.db" | head -1) --kernel $k 2>&1 | grep -v "^#"; done
done > $GRAFT_REPO_ROOT/gpurun_out/r4c/pmc_split.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r4c/pmc_split.txt
