mkdir -p gpurun_out/r4k
export RAYMARCH_SKIP_LINT=1
R=$(pwd)
(for so in libraymarch_hip_ab_s3.so libraymarch_hip_ab_s5.so libraymarch_hip_ab_s6.so; do echo -n "$so: "; RAYMARCH_LIB=$so python tools/sdf_bench.py 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" \
  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  timeout 300 rocprofv3 --pmc $SET -d /tmp/pq$i -o pmc -- python $R/tools/sdf_bench.py > /tmp/pq$i.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pq$i -name "*_results.db" | head -1) --kernel render_frame 2>&1
done) > $R/gpurun_out/r4k/sdf_pmc3.txt 2>&1
cat $R/gpurun_out/r4k/sdf_pmc3.txt
