mkdir -p gpurun_out/r4u
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in c3 c5; do
  rm -rf /tmp/kt_$w
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o kt -- python $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > $R/gpurun_out/r4u/kt_$w.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/kt_$w -name "*_results.db" | head -1) > $R/gpurun_out/r4u/kernel_stats_$w.txt 2>&1
  rm -rf /tmp/pm_$w
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pm_$w -o pmc -- python $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > $R/gpurun_out/r4u/pm_$w.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pm_$w -name "*_results.db" | head -1) --kernel render_frame > $R/gpurun_out/r4u/pmc_$w.txt 2>&1
done
head -8 $R/gpurun_out/r4u/kernel_stats_c3.txt $R/gpurun_out/r4u/kernel_stats_c5.txt; cat $R/gpurun_out/r4u/pmc_c3.txt $R/gpurun_out/r4u/pmc_c5.txt
