mkdir -p gpurun_out/r4u
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf /tmp/pf; timeout 600 rocprofv3 --pmc $c -d /tmp/pf -o pmc -- python $R/bench.py --workload c5 --steps 4 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > /tmp/pf.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pf -name "*_results.db" | head -1) --kernel "render_frame_kernel<true, 7, false, 4, 2>"
done > $R/gpurun_out/r4u/pmc_c5_mem.txt 2>&1
cat $R/gpurun_out/r4u/pmc_c5_mem.txt
