R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pd
  timeout 400 rocprofv3 --pmc $SET -d /tmp/pd -o pmc -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --frames-in-flight 1 --workload c5 > /tmp/pd.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pd -name "*_results.db" | head -1) --kernel render_frame 2>&1 | grep -v "^#\|kernel:"
done
grep -o '"kernel_ms": [0-9.]*' /tmp/pd.log | tail -1
