R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for W in c2 c5; do
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RDREQ_32B_sum TCC_READ_sum" ; do
  rm -rf /tmp/pd
  timeout 400 rocprofv3 --pmc $SET -d /tmp/pd -o pmc -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --frames-in-flight 1 --workload $W > /tmp/pd.log 2>&1
  echo "$W: $SET"
  python $R/tools/pmc_summary.py $(find /tmp/pd -name "*_results.db" | head -1) --kernel render_frame 2>&1 | grep -v "^#\|kernel:"; tail -2 /tmp/pd.log | grep -i "error\|invalid" | head -2
done; done
