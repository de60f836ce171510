#!/bin/bash
# config 5 (1024^3, HBM bound): frame time and fabric read bytes of every A/B library present
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$(pwd)
for so in libraymarch_hip.so $(cd raymarchcl_amd && ls libraymarch_hip_ab_*.so 2>/dev/null); do
  printf "%-36s c5 " $so
  RAYMARCH_LIB=$so python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd && RAYMARCH_LIB=$so timeout 400 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/pd -o pmc -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --frames-in-flight 1 --workload c5 > /tmp/pd.log 2>&1
   python $R/tools/pmc_summary.py $(find /tmp/pd -name "*_results.db" | head -1) --kernel render_frame 2>&1 | grep "FETCH\|RDREQ" | awk '{printf "%s %.1fM  ", $1, $3/1e6}')
  echo
done
RAYMARCH_LIB=libraymarch_hip_ab_nt1.so python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c5" 2>&1 | tail -1
