#!/bin/bash
# parity (C2 at 16 spp + the pass-packed small frames + C5) and serial bench of every A/B variant
# present (tools/ab_build.py) and of the product library; PMC=1 adds the VMEM / VALU counts
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$(pwd)
for so in libraymarch_hip.so $(cd raymarchcl_amd && ls libraymarch_hip_ab_*.so 2>/dev/null); do
  printf "%-40s " $so
  RAYMARCH_LIB=$so python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c2 or pass_packed or c5" 2>&1 | tail -1 | tr '\n' ' '
  RAYMARCH_LIB=$so python bench.py --steps 30 --warmup 4 --no-cpu-baseline --frames-in-flight ${FIF:-1} 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '
  if [ "${PMC:-0}" = "1" ]; then
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pq && RAYMARCH_LIB=$so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/pq -o pmc -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 --frames-in-flight 1 > /tmp/pq.log 2>&1
     python $R/tools/pmc_summary.py $(find /tmp/pq -name "*_results.db" | head -1) --kernel render_frame 2>&1 | grep "SQ_INSTS" | awk '{printf "%s %.1fM  ", $1, $3/1e6}')
  fi
  echo
done
