#!/bin/bash
# bench every A/B variant present (see tools/ab_build.py) + the product library
cd ${GRAFT_REPO_ROOT:-/root/repo}
FIF=${FIF:-1}
for so in libraymarch_hip.so $(cd raymarchcl_amd && ls libraymarch_hip_ab_*.so 2>/dev/null); do
  printf "%-40s " $so
  RAYMARCH_LIB=$so python bench.py --steps 30 --warmup 4 --no-cpu-baseline --frames-in-flight $FIF 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
