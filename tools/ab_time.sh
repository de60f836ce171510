#!/bin/bash
# serial kernel time of every A/B variant present (tools/ab_build.py) and of the product library:
#   tools/ab_time.sh [outfile]      (FIF=n frames in flight, default 1; WL=workload, default c2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-/dev/stdout}
for so in libraymarch_hip.so $(cd raymarchcl_amd && ls libraymarch_hip_ab_*.so 2>/dev/null | grep -v stats); do
  line=$(RAYMARCH_LIB=$so timeout 600 python bench.py --workload ${WL:-c2} --steps ${STEPS:-30} --warmup 4 --no-cpu-baseline --frames-in-flight ${FIF:-1} 2>&1 | tail -1)
  printf "%-44s %s\n" $so "$(echo "$line" | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"ms_per_frame": [0-9.]*' | tr '\n' ' ')" >> $OUT
done
