#!/bin/bash
# parity (C2 at 16 spp + the pass-packed small frames) and serial bench of every A/B variant
# present (tools/ab_build.py) and of the product library
cd ${GRAFT_REPO_ROOT:-/root/repo}
for so in libraymarch_hip.so $(cd raymarchcl_amd && ls libraymarch_hip_ab_*.so 2>/dev/null); do
  printf "%-44s " $so
  RAYMARCH_LIB=$so python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c2 or pass_packed or c5" 2>&1 | tail -1 | tr '\n' ' '
  RAYMARCH_LIB=$so python bench.py --steps 30 --warmup 4 --no-cpu-baseline --frames-in-flight ${FIF:-1} 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '
  echo
done
