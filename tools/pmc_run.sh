#!/bin/bash
# Collect PMC counters for bench.py in separate rocprofv3 passes (counters only:
# no --kernel-trace/--stats mixed in, as the GPU pool requires).
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" \
  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $SET"
done
ls -R $OUT | head -40
