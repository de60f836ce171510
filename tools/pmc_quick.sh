#!/bin/bash
# One rocprofv3 --pmc pass (instruction counts + lane cycles) of bench.py, for A/B builds:
#   RAYMARCH_LIB=libraymarch_hip_ab_x.so tools/pmc_quick.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY \
  -d $OUT/p1 -o pmc -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --frames-in-flight 1 "$@" > $OUT/p1.log 2>&1
cd $R
python tools/pmc_summary.py $OUT/p1/pmc_results.db --kernel "render_frame_kernel<true, 7, false, 2, 2>" | grep avg
rm -rf $OUT/p1
