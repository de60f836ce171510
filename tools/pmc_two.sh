#!/bin/bash
# SQ + cache counter passes for bench.py (counters only).  usage: tools/pmc_two.sh <outdir> [bench args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/p1 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/p3 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/p3.log 2>&1
echo done
