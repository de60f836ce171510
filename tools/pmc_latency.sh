#!/bin/bash
# memory-latency view of the frame kernel (separate rocprofv3 --pmc passes, counters only):
# average L1->L2 read latency, how many L2 read misses go on to DRAM (vs the Infinity Cache),
# address-translation traffic, vector-memory instructions in flight.  -> stdout
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum" \
  "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
  "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1)); rm -rf /tmp/pl$i
  timeout 300 rocprofv3 --pmc $SET -d /tmp/pl$i -o pmc -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 --frames-in-flight 1 "$@" > /tmp/pl$i.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pl$i -name "*_results.db" | head -1) --kernel render_frame 2>&1 | grep -v "^#\|kernel:"
done
