#!/bin/bash
# instruction-mix and cache counters of the frame kernel for the library named by RAYMARCH_LIB
# (two rocprofv3 --pmc passes, counters only); prints the summary
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pq1 /tmp/pq2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU -d /tmp/pq1 -o pmc -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --frames-in-flight 1 > /tmp/pq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pq2 -o pmc -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --frames-in-flight 1 > /tmp/pq2.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pq1 -name "*_results.db" | head -1) $(find /tmp/pq2 -name "*_results.db" | head -1) --kernel render_frame 2>&1 | grep -v "^#"
