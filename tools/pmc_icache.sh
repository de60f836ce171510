cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | tr '\n' ' '
echo
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d /tmp/ic -o ic -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --frames-in-flight 1 > /tmp/ic.log 2>&1
echo rc=$?
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/ic -name "*_results.db" | head -1) --kernel render_frame 2>&1 | tail -12
