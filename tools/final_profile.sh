#!/bin/bash
# Everything profiles/ holds for the current build, in one GPU call:
#   kernel trace (rocprofv3 --kernel-trace --stats), the PMC passes, the traffic json (with the
#   digest of the sources), the default bench line, the other workloads, the tile-partition
#   timing.  Output -> gpurun_out/final/
# (bench.py runs strictly serial frames under the profilers so that per-launch numbers are those
#  of one launch; the bench line itself uses the default frames in flight.)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 40 --warmup 5 --lean > $OUT/kt.log 2>&1
DB=$(find /tmp/kt -name "*_results.db" | head -1)
python $R/tools/rocprof_summary.py $DB > $OUT/kernel_stats.txt 2>&1
cd $R
tools/pmc_run.sh final/pmc --steps 8 --warmup 2 --lean > $OUT/pmc_run.log 2>&1
python tools/pmc_summary.py $OUT/pmc/p1/pmc_results.db $OUT/pmc/p2/pmc_results.db $OUT/pmc/p3/pmc_results.db \
  $OUT/pmc/p4/pmc_results.db $OUT/pmc/p5/pmc_results.db --kernel "render_frame_kernel<true, 7, false, 5, 3>" > $OUT/pmc.txt 2>&1
python tools/pmc_summary.py $OUT/pmc/p4/pmc_results.db $OUT/pmc/p5/pmc_results.db --kernel "render_frame_kernel<true, 7, false, 5, 3>" \
  --traffic-json $OUT/pmc_traffic.json > $OUT/traffic.log 2>&1
python - <<'PY'
import json, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench
p = os.path.join(R, "gpurun_out", "final", "pmc_traffic.json")
j = json.load(open(p))
j["source_digest"] = bench.source_digest()
json.dump(j, open(p, "w"), indent=1)
# the bench line below reads it from profiles/
json.dump(j, open(os.path.join(R, "profiles", "r06_pmc_traffic.json"), "w"), indent=1)
PY
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
for w in c1 c3 c4 c5; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1; done > $OUT/other_workloads.txt
python tools/part_timing.py --frames-in-flight 1 --reps 30 > $OUT/part_timing.txt 2>&1   # blocking shares: what `value` times
python tools/part_timing.py --frames-in-flight 2 --reps 30 > $OUT/part_timing_2inflight.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/scale_n1_as_driver.json 2>/dev/null  # the command line of the driver's SCALE run at N = 1
BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/rehearsal_ranks2.json 2>/dev/null
BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --backend library --steps 5 --warmup 2 > $OUT/rehearsal_library2.json 2>/dev/null
python tools/sdf_bench.py > $OUT/sdf_bench.txt 2>&1
rm -rf $OUT/pmc/p*/pmc_results.db
head -c 700 $OUT/bench_line.json; echo; tail -3 $OUT/part_timing.txt; head -6 $OUT/kernel_stats.txt
