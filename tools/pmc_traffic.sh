#!/bin/bash
# HBM-side traffic of the frame kernel for the CURRENT build: rocprofv3 --pmc FETCH_SIZE and
# WRITE_SIZE in separate passes (counters only), summarised into gpurun_out/pmc_traffic.json
# together with the digest of the sources (bench.py reports roofline.traffic only for a
# matching digest).  Copy the JSON to profiles/r06_pmc_traffic.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt1 /tmp/pt2
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pt1 -o pmc -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --frames-in-flight 1 > /tmp/pt1.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pt2 -o pmc -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --frames-in-flight 1 > /tmp/pt2.log 2>&1
cd $R
python tools/pmc_summary.py $(find /tmp/pt1 -name "*_results.db" | head -1) $(find /tmp/pt2 -name "*_results.db" | head -1) \
  --kernel render_frame --traffic-json $R/gpurun_out/pmc_traffic.json
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "pmc_traffic.json")
j = json.load(open(p))
j["source_digest"] = bench.source_digest()
json.dump(j, open(p, "w"), indent=1)
print(json.dumps(j))
PY
