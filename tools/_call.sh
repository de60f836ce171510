mkdir -p gpurun_out/r4v
export RAYMARCH_SKIP_LINT=1
(for i in 1 2; do bash tools/ab_time.sh; done) > gpurun_out/r4v/ud.txt 2>&1
cat gpurun_out/r4v/ud.txt
