mkdir -p gpurun_out/r4j
export RAYMARCH_SKIP_LINT=1
(WL=c5 STEPS=6 bash tools/ab_time.sh; WL=c3 STEPS=10 bash tools/ab_time.sh) > gpurun_out/r4j/c5_waves.txt 2>&1
cat gpurun_out/r4j/c5_waves.txt
