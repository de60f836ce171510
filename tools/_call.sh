mkdir -p gpurun_out/r4s
export RAYMARCH_SKIP_LINT=1
(for so in sn5 sw5 sw4 sn5 sw5; do echo -n "$so: "; RAYMARCH_LIB=libraymarch_hip_ab_$so.so python tools/sdf_bench.py 2>&1 | tail -1; done
RAYMARCH_LIB=libraymarch_hip_ab_sw5.so python -m pytest tests/test_gpu_quality.py -q -m gpu 2>&1 | tail -2
) > gpurun_out/r4s/sdfwave.txt 2>&1
cat gpurun_out/r4s/sdfwave.txt
