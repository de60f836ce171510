mkdir -p gpurun_out/r4m
export RAYMARCH_SKIP_LINT=1
(for i in 1 2; do bash tools/ab_time.sh; done; FIF=3 bash tools/ab_time.sh
RAYMARCH_LIB=libraymarch_hip_ab_f8.so python -m pytest tests/test_gpu_configs.py tests/test_gpu_device_contract.py -q -m gpu -k "c2 or c4" 2>&1 | tail -1) > gpurun_out/r4m/f8.txt 2>&1
cat gpurun_out/r4m/f8.txt
