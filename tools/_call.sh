mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_configs.py -q -m gpu -k "c3 or c5" 2>&1 | tail -3
