mkdir -p gpurun_out/r4s
python -m pytest tests/test_gpu_quality.py -q -m gpu 2>&1 | tail -2
bash tools/final_profile.sh 2>&1 | tail -8
