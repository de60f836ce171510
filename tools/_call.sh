mkdir -p gpurun_out/r4v
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3
bash tools/final_profile.sh 2>&1 | tail -6
