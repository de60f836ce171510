# scratch: the command of the last gpurun call of the round (kept for reference)
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3
