# scratch: the command of the last gpurun call of the round (kept for reference)
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 | cut -c1-400
