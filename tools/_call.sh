mkdir -p gpurun_out/r4r
(for p in 1 0 1 0; do echo -n "c2 RAYMARCH_OCTANTS... layout 5 vs 2 via RAYMARCH_L5=$p: "; RAYMARCH_NO_L5=$((1-p)) python bench.py --steps 30 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | head -3 | tr '\n' ' '; echo; done
python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*' | head -2 | tr '\n' ' '; echo
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3
) > gpurun_out/r4r/l5.txt 2>&1
cat gpurun_out/r4r/l5.txt
