mkdir -p gpurun_out/r4w
(for g in 1 2 3 4 6 1 2; do echo -n "c5 RAYMARCH_XCD_GROUP=$g: "; RAYMARCH_XCD_GROUP=$g python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; done
for g in 1 2 4; do echo -n "c3 RAYMARCH_XCD_GROUP=$g: "; RAYMARCH_XCD_GROUP=$g python bench.py --workload c3 --steps 8 --warmup 2 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; done
for g in 1 2; do echo -n "c2 RAYMARCH_XCD_GROUP=$g: "; RAYMARCH_XCD_GROUP=$g python bench.py --steps 20 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | head -1; done
RAYMARCH_XCD_GROUP=3 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "c5 or c2 or pass_packed" 2>&1 | tail -1
) > gpurun_out/r4w/xg.txt 2>&1
cat gpurun_out/r4w/xg.txt
