mkdir -p gpurun_out/r4k
(python -m pytest tests/test_gpu_quality.py -x -q -m gpu 2>&1 | tail -2
python tools/sdf_bench.py 2>&1 | tail -1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | head -1
) > gpurun_out/r4k/sdf2.txt 2>&1
cat gpurun_out/r4k/sdf2.txt
