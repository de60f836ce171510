mkdir -p gpurun_out/r4k
(python -m pytest tests/test_gpu_configs.py tests/test_gpu_device_contract.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for w in c5 c4 c3 c2; do echo -n "$w: "; python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; done) > gpurun_out/r4k/pack5.txt 2>&1
cat gpurun_out/r4k/pack5.txt
