mkdir -p gpurun_out/r4q
(for p in 1 0 1 0; do echo -n "c5 RAYMARCH_POW2=$p: "; RAYMARCH_POW2=$p python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; done
echo -n "c5 cpu contract pow2 1/0: "; for p in 1 0; do RAYMARCH_POW2=$p python bench.py --workload c5 --contract cpu --steps 6 --warmup 2 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '; done; echo
python -m pytest tests/test_gpu_configs.py tests/test_gpu_device_contract.py -q -m gpu -k "c5" 2>&1 | tail -2
) > gpurun_out/r4q/l4c5.txt 2>&1
cat gpurun_out/r4q/l4c5.txt
