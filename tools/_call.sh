for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; done
python -m pytest tests/test_gpu_device_contract.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
for w in c3 c5 c4; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' '; echo " $w"; done
