mkdir -p gpurun_out/r4q
(python -m pytest tests/ -q -m gpu -x 2>&1 | tail -4
for w in c3 c2 c5; do echo -n "$w: "; python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"frac": [0-9.]*' | head -3 | tr '\n' ' '; echo; done
python -c "
import bench, json
" ) > gpurun_out/r4q/final_tests.txt 2>&1
cat gpurun_out/r4q/final_tests.txt
