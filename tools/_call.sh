mkdir -p gpurun_out/r4o
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py > gpurun_out/r4o/bench_line.json 2> gpurun_out/r4o/bench.err; tail -c 1500 gpurun_out/r4o/bench_line.json
