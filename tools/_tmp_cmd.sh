mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/full_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.txt 2>&1
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err
