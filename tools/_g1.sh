cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05f/gputests.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r05f/gputests.log | tail -3
python bench.py > gpurun_out/r05f/bench_line.json 2> gpurun_out/r05f/bench.err; tail -c 600 gpurun_out/r05f/bench_line.json
python tools/run_configs.py > gpurun_out/r05f/configs.json 2> gpurun_out/r05f/configs.err; echo configs rc=$?
python tools/perf_ops.py --n 1e8 --ops setop,sort,unique,encode,nthash,tax > gpurun_out/r05f/perf_ops.json 2> gpurun_out/r05f/perf_ops.err; echo perf rc=$?
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
