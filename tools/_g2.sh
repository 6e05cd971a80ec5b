cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r05 > /dev/null 2>&1
mkdir -p gpurun_out/r05f
bash tools/pmc_cmd.sh sort ls_sort -- python $GRAFT_REPO_ROOT/tools/perf_ops.py --n 1e8 --ops sort --reps 2 > gpurun_out/r05f/sort_local_pmc.txt 2>&1
bash tools/trace_call.sh c2 python $GRAFT_REPO_ROOT/tools/run_configs.py --configs 2 > gpurun_out/r05f/config2_trace.txt 2>&1
bash tools/pmc_valu.sh setop_tile -- python $GRAFT_REPO_ROOT/tools/perf_ops.py --n 1e8 --ops tax --reps 2 > gpurun_out/r05f/setop_tax_pmc.txt 2>&1
ls gpurun_out/prof_r05 | head
