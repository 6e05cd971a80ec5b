mkdir -p gpurun_out
for t in base plb8 plb4; do
  if [ "$t" = base ]; then L=unikmer_amd/libunikmer_hip.so; else L=unikmer_amd/libukm_exp_$t.so; fi
  echo "== $t"
  for P in 0.9 0.5 0.2; do UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L UKM_PUNION_DEBUG=1 python tools/srmerge_bench.py 1000 1e6 $P tax merge 3 2>&1 | grep -v amdgpu.ids | grep "place  \|merge_k" | tail -2; done
done > gpurun_out/pl.txt 2>&1
