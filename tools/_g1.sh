cd $GRAFT_REPO_ROOT
for P in 0.1 0.02 0.005 0.001; do
 echo "== P=$P"
 python tools/srmerge_bench.py 1000 1e6 $P tax both 3 2>/dev/null | tail -3
done
