#!/bin/bash
# Run ON THE GPU BOX: `unikmer count -k 31 -K -s` on a synthetic 100 Mbp FASTA (config 2 through the CLI);
# prints the device-pipeline line of --verbose (upload+encode / sort+unique / download) for a few chunk sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$(mktemp -d)
python - "$T/s.fa" <<'PY'
import sys, numpy as np
rng = np.random.default_rng(1)
with open(sys.argv[1], "wb") as fh:
    for r in range(100):
        fh.write(b">r%d\n" % r)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 1_000_000)]
        fh.write(seq.tobytes()); fh.write(b"\n")
PY
for mb in ${CHUNKS:-8 64 1024}; do
  for rep in 1 2; do
    t0=$(date +%s.%N)
    UNIKMER_CHUNK_MB=$mb $R/unikmer_amd/bin/unikmer count -k 31 -K -s -C --verbose "$T/s.fa" -o "$T/o" > "$T/log" 2>&1 || cat "$T/log"
    t1=$(date +%s.%N)
    grep -E "device pipeline|saved" "$T/log" | sed "s/^/chunk=${mb}MB rep$rep: /"
    echo "chunk=${mb}MB rep$rep: whole command $(python -c "print('%.2f' % ($t1 - $t0))") s (FASTA parse + device + .unik write)"
  done
done
rm -rf "$T"
