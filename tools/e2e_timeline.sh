#!/bin/bash
# Where does the wall time of `sigfish-amd dtw` go on a compressed file?  (run on the GPU box)  COPIES reads x 5 of the DNA
# fixture; per (K, streams, route): whole-process wall and the command's own timeline (--verbose 4)
COPIES=${COPIES:-80000}
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 /tmp/c.blow5 --copies $COPIES --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/tmp/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
for cfg in ${CFGS:-4096:2 8192:2 8192:3 16384:2}; do  # K:streams
  set -- ${cfg/:/ }
  for rep in 1 2; do
    T0=$(date +%s.%N)
    sigfish_amd/bin/sigfish-amd dtw --kmer-model /tmp/syn6.model -t 16 -K $1 -B 2G --verbose 4 --streams $2 $EXTRA tests/golden/data/nCoV-2019.reference.fasta /tmp/c.blow5 > /tmp/c.paf 2> /tmp/c.err
    T1=$(date +%s.%N)
    python -c "n=sum(1 for _ in open('/tmp/c.paf')); dt=$T1-$T0; print(f'K $1 streams $2 $EXTRA: {n} reads in {dt:.3f} s = {n/dt:.0f} reads/s')"
  done
  grep -v "Entries" /tmp/c.err | cut -c1-160
done
