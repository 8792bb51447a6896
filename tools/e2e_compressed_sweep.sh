#!/bin/bash
# compressed-file end-to-end: contexts per device x batch size (run on the GPU box)
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 /tmp/c.blow5 --copies 160000 --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/tmp/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
for extra in "" "--host-parse"; do
for S in 2 3 4; do for K in 4096 8192 16384; do
  T0=$(date +%s.%N)
  sigfish_amd/bin/sigfish-amd dtw --kmer-model /tmp/syn6.model -t 16 -K $K -B 2G --verbose 0 --streams $S $extra tests/golden/data/nCoV-2019.reference.fasta /tmp/c.blow5 > /tmp/c.paf
  T1=$(date +%s.%N)
  python -c "n=sum(1 for _ in open('/tmp/c.paf')); dt=$T1-$T0; print(f'streams $S K $K $extra: {n} reads in {dt:.2f} s = {n/dt:.0f} reads/s')"
done; done; done
