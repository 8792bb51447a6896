#!/bin/bash
# End-to-end throughput of the command line on a replicated fixture (run on the GPU box):  [BLOW5_FLAGS=--compress] [KS="4096 16384"] tools/e2e_cli.sh [copies] [threads]
set -e
COPIES=${1:-4000}; THREADS=${2:-16}
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 /tmp/big.blow5 --copies $COPIES $BLOW5_FLAGS
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/tmp/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
for K in ${KS:-4096 16384}; do
  T0=$(date +%s.%N)
  sigfish_amd/bin/sigfish-amd dtw --kmer-model /tmp/syn6.model -t $THREADS -K $K -B 2G --verbose 3 \
      tests/golden/data/nCoV-2019.reference.fasta /tmp/big.blow5 > /tmp/big.paf
  T1=$(date +%s.%N)
  python -c "n=sum(1 for _ in open('/tmp/big.paf')); dt=$T1-$T0; print(f'K=$K threads=$THREADS: {n} reads in {dt:.2f} s = {n/dt:.0f} reads/s end to end')"
done
head -5 /tmp/big.paf | cut -f1-12 | diff - <(cut -f1-12 tests/golden/cases/dna_default.out | sed 's/\t/_0\t/') && echo "first five rows equal the fixture rows"
