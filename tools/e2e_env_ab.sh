#!/bin/bash
# whole-process A/B of environment settings for the command line on a compressed file in /dev/shm (run on the GPU box):
#   bash tools/e2e_env_ab.sh "VAR=value" ["VAR2=value" ...]      (COPIES=80000 -> 400 000 reads)
D=/dev/shm/sfa_env; mkdir -p $D
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 $D/c.blow5 --copies ${COPIES:-80000} --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/dev/shm/sfa_env/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
cat $D/c.blow5 > /dev/null
for rep in 1 2 3; do
for E in "SFA_NOP=1" "$@"; do
sleep 1
T0=$(date +%s.%N)
env $E sigfish_amd/bin/sigfish-amd dtw --kmer-model $D/syn6.model -t 16 -B 2G -K 4096 --verbose 4 tests/golden/data/nCoV-2019.reference.fasta $D/c.blow5 > $D/out.paf 2> $D/err.txt
T1=$(date +%s.%N)
python -c "print('$E: wall %.3f' % ($T1-$T0))"
grep "host stages\|waited\|initialised" $D/err.txt | cut -c1-170
done
done
rm -rf $D
