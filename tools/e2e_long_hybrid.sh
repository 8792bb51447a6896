#!/bin/bash
# whole-process reads/s on a 1 600 000-read compressed file: host route against every Nth batch decoded on the device (run on the GPU box)
D=/dev/shm/sfa_lh; mkdir -p $D
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 $D/c.blow5 --copies ${COPIES:-320000} --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/dev/shm/sfa_lh/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
cat $D/c.blow5 > /dev/null
for rep in 1 2; do
for S in "--hybrid-parse 0" "--hybrid-parse 8" "--hybrid-parse 5" "--hybrid-parse 3" "--hybrid-parse 0 -K 8192" "--hybrid-parse 4 -K 8192"; do
sleep 1
T0=$(date +%s.%N)
sigfish_amd/bin/sigfish-amd dtw --kmer-model $D/syn6.model -t 16 -B 2G -K 4096 --verbose 4 $S tests/golden/data/nCoV-2019.reference.fasta $D/c.blow5 > $D/out.paf 2> $D/err.txt
T1=$(date +%s.%N)
python -c "print('$S: wall %.3f' % ($T1-$T0))"
grep "host stages\|waited" $D/err.txt | cut -c1-150
done
done
rm -rf $D
