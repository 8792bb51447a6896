D=/dev/shm/sfa_tail; mkdir -p $D
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 $D/c.blow5 --copies ${COPIES:-80000} --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/dev/shm/sfa_tail/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
cat $D/c.blow5 > /dev/null
# KS="4096 8192" EXTRA="--streams 3": batch sizes / further options to sweep (each REPS times, interleaved)
for rep in $(seq 1 ${REPS:-3}); do
for K in ${KS:-4096}; do
T0=$(date +%s.%N)
sigfish_amd/bin/sigfish-amd dtw --kmer-model $D/syn6.model -t ${T:-16} -B 2G -K $K $EXTRA --verbose 4 tests/golden/data/nCoV-2019.reference.fasta $D/c.blow5 > $D/out.paf 2> $D/err.txt
T1=$(date +%s.%N)
python -c "print('-K $K $EXTRA: wall %.3f' % ($T1-$T0))"
grep "initialised\|all output\|released\|Data\|waited" $D/err.txt
done
done
rm -rf $D
