#!/bin/bash
# whole-process reads/s of the command line on a compressed (and an uncompressed) 400 000-read file, three interleaved repetitions
# of a few option sets, with the stage timers of one run each (run on the GPU box):  bash tools/e2e_quick.sh [option sets, ';'-separated]
D=/dev/shm/sfa_e2eq; mkdir -p $D
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 $D/c.blow5 --copies 80000 --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/dev/shm/sfa_e2eq/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
cat $D/c.blow5 > /dev/null
run() {  # label, args...
  local label=$1; shift
  T0=$(date +%s.%N)
  sigfish_amd/bin/sigfish-amd dtw --kmer-model $D/syn6.model -t 16 -B 2G "$@" tests/golden/data/nCoV-2019.reference.fasta $D/c.blow5 > $D/out.paf 2> $D/err.txt
  T1=$(date +%s.%N)
  python -c "import hashlib; b=open('$D/out.paf','rb').read(); n=b.count(b'\n'); dt=$T1-$T0; print(f'$label: {n} reads in {dt:.3f} s = {n/dt:.0f} reads/s  md5 {hashlib.md5(b).hexdigest()[:12]}', flush=True)"
}
IFS=';' read -ra SETS <<< "${1:---hybrid-parse 0;--hybrid-parse 6;--hybrid-parse 4}"
for rep in 1 2 3; do
  for S in "${SETS[@]}"; do run "rep $rep -K 4096 $S" -K 4096 --verbose 0 $S; done
done
for S in "${SETS[@]}"; do run "timers -K 4096 $S" -K 4096 --verbose 3 $S; grep "Data\|initialised\|all output" $D/err.txt; done
rm -rf $D
