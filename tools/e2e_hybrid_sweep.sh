#!/bin/bash
# whole-process reads/s of the command line on a compressed 400 000-read file against the share of batches that take the device
# route (--hybrid-parse N: every Nth batch; 0 = host route only; --gpu-parse = all), three interleaved repetitions (run on the GPU box)
D=/dev/shm/sfa_hyb; mkdir -p $D
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 $D/c.blow5 --copies 80000 --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/dev/shm/sfa_hyb/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
cat $D/c.blow5 > /dev/null
run() {  # label, args...
  local label=$1; shift
  T0=$(date +%s.%N)
  sigfish_amd/bin/sigfish-amd dtw --kmer-model $D/syn6.model -t 16 -B 2G --verbose 0 "$@" tests/golden/data/nCoV-2019.reference.fasta $D/c.blow5 > $D/out.paf
  T1=$(date +%s.%N)
  python -c "import hashlib; b=open('$D/out.paf','rb').read(); n=b.count(b'\n'); dt=$T1-$T0; print(f'$label: {n} reads in {dt:.3f} s = {n/dt:.0f} reads/s  md5 {hashlib.md5(b).hexdigest()[:12]}', flush=True)"
}
for rep in 1 2 3; do
  for K in ${KS:-4096}; do
    run "rep $rep -K $K host route only (--hybrid-parse 0)" -K $K --hybrid-parse 0
    for H in 6 4 3 2; do run "rep $rep -K $K every ${H}th batch on the device" -K $K --hybrid-parse $H; done
    run "rep $rep -K $K device route only (--gpu-parse)" -K $K --gpu-parse
  done
done
run "verbose timers, host route" -K 4096 --hybrid-parse 0 --verbose 3 2> $D/v0.err; grep "Data\|initialised" $D/v0.err
run "verbose timers, every 3rd" -K 4096 --hybrid-parse 3 --verbose 3 2> $D/v3.err; grep "Data\|initialised" $D/v3.err
rm -rf $D
