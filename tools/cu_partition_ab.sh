#!/bin/bash
# CU partition for the device route of the command line (run on the GPU box): record decoding + event detection on
# SFA_RESERVED_CUS CUs of their own (CU-masked stream), alignment on the rest -- against the unpartitioned device route and
# the host route, whole process, compressed 400 000-read file in /dev/shm, warm.  PAF of every run compared by checksum.
D=/dev/shm/sfa_cu_ab; mkdir -p $D
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 $D/c.blow5 --copies 80000 --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/dev/shm/sfa_cu_ab/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
cat $D/c.blow5 > /dev/null
run() {  # label, env, args...
  local label=$1 envv=$2; shift 2
  for rep in 1 2; do
    T0=$(date +%s.%N)
    env $envv sigfish_amd/bin/sigfish-amd dtw --kmer-model $D/syn6.model -t 16 -B 2G --verbose 0 "$@" tests/golden/data/nCoV-2019.reference.fasta $D/c.blow5 > $D/out.paf
    T1=$(date +%s.%N)
    python -c "import hashlib; b=open('$D/out.paf','rb').read(); n=b.count(b'\n'); dt=$T1-$T0; print(f'$label: {n} reads in {dt:.3f} s = {n/dt:.0f} reads/s  md5 {hashlib.md5(b).hexdigest()[:12]}', flush=True)"
  done
}
run "host route -K 4096" X=1 -K 4096
run "host route -K 8192" X=1 -K 8192
for S in 2 4; do
  run "device route, no partition, --streams $S -K 8192" X=1 --gpu-parse --streams $S -K 8192
  for R in 32 64 96 128; do
    run "device route, $R CUs reserved, --streams $S -K 8192" SFA_RESERVED_CUS=$R --gpu-parse --streams $S -K 8192
  done
done
run "device route, 64 CUs reserved, --streams 4 -K 4096" SFA_RESERVED_CUS=64 --gpu-parse --streams 4 -K 4096
run "device route, 64 CUs reserved, --streams 4 -K 16384" SFA_RESERVED_CUS=64 --gpu-parse --streams 4 -K 16384
rm -rf $D
