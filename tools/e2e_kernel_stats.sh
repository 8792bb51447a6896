#!/bin/bash
# per-kernel device time of the command line on a compressed file (host route, -K 4096): rocprofv3 --kernel-trace --stats of the
# binary itself (run on the GPU box).  COPIES=80000 -> 400 000 reads
D=/dev/shm/sfa_ks; mkdir -p $D
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 $D/c.blow5 --copies ${COPIES:-80000} --compress --jobs 16 | tail -1
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/dev/shm/sfa_ks/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
cat $D/c.blow5 > /dev/null
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/e2e_kernel_stats; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- $(pwd)/sigfish_amd/bin/sigfish-amd dtw --kmer-model $D/syn6.model -t 16 -B 2G -K ${K:-4096} --verbose 0 -o $D/out.paf tests/golden/data/nCoV-2019.reference.fasta $D/c.blow5 > $OUT/run.log 2>&1
wc -l $D/out.paf
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160
rm -rf $D
