#!/bin/bash
# kernel-level profile of the command line on a replicated fixture (run on the GPU box): tools/e2e_profile.sh [copies] [K]
set -e
COPIES=${1:-20000}; K=${2:-16384}
ROOT=$(pwd)
python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 /tmp/big.blow5 --copies $COPIES
python - <<'PY'
import itertools, numpy as np
lv = np.fromfile("tests/golden/models/syn6.f32", np.float32)
with open("/tmp/syn6.model", "w") as f:
    f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
    for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
        f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
PY
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/e2e_prof_K$K
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o runc -- $ROOT/sigfish_amd/bin/sigfish-amd dtw --kmer-model /tmp/syn6.model -t 16 -K $K -B 2G --streams 1 --verbose 3 \
    tests/golden/data/nCoV-2019.reference.fasta /tmp/big.blow5 > /tmp/big.paf 2> $OUT/log.txt
grep dtw_main $OUT/log.txt
cut -d, -f1-4 $OUT/runc_kernel_stats.csv | head -14
