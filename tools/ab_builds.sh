#!/bin/bash
# usage: [WL=workload] tools/ab_builds.sh name1 name2 ...   (A/B of libsfa_<name>.so builds, interleaved rounds in one call)
WL=${WL:-ncov_r9_dna_q250}
for round in 1 2; do
for n in "$@"; do
  SFA_LIB=$GRAFT_REPO_ROOT/sigfish_amd/lib/libsfa_$n.so timeout -k 10 200 python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL', '$n', d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['trace_kernel_ms_per_step'])"
done; done
