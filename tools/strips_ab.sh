#!/bin/bash
# row strips (queries beyond 2048 events), run on the GPU box: per workload the whole-step rate under
#   pipelined pass 1 / one wave per (read, job)  x  chained pass 2 / all strips over the whole range,
# then any libsfa_<name>.so builds named on the command line (A/B of compile-time variants, e.g. SFA_STRIP_WT=0)
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --steps 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['reads_per_gpu'], 'reads/s', d['value'], 'cells/s %.3e' % d['dp_cells_per_s'], 'ms/step', d['ms_per_step'], flush=True)"; }
for o in "1 1" "1 0" "0 1" "0 0"; do
  set -- $o "${@:3}"
  echo "== strip_pipeline $1 strip_chain $2"
  for w in ${WL:-ncov_r9_dna_q3000 ncov_r9_dna_q4000 ncov_r9_dna_q8000}; do
    run --workload $w --opt strip_pipeline=$1 --opt strip_chain=$2
  done
done
echo "== balanced_strips 0 (64 x 32 rows per strip and a short last one)"
for w in ${WL:-ncov_r9_dna_q3000 ncov_r9_dna_q4000 ncov_r9_dna_q8000}; do
  run --workload $w --opt balanced_strips=0
done
for n in ${LIBS}; do
  echo "== libsfa_$n.so (defaults)"
  for w in ${WL:-ncov_r9_dna_q3000 ncov_r9_dna_q4000 ncov_r9_dna_q8000}; do
    SFA_LIB=$GRAFT_REPO_ROOT/sigfish_amd/lib/libsfa_$n.so run --workload $w
  done
done
