run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --steps 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['reads_per_gpu'], 'reads/s', d['value'], 'cells/s %.3e' % d['dp_cells_per_s'], 'ms/step', d['ms_per_step'], flush=True)"; }
for rep in 1 2; do
for n in sigfish_amd sfa_pw4; do
  echo "== lib$n.so"
  for w in ncov_r9_dna_q3000 ncov_r9_dna_q4000 ncov_r9_dna_q8000 ncov_r9_dna_q2500 ncov_r9_dna_q6000; do
    SFA_LIB=$GRAFT_REPO_ROOT/sigfish_amd/lib/lib$n.so run --workload $w
  done
done
done
