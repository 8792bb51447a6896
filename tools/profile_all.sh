#!/bin/bash
# kernel-trace + four PMC passes (tools/profile_round.sh) for every BASELINE.json workload shape, one 32-row shape and one row-strip shape (8 000 events)
# (run on the GPU box):  bash tools/profile_all.sh <round tag, e.g. r03>   ->  gpurun_out/prof_<tag>_<workload>/
R=${1:?round tag}
set -x
bash tools/profile_round.sh ${R}_ncov_q250
bash tools/profile_round.sh ${R}_sequin_rna_q250 --workload sequin_r9_rna_q250
bash tools/profile_round.sh ${R}_rna004_dtwstd_q250 --workload rna004_fullref_dtwstd_q250
bash tools/profile_round.sh ${R}_r10_1mb_q250 --workload r10_dna_1mb_q250 --reads 125000
bash tools/profile_round.sh ${R}_ncov_q500 --workload ncov_r9_dna_q500
bash tools/profile_round.sh ${R}_ncov_q1000 --workload ncov_r9_dna_q1000
bash tools/profile_round.sh ${R}_ncov_q8000 --workload ncov_r9_dna_q8000
