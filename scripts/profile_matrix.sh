#!/bin/bash
# rocprof evidence of the exact distance matrix (scl_sc_distance_matrix, D.h:1538-1569 per pair): 64 rows x 9 900 keyframes.
#   gpurun -- 'scripts/profile_matrix.sh'   -> gpurun_out/prof_matrix_64x120/, gpurun_out/prof_matrix_80x180/
# Copy kernel_stats.csv, kernel_stats_short.txt, pmc_summary.json, bench.json to profiles/rNN/matrix[_80x180]/.
D=$(dirname "$0")
$D/profile_cmd.sh matrix_64x120 ${MATRIX_FILTER:-sc_matrix,sc_masked} scripts/bench_matrix.py 64 64x120 && \
$D/profile_cmd.sh matrix_80x180 ${MATRIX_FILTER:-sc_matrix,sc_masked} scripts/bench_matrix.py 64 80x180
