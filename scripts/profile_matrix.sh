#!/bin/bash
# rocprof evidence of the exact distance matrix (scl_sc_distance_matrix, D.h:1538-1569 per pair): 64 rows x 9 900 keyframes.
#   gpurun -- 'scripts/profile_matrix.sh'   -> gpurun_out/prof_matrix_64x120/, gpurun_out/prof_matrix_80x180/
# Copy kernel_stats.csv, kernel_stats_short.txt, pmc_summary.json, bench.json to profiles/rNN/matrix[_80x180]/ and
# traffic_matrix.json (HBM bytes of a group of 16 rows: all four kernels of the group) to profiles/traffic_matrix.json.
D=$(dirname "$0")
F=${MATRIX_FILTER:-sc_matrix_kernel,sc_screen2_kernel,sc_align2_kernel,sc_screen2_finish_kernel}
$D/profile_cmd.sh matrix_64x120 $F scripts/bench_matrix.py 64 64x120 && \
python3 - <<'P' && $D/profile_cmd.sh matrix_80x180 $F scripts/bench_matrix.py 64 80x180
import json
j = json.load(open("gpurun_out/prof_matrix_64x120/pmc_summary.json"))
out = {"source": "scripts/profile_matrix.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of scripts/bench_matrix.py 64 64x120; FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, "
                 "summed over the four kernels of a group of 16 rows (sc_align2, sc_screen2, sc_screen2_finish, sc_matrix)",
       "hbm_bytes_per_group": j["hbm_bytes_per_launch"], "rows_per_group": 16, "eligible_keyframes": 9900,
       "per_kernel": {k["kernel_filter"]: k.get("hbm_bytes_per_launch") for k in j["kernels"]}}
json.dump(out, open("gpurun_out/prof_matrix_64x120/traffic_matrix.json", "w"), indent=1)
print(json.dumps(out))
P
