#!/bin/bash
# rocprof evidence of the 80x180 stream (BASELINE configs[4]'s grid, 10 k keyframes): bench.py's secondary on its own.
#   gpurun -- 'scripts/profile_80x180.sh'   -> gpurun_out/prof_k1_80x180/
D=$(dirname "$0")
$D/profile_cmd.sh k1_80x180 ${K1_FILTER:-sc_screen2_kernel,sc_screen2_tail2_kernel} scripts/bench_80x180.py
