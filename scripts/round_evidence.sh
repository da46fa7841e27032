#!/bin/bash
# The round's evidence in two gpurun calls (each under the 20-minute limit):
#   scripts/round_evidence.sh a   -> gpurun_out/r05a/: whole -m gpu suite, scripts/profile_front.sh (kernel trace + PMC), soak_front
#   scripts/round_evidence.sh b   -> gpurun_out/r05b/: scripts/profile_k1.sh (headline group), bench lines (driver style + default), soak_parity
#   scripts/round_evidence.sh c   -> gpurun_out/r05c/: scripts/profile_icp.sh (configs[2] batch: kernel trace, timeline, counters), soak_icp
# Copy the summaries to profiles/r05/ (README there).
P=$1
cd "$GRAFT_REPO_ROOT"
if [ "$P" = "a" ]; then
  O=gpurun_out/r05a; mkdir -p $O
  timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
  scripts/profile_front.sh && cp -r gpurun_out/prof_front $O/front
  timeout -k 10 200 python3 scripts/soak_front.py 150 > $O/soak_front.txt 2>&1; tail -2 $O/soak_front.txt
elif [ "$P" = "c" ]; then
  O=gpurun_out/r05c; mkdir -p $O
  scripts/profile_icp.sh && cp -r gpurun_out/prof_icp $O/icp
  timeout -k 10 260 python3 scripts/soak_icp.py 200 > $O/soak_icp.txt 2>&1; tail -3 $O/soak_icp.txt
else
  O=gpurun_out/r05b; mkdir -p $O
  STEPS=128 scripts/profile_k1.sh && cp -r gpurun_out/prof_k1 $O/k1 && rm -rf $O/k1/trace $O/k1/p[0-9]*
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; tail -c 600 $O/bench_driver_style.json
  timeout -k 10 300 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
  timeout -k 10 300 python3 scripts/soak_parity.py 240 > $O/soak_parity.txt 2>&1; tail -2 $O/soak_parity.txt
fi
