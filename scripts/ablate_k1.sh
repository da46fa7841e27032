#!/bin/bash
# Phase ablation of the exact SC-distance wave kernel (diagnostic build only: results are wrong on purpose).
# Rebuilds the library with -DSCL_DIAGNOSTICS, runs the bench with the screening pass off, restores the product build.
set -e
cd "$(dirname "$0")/.."
touch scl_slam_amd/csrc/sc_distance.hip && make -j8 EXTRA=-DSCL_DIAGNOSTICS > /dev/null
for f in ${ABLATE_SET:-0 1 2 4 7 8 16 24}; do echo "ablate=$f"; SCL_SCREEN=0 SCL_ABLATE=$f python bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary --keyframes 10000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'])"; done
touch scl_slam_amd/csrc/sc_distance.hip && make -j8 > /dev/null
