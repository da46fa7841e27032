#!/bin/bash
# What bounds the screening kernel?  Diagnostic build (-DSCL_DIAGNOSTICS): the kernel without the alignment (probe 1) and
# without alignment, conversion and MFMA (probe 2: the access pattern alone), the same bytes as one contiguous stream per wave
# (probe 3), and both with 15 instead of 5 k-steps of loads in flight (probes 4, 5).  Results are wrong on purpose.
set -e
cd "$(dirname "$0")/.."
touch scl_slam_amd/csrc/sc_screen.hip scl_slam_amd/csrc/sc_distance.hip && make -j8 EXTRA=-DSCL_DIAGNOSTICS > /dev/null 2>&1
for p in ${PROBES:-0 2 3 4 5}; do echo "probe=$p"; SCL_ABLATE=8 SCL_SCREEN_PROBE=$p python bench.py --steps 100 --warmup 8 --repeats 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))"; done
touch scl_slam_amd/csrc/sc_screen.hip scl_slam_amd/csrc/sc_distance.hip && make -j8 > /dev/null 2>&1
