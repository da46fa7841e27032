#!/bin/bash
# What bounds the FIRST form of the screening products (SCL_SCREEN_FORM=1; the probes live in its kernel)?  Diagnostic build (-DSCL_DIAGNOSTICS), SCL_SCREEN_PROBE: 0 = the product kernels,
# 2 = the screening kernel without staging and MFMA (its loads alone), 3 = without the alignment kernel (stale first shifts),
# 4 = the first launch's own alignment kernel alone (outside the event pair: see rocprofv3), 5 / 6 = the alignment role inside the products' launch reduced to its sector-key reads /
# to its matrix products on stale keys.  Results are wrong on purpose.
set -e
cd "$(dirname "$0")/.."
touch scl_slam_amd/csrc/sc_screen.hip scl_slam_amd/csrc/sc_distance.hip && make -j8 EXTRA=-DSCL_DIAGNOSTICS > /dev/null 2>&1
for p in ${PROBES:-0 2 3 4}; do echo "probe=$p"; SCL_SCREEN_FORM=1 SCL_ABLATE=8 SCL_SCREEN_PROBE=$p python bench.py --steps 8 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))"; done
touch scl_slam_amd/csrc/sc_screen.hip scl_slam_amd/csrc/sc_distance.hip && make -j8 > /dev/null 2>&1
