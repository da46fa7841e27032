#!/bin/bash
# Where does the exact pass over the survivors (sc_distance_survivors_kernel, the tail of a short call) spend its 44 us?
# Diagnostic build; SCL_ABLATE: 1 = no alignment, 2 = no ring products, 4 = no sector sums, 7 = all three (what is left is
# staging the query, selecting the survivors, the ring-key top-k and the tail).  Results are wrong on purpose.
set -e
cd "$(dirname "$0")/.."
touch scl_slam_amd/csrc/sc_distance.hip scl_slam_amd/csrc/sc_screen.hip && make -j8 EXTRA=-DSCL_DIAGNOSTICS > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in 0 1 2 4 7; do
  rm -rf gpurun_out/ps && SCL_ABLATE=$f timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps -o t -- python3 scripts/probes/short_block.py 20 > /dev/null 2>&1 || true
  echo "ablate=$f $(python3 scripts/summarize_kernel_stats.py $(find gpurun_out/ps -name '*kernel_stats.csv' | head -1) | grep survivors)"
done
touch scl_slam_amd/csrc/sc_distance.hip scl_slam_amd/csrc/sc_screen.hip && make -j8 > /dev/null 2>&1
