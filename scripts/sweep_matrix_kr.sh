#!/bin/bash
# keyframes per workgroup of the exact matrix kernel (diagnostics build of engine.hip: SCL_MATRIX_KR), on one box
cd "$GRAFT_REPO_ROOT"
export SCL_ENGINE_LIB=scl_slam_amd/lib/variants/libscl_engine_diag.so
for kr in 16 32 48 64 96 128; do
  SCL_MATRIX_KR=$kr timeout -k 10 200 python3 scripts/bench_matrix.py 64 ${GRID:-64x120} > gpurun_out/kr_$kr.json 2> gpurun_out/kr_$kr.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/kr_$kr.json'))['${GRID:-64x120}']
print('kr $kr: %.1f M pairs/s  %.1f us per row  group %.0f us' % (d['pairs_per_s']/1e6, d['ms_per_row']*1e3, d['group_us']))"
done
