#!/bin/bash
# keyframes per workgroup of the exact matrix kernel against BOTH its rate and its HBM traffic (rocprofv3 --pmc FETCH_SIZE, separate pass):
# the sets of one range re-read keyframes their siblings have pushed out of the XCD's L2 -- a shorter range bounds how far they can drift.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SCL_ENGINE_LIB=scl_slam_amd/lib/variants/libscl_engine_diag.so
for kr in ${KRS:-16 24 32 48 64}; do
  export SCL_MATRIX_KR=$kr
  python3 scripts/bench_matrix.py 64 ${GRID:-64x120} > gpurun_out/kr_$kr.json 2> gpurun_out/kr_$kr.err
  rm -rf gpurun_out/kr_pmc_$kr
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/kr_pmc_$kr -o pmc -- python3 scripts/bench_matrix.py 64 ${GRID:-64x120} > /dev/null 2> gpurun_out/kr_pmc_$kr.err
  python3 - <<P
import json, csv, glob
d = json.load(open('gpurun_out/kr_$kr.json'))['${GRID:-64x120}']
v = []
for f in glob.glob('gpurun_out/kr_pmc_$kr/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'sc_matrix_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            v.append(float(r['Counter_Value']))
fetch = sum(v) / max(1, len(v)) * 1024 * 2 / 1e6
print('kr $kr: %.1f M pairs/s  group %.0f us  sc_matrix_kernel reads %.0f MB per group of 16 rows (x2-corrected FETCH_SIZE, %d launches)' % (d['pairs_per_s'] / 1e6, d['group_us'], fetch, len(v)))
P
  rm -rf gpurun_out/kr_pmc_$kr
done
