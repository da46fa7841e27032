#!/bin/bash
# rocprof evidence of the front half of the per-incoming-scan path (K3 descriptor + ingest in batches of 16, K2 ring-key scan, the
# voxel filter's kernels when a filtered call is in the command):
#   gpurun -- 'scripts/profile_front.sh'   -> gpurun_out/prof_front/{kernel_stats.csv,kernel_stats_short.txt,pmc_summary.json,bench.json}
# Copy those to profiles/rNN/front/.
D=$(dirname "$0")
F=${FRONT_FILTER:-make_sc_batch_scatter_kernel,ingest_kernel,front_fused_kernel,ringkey_lists_kernel,topk_merge_pack_kernel}
$D/profile_cmd.sh front $F scripts/bench_front.py all
