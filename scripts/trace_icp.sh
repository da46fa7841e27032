#!/bin/bash
# Kernel trace of the configs[2] batch from the keyframe store (scripts/trace_icp_batch.py), summarised per phase.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_icp
rm -rf $OUT && mkdir -p $OUT
LEAF=${LEAF:-0.05} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 scripts/trace_icp_batch.py run > $OUT/run.log 2> $OUT/run.err
DUMP=${DUMP:-} python3 scripts/trace_icp_batch.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/timeline.txt
rm -rf $OUT/trace
cat $OUT/run.log
