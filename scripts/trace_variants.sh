#!/bin/bash
# Kernel traces of the headline stream under several builds / switches (GPU box): scripts/trace_variants.sh "name|lib|ENV=..;ENV2=.." ...
# lib = path of a variant library or "-" for the product build.  One rocprofv3 --kernel-trace run per configuration (no counters).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trc
mkdir -p $OUT
for cfg in "$@"; do
  name=${cfg%%|*}; rest=${cfg#*|}; lib=${rest%%|*}; envs=${rest#*|}
  (
    [ "$lib" != "-" ] && export SCL_ENGINE_LIB=$lib
    IFS=';' read -ra kv <<< "$envs"
    for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    rm -rf $OUT/$name
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 bench.py --steps 48 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/$name.json 2> $OUT/$name.err
    python3 bench.py --steps 128 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/$name.plain.json 2>> $OUT/$name.err
    echo "== $name" >> $OUT/summary.txt
    python3 scripts/trace_summary.py $OUT/$name 200 >> $OUT/summary.txt 2>&1
    python3 -c "import json,sys; j=json.load(open('$OUT/$name.plain.json')); print('plain bench: value %.3f G pairs/s, ms_per_step %.4f, kernel_ms %.4f, frac %.3f' % (j['value']/1e9, j['ms_per_step'], j['kernel_ms']['sc_distance'], j['roofline']['frac']))" >> $OUT/summary.txt 2>&1
    find $OUT/$name -name "*.csv" ! -name "*kernel_trace.csv" -delete
  )
done
cat $OUT/summary.txt | cut -c1-200
