#!/bin/bash
# quick kernel trace of the headline bench (run on the GPU box through gpurun): per-kernel durations -> gpurun_out/prof_quick/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_quick
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k1 -- python3 bench.py --steps 16 --warmup 2 --repeats 2 --no-cpu-baseline --no-secondary > $OUT/bench_under_trace.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 scripts/summarize_kernel_stats.py $OUT/kernel_stats.csv > $OUT/kernel_stats_short.txt
# gaps: consecutive kernels on the trace
python3 - <<'PY' > $OUT/gaps.txt
import csv, glob
f = glob.glob('gpurun_out/prof_quick/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if any(k in r['Kernel_Name'] for k in ('sc_screen', 'sc_select', 'sc_distance_wave'))]
rows = rows[len(rows)//2:]
import collections
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    ka = a['Kernel_Name'].split('<')[0].split('(')[0][-28:]; kb = b['Kernel_Name'].split('<')[0].split('(')[0][-28:]
    dur[ka].append((int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3)
    gap[ka + ' -> ' + kb].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[:16]: print('row', r['Kernel_Name'][:40].replace('void ',''), 'start', (int(r['Start_Timestamp']) - t0) / 1e3, 'end', (int(r['End_Timestamp']) - t0) / 1e3, 'queue', r.get('Queue_Id'), 'stream', r.get('Stream_Id'))
for k, v in dur.items(): print('dur us', k, round(sum(v) / len(v), 1), 'n', len(v))
for k, v in gap.items(): print('gap us', k, round(sum(v) / len(v), 1), 'n', len(v))
PY
rm -rf $OUT/trace
echo done
