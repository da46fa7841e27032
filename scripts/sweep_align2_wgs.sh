cd "$GRAFT_REPO_ROOT"
for w in 0 192 256 320 384 640 768 1024; do
  for r in 1 2; do
    if [ "$w" = "0" ]; then unset SCL_ALIGN2_WGS; else export SCL_ALIGN2_WGS=$w; fi
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs $w run $r: %.3f G pairs/s  group %.1f us  frac %.3f' % (d['value']/1e9, d['kernel_ms']['sc_distance']*1e3, d['roofline']['frac']))"
  done
done
