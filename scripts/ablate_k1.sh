# diagnostic: K1 time with phases switched off (results are wrong on purpose), see DESIGN.md
for f in ${ABLATE_SET:-0 1 2 4 7 8 16 24}; do echo "ablate=$f"; SCL_ABLATE=$f python bench.py --steps 60 --warmup 5 --no-cpu-baseline --keyframes 10000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'])"; done
