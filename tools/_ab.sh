python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for g in 1 2 3; do
REM2D_STEP_GROUPS=$g python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lsystem groups=$g', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), d['roofline']['launches'], round(d['roofline']['frac'],5))"
done
for wl in chain8 cppn_hardcore chain4; do
for g in 1 2; do
REM2D_STEP_GROUPS=$g python bench.py --no-cpu-baseline --steps 100 --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl groups=$g', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
done; done
