REM2D_PIPELINE=1 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for pl in 0 1; do for g in 2 3; do
python bench.py --no-cpu-baseline --steps 100 --pipeline $pl --step-groups $g 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lsystem pipeline=$pl groups=$g', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
python bench.py --no-cpu-baseline --steps 100 --pipeline $pl --step-groups $g --workload cppn_hardcore 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cppn pipeline=$pl groups=$g', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
done; done
