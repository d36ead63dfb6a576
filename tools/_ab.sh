python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for ml in 1 0; do
REM2D_MERGED_LAUNCH=$ml python bench.py --no-cpu-baseline --steps 60 --pipeline 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lsystem fused merged=$ml', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
done
REM2D_WAVES_PER_SIMD=2 python bench.py --no-cpu-baseline --steps 60 --pipeline 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lsystem fused merged W=2', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
python bench.py --no-cpu-baseline --steps 60 --pipeline 0 --discrete 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('discrete fused merged', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
python bench.py --no-cpu-baseline --steps 60 --pipeline 0 --workload cppn_hardcore 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cppn fused merged', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
