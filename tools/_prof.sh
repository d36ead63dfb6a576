cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof7 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipeline 0 > gpurun_out/prof7.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof7/*/*kernel_stats.csv")[0]
for r in list(csv.reader(open(f)))[:16]:
    print(r[0][:62], r[1], r[3][:9], r[4])
PY
