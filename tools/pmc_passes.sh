#!/bin/bash
# Round 5: instruction mix and LDS behaviour of the step kernels (one --pmc pass per counter group; counters only + --kernel-trace)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_passes; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="--steps 10 --warmup 2 --settle 80 --no-cpu-baseline --no-secondary --min-time 0"
python3 bench.py $ARGS > /dev/null 2>&1
i=0
# PMC_GROUPS="A B C;D E": other counter groups (';' between passes), e.g. the instruction-cache pass
#   PMC_GROUPS="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE;SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
DEFAULT_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES;SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
IFS=';' read -ra GROUPS_ <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 420 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/raw$i -- python3 bench.py $ARGS > /dev/null 2> $O/err$i.txt
  f=$(find $O/raw$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$grp" <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "rem2d_velpost" in k or "rem2d_pre" in k or "rem2d_toi_heavy" in k or "rem2d_step_tile" in k:
        acc[k.split("(")[0]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, cs in acc.items():
    out = []
    for c, v in cs.items():
        per = collections.defaultdict(float)
        for d, x in v: per[d] += x
        last = [per[d] for d in sorted(per)[-30:]]
        out.append("%s %.4g" % (c, sum(last) / len(last)))
    print(k, "per launch (mean of the last 30):", ", ".join(out))
PY
  else tail -3 $O/err$i.txt; fi
  rm -rf $O/raw$i
done
