#!/bin/bash
# wait / issue counters of the path kernel for one bench configuration (run on the GPU box):
#   tools/pmc_wait.sh <tag> "<bench args>"   ->  gpurun_out/pmcw_<tag>.txt
TAG=${1:-x}
EXTRA=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH \
    --output-format csv -d $O/pmcw_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > $O/pmcw_$TAG.log 2>&1
python3 - "$O/pmcw_$TAG" > $O/pmcw_$TAG.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "persistent" not in k and "path_pixel" not in k: continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
cat $O/pmcw_$TAG.txt
