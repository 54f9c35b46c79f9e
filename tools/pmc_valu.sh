#!/bin/bash
# VALU / SALU / LDS instruction counts of one bench configuration (run on the GPU box):
#   tools/pmc_valu.sh <tag> "<bench args>"   ->  gpurun_out/pmc_<tag>.txt
TAG=${1:-x}
EXTRA=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU \
    --output-format csv -d $O/pmc_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > $O/pmc_$TAG.log 2>&1
python3 - "$O/pmc_$TAG" > $O/pmc_$TAG.txt <<'PY'
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
cat $O/pmc_$TAG.txt
