#!/bin/bash
# HBM traffic counters of one bench configuration with a chosen library (run on the GPU box):
#   tools/pmc_traffic.sh <tag> <lib.so> "<bench args>"  -> prints mean FETCH_SIZE / WRITE_SIZE (KiB) of the path kernel
TAG=${1:-x}; LIB=$2; EXTRA=${3:-}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
export TMPDIR=/tmp RTIOW_LIB=$R/$LIB
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pt_${TAG}_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $EXTRA > $O/pt_${TAG}_$C.log 2>&1
done
python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys
o, tag = sys.argv[1:3]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"{o}/pt_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "persistent" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    print(tag, c, "KiB mean", sum(vals) / max(1, len(vals)), "launches", len(vals))
PY
