#!/bin/bash
# The counters of the END of a small frame: one eighth of the cover frame with max_depth 50 and 400 under --pmc -- the
# difference is 350 more bounces of the paths trapped in the glass ball, i.e. sparse iterations only (run on the GPU box):
#   tools/tail_pmc.sh  ->  gpurun_out/tail_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
cat > /tmp/tail_one.py <<PY
import os, sys
sys.path.insert(0, "$R")
import vulkan_rtiow_amd as V
d = int(sys.argv[1])
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=100, max_depth=d, seed=1, row_block=4, tile_rank=0, tile_count=8)
    for _ in range(5):
        ctx.render(cam, prm)
    print(d, ctx.stats().kernel_ms, ctx.stats().segments)
PY
for d in 50 400; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
      --output-format csv -d $O/tail_pmc_a_$d -- python3 /tmp/tail_one.py $d > $O/tail_pmc_a_$d.log 2>&1
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
      --output-format csv -d $O/tail_pmc_b_$d -- python3 /tmp/tail_one.py $d > $O/tail_pmc_b_$d.log 2>&1
done
python3 - $O > $O/tail_pmc.txt <<'PY'
import csv, glob, sys, collections
res = {}
for d in (50, 400):
    acc = collections.defaultdict(list)
    for tag in "ab":
        for f in glob.glob(f"{sys.argv[1]}/tail_pmc_{tag}_{d}/**/*counter_collection.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "persistent" in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})
            keep = set(ids[2:])  # (the first frames of a shape have no chunk order yet)
            for r in rows:
                if int(r["Dispatch_Id"]) in keep:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # a counter appears once per dispatch and (for some) per XCD/SE dimension: sum over dimensions, mean over dispatches
    n = 3
    res[d] = {k: sum(v) / n for k, v in acc.items()}
print(f"{'counter':28s} {'depth 50':>14s} {'depth 400':>14s} {'difference':>14s}")
for k in sorted(res[50]):
    a, b = res[50][k], res[400].get(k, 0.0)
    print(f"{k:28s} {a:14.4g} {b:14.4g} {b - a:14.4g}")
PY
cat $O/tail_pmc_a_50.log $O/tail_pmc_a_400.log | grep -v amdgpu | tail -4
cat $O/tail_pmc.txt
