"""Condenses gpurun_out/prof_<tag>_ch* (tools/pmc_ch.sh) into profiles/<tag>_ch_pmc.json and profiles/<tag>_ch_kernel_stats.csv:
per launch of ch_kernel_rows (grouped by grid size = frame size) the counters, and derived per-pixel figures."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")

def one(pattern):
    hits = sorted(glob.glob(os.path.join(out, pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None

res = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("sq", "mix", "fetch", "write"):
    f = one(f"prof_{tag}_ch_{kind}/**/*_counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if "ch_kernel" not in r["Kernel_Name"]:
            continue
        key = f'{"rows" if "rows" in r["Kernel_Name"] else "tiles"} grid={r["Grid_Size"]} vgpr={r["VGPR_Count"]}'
        res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
doc = {}
for key, ctrs in res.items():
    d = {c: sum(v) / len(v) for c, v in ctrs.items()}
    d["launches"] = max(len(v) for v in ctrs.values())
    doc[key] = d
ks = one(f"prof_{tag}_ch/**/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(prof, f"{tag}_ch_kernel_stats.csv"))
log = os.path.join(out, f"prof_{tag}_ch.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("CH0")]
    open(os.path.join(prof, f"{tag}_ch_bandwidth.txt"), "w").writelines(lines)
    doc["_bandwidth_under_rocprof"] = [l.strip() for l in lines]
json.dump(doc, open(os.path.join(prof, f"{tag}_ch_pmc.json"), "w"), indent=1, sort_keys=True)
for k, v in doc.items():
    print(k, v)
