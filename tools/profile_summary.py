"""Condenses gpurun_out/prof_<tag>_* (rocprofv3 csv) into profiles/<tag>_kernel_stats.csv,
profiles/<tag>_pmc.json and profiles/traffic_<tag>.json (read back by bench.py as roofline.traffic)."""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_sha16  # identity of the kernels these counters belong to (bench.py checks it)
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def one(pattern):
    hits = sorted(glob.glob(os.path.join(out, pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


ks = one(f"prof_{tag}_stats/**/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(prof, f"{tag}_kernel_stats.csv"))
bench_line = None
log = os.path.join(out, f"prof_{tag}_stats.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{\"metric\""):
            bench_line = json.loads(line)
            open(os.path.join(prof, f"{tag}_bench_under_rocprof.json"), "w").write(line)

pmc = {}
for kind in ("fetch", "write", "sq", "lds", "lane", "mixa", "mixb"):
    f = one(f"prof_{tag}_{kind}/**/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*", "", name)                       # drop the argument list, keep <template args>
        name = name.split("::")[-1] if "<" not in name else name[name.rfind("::", 0, name.index("<")) + 2:]
        name = name.replace(" ", "")
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        if "path_persistent" in name:
            pmc.setdefault("_dispatch_" + name, {"vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"],
                                         "scratch": r["Scratch_Size"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]})
    warm = int(((bench_line or {}).get("warmup")) or 0)
    for (name, ctr), vals in agg.items():
        # the path kernel's warm-up launches (cold caches, natural chunk order, cost reports per pixel) are kept apart:
        # bench.py's timed region does not contain them either
        timed = vals[warm:] if name.startswith("path_persistent") and len(vals) > warm else vals
        pmc.setdefault(name, {})[ctr] = {"mean_per_launch": sum(timed) / len(timed), "launches": len(timed),
                                         "per_launch_incl_warmup": vals if name.startswith("path_persistent") else None}

traffic = None
want_accel = (bench_line or {}).get("config", {}).get("kernel") == "persistent_clustered_list"
def is_path_kernel(name, accel):  # path_persistent_kernel<shading records in LDS, clustered list, flat-axis box test>
    if not name.startswith("path_persistent_kernel<"):
        return False
    args = name[name.index("<") + 1:name.rindex(">")].split(",")
    return len(args) >= 2 and (args[1] == "true") == accel
pk = next((k for k in pmc if is_path_kernel(k, want_accel)), None)
rk = next((k for k in pmc if k.startswith("resolve")), None)
if pk and "FETCH_SIZE" in pmc[pk] and "WRITE_SIZE" in pmc[pk]:
    # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
    # so a wide streaming read is doubled; WRITE_SIZE is exact.
    def bytes_of(k):
        return (2.0 * pmc[k]["FETCH_SIZE"]["mean_per_launch"] + pmc[k]["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    traffic = {"workload": (bench_line or {}).get("config", {}).get("workload", "cover_1200x800_100spp"),
               "bench_kernel": (bench_line or {}).get("config", {}).get("kernel", "persistent_flat_list"),
               "kernel": pk, "hbm_bytes_per_launch": bytes_of(pk),
               "fetch_kib_raw": pmc[pk]["FETCH_SIZE"]["mean_per_launch"], "write_kib_raw": pmc[pk]["WRITE_SIZE"]["mean_per_launch"],
               "resolve_kernel_hbm_bytes_per_launch": bytes_of(rk) if rk and "FETCH_SIZE" in pmc[rk] else None,
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads half of a wide stream)",
               "kernel_source_sha16": kernel_source_sha16()}
    json.dump(traffic, open(os.path.join(prof, f"traffic_{tag}.json"), "w"), indent=1)
for k, v in pmc.items():  # derived: share of the 64 lanes active in an average vector instruction
    if isinstance(v, dict) and "SQ_THREAD_CYCLES_VALU" in v and "SQ_ACTIVE_INST_VALU" in v:
        v["lane_occupancy_valu"] = v["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / (64.0 * max(1.0, v["SQ_ACTIVE_INST_VALU"]["mean_per_launch"]))
pmc["kernel_source_sha16"] = kernel_source_sha16()
json.dump(pmc, open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
# per-launch durations of the path kernel (the stats csv averages the warm-up launch in: cold caches, natural chunk order)
kt = one(f"prof_{tag}_stats/**/*_kernel_trace.csv")
if kt:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))
         if "path_persistent" in r["Kernel_Name"]]
    warm = int(((bench_line or {}).get("warmup")) or 0)
    timed = d[warm:] if len(d) > warm else d
    json.dump({"path_kernel_ms_per_launch": d, "warmup_launches": warm, "mean_ms_of_timed_launches": sum(timed) / max(1, len(timed)),
               "bench_event_ms_same_run": (bench_line or {}).get("config", {}).get("kernel_ms_rank0")},
              open(os.path.join(prof, f"{tag}_kernel_launches.json"), "w"), indent=1)
# (the reference's own kernels: tools/pmc_ch.sh + tools/pmc_ch_summary.py)
print("kernel stats:", open(os.path.join(prof, f"{tag}_kernel_stats.csv")).read() if ks else "missing")
print("traffic:", traffic)
