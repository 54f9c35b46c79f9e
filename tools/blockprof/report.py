"""Executed-instruction mix of one frame from basic-block counters (tools/blockprof/instrument.py, build.sh, run.py).
usage: report.py map.json counts.txt [pmc.json kernel-key] [--lines N]
Prints (1) wave-instructions executed per class, beside the SQ_INSTS_* counters of the un-instrumented kernel where a profile is given
(the tool's own check: the sums must agree), (2) the same by source region (function / lambda / stage of rtiow_kernels.hip the
instruction's line table entry points at; an inlined function counts under its own name wherever it was inlined), (3) the hottest lines."""
import collections, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_lines = int(sys.argv[sys.argv.index("--lines") + 1]) if "--lines" in sys.argv else 30
m = json.load(open(args[0]))
rows_ = [l.split() for l in open(args[1]).read().split("\n") if l.strip()]
counts = [int(r[0]) for r in rows_]
lanes = [int(r[1]) for r in rows_] if m.get("lanes") and all(len(r) > 1 for r in rows_) else None  # active lanes summed over the entries
blocks = m["blocks"]
assert len(counts) >= len(blocks)

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op, rest):
    if op.startswith("v_"):
        if "readlane" in op or "writelane" in op or op.startswith("v_readfirstlane"):
            return "valu", "readlane/writelane"
        if "dpp" in op or "row_shr" in rest or "row_bcast" in rest or "quad_perm" in rest or "row_shl" in rest or "wave_shr" in rest or "row_ror" in rest or "row_mirror" in rest or "row_half_mirror" in rest or "row_newbcast" in rest:
            return "valu", "dpp"
        if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_mac_f32", "v_pk_fma_f32", "v_fmaak_f32", "v_fmamk_f32")):
            return "valu", "fma_f32"
        if op.startswith(("v_mul_f32", "v_pk_mul_f32", "v_mul_legacy_f32")):
            return "valu", "mul_f32"
        if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32")):
            return "valu", "add_f32"
        if op.startswith(TRANS):
            return "valu", "trans"
        if op.startswith("v_cvt_"):
            return "valu", "cvt"
        if op.startswith(("v_cmp_", "v_cmpx_")):
            return "valu", "cmp"
        if op.startswith("v_cndmask"):
            return "valu", "cndmask"
        if op.startswith(("v_mov_b", "v_accvgpr", "v_pk_mov")):
            return "valu", "mov"
        if op.startswith(("v_min", "v_max", "v_med3")) and "_f32" in op:
            return "valu", "minmax_f32"
        if op.startswith(("v_fract", "v_floor", "v_trunc", "v_rndne", "v_ceil", "v_ldexp", "v_frexp", "v_div_", "v_fma_f64", "v_add_f64", "v_mul_f64")):
            return "valu", "other_float"
        return "valu", "int/bit"
    if op.startswith("s_"):
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
            return "branch", "branch"
        if op.startswith(("s_load", "s_buffer_load", "s_atomic", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
            return "smem", "smem"
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_sethalt", "s_trap", "s_code_end", "s_incperflevel", "s_decperflevel", "s_ttracedata")):
            return "internal", op
        return "salu", "salu"
    if op.startswith("ds_"):
        k = "lds_atomic" if re.match(r"ds_(add|min|max|and|or|xor|inc|dec|cmpst|wrxchg|sub|rsub)", op) else ("lds_store" if op.startswith("ds_write") or op.startswith("ds_store") else ("lds_permute" if "permute" in op or "swizzle" in op else "lds_load"))
        return "lds", k
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem", "vmem"
    return "other", op


# ---- source regions: nearest preceding anchor of rtiow_kernels.hip ----
srcfile = os.path.join(root, "vulkan-rtiow_amd", "csrc", "rtiow_kernels.hip")
anchors = []  # (line, name)
kernel_body = False
for i, l in enumerate(open(srcfile).read().split("\n"), 1):
    mm = re.match(r"^(?:\[\[maybe_unused\]\]\s+)?(?:DI|HDI|static|__global__.*?\)\s*|template.*?>\s*(?:DI|HDI))?\s*(?:[\w:<>,\s\*&]+?)\s+(\w+)\s*\([^;]*$", l)
    if re.match(r"^(DI|HDI|\[\[maybe_unused\]\] DI|__global__|void path_persistent_kernel)", l):
        mm2 = re.search(r"(\w+)\s*\(", l.split("__launch_bounds__")[0] if "__launch_bounds__" not in l else "")
        nm = re.findall(r"(\w+)\s*\(", l)
        nm = [x for x in nm if x not in ("__launch_bounds__", "__attribute__", "amdgpu_waves_per_eu", "defined")]
        if nm:
            anchors.append((i, nm[0]))
    mm = re.match(r"^\s+(?:\[\[maybe_unused\]\]\s+)?auto (\w+) = \[&\]", l)
    if mm:
        anchors.append((i, "λ " + mm.group(1)))
    mm = re.match(r"^\s+// ---- (.*?)\s*-*\s*$", l)
    if mm:
        anchors.append((i, "§ " + mm.group(1)[:40]))
anchors.sort()
names = [a[1] for a in anchors]
firsts = [a[0] for a in anchors]
import bisect


HELPERS = {"fma_", "dot3", "mk", "psqrt", "precip", "psqrt_signed", "lane_rank", "keep_b128", "meta_of", "meta_entry", "meta_depth", "meta_line"}
files = {int(k): v for k, v in m.get("files", {}).items()}


def region_of(line):
    if line < 0:  # another file
        return "[" + files.get((-line) // 100000, "file %d" % ((-line) // 100000)) + "]"
    if line == 0:
        return "?"
    k = bisect.bisect_right(firsts, line) - 1
    return names[k] if k >= 0 else "?"


def context_regions(ins):
    """Region of every instruction of a block; compiler-generated code (line 0) and one-line helpers inlined everywhere (fma_, dot3 ...)
    count under the region of the nearest instruction of the block that has one (the one before it, else the one after it)."""
    own = [region_of(line) for _, _, line in ins]
    res = list(own)
    last = None
    for i, r in enumerate(own):
        if r != "?" and r not in HELPERS:
            last = r
        elif last is not None:
            res[i] = last
    nxt = None
    for i in range(len(own) - 1, -1, -1):
        if own[i] != "?" and own[i] not in HELPERS:
            nxt = own[i]
        elif res[i] == own[i] and nxt is not None:
            res[i] = nxt
    return res


by_class = collections.Counter()
by_kind = collections.Counter()
by_region = collections.defaultdict(collections.Counter)
by_line = collections.defaultdict(collections.Counter)
static_kind = collections.Counter()
rw_blocks = []
for b, ins in enumerate(blocks):
    c = counts[b]
    ctx_r = context_regions(ins)
    n_rw = sum(1 for op, rest, line in ins if classify(op, rest)[1] == "readlane/writelane")
    if c and n_rw:
        rw_blocks.append((c * n_rw, c, n_rw, b, collections.Counter(ctx_r).most_common(2), sorted({l for _, _, l in ins if l > 0})[:1] + sorted({l for _, _, l in ins if l > 0})[-1:]))
    for k_i, (op, rest, line) in enumerate(ins):
        cl, kind = classify(op, rest)
        static_kind[(cl, kind)] += 1
        if c == 0:
            continue
        by_class[cl] += c
        by_kind[(cl, kind)] += c
        r = ctx_r[k_i]
        by_region[r][cl] += c
        if lanes and cl == "valu" and kind != "readlane/writelane":  # (lane-select instructions run whatever EXEC is)
            by_region[r]["valu_x"] += c
            by_region[r]["lanes"] += lanes[b]
            by_line[line]["valu_x"] += c
            by_line[line]["lanes"] += lanes[b]
            by_class["valu_x"] += c
            by_class["lanes"] += lanes[b]
        if kind == "readlane/writelane":
            by_region[r]["rw"] += c
        by_line[line][cl] += c

pmc = None
if len(args) > 3:
    d = json.load(open(args[2]))
    pmc = {k: v["mean_per_launch"] for k, v in d[args[3]].items() if isinstance(v, dict)}
PMC_OF = {"valu": "SQ_INSTS_VALU", "salu": "SQ_INSTS_SALU", "branch": "SQ_INSTS_BRANCH", "smem": "SQ_INSTS_SMEM", "lds": "SQ_INSTS_LDS", "vmem": "SQ_INSTS_VMEM"}
PMC_KIND = {"fma_f32": "SQ_INSTS_VALU_FMA_F32", "mul_f32": "SQ_INSTS_VALU_MUL_F32", "add_f32": "SQ_INSTS_VALU_ADD_F32", "trans": "SQ_INSTS_VALU_TRANS_F32", "cvt": "SQ_INSTS_VALU_CVT",
            "lds_atomic": "SQ_INSTS_LDS_ATOMIC", "lds_store": "SQ_INSTS_LDS_STORE"}
print(f"kernel {m['kernel']}\n{len(blocks)} basic blocks, {sum(len(b) for b in blocks)} instructions; {sum(1 for b in range(len(blocks)) if counts[b])} blocks entered, {sum(counts[:len(blocks)]):.4g} block entries\n")
print("(1) wave-instructions executed per frame, by class" + ("   [SQ counter of the un-instrumented kernel, ratio]" if pmc else ""))
for cl in ("valu", "salu", "branch", "smem", "lds", "vmem", "internal"):
    line = f"  {cl:9s} {by_class[cl]:14.5g}"
    if pmc and cl in PMC_OF and PMC_OF[cl] in pmc:
        line += f"   [{PMC_OF[cl]} {pmc[PMC_OF[cl]]:.5g}, {by_class[cl] / max(1.0, pmc[PMC_OF[cl]]):.4f}]"
    print(line)
tv = max(1, by_class["valu"])
print("\n    vector instructions by kind            executed   share of VALU   static")
for (cl, kind), c in sorted(by_kind.items(), key=lambda kv: -kv[1]):
    if cl != "valu":
        continue
    line = f"    {kind:28s} {c:14.5g}   {c / tv:6.3f}        {static_kind[(cl, kind)]:5d}"
    if pmc and kind in PMC_KIND and PMC_KIND[kind] in pmc:
        line += f"   [{PMC_KIND[kind]} {pmc[PMC_KIND[kind]]:.5g}, {c / max(1.0, pmc[PMC_KIND[kind]]):.4f}]"
    print(line)
print("\n    LDS / scalar / other kinds")
for (cl, kind), c in sorted(by_kind.items(), key=lambda kv: -kv[1]):
    if cl in ("valu",):
        continue
    line = f"    {cl + ':' + kind:28s} {c:14.5g}                  {static_kind[(cl, kind)]:5d}"
    if pmc and kind in PMC_KIND and PMC_KIND[kind] in pmc:
        line += f"   [{PMC_KIND[kind]} {pmc[PMC_KIND[kind]]:.5g}, {c / max(1.0, pmc[PMC_KIND[kind]]):.4f}]"
    print(line)
if lanes:
    occ = by_class["lanes"] / (64.0 * max(1, by_class["valu_x"]))
    line = f"\n    lane occupancy of the vector instructions (active lanes at the entry of each stretch under one EXEC / 64): {occ:.4f}"
    print(line)
    if pmc and "SQ_THREAD_CYCLES_VALU" in pmc:
        # the counter takes a v_readlane / v_writelane / v_readfirstlane (no EXEC: one lane read or written) as ONE thread
        n_rw = by_kind[("valu", "readlane/writelane")]
        occ_hw = (by_class["lanes"] + n_rw) / (64.0 * (by_class["valu_x"] + n_rw))
        ref = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
        print(f"    with the lane-select instructions as one lane each, as the SQ counters take them: {occ_hw:.4f}   [SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) {ref:.4f}, ratio {occ_hw / ref:.4f}]")
print("\n(2) by source region (rtiow_kernels.hip; inlined code counts under its own function)      valu    share   readlane/writelane     salu      lds")
for r, c in sorted(by_region.items(), key=lambda kv: -kv[1]["valu"]):
    if c["valu"] + c["salu"] < 0.002 * tv:
        continue
    extra = f"   lanes {c['lanes'] / (64.0 * c['valu_x']):5.3f}  idle lane-slots {(64.0 * c['valu_x'] - c['lanes']) / (64.0 * tv):6.4f} of all" if lanes and c["valu_x"] else ""
    print(f"    {r:52s} {c['valu']:12.4g}  {c['valu'] / tv:6.3f}   {c['rw']:12.4g}     {c['salu']:12.4g} {c['lds']:10.4g}{extra}")
print(f"\n(3) the {n_lines} hottest source lines (vector instructions executed)")
src = open(srcfile).read().split("\n")
for line, c in sorted(by_line.items(), key=lambda kv: -kv[1]["valu"])[:n_lines]:
    text = src[line - 1].strip()[:110] if 0 < line <= len(src) else ""
    extra = f" lanes {c['lanes'] / (64.0 * c['valu_x']):5.3f} |" if lanes and c["valu_x"] else ""
    print(f"    {line:5d} {c['valu']:12.4g} {c['valu'] / tv:6.3f}  salu {c['salu']:10.4g}  |{extra} {region_of(line)[:24]:24s} | {text}")

print("\n(4) v_readlane / v_writelane / v_readfirstlane by basic block: executed, block entries x instructions in it, block, its regions, its source lines")
for tot, c, n_rw, b, regs, span in sorted(rw_blocks, reverse=True)[:40]:
    print(f"    {tot:12.4g} = {c:10d} x {n_rw:3d}   block {b:4d} ({len(blocks[b]):4d} instructions)  {', '.join(f'{r} ({k})' for r, k in regs):60s} lines {span}")

if lanes:
    print("\n(4b) idle lane-slots by stretch (vector instructions x (64 x entries - active lanes)): share of all lane-slots, occupancy, entries, vector instructions, block, regions, lines")
    rows = []
    for b, ins in enumerate(blocks):
        n = sum(1 for op, rest, line in ins if classify(op, rest)[0] == "valu" and classify(op, rest)[1] != "readlane/writelane")
        if n and counts[b]:
            rows.append((n * (64 * counts[b] - lanes[b]), lanes[b] / (64.0 * counts[b]), counts[b], n, b))
    for lost, occ_b, c, n, b in sorted(rows, reverse=True)[:40]:
        regs = collections.Counter(context_regions(blocks[b])).most_common(2)
        ls = sorted({l for _, _, l in blocks[b] if l > 0})
        print(f"    {lost / (64.0 * tv):7.4f}  {occ_b:5.3f}  {c:10d} x {n:3d}   block {b:4d}  {', '.join(f'{r} ({k})' for r, k in regs):60s} lines {ls[:1] + ls[-1:]}")

# --kind K: the blocks that execute most instructions of one kind (mov, cndmask, cmp, int/bit, salu ...); --block B: a block's listing
if "--kind" in sys.argv:
    want = sys.argv[sys.argv.index("--kind") + 1]
    rows = []
    for b, ins in enumerate(blocks):
        n = sum(1 for op, rest, line in ins if classify(op, rest)[1] == want or classify(op, rest)[0] == want)
        if n and counts[b]:
            rows.append((n * counts[b], counts[b], n, b))
    print(f"\n(5) '{want}' by basic block")
    for tot, c, n, b in sorted(rows, reverse=True)[:30]:
        regs = collections.Counter(context_regions(blocks[b])).most_common(2)
        ls = sorted({l for _, _, l in blocks[b] if l > 0})
        print(f"    {tot:12.4g} = {c:10d} x {n:3d}   block {b:4d} ({len(blocks[b]):4d} instructions)  {', '.join(f'{r} ({k})' for r, k in regs):60s} lines {ls[:1] + ls[-1:]}")
if "--block" in sys.argv:
    for b in [int(x) for x in sys.argv[sys.argv.index("--block") + 1].split(",")]:
        print(f"\nblock {b}: entered {counts[b]} times")
        for op, rest, line in blocks[b]:
            print(f"    {line:6d}  {op} {rest}")
