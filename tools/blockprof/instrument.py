"""Basic-block execution counters for one gfx950 kernel, put into its ASSEMBLY (tools/blockprof/build.sh drives this).

rocprofv3's PC sampling and thread trace are not available on this GPU pool, and the SQ counters class instructions only coarsely
(fma / mul / add / cvt / trans / int: 61 % of the path kernel's vector instructions; the rest -- moves, selects, compares, v_readlane /
v_writelane of spilled scalars, DPP -- is one lump).  This tool counts EXACTLY instead: in front of every basic block of the kernel it
inserts one scalar memory atomic (s_atomic_add, which gfx950 executes: tools/blockprof/satomic_test.hip) on the block's own counter, so
that after a frame `count[b]` is the number of times a wave entered block b; tools/blockprof/report.py multiplies that by the block's
static instruction list.  The sums must reproduce the SQ_INSTS_* counters of the un-instrumented kernel, which is the tool's own check.

Inserted per block (wave-uniform, EXEC-independent, SCC / VCC / EXEC / M0 untouched):
    v_writelane_b32 vS, s0..s2 -> lanes 0..2      save three scalars in a spare vector register
    v_readlane_b32  s0, s1 <- vB lanes 0, 1       the counter array's address (put there by the kernel's first instructions)
    s_mov_b32 s2, 1 ; s_atomic_add s2, s[0:1], 128 * b ; s_waitcnt lgkmcnt(0)     (and one s_waitcnt lgkmcnt(0) in front of it all)
    v_readlane_b32  s0..s2 <- vS ; s_nop 4        restore (the nops: a VALU-written SGPR must age before VMEM / lane-select reads it)
vS, vB are the two vector registers after the kernel's own (.amdhsa_next_free_vgpr is raised by two).

With a seventh argument `lanes` every counter also sums the ACTIVE LANES at its block's entry (s_bcnt1_i32_b64 of EXEC into a 64-bit sum eight
bytes after the entry count; SCC is saved around it), and a block is also cut after every instruction that writes EXEC, so that every counted
stretch of instructions runs under one EXEC: report.py then gives the lane occupancy of the vector instructions region by region, which must
reproduce SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) of the un-instrumented kernel.

usage: instrument.py in.s out.s map.json <kernel-name-substring> <kernarg offset of PathArgs::counters> <offset of Counters::block_counts> [lanes]
"""
import json, re, sys

src, dst, map_path, key, off_counters, off_blocks = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5], 0), int(sys.argv[6], 0)
LANES = len(sys.argv) > 7 and sys.argv[7] == "lanes"
STRIDE = 128  # bytes between counters: one 128-byte line each, so that hot blocks do not queue behind each other in one L2 channel
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
name = lines[start].split(":")[0]
# the kernel's descriptor (it follows the code, in front of .Lfunc_end): registers in use
desc = next(i for i in range(start, len(lines)) if ".amdhsa_kernel " + name in lines[i])
end = next(i for i in range(start, desc) if lines[i].lstrip().startswith(".section"))  # (the code ends where .rodata begins)
nfv_i = next(i for i in range(desc, len(lines)) if ".amdhsa_next_free_vgpr" in lines[i])
nfv = int(lines[nfv_i].split()[-1])
acc_i = next(i for i in range(desc, len(lines)) if ".amdhsa_accum_offset" in lines[i])
vS, vB = nfv, nfv + 1
LS, LB = 0, 0      # first lane of the save area in vS, of the counter array's address in vB
# the two spare registers must not cost a launch its largest workgroup: a 1024-thread group is four waves per SIMD, 512 registers between them
meta = "\n".join(lines)
mwg = re.search(r"\.max_flat_workgroup_size:\s*(\d+)\s*\n\s*\.name:\s*" + re.escape(name), meta)
if mwg:
    waves_per_simd = (int(mwg.group(1)) + 255) // 256
    alloc = (nfv + 2 + 7) // 8 * 8
    if waves_per_simd * alloc > 512:
        # No room for two registers -- but a kernel that spills scalars keeps them in the lanes of a vector register of its own, which nothing
        # else touches (v_writelane / v_readlane only, from the first block to the last), and it rarely uses all 64: lanes 56-63 of such a
        # register do for the save area and the address.
        body = lines[start:desc]
        lane_use, other_use = {}, set()
        for l in body:
            mm = re.match(r"\s+v_writelane_b32\s+v(\d+),\s*\w+,\s*(\d+)", l) or re.match(r"\s+v_readlane_b32\s+\w+,\s*v(\d+),\s*(\d+)", l)
            if mm:
                lane_use.setdefault(int(mm.group(1)), set()).add(int(mm.group(2)))
                continue
            for r in re.findall(r"\bv(\d+)\b", l.split(";")[0]):
                other_use.add(int(r))
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l.split(";")[0]):
                other_use.update(range(int(a), int(b) + 1))
        spare = [r for r, ls in lane_use.items() if r not in other_use and max(ls) < 56]
        if spare:
            vS = vB = spare[0]
            LB, LS = 56, 58
            nfv -= 2  # (the descriptor keeps its register count)
            print(f"blockprof: no room for two registers more; lanes 56-63 of v{vS} (scalar spills in lanes 0-{max(lane_use[vS])}) take their place")
    if vS != vB and waves_per_simd * alloc > 512:
        sys.exit(f"blockprof: {name} may run {mwg.group(1)}-thread groups ({waves_per_simd} waves per SIMD) and uses {nfv} VGPRs: two more would not fit "
                 f"512 registers per SIMD lane -- the launch would be refused (HSA_STATUS_ERROR_INVALID_ISA).  Not instrumented.")
lines[nfv_i] = f"\t\t.amdhsa_next_free_vgpr {nfv + 2}"
lines[acc_i] = f"\t\t.amdhsa_accum_offset {(nfv + 2 + 3) // 4 * 4}"

INSTR = re.compile(r"^\s+([a-z][a-z0-9_]+)\b(.*)$")
BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch|s_setpc_b64|s_swappc_b64|s_endpgm)$")
LABEL = re.compile(r"^(\.LBB\d+_\d+|_Z\w+):")


def writes_exec(op, rest):
    dest = rest.split(",")[0].strip()
    return dest in ("exec", "exec_lo", "exec_hi") or "saveexec" in op or op.startswith("v_cmpx")


def counter_code_lanes(b):
    # s3: SCC while s_bcnt1 overwrites it; s[4:5]: the lanes of EXEC as a 64-bit addend
    sv = [f"\tv_writelane_b32 v{vS}, s{k}, {LS + k}" for k in range(6)]
    rs = [f"\tv_readlane_b32 s{k}, v{vS}, {LS + k}" for k in range(6)]
    return [f"\t; ---- blockprof: block {b}", "\ts_waitcnt lgkmcnt(0)"] + sv + [
            "\ts_cselect_b32 s3, 1, 0", f"\tv_readlane_b32 s0, v{vB}, {LB}", f"\tv_readlane_b32 s1, v{vB}, {LB + 1}", "\ts_mov_b32 s2, 1", "\ts_mov_b32 s5, 0",
            "\ts_nop 4", "\ts_bcnt1_i32_b64 s4, exec",
            f"\ts_atomic_add s2, s[0:1], 0x{b * STRIDE:x}", f"\ts_atomic_add_x2 s[4:5], s[0:1], 0x{b * STRIDE + 8:x}", "\ts_waitcnt lgkmcnt(0)",
            "\ts_cmp_lg_u32 s3, 0"] + rs + ["\ts_nop 4"]


def counter_code(b):
    if LANES:
        return counter_code_lanes(b)
    # (the first wait: a scalar load still in flight may have s0..s2 as its destination -- saved before it lands and restored after, they
    # would lose what it loaded)
    return [f"\t; ---- blockprof: block {b}", "\ts_waitcnt lgkmcnt(0)",
            f"\tv_writelane_b32 v{vS}, s0, {LS}", f"\tv_writelane_b32 v{vS}, s1, {LS + 1}", f"\tv_writelane_b32 v{vS}, s2, {LS + 2}",
            f"\tv_readlane_b32 s0, v{vB}, {LB}", f"\tv_readlane_b32 s1, v{vB}, {LB + 1}", "\ts_mov_b32 s2, 1", "\ts_nop 4",
            f"\ts_atomic_add s2, s[0:1], 0x{b * STRIDE:x}", "\ts_waitcnt lgkmcnt(0)",
            f"\tv_readlane_b32 s0, v{vS}, {LS}", f"\tv_readlane_b32 s1, v{vS}, {LS + 1}", f"\tv_readlane_b32 s2, v{vS}, {LS + 2}", "\ts_nop 4"]


out = lines[:start + 1]
blocks = []          # per block: list of [mnemonic, operands, source line]
cur_line = 0         # last .loc line seen
at_head = True       # the next instruction opens a block
prologue_done = False
for l in lines[start + 1:end]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur_line = int(m.group(2)) if int(m.group(1)) <= 1 else -(int(m.group(1)) * 100000 + int(m.group(2)))  # (< 0: a line of another file)
    if LABEL.match(l):
        at_head = True
        out.append(l)
        continue
    m = INSTR.match(l)
    if not m or l.lstrip().startswith((".", ";")):
        out.append(l)
        continue
    op, rest = m.group(1), m.group(2).split(";")[0].strip()
    if at_head:
        if not prologue_done:
            # s[0:1] = kernarg segment; s[90:91] are not initialised at wave start
            out += ["\t; ---- blockprof: address of Counters::block_counts into lanes 0, 1 of the spare register",
                    f"\ts_load_dwordx2 s[90:91], s[0:1], 0x{off_counters:x}", "\ts_waitcnt lgkmcnt(0)",
                    f"\ts_add_u32 s90, s90, 0x{off_blocks:x}", "\ts_addc_u32 s91, s91, 0",
                    f"\tv_writelane_b32 v{vB}, s90, {LB}", f"\tv_writelane_b32 v{vB}, s91, {LB + 1}", "\ts_nop 4"]
            prologue_done = True
        out += counter_code(len(blocks))
        blocks.append([])
        at_head = False
    blocks[-1].append([op, rest, cur_line])
    out.append(l)
    if BRANCH.match(op) or (LANES and writes_exec(op, rest)):
        at_head = True
out += lines[end:]
open(dst, "w").write("\n".join(out))
files = {int(mm.group(1)): mm.group(2) for mm in (re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]*)"', l) for l in lines) if mm}
json.dump({"files": files, "kernel": name, "stride_bytes": STRIDE, "blocks": blocks, "spare_vgprs": [vS, vB], "lanes": LANES}, open(map_path, "w"))
n_ins = sum(len(b) for b in blocks)
print(f"{name}: {len(blocks)} blocks, {n_ins} instructions, {len(blocks) * len(counter_code(0)) - len(blocks) + 8} inserted; spare registers v{vS}, v{vB}; counters need {len(blocks) * STRIDE} bytes")
