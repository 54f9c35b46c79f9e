"""Register / spill report of the kernels from the assembly `make -C vulkan-rtiow_amd/csrc asm` leaves behind (the
metadata notes at the end of each .s file).  usage: asm_report.py [file.s ...]"""
import os, re, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "csrc")
files = sys.argv[1:] or [os.path.join(root, n) for n in ("rtiow_kernels.s", "rtiow_kernels_small.s", "rtiow_kernels_large.s")]
keys = (".vgpr_count", ".vgpr_spill_count", ".sgpr_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".max_flat_workgroup_size")
for f in files:
    text = open(f).read()
    meta = text[text.rfind("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        vals = {k: int(re.search(re.escape(k) + r":\s+(\d+)", blk).group(1)) for k in keys}
        short = re.sub(r"_ZN5rtiow12_GLOBAL__N_1\d+", "", name)
        short = re.sub(r"ENS_8PathArgsENS_11PersistArgsE|ENS_6ChArgsE", "", short)
        # scratch instructions inside the kernel's body
        body = text[text.find(name + ":"):]
        body = body[:body.find(".Lfunc_end")]
        ld = len(re.findall(r"scratch_load", body)); st = len(re.findall(r"scratch_store", body))
        rl = len(re.findall(r"v_readlane_b32", body)); wl = len(re.findall(r"v_writelane_b32", body))
        print(f"{os.path.basename(f):24s} {short:48s} vgpr {vals['.vgpr_count']:3d} spill {vals['.vgpr_spill_count']:3d}  sgpr {vals['.sgpr_count']:3d} spill {vals['.sgpr_spill_count']:3d}"
              f"  scratch {vals['.private_segment_fixed_size']:3d} B  ld/st {ld}/{st}  readlane/writelane {rl}/{wl}  lines {body.count(chr(10))}")
