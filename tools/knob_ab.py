"""Frame time under settings of the RTIOW_DEBUG_* knobs, interleaved in one process on the knobs build (one context per
setting: its chunk order is made under that setting).  usage: knob_ab.py [--grid 32 --width 3840 --height 2160] [--tile G] [--rank r] [--spp N] [--rounds R] "A=1,B=2" "default" ..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTIOW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-rtiow_amd", "librtiow_hip_knobs.so"))  # the RTIOW_DEBUG_* knobs exist in this build only
import vulkan_rtiow_amd as V
argv = sys.argv[1:]
def opt(name, default):
    if name in argv:
        i = argv.index(name); v = argv[i + 1]; del argv[i:i + 2]; return int(v)
    return default
G, spp, rounds, rank = opt("--tile", 1), opt("--spp", 100), opt("--rounds", 9), opt("--rank", 0)
grid, w, h = opt("--grid", 11), opt("--width", 1200), opt("--height", 800)
settings = argv or ["default"]
sph, mat = V.make_cover_scene(1, grid)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
prm = V.make_params(w, h, spp=spp, max_depth=50, seed=1, row_block=4, tile_rank=rank, tile_count=G)
def apply(s):
    for k in [k for k in os.environ if k.startswith("RTIOW_DEBUG_")]:
        del os.environ[k]
    if s != "default":
        for kv in s.split(","):
            k, v = kv.split("=")
            os.environ["RTIOW_DEBUG_" + k] = v
ctxs = {}
for s in settings:
    apply(s)
    c = V.Context(0); c.set_scene(sph, mat); ctxs[s] = c
times = {s: [] for s in settings}
for r in range(rounds + 3):
    for s in settings:
        apply(s)
        ctxs[s].render(cam, prm)
        if r >= 3:
            times[s].append(ctxs[s].stats().kernel_ms)
for s in settings:
    print(f"{s:36s} median {statistics.median(times[s]):7.3f} ms  min {min(times[s]):7.3f}")
