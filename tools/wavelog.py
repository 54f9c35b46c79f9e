"""The end of a small frame, wave by wave (a -DRTIOW_DEBUG_TIMELINE build writes one record per wave: Counters::tl_wave).
usage: RTIOW_LIB=$PWD/vulkan-rtiow_amd/librtiow_hip_tl.so python tools/wavelog.py [tile_count] [spp] [max_depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_rtiow_amd as V
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
w, h = 1200, 800
sph, mat = V.make_cover_scene(1, 11)
cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
log = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wavelog.txt")
os.environ["RTIOW_DEBUG_WAVELOG"] = log
with V.Context(0) as ctx:
    ctx.set_scene(sph, mat)
    prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, row_block=4, tile_rank=0, tile_count=G)
    for _ in range(4):
        ctx.render(cam, prm)
        st = ctx.stats()
    print(f"G={G} spp={spp} depth={depth}: {st.kernel_ms:.3f} ms, {st.segments} segments")
r = np.loadtxt(log, dtype=np.int64)
us = lambda t: t * 0.01
dry, sparse, end, tail_it, sp_it, live_dry, deepest, deep_end = (r[:, k] for k in range(1, 9))
sparse_paths, live_dry = live_dry >> 8, live_dry & 0xFF  # (paths summed over the wave's sparse iterations)
adopted = gave = np.zeros_like(live_dry)
mean_p = sparse_paths / np.maximum(sp_it, 1)
print(f"waves {len(r)}; dry: first {us(dry[dry > 0].min()):.0f} median {us(np.median(dry[dry > 0])):.0f} last {us(dry.max()):.0f} us; "
      f"end: median {us(np.median(end)):.0f} p90 {us(np.percentile(end, 90)):.0f} p99 {us(np.percentile(end, 99)):.0f} last {us(end.max()):.0f} us")
print(f"iterations after dry: mean {tail_it.mean():.1f} (sparse {sp_it.mean():.1f}); live at dry: mean {live_dry[live_dry < 1000].mean():.0f}")
main_it = tail_it - sp_it
tail_t = end - dry
ok = (dry > 0) & (tail_it > 0)
print(f"time from dry to end: mean {us(tail_t[ok].mean()):.0f} us = {main_it[ok].mean():.1f} main-loop iterations + {sp_it[ok].mean():.1f} sparse ones")
# a two-parameter fit: tail time = a * main iterations + b * sparse iterations
A = np.stack([main_it[ok], sp_it[ok]], 1).astype(float)
coef, *_ = np.linalg.lstsq(A, us(tail_t[ok]).astype(float), rcond=None)
print(f"least squares: {coef[0]:.1f} us per main-loop iteration after dry, {coef[1]:.1f} us per sparse iteration")
if adopted.any() or gave.any():
    per_it = us(end - sparse) / np.maximum(sp_it, 1)
    for name, m in (("gave their paths away", gave > 0), ("adopted", adopted > 0), ("neither", (gave == 0) & (adopted == 0) & (sp_it > 0))):
        if m.any():
            print(f"waves that {name}: {m.sum()}, sparse iterations {sp_it[m].mean():.1f}, {per_it[m].mean():.2f} us each, end median {us(np.median(end[m])):.0f} "
                  f"last {us(end[m].max()):.0f}; paths adopted {adopted[m].mean():.1f} given {gave[m].mean():.1f}")
order = np.argsort(-end)
print("the last waves:  wave   dry  sparse    end | iters after dry (sparse) | live at dry | deepest after dry | last 40+ path ended | mean paths per sparse iteration")
for k in order[:25]:
    print(f"               {r[k,0]:5d} {us(dry[k]):5.0f} {us(sparse[k]):6.0f} {us(end[k]):6.0f} | {tail_it[k]:4d} ({sp_it[k]:3d}) | {live_dry[k]:4d} +{adopted[k]:<3d} | {deepest[k]:3d} | {us(deep_end[k]):6.0f} | {mean_p[k]:5.1f}")
late = end > np.percentile(end, 90)
print(f"mean paths per sparse iteration: all waves {mean_p[sp_it > 0].mean():.1f}, the last tenth {mean_p[late].mean():.1f}, the last hundredth {mean_p[end > np.percentile(end, 99)].mean():.1f}; "
      f"correlation of a wave's end with its mean paths {np.corrcoef(end[sp_it > 0], mean_p[sp_it > 0])[0, 1]:.2f}, with its sparse iterations {np.corrcoef(end[sp_it > 0], sp_it[sp_it > 0])[0, 1]:.2f}")
print(f"the last tenth of the waves: deepest path after dry: mean {deepest[late].mean():.1f}, share with a 40+ path {np.mean(deepest[late] >= 40):.2f}; "
      f"all waves: {deepest.mean():.1f}, {np.mean(deepest >= 40):.2f}")
