"""Straggler waves by workgroup, from the wave log tools/wavelog.py leaves in gpurun_out/wavelog.txt (timeline build)."""
import numpy as np, sys, os
r = np.loadtxt(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wavelog.txt"), dtype=np.int64)
wpb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
wave, dry, end = r[:, 0], r[:, 1] * 0.01, r[:, 3] * 0.01
blk = wave // wpb
ub = np.unique(blk)
be = np.array([end[blk == b].max() for b in ub])
b2 = np.array([np.sort(end[blk == b])[-2] if (blk == b).sum() > 1 else end[blk == b].max() for b in ub])
print(f"waves {len(r)} in {len(ub)} workgroups; wave end: mean {end.mean():.0f} median {np.median(end):.0f} p95 {np.percentile(end, 95):.0f} last {end.max():.0f} us")
print(f"workgroup end: median {np.median(be):.0f} p90 {np.percentile(be, 90):.0f} last {be.max():.0f}; its second-latest wave: median {np.median(b2):.0f} p90 {np.percentile(b2, 90):.0f} last {b2.max():.0f}")
print(f"busy share of the waves up to the frame's end: {end.sum() / (len(end) * end.max()):.3f}")
for b in np.argsort(-be)[:8]:
    e = np.sort(end[blk == ub[b]]); d = np.sort(dry[blk == ub[b]])
    print(f"  workgroup {ub[b]:4d}: last ends {np.round(e[-5:]).astype(int)}  last drys {np.round(d[-5:]).astype(int)}")
