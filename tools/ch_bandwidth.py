"""HBM-write rate of the reference's own kernels (CH05/CH06: 4 bytes per pixel) at sizes where they are bandwidth- rather than
launch-bound (RtStats.kernel_ms: HIP events around the kernel on its stream).  Two regimes, because this GPU lowers its clock a
few milliseconds into a run of arithmetic-heavy launches (a fill of the same buffer does not slow down):
  spaced     median of 12 launches 50 ms apart (what rounds 2 and 3 quoted: the first launches after a warm-up)
  sustained  median of launches 25..48 of 48 back to back"""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vulkan_rtiow_amd as V
sizes = ((800, 608), (4096, 4096), (16384, 8192), (16384, 16384))
if len(sys.argv) > 1:
    sizes = tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[1:])
with V.Context(0) as ctx:
    for w, h in sizes:
        buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        prm = V.make_params(w, h, mode=V.RT_MODE_CH06)
        def launch():
            ctx.render_device(None, prm, buf.data_ptr(), w * 4, s.cuda_stream)
            return ctx.stats().kernel_ms
        launch(); launch()
        spaced = []
        for _ in range(12):
            time.sleep(0.05)
            spaced.append(launch())
        time.sleep(0.2)
        run = [launch() for _ in range(48)]
        ms, sus = statistics.median(spaced), statistics.median(run[24:])
        rate = lambda t: w * h * 4 / t / 1e6
        print(f"CH06 {w}x{h}: {ms:.4f} ms  {rate(ms):.1f} GB/s  ({rate(ms)/80:.1f}% of 8 TB/s)   sustained {sus:.4f} ms  {rate(sus):.1f} GB/s  ({rate(sus)/80:.1f}%)")
