"""HBM-write rate of the reference's own kernels (CH05/CH06: 4 bytes per pixel) at sizes where they are bandwidth- rather than
launch-bound (RtStats.kernel_ms: HIP events around the kernel on its stream).  Two regimes, because this GPU lowers its clock a
few milliseconds into a run of arithmetic-heavy launches (a fill of the same buffer does not slow down):
  spaced     median of 12 launches 50 ms apart (what rounds 2 and 3 quoted: the first launches after a warm-up)
  sustained  median of launches 25..48 of 48 back to back
raytrace05's sphere is one flat colour: CH05 is quoted at the last size only."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vulkan_rtiow_amd as V
sizes = ((800, 608), (4096, 4096), (16384, 8192), (16384, 16384))
if len(sys.argv) > 1:
    sizes = tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[1:])
with V.Context(0) as ctx:
    cases = [(w, h, V.RT_MODE_CH06, "CH06") for w, h in sizes]
    if not os.environ.get("CH_BW_NO_CH05"):  # (tools/pmc_ch.sh: counters are grouped by grid size)
        cases.append((*sizes[-1], V.RT_MODE_CH05, "CH05"))
    for w, h, mode, name in cases:
        buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        prm = V.make_params(w, h, mode=mode)

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
        best = min(spaced + run)  # (small frames: launches 50 ms apart can find the clock ramped down by the idle time instead)
        print(f"{name} {w}x{h}: {ms:.4f} ms  {rate(ms):.1f} GB/s  ({rate(ms)/80:.1f}% of 8 TB/s)   sustained {sus:.4f} ms  {rate(sus):.1f} GB/s  ({rate(sus)/80:.1f}%)"
              f"   best {best:.4f} ms")
        del buf
