"""HBM-write rate of the reference's own kernels (CH05/CH06: 4 bytes per pixel, ~60 flops) at sizes
where they are bandwidth- rather than launch-bound."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vulkan_rtiow_amd as V
with V.Context(0) as ctx:
    for w, h in ((800, 608), (4096, 4096), (16384, 8192), (16384, 16384)):
        buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        prm = V.make_params(w, h, mode=V.RT_MODE_CH06)
        ts = []
        for _ in range(6):
            ctx.render_device(None, prm, buf.data_ptr(), w * 4, s.cuda_stream)
            ts.append(ctx.stats().kernel_ms)
        ms = statistics.median(ts[1:])
        print(f"CH06 {w}x{h}: {ms:.4f} ms  {w*h*4/ms/1e6:.1f} GB/s  ({w*h*4/ms/1e6/8000*100:.1f}% of 8 TB/s)")
