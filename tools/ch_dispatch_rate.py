"""Dispatch rate of the reference's own per-frame path (RTCHAP06/main.cpp:313-325: one compute dispatch of
raytrace06.comp over 800x608 per frame) through rtRenderUbo with a device destination: frames per second
of back-to-back asynchronous dispatches, and the host-destination (synchronous, copy included) rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vulkan_rtiow_amd as V

w, h = 800, 608
ubo = V.ubo_from_image(w, h)
with V.Context(0) as ctx:
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    for mode, name in ((V.RT_MODE_CH06, "CH06"), (V.RT_MODE_CH05, "CH05")):
        for _ in range(10):
            ctx.render_ubo_device(ubo, mode, buf.data_ptr(), w * 4, s.cuda_stream)
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.render_ubo_device(ubo, mode, buf.data_ptr(), w * 4, s.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name} {w}x{h} device destination: {dt * 1e6:.1f} us/frame ({1 / dt:.0f} frames/s), kernel {ctx.stats().kernel_ms * 1e3:.1f} us")
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.render_ubo(ubo, mode)
        dt = (time.perf_counter() - t0) / 200
        print(f"{name} {w}x{h} host destination (copy + sync included): {dt * 1e6:.1f} us/frame")
