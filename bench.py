#!/usr/bin/env python3
"""bench.py — Mray/s of the path-tracing hot path on the BASELINE cover scene.

A step is one frame: one pass of the hot path (ray generation, ray-sphere list
intersection, scatter, multi-sample accumulate, RGBA8 store) over the
1200x800x100spp cover scene (~485 spheres, depth 50), synthetic scene from the
shared counter RNG, inputs resident in HBM.  With --gpus N > 1 the SAME frame is
row-tiled over N ranks (block-cyclic) and gathered to rank 0 over RCCL: total
work is fixed, so scaling is "strong".

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector (= f32 MFMA rate)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.53  # wave instructions / s: profiles/r01_ubench_valu.txt (v_fma_f32, 8 blocks/CU)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
FLOPS_PER_TEST = 16             # oc 3, hb 5, cc 6, disc 2 (unit direction: a == 1)
FLOPS_PER_SEGMENT_SHADE = 60

KERNEL_NAMES = {1: "pixel_per_lane", 2: "persistent_flat_list", 3: "persistent_clustered_list"}

WORKLOADS = {
    # name: (scene, grid_half, width, height, spp, depth)
    "cover_1200x800_100spp": ("cover", 11, 1200, 800, 100, 50),
    "cover_1200x800_500spp": ("cover", 11, 1200, 800, 500, 50),
    "three_400x225_100spp": ("three", 0, 400, 225, 100, 50),
    "cover4096_3840x2160_1024spp": ("cover", 32, 3840, 2160, 1024, 50),
    "cover4096_3840x2160_64spp": ("cover", 32, 3840, 2160, 64, 50),   # (C5's scene and frame at a sample count the PMC passes can afford)
    "cover_300x200_10spp": ("cover", 11, 300, 200, 10, 50),
}


def build_scene(V, scene, grid_half, w, h):
    if scene == "three":
        sph, mat = V.make_three_sphere_scene(False)
        cam = V.camera_from_ubo(V.ubo_from_image(w, h))
    else:
        sph, mat = V.make_cover_scene(1, grid_half)
        cam = V.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    return sph, mat, cam


def usable_cores(reported):
    """Host cores this process may actually use: min(OpenMP's count, affinity mask, cgroup quota)."""
    n = reported
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


KERNEL_SOURCES = ("vulkan-rtiow_amd/csrc/rtiow_kernels.hip", "vulkan-rtiow_amd/csrc/rtiow_device.h",
                  "vulkan-rtiow_amd/csrc/rtiow_rng.h", "vulkan-rtiow_amd/csrc/Makefile")


def kernel_source_sha16():
    """Identity of the kernels a PMC profile was taken with: sha256 over the kernel sources and their build flags.
    tools/profile_summary.py records it in profiles/r*_pmc.json; figures derived from such a file are printed only
    while the tree still holds the same kernels (a stale instruction count over a fresh time would be no measurement)."""
    import hashlib
    hsh = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()[:16]


def golden_frame_check(workload, frame_bytes, segments, tile_rows=None):
    """The timed frame against the oracle's CRC / segment count of the same frame (tests/golden/frame_golden.json, made
    in the build container by tests/golden/make_frame_golden.py).  "golden": both equal; "DIFFERS": not; None: no golden
    for this workload."""
    import zlib
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_golden.json"))).get(workload)
    except (OSError, ValueError):
        gold = None
    if not gold:
        return None, None
    if "crc32" in gold:
        ok = zlib.crc32(frame_bytes) == gold["crc32"] and int(segments) == gold["segments"]
        return ("golden" if ok else "DIFFERS"), f"whole frame: crc32 + segment count of the oracle's frame ({gold['segments']} segments)"
    if "rows" in gold and len(frame_bytes) == gold["width"] * gold["height"] * 4:  # a frame the CPU cannot finish: some of its rows
        pitch = gold["width"] * 4
        ok = all(zlib.crc32(frame_bytes[int(r) * pitch:(int(r) + 1) * pitch]) == g["crc32"] for r, g in gold["rows"].items())
        return ("golden" if ok else "DIFFERS"), "rows " + ", ".join(sorted(gold["rows"], key=int)) + ": crc32 of the oracle's rows"
    return None, None


def cpu_baseline(V, sph, mat, cam, w, h, spp, depth, chunk, target_s=15.0):
    """Times the CPU oracle (kind "port": the reference has no CPU path) on a bounded sample:
    every k-th row of the same frame, all host cores, sized for ~target_s seconds; and four rows of it on ONE thread
    (SURVEY 8d / BASELINE.md 3 ask for both)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind
    orc = oracle_bind.load()
    cores = usable_cores(orc.num_procs())
    # probe: 4 evenly spread rows
    probe_rows = 4
    stride = max(1, h // probe_rows)
    prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, chunk_spp=chunk, row_block=1,
                        tile_rank=0, tile_count=stride)
    t0 = time.perf_counter()
    img, _ = orc.render(sph, mat, cam, prm, nthreads=cores)
    probe_t = time.perf_counter() - t0
    per_row = probe_t / max(1, img.shape[0])
    rows = int(min(h, max(probe_rows, target_s / max(per_row, 1e-9))))
    stride = max(1, h // rows)
    prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, chunk_spp=chunk, row_block=1,
                        tile_rank=0, tile_count=stride)
    t0 = time.perf_counter()
    img, segs = orc.render(sph, mat, cam, prm, nthreads=cores)
    dt = time.perf_counter() - t0
    n_rows = img.shape[0]
    # one thread: four evenly spread rows (sky, horizon, field, foreground), or fewer if a row takes long
    rows1 = 4 if per_row * cores * 4 < 2.0 * target_s else 1
    prm1 = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, chunk_spp=chunk, row_block=1,
                         tile_rank=0, tile_count=max(1, h // rows1))
    t0 = time.perf_counter()
    img1, segs1 = orc.render(sph, mat, cam, prm1, nthreads=1)
    dt1 = time.perf_counter() - t0
    return {
        "value": n_rows * w * spp * depth / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
        "sample": f"every {stride}th row ({n_rows} of {h} rows) of the same frame at full spp/depth, "
                  f"{dt:.1f} s, OpenMP dynamic over rows",
        "segments_per_s": segs / dt,
        "one_thread": {"value": img1.shape[0] * w * spp * depth / dt1 / 1e6, "unit": "Mray/s", "cores": 1,
                       "sample": f"every {max(1, h // rows1)}th row ({img1.shape[0]} of {h} rows) at full spp/depth, {dt1:.1f} s",
                       "segments_per_s": segs1 / dt1},
    }


def reference_shader_rates(V, torch, ctx, stream):
    """The reference's own two compute shaders (CH05 / CH06: rtRenderUbo's path) where bytes matter -- a 16384^2
    frame, 1 GiB, 4 bytes per pixel written once -- and at the reference's own 800 x 608: kernel time from the library's HIP events
    (RtStats.kernel_ms), the median of launches 50 ms apart and of the second half of 32 launches back to back (the clock drops a few
    milliseconds into a run of such launches: DESIGN 4.1).  Untimed extra.  Cross-check on the GPU: the row kernel's frame equals the
    statement-for-statement tiled form's (RtParams.kernel = 1) at 2048 x 1024, both shaders."""
    import statistics
    out = {}
    same = True
    for mode in (V.RT_MODE_CH05, V.RT_MODE_CH06):
        a = torch.empty((1024, 2048), dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        ctx.render_device(None, V.make_params(2048, 1024, mode=mode), a.data_ptr(), 2048 * 4, stream.cuda_stream)
        ctx.render_device(None, V.make_params(2048, 1024, mode=mode, kernel=1), b.data_ptr(), 2048 * 4, stream.cuda_stream)
        ctx.synchronize()
        same = same and bool(torch.equal(a, b))
    out["row_kernel_equals_tiled_form"] = same
    for name, mode, w, h in (("ch06_16384x16384", V.RT_MODE_CH06, 16384, 16384), ("ch05_16384x16384", V.RT_MODE_CH05, 16384, 16384),
                             ("ch06_800x608", V.RT_MODE_CH06, 800, 608)):
        buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
        prm = V.make_params(w, h, mode=mode)

        def launch():
            ctx.render_device(None, prm, buf.data_ptr(), w * 4, stream.cuda_stream)
            return ctx.stats().kernel_ms

        launch(), launch()
        spaced = []
        for _ in range(8):
            time.sleep(0.05)
            spaced.append(launch())
        time.sleep(0.2)
        run = [launch() for _ in range(32)]
        ms, sus = statistics.median(spaced), statistics.median(run[16:])
        out[name] = {"kernel_ms": ms, "gb_per_s": w * h * 4 / ms / 1e6, "hbm_frac": w * h * 4 / ms / 1e6 / HBM_PEAK_GBS,
                     "sustained_kernel_ms": sus, "sustained_gb_per_s": w * h * 4 / sus / 1e6}
        del buf
    return out


def cold_and_moving_frames(V, torch, dev, device_id, sph, mat, cam, prm, w, h, timed_frame, warm_ctx, stream, scratch):
    """Outside the timed region, N = 1: what the steady state of a static view leaves out.
    first_frame_ms: a FRESH context's first frame of the same view -- scene just uploaded, no chunk order yet, counters set by a
    memset -- checked against the timed frame.  moving_camera_ms_per_step: 16 frames of an orbit around the look-at point, a new
    camera every frame, on the warm context (whose chunk order then belongs to the PREVIOUS view), by the wall clock around the
    loop; the same loop with the camera standing still beside it; every orbit frame checked against a single-shot render of the
    same view by a fresh context."""
    import math
    import time
    res = {}
    with V.Context(device_id) as fresh:
        t0 = time.perf_counter()
        fresh.set_scene(sph, mat)
        res["scene_upload_ms"] = (time.perf_counter() - t0) * 1e3
        ss = fresh.scene_stats()
        res["scene_build_ms"] = ss.scene_build_ms
        res["cluster_builds"] = int(ss.cluster_builds)
        cold = torch.zeros((h, w), dtype=torch.int32, device=dev)
        again = torch.zeros((h, w), dtype=torch.int32, device=dev)
        # the first two frames back to back on one stream, timed by events on that stream (an idle gap between them -- a host-side
        # comparison, say -- lets the clock drop and costs the next frame 5-10 % whatever its chunk order: tools/first_frames.py)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        with torch.cuda.stream(stream):
            ev[0].record()
            fresh.render_device(cam, prm, cold.data_ptr(), w * 4, stream.cuda_stream)
            ev[1].record()
            fresh.render_device(cam, prm, again.data_ptr(), w * 4, stream.cuda_stream)
            ev[2].record()
        torch.cuda.synchronize()
        res["first_frame_ms"] = ev[0].elapsed_time(ev[1])
        res["second_frame_ms"] = ev[1].elapsed_time(ev[2])
        same = bool(torch.equal(cold, timed_frame.to(dev))) and bool(torch.equal(again, cold))
        res["first_frame_vs_timed_frame"] = "identical" if same else "DIFFERS"
    # what one frame costs the HOST: the rtRender call itself (ctypes, three event records, the launch), returning before the GPU is done
    calls = []
    for k in range(48):
        t0 = time.perf_counter()
        warm_ctx.render_device(cam, prm, scratch.data_ptr(), w * 4, stream.cuda_stream)
        calls.append(time.perf_counter() - t0)
        if k % 8 == 7:
            torch.cuda.synchronize()
    calls.sort()
    res["host_enqueue_us_per_frame"] = calls[len(calls) // 2] * 1e6
    n = 16
    r0 = math.hypot(13.0, 3.0)
    cams = [V.make_camera((r0 * math.cos(a), 2.0, r0 * math.sin(a)), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
            for a in (math.atan2(3.0, 13.0) + 2.0 * math.pi * k / 96.0 for k in range(1, n + 1))]  # 3.75 degrees a frame
    bufs = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(n)]

    def loop(cameras):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, c in enumerate(cameras):
            warm_ctx.render_device(c, prm, bufs[k].data_ptr(), w * 4, stream.cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / len(cameras) * 1e3

    loop([cam] * 4)
    res["static_camera_same_loop_ms_per_step"] = loop([cam] * n)
    res["moving_camera_ms_per_step"] = loop(cams)
    bad = 0  # every orbit frame against a single-shot render of the same view by a fresh context
    with V.Context(device_id) as fresh:
        fresh.set_scene(sph, mat)
        for k, c in enumerate(cams):
            fresh.render_device(c, prm, scratch.data_ptr(), w * 4, stream.cuda_stream)
            torch.cuda.synchronize()
            bad += 0 if bool(torch.equal(scratch, bufs[k])) else 1
    res["moving_camera_frames_vs_single_shot"] = "identical" if bad == 0 else f"{bad} of {n} DIFFER"
    # ... the orbit's views are other frames than the timed one (more of the glass ball, other path lengths): what moving costs is
    # read against the SAME sixteen views, each standing still until its own chunk order is there
    settled = 0.0
    for c in cams:
        loop([c] * 5)
        settled += loop([c] * 3)
    res["moving_camera_views_standing_still_ms_per_step"] = settled / n
    res["moving_camera_penalty"] = res["moving_camera_ms_per_step"] / (settled / n) - 1.0
    res["moving_camera"] = f"{n} frames of an orbit around the look-at point, 3.75 degrees a frame, a new RtCamera every frame"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cover_1200x800_100spp", choices=sorted(WORKLOADS))
    ap.add_argument("--chunk-spp", type=int, default=10)
    ap.add_argument("--row-block", type=int, default=4)
    ap.add_argument("--kernel", type=int, default=0,
                    help="0 library default (clustered list from 64 spheres on), 2 persistent flat list, 3 persistent clustered list, 1 one lane per pixel")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="frames rendered concurrently on separate contexts/streams for the TIMED loop, as the reference "
                         "keeps one compute fence per swapchain image (RTCHAP06/main.cpp:94-98,313-316).  Default 1 at "
                         "every N, so that `value` compares like with like across N; the rate with three frames in "
                         "flight is measured after the timed loop and reported beside it (config.three_frames_in_flight)")
    ap.add_argument("--no-gather-overlap", action="store_true",
                    help="N > 1: run the frame gather on the rendering stream instead of a side stream (where it runs beside "
                         "the next frame's rendering, the tile double-buffered)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (two frames in flight, the other kernel): profiling runs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import vulkan_rtiow_amd as V

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (not used by the driver): BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # BENCH_BACKEND=gloo replaces RCCL, so the N>1 code path can run on a one-GPU box
    if os.environ.get("BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    D = __import__("importlib").import_module("vulkan-rtiow_amd.dist")
    if world > 1:
        # A collective that cannot complete ends the job: the timeout turns a hang into an error, and D.fail_loudly
        # (around everything below) turns an error on any rank into a non-zero exit of that rank at once -- no other
        # collective is tried in its place (vulkan-rtiow_amd/dist.py).
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    with (D.fail_loudly("bench.py") if world > 1 else __import__("contextlib").nullcontext()):
        run(args, np, torch, dist, V, D, world, rank, local_rank, backend, dev)


def run(args, np, torch, dist, V, D, world, rank, local_rank, backend, dev):

    scene, grid_half, w, h, spp, depth = WORKLOADS[args.workload]
    sph, mat, cam = build_scene(V, scene, grid_half, w, h)
    F = max(1, args.frames_in_flight)
    prm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, chunk_spp=args.chunk_spp,
                        quantiser=V.RT_QUANT_BOOK, row_block=args.row_block if world > 1 else 0,
                        tile_rank=rank if world > 1 else 0, tile_count=world if world > 1 else 0,
                        kernel=args.kernel)
    rows = V.tile_row_count(h, prm.row_block, prm.tile_rank, prm.tile_count)
    # One context + one real (non-null) torch stream + one tile buffer per frame in flight: the kernels,
    # the timing events and the RCCL gather of a frame are all ordered on its stream (a NULL handle
    # would mean "the context's own stream" to the C ABI).
    ctxs, streams, locals_ = [], [], []
    for _ in range(F):
        c = V.Context(local_rank)
        c.set_scene(sph, mat)
        ctxs.append(c)
        streams.append(torch.cuda.Stream(device=dev))
        locals_.append(torch.zeros((rows, w), dtype=torch.int32, device=dev))
    ctx = ctxs[0]

    # N > 1, one frame in flight: the gather of frame k (and its de-interleave on rank 0) runs on a side stream beside the
    # RENDERING of frame k + 1 -- communication under computation, still one frame being rendered at a time.  Two tile
    # buffers; a tile is rendered into again only when its previous gather has read it.
    overlap = world > 1 and F == 1 and not args.no_gather_overlap
    if overlap:
        gstream = torch.cuda.Stream(device=dev)
        tiles = [locals_[0], torch.zeros_like(locals_[0])]
        rendered = [torch.cuda.Event(), torch.cuda.Event()]
        gathered = [None, None]

    def step(k, events=None):
        if overlap:
            b = k % 2
            with torch.cuda.stream(streams[0]):
                if gathered[b] is not None:
                    streams[0].wait_event(gathered[b])
                if events is not None:
                    events[0].record()
                ctxs[0].render_device(cam, prm, tiles[b].data_ptr(), w * 4, streams[0].cuda_stream)
                if events is not None:
                    events[1].record()
                rendered[b].record()
            with torch.cuda.stream(gstream):
                gstream.wait_event(rendered[b])
                out = D.gather_frame(tiles[b], h, prm.row_block, rank, world)
                done = torch.cuda.Event()
                done.record()
                gathered[b] = done
            return out
        i = k % F
        with torch.cuda.stream(streams[i]):
            if events is not None:
                events[0].record()              # same stream the kernels are launched on
            ctxs[i].render_device(cam, prm, locals_[i].data_ptr(), w * 4, streams[i].cuda_stream)
            if events is not None:
                events[1].record()
            if world > 1:
                return D.gather_frame(locals_[i], h, prm.row_block, rank, world)
            return locals_[i]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(max(args.warmup, F) if args.warmup > 0 else 0):  # every context in flight warm before the clock starts
        step(k)
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    frame = None
    for k in range(args.steps):
        frame = step(k, ev[k])
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    st = ctx.stats()
    ctx_clock_mhz = int(st.shader_clock_mhz)  # the shader clock the last timed frame's kernel measured for itself (RtStats)

    t = torch.tensor([elapsed, kernel_ms, float(st.segments)], dtype=torch.float64,
                     device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms_max, segments = tmax[0].item(), tmax[1].item(), tsum[2].item()
    else:
        kernel_ms_max, segments = kernel_ms, float(st.segments)

    # N = 1: the last timed frame against the oracle's (CRC-32 + segment count made in the build container); N > 1: the
    # last gathered frame against the same frame rendered whole by rank 0 alone, and that one against the oracle's.
    # All of it outside the timed region.
    frame_check = golden_check = golden_what = None
    if world == 1 and frame is not None:
        golden_check, golden_what = golden_frame_check(args.workload, frame.cpu().numpy().tobytes(), st.segments)
    if world > 1 and rank == 0 and frame is not None:
        whole = torch.zeros((h, w), dtype=torch.int32, device=dev)
        wprm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, chunk_spp=args.chunk_spp, quantiser=V.RT_QUANT_BOOK,
                             kernel=args.kernel)
        with torch.cuda.stream(streams[0]):
            ctx.render_device(cam, wprm, whole.data_ptr(), w * 4, streams[0].cuda_stream)
        torch.cuda.synchronize()
        frame_check = "identical" if bool(torch.equal(frame.to(dev), whole)) else "DIFFERS"
        golden_check, golden_what = golden_frame_check(args.workload, frame.cpu().numpy().tobytes(), segments)

    # The same loop with three frames in flight (three contexts / streams / tile buffers per rank, as the
    # reference's per-swapchain-image fences allow): measured at EVERY N, after the timed region, never `value`.
    inflight3 = None
    if F == 1 and not args.no_extras and elapsed / args.steps < 0.5:
        F3 = 3
        while len(ctxs) < F3:
            c = V.Context(local_rank)
            c.set_scene(sph, mat)
            ctxs.append(c)
            streams.append(torch.cuda.Stream(device=dev))
            locals_.append(torch.zeros((rows, w), dtype=torch.int32, device=dev))

        def step3(k):
            i = k % F3
            with torch.cuda.stream(streams[i]):
                ctxs[i].render_device(cam, prm, locals_[i].data_ptr(), w * 4, streams[i].cuda_stream)
                if world > 1:
                    D.gather_frame(locals_[i], h, prm.row_block, rank, world)

        for k in range(2 * F3):
            step3(k)
        fence()
        n3 = max(6, args.steps)
        t3 = time.perf_counter()
        for k in range(n3):
            step3(k)
        fence()
        e3 = torch.tensor([time.perf_counter() - t3], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(e3, op=dist.ReduceOp.MAX)
        inflight3 = {"frames_in_flight": F3, "steps": n3, "ms_per_step": e3.item() / n3 * 1e3,
                     "value": w * h * spp * depth / (e3.item() / n3) / 1e6}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        nominal = w * h * spp * depth
        value = nominal / (elapsed / args.steps) / 1e6
        n = len(sph)
        eff_kernel = ctx.last_kernel()   # what RtParams.kernel 0 resolved to
        # dominant kernel: the path-trace kernel of rank 0's tile (per launch).  Flops per segment =
        # 16 per ray-sphere / ray-box test the kernel's algorithm performs + 60 of shading (SURVEY 8d with
        # a = 1).  The flat list tests all N spheres per segment; the clustered list's count depends on
        # the rays and is taken from the kernel's own counter.  `flat_list_equivalent` rates the same
        # frame as if every segment had met the whole list (SURVEY 8d's literal 16 N + 60).
        flops = st.sphere_tests * FLOPS_PER_TEST + st.segments * FLOPS_PER_SEGMENT_SHADE
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        flat_equiv = st.segments * (n * FLOPS_PER_TEST + FLOPS_PER_SEGMENT_SHADE) / (kernel_ms * 1e-3) / 1e12
        fb_bytes = st.bytes_written
        traffic = None
        if world == 1:  # PMC traffic of this workload + kernel, newest profile first
            for tpath in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):
                tj = json.load(open(tpath))
                # (of this workload and kernel AND of these kernel sources: a profile of an earlier round's kernels, or of the A/B baseline
                # a round began with, is passed over)
                if (tj.get("workload") == args.workload and tj.get("bench_kernel") == KERNEL_NAMES.get(eff_kernel)
                        and tj.get("kernel_source_sha16") == kernel_source_sha16()):
                    traffic = tj.get("hbm_bytes_per_launch")
                    break
        # VALU issue utilisation of the same kernel from the committed PMC profile (SQ_INSTS_VALU per launch)
        # against the issue rate tools/ubench_valu.hip measures for back-to-back v_fma_f32 on this part
        # (2.53 cycles per wave instruction per SIMD at 2.4 GHz: 0.97e12 wave instructions / s)
        valu_issue = None
        def pmc_kernel(pj):  # path_persistent_kernel<shading records in LDS, clustered list, flat-axis box test> of this variant
            for name in pj:
                if name.startswith("path_persistent_kernel<"):
                    targs = name[name.index("<") + 1:name.rindex(">")].split(",")
                    if len(targs) >= 2 and (targs[1] == "true") == (eff_kernel == 3):
                        return name
            raise KeyError("no path kernel in the profile")
        if world == 1 and eff_kernel in (2, 3) and args.workload == "cover_1200x800_100spp":
            for ppath in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):  # newest round first
                try:
                    pj = json.load(open(ppath))
                    if pj.get("kernel_source_sha16") != kernel_source_sha16():
                        break  # the newest profile is of other kernels than the tree holds: no figure rather than a stale one
                    pmc_name = pmc_kernel(pj)
                    insts = pj[pmc_name]["SQ_INSTS_VALU"]["mean_per_launch"]
                    valu_issue = {"valu_wave_instructions_per_launch": insts,
                                  "rate": insts / (kernel_ms * 1e-3), "peak": VALU_ISSUE_PEAK, "unit": "wave instructions/s",
                                  "frac": insts / (kernel_ms * 1e-3) / VALU_ISSUE_PEAK,
                                  "lane_occupancy": pj[pmc_name].get("lane_occupancy_valu"),
                                  "kernel": pmc_name, "source": "profiles/" + os.path.basename(ppath),
                                  "kernel_source_sha16": pj.get("kernel_source_sha16")}
                    break
                except (OSError, KeyError, ValueError):
                    continue
        out = {
            "metric": "Mray/s (w*h*spp*depth / s), cover scene 1200x800", "value": value, "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "kernel": KERNEL_NAMES.get(eff_kernel, str(eff_kernel)), "spheres": n, "width": w, "height": h, "spp": spp,
                       "max_depth": depth, "chunk_spp": args.chunk_spp, "seed": 1,
                       "partition": (f"row-tiles block-cyclic x{args.row_block} over {world} ranks; frame gathered by {D.transport_label()}"
                                     if world > 1 else "single GPU"),
                       "backend": (dist.get_backend() if world > 1 else None),
                       "collective": (D.collective() if world > 1 else None),
                       "collective_ranks": (dist.get_world_size() if world > 1 else 1),
                       "rccl_ranks": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else 0),
                       "ranks_on_distinct_gpus": (os.environ.get("BENCH_ONE_DEVICE") != "1") if world > 1 else None,
                       "frames_in_flight": F,
                       "gather": ("side stream: frame k's gather runs beside the rendering of frame k + 1 (two tile buffers)"
                                  if overlap else ("on the rendering stream" if world > 1 else "none (one GPU)")),
                       "segments_per_frame": int(segments), "segments_per_s": segments / (elapsed / args.steps),
                       "sphere_tests_per_s": st.sphere_tests * (segments / max(1, st.segments)) / (elapsed / args.steps),
                       "tests_per_segment": st.sphere_tests / max(1, st.segments),
                       "kernel_ms_rank0": kernel_ms, "kernel_ms_max": kernel_ms_max},
            "roofline": {
                "bound": "valu", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_VALU_PEAK_TFLOPS, "traffic": traffic,
                "note": "fp32 vector-ALU bound (no dense contraction: MFMA unused by design); flops = "
                        f"tests*{FLOPS_PER_TEST} + segments*{FLOPS_PER_SEGMENT_SHADE}, tests from the kernel's counter "
                        "(flat list: segments*N; clustered list: large spheres + cluster boxes + members of the boxes "
                        "a ray reaches, and for camera rays -- traced in the primary pass -- the cone tests of their pixel "
                        "and the spheres its cone reaches); kernel time = HIP events on the launch stream over the timed steps; "
                        "the fraction rates EXECUTED tests: the primary pass of round 2 removed a third of them (58.3 -> 39.0 per "
                        "segment on the cover frame) while the frame got 9-14 % faster, so it fell from 0.176 to 0.133 -- "
                        "valu_issue (instruction issue against the measured v_fma_f32 rate) is the utilisation figure",
                "tests_per_segment": st.sphere_tests / max(1, st.segments),
                "valu_issue": valu_issue,
                "flat_list_equivalent": {"achieved": flat_equiv, "frac": flat_equiv / FP32_VALU_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "note": "segments*(16 N + 60): the flat list's work for "
                                         "the same frame over this kernel's time; above 1 means the acceleration "
                                         "structure beats any possible flat-list kernel"},
                "hbm_write": {"achieved": fb_bytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": fb_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "bytes_per_launch": fb_bytes},
            },
        }
        if F > 1:
            out["roofline"]["overlap_note"] = (f"{F} frames in flight share the GPU: the event-bracketed kernel time of a "
                                               "frame includes the time its waves wait behind the other frames', so this "
                                               "per-kernel fraction understates the kernel; the N=1 line is the clean one")
        quick = ms_per_step < 500.0 and not args.no_extras  # the extras below re-render the frame a few times
        if inflight3 is not None:
            out["config"]["three_frames_in_flight"] = inflight3
        if frame_check is not None:
            out["config"]["gathered_frame_vs_single_gpu_frame"] = frame_check
        out["frame_check"] = golden_check  # "golden": the timed frame IS the oracle's frame (crc32 + segments)
        out["config"]["frame_check_against"] = golden_what
        if world == 1 and quick:  # the other persistent kernel on the same frame, outside the timed region
            other = 3 if eff_kernel == 2 else 2
            oprm = V.make_params(w, h, spp=spp, max_depth=depth, seed=1, quantiser=V.RT_QUANT_BOOK, kernel=other)
            oms = []
            for _ in range(3):
                ctx.render_device(cam, oprm, locals_[0].data_ptr(), w * 4, streams[0].cuda_stream)
                oms.append(ctx.stats().kernel_ms)
            ost = ctx.stats()
            oflops = ost.sphere_tests * FLOPS_PER_TEST + ost.segments * FLOPS_PER_SEGMENT_SHADE
            out["config"]["other_kernel"] = {"kernel": KERNEL_NAMES[other], "kernel_ms": min(oms),
                                             "tests_per_segment": ost.sphere_tests / max(1, ost.segments),
                                             "roofline_frac": oflops / (min(oms) * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS}
        if world == 1 and quick and scene == "cover":
            out["config"].update(cold_and_moving_frames(V, torch, dev, local_rank, sph, mat, cam, prm, w, h, frame, ctx, streams[0],
                                                        locals_[0]))
        if world == 1 and quick:
            try:  # (an untimed extra on a 1 GiB frame: it must not cost the run its line)
                out["config"]["reference_shaders"] = reference_shader_rates(V, torch, ctx, streams[0])
            except Exception as e:  # noqa: BLE001
                out["config"]["reference_shaders"] = {"error": f"{type(e).__name__}: {e}"}
        clock = ctx_clock_mhz
        out["roofline"]["shader_clock_mhz"] = clock or None
        out["roofline"]["peak_at_held_clock"] = FP32_VALU_PEAK_TFLOPS * clock / 2400.0 if clock else None
        out["roofline"]["frac_at_held_clock"] = achieved / (FP32_VALU_PEAK_TFLOPS * clock / 2400.0) if clock else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(V, sph, mat, cam, w, h, spp, depth, args.chunk_spp, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
