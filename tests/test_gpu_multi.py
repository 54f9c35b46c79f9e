"""GPU (-m gpu): the native multi-GPU path behind the C ABI (rtCreateMulti .. rtMultiRender) on the one GPU
this box has.  N = 1 is the degenerate path; a device list that repeats device 0 runs the whole N-tile path --
N contexts and streams, block-cyclic tiles, the [N][rows_max][width] gather buffer, the de-interleave kernel --
with hipMemcpyPeerAsync moving the tiles (RCCL refuses a duplicated device).  The RCCL transport itself needs
distinct GPUs: it is exercised only by `rtiow_main --gpus N` on a multi-GPU node."""
import os
import subprocess

import numpy as np
import pytest

import vulkan_rtiow_amd as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "vulkan-rtiow_amd", "rtiow_main")


def _case(oracle, scene, w, h):
    from golden.make_golden import build_case
    return build_case(V, oracle, scene, w, h)


def test_multi_n1_is_the_single_gpu_render(gpu_ctx, oracle):
    w, h = 150, 100
    sph, mat, cam = _case(oracle, "cover11", w, h)
    prm = V.make_params(w, h, spp=4, max_depth=50, seed=11)
    want, segs = oracle.render(sph, mat, cam, prm)
    with V.MultiContext([0]) as m:
        assert m.transport == "single"
        m.set_scene(sph, mat)
        assert np.array_equal(m.render(cam, prm), want)
        st, ms = m.stats(0)
        assert st.segments == segs and ms > 0


@pytest.mark.parametrize("n,block", [(2, 4), (4, 4), (8, 4), (3, 1), (5, 16)])
def test_multi_tiles_gathered_and_deinterleaved_equal_the_oracle(oracle, n, block):
    """Ragged sizes: tiles of unequal height (padded slots), a width that is not a multiple of 4 (word copies
    in the de-interleave) and one that is (128-bit copies)."""
    for w, h in ((121, 83), (200, 64)):
        sph, mat, cam = _case(oracle, "cover11", w, h)
        prm = V.make_params(w, h, spp=3, max_depth=20, seed=2, row_block=block)
        want, segs = oracle.render(sph, mat, cam, V.make_params(w, h, spp=3, max_depth=20, seed=2))
        with V.MultiContext([0] * n) as m:
            assert m.transport == "peer-copy"
            m.set_scene(sph, mat)
            for _ in range(3):     # frames 2 and 3 are dealt in the cost order of the frame before
                assert np.array_equal(m.render(cam, prm), want), (n, block, w, h)
            total = sum(m.stats(g)[0].segments for g in range(n))
            assert total == segs
            rows = sum(m.stats(g)[0].rows_rendered for g in range(n))
            assert rows == h


def test_multi_device_destination_back_to_back(oracle):
    """Frames enqueued back to back into device memory: tile g of frame k+1 must not land in the gather buffer
    while the root still de-interleaves frame k."""
    torch = pytest.importorskip("torch")
    w, h = 240, 160
    sph, mat, cam = _case(oracle, "cover11", w, h)
    prms = [V.make_params(w, h, spp=2, max_depth=50, seed=s) for s in (5, 6)]
    wants = [oracle.render(sph, mat, cam, p)[0] for p in prms]
    bufs = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    with V.MultiContext([0, 0, 0, 0]) as m:
        m.set_scene(sph, mat)
        for k in range(6):
            m.render_device(cam, prms[k % 2], bufs[k].data_ptr(), w * 4)
        m.synchronize()
        for k in range(6):
            assert np.array_equal(bufs[k].cpu().numpy().view(np.uint8).reshape(h, w, 4), wants[k % 2]), k


def test_multi_rejects_what_it_cannot_do(oracle):
    sph, mat, cam = _case(oracle, "three", 32, 16)
    with V.MultiContext([0, 0]) as m:
        m.set_scene(sph, mat)
        with pytest.raises(V.RtError) as e:      # the tiling is the call's own
            m.render(cam, V.make_params(32, 16, spp=1, tile_rank=1, tile_count=2))
        assert e.value.code == V.RT_ERR_INVALID
        with pytest.raises(V.RtError) as e:      # the reference's kernels are one 7-us dispatch
            m.render(cam, V.make_params(32, 16, mode=V.RT_MODE_CH06))
        assert e.value.code == V.RT_ERR_INVALID
    with pytest.raises(V.RtError) as e:
        V.MultiContext([0, 99])
    assert e.value.code == V.RT_ERR_NO_DEVICE


def test_harness_gpus_flag(oracle, tmp_path):
    """rtiow_main --gpus 1 and --devices 0,0,0 (C++ over the same ABI) write the frame the oracle renders."""
    w, h = 120, 80
    sph, mat = V.make_cover_scene(1, 11)
    cam = oracle.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    want, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=3, max_depth=50, seed=1))
    for flag, val in (("--gpus", "1"), ("--devices", "0,0,0")):
        out = str(tmp_path / f"multi_{val.replace(',', '')}.ppm")
        res = subprocess.run([MAIN, "--scene", "cover", "--width", str(w), "--height", str(h), "--spp", "3", flag, val,
                              "--frames", "2", "--out", out], check=True, capture_output=True, text=True)
        assert ("transport single" if flag == "--gpus" else "transport peer-copy") in res.stdout
        data = open(out, "rb").read()
        _, dims, _, body = data.split(b"\n", 3)
        img = np.frombuffer(body, np.uint8).reshape(h, w, 3)
        assert np.array_equal(img, want[::-1, :, :3])


def test_harness_multi_gpu_bench_line(tmp_path):
    """`rtiow_main --gpus N --json 1` ends with bench.py's fields for the C-ABI path (rtMultiRender) --
    rehearsed here as four tiles on device 0 (peer-copy transport): ms per frame with one frame in flight, every device's
    tile kernel time, the transport by its real name, the gathered frame equal to the one device 0 renders alone, and its
    CRC-32 equal to the ORACLE's for the same workload (tests/golden/frame_golden.json: cover_300x200_10spp)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_golden.json")))["cover_300x200_10spp"]
    res = subprocess.run([MAIN, "--scene", "cover", "--width", "300", "--height", "200", "--spp", "10", "--devices", "0,0,0,0",
                          "--frames", "5", "--warmup", "2", "--json", "1", "--out", str(tmp_path / "b.ppm")],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-1500:])
    lines = [json.loads(x) for x in res.stdout.strip().splitlines()]
    line = lines[-1]
    assert [x["frame"] for x in lines[:-1]] == [0, 1, 2, 3, 4]
    assert line["n_gpus"] == 4 and line["steps"] == 3 and line["warmup"] == 2 and line["unit"] == "Mray/s"
    assert line["ms_per_step"] > 0 and abs(line["value"] - 300 * 200 * 10 * 50 / (line["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * line["value"]
    cfg = line["config"]
    assert cfg["transport"] == "peer-copy" and cfg["devices"] == [0, 0, 0, 0] and cfg["frames_in_flight"] == 1
    assert len(cfg["device_kernel_ms"]) == 4 and all(t > 0 for t in cfg["device_kernel_ms"])
    assert cfg["gathered_frame_vs_single_gpu_frame"] == "identical"
    assert cfg["frame_crc32"] == gold["crc32"] and cfg["segments_per_frame"] == gold["segments"]


def test_rccl_bindings_with_a_communicator_of_one(oracle, tmp_path):
    """RTIOW_MULTI_TRANSPORT=rccl with one device: librccl.so.1 is opened, ncclCommInitAll makes a one-rank communicator
    and every frame goes through ncclGroupStart / ncclGather (in place) / ncclGroupEnd and the de-interleave kernel.
    What this box cannot exercise is a transfer between two GPUs; the entry points, their signatures and the call
    sequence it can.  Run in the C++ harness (a process of its own, no second RCCL from torch in it)."""
    w, h = 120, 80
    sph, mat = V.make_cover_scene(1, 11)
    cam = oracle.make_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)
    want, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=3, max_depth=50, seed=1))
    out = str(tmp_path / "rccl1.ppm")
    env = dict(os.environ, RTIOW_MULTI_TRANSPORT="rccl")
    res = subprocess.run([MAIN, "--scene", "cover", "--width", str(w), "--height", str(h), "--spp", "3", "--gpus", "1",
                          "--frames", "3", "--out", out], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-1500:])
    assert "transport rccl" in res.stdout
    _, dims, _, body = open(out, "rb").read().split(b"\n", 3)
    assert np.array_equal(np.frombuffer(body, np.uint8).reshape(h, w, 3), want[::-1, :, :3])



def test_torch_rccl_backend_with_one_rank():
    """bench.py's N > 1 calls on the `nccl` (= RCCL) backend with a world of one rank: process-group init with a device
    id, the rooted gather into views of one receive buffer exactly as dist.gather_frame issues it, the all-gather form,
    the all_reduce of the timing tensor, barrier, teardown.  A second rank needs a second GPU; the API usage does not."""
    import sys
    code = (
        "import os, torch, torch.distributed as dist\n"
        "from importlib import import_module\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29544')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
        "D = import_module('vulkan-rtiow_amd.dist')\n"
        "h, w, block = 37, 19, 4\n"
        "local = torch.arange(h * w, dtype=torch.int32, device=dev).reshape(h, w)\n"
        "side = torch.cuda.Stream()\n"
        "with torch.cuda.stream(side):\n"
        "    recv = torch.empty((1, h, w), dtype=torch.int32, device=dev)\n"
        "    dist.gather(local, gather_list=list(recv.unbind(0)), dst=0)\n"
        "    frame = D._deinterleave(recv, h, block, 1)\n"
        "    recv2 = torch.empty((1, h, w), dtype=torch.int32, device=dev)\n"
        "    dist.all_gather_into_tensor(recv2.view(h, w), local)\n"
        "t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)\n"
        "dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()\n"
        "assert torch.equal(frame, local) and torch.equal(recv2[0], local) and t.tolist() == [1.5, 2.5]\n"
        "dist.destroy_process_group(); print('rccl-one-rank ok')\n")
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "rccl-one-rank ok" in res.stdout, res.stderr[-3000:]


def test_multi_progressive_accumulation(oracle):
    """The frame loop with a running average on N tiles: every device keeps the accumulators of its own rows, k dispatches
    of spp samples are one dispatch of k * spp, and every intermediate frame is the oracle's at that sample count."""
    w, h = 131, 70
    sph, mat, cam = _case(oracle, "cover11", w, h)
    with V.MultiContext([0, 0, 0]) as m:
        m.set_scene(sph, mat)
        done = 0
        for spp in (2, 3, 1):
            prm = V.make_params(w, h, spp=spp, max_depth=50, seed=9, row_block=4, sample_offset=done, accumulate=1)
            got = m.render(cam, prm)
            done += spp
            want, _ = oracle.render(sph, mat, cam, V.make_params(w, h, spp=done, max_depth=50, seed=9))
            assert np.array_equal(got, want), done
