"""CPU: the N>1 path (row-tile partition + framebuffer gather) with world_size 2 and 3 over
gloo.  Rendering is replaced by a synthetic pattern f(global_row, x): only the plumbing runs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vulkan_rtiow_amd as V
from importlib import import_module

D = import_module("vulkan-rtiow_amd.dist")


def _pattern(rows, width):
    x = np.arange(width, dtype=np.int64)[None, :]
    return ((rows[:, None] * 7919 + x * 104729) % (2**31 - 1)).astype(np.int32)


def _worker(rank, world, port, height, width, block, out_path, collective="gather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    D._collective = collective
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = D.tile_rows(height, block, rank, world)
    assert len(rows) == V.tile_row_count(height, block, rank, world)
    assert [V.tile_global_row(i, block, rank, world) for i in range(len(rows))] == rows.tolist()
    local = torch.from_numpy(_pattern(rows, width))
    frame = D.gather_frame(local, height, block, rank, world)
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,height,width,block", [(2, 37, 19, 4), (2, 800, 32, 16), (3, 10, 5, 1)])
def test_gather_reassembles_frame(tmp_path, world, height, width, block):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, block, out), nprocs=world, join=True)
    got = np.load(out)
    assert np.array_equal(got, _pattern(np.arange(height), width))


def test_all_gather_form_reassembles_the_same_frame(tmp_path):
    """The form dist.gather_frame falls back to when the backend refuses a rooted gather (RTIOW_COLLECTIVE=all_gather
    selects it outright): same padded slots, same de-interleave, same frame."""
    world, height, width, block = 3, 41, 11, 4
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, block, out, "all_gather"), nprocs=world, join=True)
    assert np.array_equal(np.load(out), _pattern(np.arange(height), width))


def test_single_rank_is_identity():
    t = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    assert D.gather_frame(t, 3, 8, 0, 1) is t
    assert D.tile_rows(5, 2, 0, 1).tolist() == [0, 1, 2, 3, 4]
