"""Row-tile partition of a frame across the GPUs of one node and the framebuffer
gather over RCCL (the reference is single-device:
RTCHAP06/Vulkan.cpp:87-97,122).

One process per GPU.  Rank r renders the rows with (row // row_block) % world == r
(block-cyclic: sky rows are far cheaper than sphere-dense rows, so contiguous
bands would leave the sky ranks idle), packed in ascending order, into its own
HBM buffer; nothing is exchanged while rendering.  The only collective is one
gather of the finished RGBA8 rows to the root, followed by a de-interleave.
Pixels do not depend on the partition (RNG keyed by the global pixel index), so
1/2/4/8-GPU frames are byte-identical.
"""
from __future__ import annotations

import contextlib
import os
import sys
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def tile_rows(height: int, row_block: int, rank: int, world: int) -> np.ndarray:
    """Global row indices rank `rank` renders, ascending (== rtTileGlobalRow over its local rows)."""
    if world <= 1:
        return np.arange(height, dtype=np.int64)
    row_block = max(1, int(row_block))
    rows = np.arange(height, dtype=np.int64)
    return rows[(rows // row_block) % world == rank]


def max_tile_rows(height: int, row_block: int, world: int) -> int:
    return max(len(tile_rows(height, row_block, r, world)) for r in range(max(1, world)))


_scatter_rows_cache: dict = {}


def _scatter_rows(height: int, row_block: int, world: int, device) -> torch.Tensor:
    """Destination row of every (rank, padded local row) of the gathered buffer; padding rows go to the
    scratch row `height`.  Built once per partition and device (no per-frame host-to-device copies)."""
    key = (height, row_block, world, str(device))
    hit = _scatter_rows_cache.get(key)
    if hit is None:
        pad_rows = max_tile_rows(height, row_block, world)
        dest = np.full((world, pad_rows), height, dtype=np.int64)
        for r in range(world):
            rows = tile_rows(height, row_block, r, world)
            dest[r, : len(rows)] = rows
        hit = torch.from_numpy(dest.reshape(-1)).to(device)
        _scatter_rows_cache[key] = hit
    return hit


def collective() -> str:
    """The collective the frame travels by: "gather" (the default: only the root receives) or "all_gather"
    (RTIOW_COLLECTIVE=all_gather: every rank receives every tile).  An explicit choice, made once per process and the
    same on every rank; a collective that fails is an error on every rank (see `fail_loudly`), never a reason to try
    the other one -- on RCCL a rank that switched collectives alone would leave its peers inside the first."""
    name = os.environ.get("RTIOW_COLLECTIVE", "gather")
    if name not in ("gather", "all_gather"):
        raise ValueError(f"RTIOW_COLLECTIVE={name!r}: expected 'gather' or 'all_gather'")
    return name


def transport_label(group=None) -> str:
    """What actually moves the frame, for bench lines and logs: backend (as torch.distributed names it; "nccl" is RCCL
    on ROCm), collective, world size."""
    if not dist.is_available() or not dist.is_initialized():
        return "single process, no collective"
    backend = dist.get_backend(group)
    name = {"nccl": "RCCL (torch backend nccl)", "gloo": "gloo (CPU rehearsal, frames staged through host memory)"}.get(backend, backend)
    how = "dist.gather to rank 0" if collective() == "gather" else "dist.all_gather_into_tensor"
    return f"{name}, {how}, {dist.get_world_size(group)} ranks"


def gather_frame(local: torch.Tensor, height: int, row_block: int, rank: int, world: int,
                 dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Gathers every rank's packed rows ([rows_r, width] int32, RGBA8 packed) to `dst` and
    returns the assembled [height, width] frame there (None elsewhere).  Errors of the collective propagate."""
    if world <= 1:
        return local
    # RCCL moves device tensors; a gloo rehearsal (tests, or two ranks sharing one GPU) stages via host
    if local.is_cuda and dist.get_backend(group) == "gloo":
        frame = gather_frame(local.cpu(), height, row_block, rank, world, dst, group)
        return frame.to(local.device) if frame is not None else None
    width = local.shape[1]
    pad_rows = max_tile_rows(height, row_block, world)
    send = local
    if local.shape[0] != pad_rows:  # equal counts for the collective: pad the short tiles
        send = torch.zeros((pad_rows, width), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if collective() == "gather":
        if rank == dst:
            # one receive buffer [world, pad_rows, width]; one scatter of its rows into the frame
            # (plus a scratch row that swallows the padding)
            recv = torch.empty((world, pad_rows, width), dtype=local.dtype, device=local.device)
            dist.gather(send, gather_list=list(recv.unbind(0)), dst=dst, group=group)
            return _deinterleave(recv, height, row_block, world)
        dist.gather(send, gather_list=None, dst=dst, group=group)
        return None
    recv = torch.empty((world, pad_rows, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv.view(world * pad_rows, width), send, group=group)
    return _deinterleave(recv, height, row_block, world) if rank == dst else None


@contextlib.contextmanager
def fail_loudly(what: str = "multi-GPU frame"):
    """Wraps the N > 1 part of a program: any exception on this rank is printed with its rank and ends the PROCESS with
    a non-zero code at once (os._exit: no communicator destructors, which may wait for peers that are themselves stuck
    in the collective this rank never joined).  The launcher (torch.distributed.run) then tears the other ranks down,
    and a peer left inside a collective fails on its own once this rank's sockets close or the process group's timeout
    expires -- it never completes a different collective instead.  A clean exit (SystemExit with code 0 or None) passes
    through untouched.  os._exit skips Python's cleanup: files the wrapped code writes must be closed (or written with
    `with open(...)`) before an error can occur -- only stdout and stderr are flushed here."""
    try:
        yield
    except SystemExit as exc:
        if exc.code in (0, None):
            raise  # a clean exit is not a failure
        sys.stderr.write(f"[rank {os.environ.get('RANK', '?')}] {what} exited with status {exc.code}\n")
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(exc.code if isinstance(exc.code, int) and 0 < exc.code < 256 else 13)
    except BaseException as exc:  # noqa: BLE001 (KeyboardInterrupt too: the peers must not be left waiting)
        import traceback
        r = os.environ.get("RANK", "?")
        sys.stderr.write(f"[rank {r}] {what} failed: {type(exc).__name__}: {exc}\n")
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(13)


def _deinterleave(recv: torch.Tensor, height: int, row_block: int, world: int) -> torch.Tensor:
    pad_rows, width = recv.shape[1], recv.shape[2]
    frame = torch.empty((height + 1, width), dtype=recv.dtype, device=recv.device)
    frame.index_copy_(0, _scatter_rows(height, row_block, world, recv.device), recv.view(world * pad_rows, width))
    return frame[:height]
