"""Python mirror of include/hr_comm.h (libhr_comm.so): the native RCCL / loopback transport of the row-tiled frame.

bench.py and tiling.py use torch.distributed for the same plan (that is how the driver launches N ranks); this mirror exists so
that the C-ABI path a C++ host links is exercised by the test-suite (loopback: all ranks in one process on one GPU), and so that
a torch.distributed job can be switched to the native path (`NativeComm.from_torch_distributed()`: rank 0 draws the RCCL unique
id, torch broadcasts it)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import api
from .api import _check, _stream_ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhr_comm.so")
HR_COMM_ID_BYTES = 128
ABI_SYMBOLS = ["hr_comm_get_unique_id", "hr_comm_create_rccl", "hr_comm_create_loopback", "hr_comm_destroy", "hr_comm_rank", "hr_comm_world",
               "hr_comm_exchange_rows", "hr_comm_wait", "hr_comm_wait_ticket", "hr_comm_allgather_rows", "hr_shadows_exchange_history", "hr_ao_exchange_history",
               "hr_reflections_exchange_history", "hr_ddgi_allgather_atlases"]
_lib = None


class hr_comm_image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("row_pitch_bytes", C.c_int64)]


def lib():
    global _lib
    if _lib is None:
        api.lib()   # the pass library first (libhr_comm links it)
        if not os.path.exists(LIB_PATH):
            raise api.HRError(f"{LIB_PATH} not built: run `python -m hybrid_rendering_amd.build`")
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _bounds(b):
    return (C.c_int32 * len(b))(*[int(v) for v in b])


class NativeComm:
    def __init__(self, ctx, world, rank, loopback_name=None, unique_id=None):
        self.h = C.c_void_p()
        self.world, self.rank = world, rank
        if loopback_name is not None:
            _check(lib().hr_comm_create_loopback(ctx.h, C.c_int32(world), C.c_int32(rank), loopback_name.encode(), C.byref(self.h)), "hr_comm_create_loopback")
        else:
            buf = (C.c_uint8 * HR_COMM_ID_BYTES)(*unique_id)
            _check(lib().hr_comm_create_rccl(ctx.h, C.c_int32(world), C.c_int32(rank), buf, C.byref(self.h)), "hr_comm_create_rccl")

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * HR_COMM_ID_BYTES)()
        _check(lib().hr_comm_get_unique_id(buf), "hr_comm_get_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, ctx, group=None):
        """RCCL communicator over the ranks of a torch.distributed job: rank 0 draws the id, torch broadcasts it"""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        t = torch.zeros(HR_COMM_ID_BYTES, dtype=torch.uint8, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, src=0, group=group)
        return cls(ctx, world, rank, unique_id=bytes(t.cpu().numpy().tobytes()))

    # every collective returns its TICKET (hr_comm.h); wait(ticket) orders the stream behind everything posted up to that ticket only
    def exchange_rows(self, tensors, bounds, rows, stream=None) -> int:
        """tensors: [H, W, ...] row-major cuda tensors addressed by absolute row"""
        ims = (hr_comm_image * len(tensors))(*[hr_comm_image(t.data_ptr(), t.stride(0) * t.element_size()) for t in tensors])
        t = C.c_int64(0)
        _check(lib().hr_comm_exchange_rows(self.h, ims, C.c_int32(len(tensors)), _bounds(bounds), C.c_int32(rows), _stream_ptr(stream), C.byref(t)), "hr_comm_exchange_rows")
        return t.value

    def allgather_rows(self, tensor, row_bounds, stream=None) -> int:
        im = hr_comm_image(tensor.data_ptr(), tensor.stride(0) * tensor.element_size())
        t = C.c_int64(0)
        _check(lib().hr_comm_allgather_rows(self.h, im, _bounds(row_bounds), _stream_ptr(stream), C.byref(t)), "hr_comm_allgather_rows")
        return t.value

    def wait(self, stream=None, ticket=None):
        if ticket is None:
            _check(lib().hr_comm_wait(self.h, _stream_ptr(stream)), "hr_comm_wait")
        else:
            _check(lib().hr_comm_wait_ticket(self.h, C.c_int64(int(ticket)), _stream_ptr(stream)), "hr_comm_wait_ticket")

    # per-pass conveniences
    def _pass_exchange(self, fn, name, p, bounds, ping_pong, rows, stream):
        t = C.c_int64(0)
        _check(fn(p.h, self.h, _bounds(bounds), C.c_int32(int(ping_pong)), C.c_int32(rows), _stream_ptr(stream), C.byref(t)), name)
        return t.value

    def exchange_shadows(self, p, bounds, ping_pong, rows, stream=None) -> int:
        return self._pass_exchange(lib().hr_shadows_exchange_history, "hr_shadows_exchange_history", p, bounds, ping_pong, rows, stream)

    def exchange_ao(self, p, bounds, ping_pong, rows, stream=None) -> int:
        return self._pass_exchange(lib().hr_ao_exchange_history, "hr_ao_exchange_history", p, bounds, ping_pong, rows, stream)

    def exchange_reflections(self, p, bounds, ping_pong, rows, stream=None) -> int:
        return self._pass_exchange(lib().hr_reflections_exchange_history, "hr_reflections_exchange_history", p, bounds, ping_pong, rows, stream)

    def allgather_ddgi(self, p, stream=None) -> int:
        t = C.c_int64(0)
        _check(lib().hr_ddgi_allgather_atlases(p.h, self.h, _stream_ptr(stream), C.byref(t)), "hr_ddgi_allgather_atlases")
        return t.value

    def close(self):
        if self.h:
            lib().hr_comm_destroy(self.h)
            self.h = C.c_void_p()
