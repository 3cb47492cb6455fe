"""Host logic of the one-process-per-GPU job: which frames a rank owns, how the model reaches
every rank, how the step time is reduced.  Backend-agnostic (`nccl` = RCCL on the GPUs, `gloo`
in the CPU tests); no compute here.

The path shards by independent frames: no data-path collective.  The only
exchange is at load time: rank 0 owns the .mars file; descriptors travel as bytes, the packed
parameter arena (everything derived from the weight blob) as ONE broadcast of device memory.
"""
import struct

import numpy as np


def shard_frames(per_rank, rank, world):
    """global frame indices of `rank` under weak scaling (per_rank frames on every rank)"""
    if not (0 <= rank < world) or per_rank < 0:
        raise ValueError("bad shard request")
    return range(rank * per_rank, (rank + 1) * per_rank)


def split_frames(total, rank, world):
    """strong-scaling variant: contiguous, balanced slices of `total` frames"""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def strip_weights(model_bytes):
    """the same file with the weight blob zeroed: what non-root ranks load (descriptors only)"""
    woff, wsz = struct.unpack_from("<QQ", model_bytes, 28)
    out = bytearray(model_bytes)
    end = min(len(out), woff + wsz)
    if woff < end:
        out[woff:end] = bytes(end - woff)
    return bytes(out)


def descriptor_bytes(model_bytes):
    """header + tensor table + layer table (everything before the blob)"""
    nl, nt = struct.unpack_from("<II", model_bytes, 12)
    return model_bytes[:76 + 124 * nt + 112 * nl]


def broadcast_bytes(dist, data, src, device="cpu"):
    """rank `src` passes `data`; every rank returns the same bytes"""
    import torch
    n = torch.tensor([len(data) if dist.get_rank() == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def max_over_ranks(dist, seconds, device="cpu"):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class DeviceBuffer:
    """zero-copy torch view of a raw HBM pointer (RCCL broadcast of the parameter arena)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def gather_rows(dist, rows, device="cpu"):
    """all ranks contribute a [n, k] uint8 array with equal shapes; returns the rank-ordered stack"""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(rows)).to(device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return np.concatenate([o.cpu().numpy() for o in outs])
