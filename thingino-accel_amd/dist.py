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


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank, world, port, base=None):
    """the environment of rank `rank` of a one-node job: what `torch.distributed.run` would have set"""
    import os
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on these hosts
    return env


def spawn_ranks(world, cmd, extra_env=None, timeout=None, poll_s=0.05):
    """One FRESH child process per rank (`cmd` = argv list, the same for every rank), started by a parent that has not
    touched the GPU (children are spawned, never exec'd over a process that owns a HIP context).  Rank 0's stdout is
    captured and returned, the other ranks' stdout goes to this process's stderr, every rank's stderr is inherited.
    If any rank exits non-zero (or the job outlives `timeout` seconds) the ranks still running are terminated -- by the
    exact PIDs started here -- and the job fails.  Returns (return code, rank 0's stdout as str, per-rank codes)."""
    import subprocess
    import sys
    import time
    if world < 1:
        raise ValueError("world size %r" % (world,))
    port = free_port()
    procs = []
    for r in range(world):
        env = rank_env(r, world, port)
        env.update(extra_env or {})
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    out0 = []
    import threading
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)  # drain the pipe while it runs
    rd.start()
    t0, failed = time.time(), None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if timeout is not None and time.time() - t0 > timeout:
            failed = "job exceeded %.0f s" % timeout
            break
        time.sleep(poll_s)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.time()
        for p in procs:
            try:
                p.wait(max(0.1, 10 - (time.time() - t1)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        print("spawn_ranks: %s; the other ranks were stopped" % failed, file=sys.stderr)
    rd.join(5)
    codes = [p.returncode for p in procs]
    rc = 0 if not failed else next((c for c in codes if c not in (0, None) and c > 0), 1)
    return rc, (out0[0] if out0 else ""), codes
