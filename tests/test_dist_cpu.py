"""World-size-2 `gloo` run of the multi-rank host logic (no GPU): the model travels from rank 0,
every rank computes its own frame shard (here with the CPU oracle standing in for the device), the
stacked result equals a single-process run over all frames."""
import importlib.util
import os
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_rank, q):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = _load("mdist", "thingino-accel_amd/dist.py")
    import marsfile
    import orcbind
    from conftest import lcg_frame
    model = None
    if rank == 0:
        model = _load("marsrt", "thingino-accel_amd/marsrt.py").synth_model(tiny=True, input_hw=24, seed=21)
    # descriptors to everyone, then the blob (on the GPUs: the packed parameter arena, device to device)
    desc = D.broadcast_bytes(dist, D.descriptor_bytes(model) if rank == 0 else b"", 0)
    blob_off = (len(desc) + 63) & ~63
    blob = D.broadcast_bytes(dist, model[blob_off:] if rank == 0 else b"", 0)
    mine = desc + bytes(blob_off - len(desc)) + blob
    hdr, tensors, _ = marsfile.parse(mine)
    nb, ob = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]]), marsfile.tensor_nbytes(tensors[hdr["outputs"][0]])
    frames = D.shard_frames(per_rank, rank, world)
    x = np.stack([lcg_frame(0x5EED0000 + f, nb) for f in frames])
    out = orcbind.run_frames(mine, x, ob)
    allout = D.gather_rows(dist, out)
    tmax = D.max_over_ranks(dist, 1.0 + rank)
    if rank == 0:
        q.put((model == mine, allout, tmax, D.strip_weights(model)[:blob_off] == model[:blob_off]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, per_rank, world = _free_port(), 3, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    same_model, allout, tmax, stripped_ok = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same_model and stripped_ok
    assert tmax == 2.0  # MAX over ranks
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import marsfile
    import orcbind
    from conftest import lcg_frame
    model = _load("marsrt", "thingino-accel_amd/marsrt.py").synth_model(tiny=True, input_hw=24, seed=21)
    hdr, tensors, _ = marsfile.parse(model)
    nb, ob = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]]), marsfile.tensor_nbytes(tensors[hdr["outputs"][0]])
    x = np.stack([lcg_frame(0x5EED0000 + f, nb) for f in range(per_rank * world)])
    want = orcbind.run_frames(model, x, ob)
    assert allout.shape == want.shape and np.array_equal(allout, want)


def test_sharding_helpers():
    D = _load("mdist", "thingino-accel_amd/dist.py")
    assert list(D.shard_frames(4, 2, 8)) == [8, 9, 10, 11]
    cover = [f for r in range(8) for f in D.shard_frames(128, r, 8)]
    assert cover == list(range(1024))  # config 4: 1024 frames = 8 x 128, no overlap, no gap
    parts = [list(D.split_frames(10, r, 4)) for r in range(4)]
    assert sum(parts, []) == list(range(10)) and [len(p) for p in parts] == [3, 3, 2, 2]
    with pytest.raises(ValueError):
        D.shard_frames(4, 8, 8)
    m = _load("marsrt", "thingino-accel_amd/marsrt.py").synth_model(tiny=True, input_hw=16)
    s = D.strip_weights(m)
    assert len(s) == len(m) and s != m and D.descriptor_bytes(s) == D.descriptor_bytes(m)
