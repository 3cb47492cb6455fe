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


# ---- the job launcher behind `bench.py --gpus N` (dist.spawn_ranks): environment, relay, failure handling -------------

_CHILD = r"""
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "fail" and r == 1:
    sys.exit(7)
if mode == "fail" and r != 1:
    time.sleep(60)  # must be stopped by the launcher, not waited for
if mode == "hang":
    time.sleep(60)
print('{"rank": %d, "world": %d, "port": %s, "x": "%s"}' % (r, w, os.environ["MASTER_PORT"], os.environ.get("EXTRA", "")))
"""


def test_spawn_ranks_env_relay_and_failure():
    import json
    import sys
    import time
    D = _load("mdist", "thingino-accel_amd/dist.py")
    rc, out, codes = D.spawn_ranks(3, [sys.executable, "-c", _CHILD, "ok"], extra_env={"EXTRA": "e"})
    assert rc == 0 and codes == [0, 0, 0]
    d = json.loads(out)  # rank 0's stdout only: the other ranks' lines went to stderr
    assert d["rank"] == 0 and d["world"] == 3 and d["x"] == "e" and d["port"] > 0
    t0 = time.time()
    rc, out, codes = D.spawn_ranks(3, [sys.executable, "-c", _CHILD, "fail"])
    assert rc == 7 and codes[1] == 7 and out == "" and time.time() - t0 < 30  # ranks 0 and 2 were terminated, not joined
    assert all(c is not None for c in codes)
    t0 = time.time()
    rc, out, codes = D.spawn_ranks(2, [sys.executable, "-c", _CHILD, "hang"], timeout=1.0)
    assert rc != 0 and time.time() - t0 < 30
    with pytest.raises(ValueError):
        D.spawn_ranks(0, [sys.executable, "-c", "pass"])
    e = D.rank_env(2, 8, 1234, base={})
    assert e["RANK"] == "2" and e["LOCAL_RANK"] == "2" and e["WORLD_SIZE"] == "8" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_entry_point_refuses_a_rank_count_it_was_not_asked_for():
    """`bench.py --gpus N`: under a launcher WORLD_SIZE must equal N; without one it starts N ranks itself -- here (no GPU)
    they fail at device selection, and the parent must fail too, printing no JSON line"""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "1"], env=dict(os.environ, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and r.stdout == ""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: the ranks find no device
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout == "" and "2-rank job failed" in r.stderr
