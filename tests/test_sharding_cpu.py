"""The N>1 path on CPU: world_size-2 gloo processes shard images by index and all-gather the final latents."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stablediffusioneo_amd.sharding import gather_latents, shard_indices, unit_index


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def fake_latent(i):
    g = torch.Generator().manual_seed(2946901 + i)
    return torch.randn((4, 8, 8), generator=g)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = shard_indices(total, rank, world)
        local = torch.stack([fake_latent(i) for i in idx]) if idx else torch.empty((0, 4, 8, 8))
        full = gather_latents(local, total)
        ref = torch.stack([fake_latent(i) for i in range(total)])
        q.put((rank, bool(torch.equal(full, ref)), idx))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5, 2])
def test_gloo_world2_image_sharding(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    seen = sorted(i for _, _, idx in res for i in idx)
    assert seen == list(range(total))          # every image exactly once, no overlap


def _bench_worker(rank, world, port, steps, batch, q):
    """the N > 1 data flow of bench.py, on CPU tensors: every rank runs `steps` units of `batch` images, unit index from
    unit_index(rank, step, world), one all-gather of the (steps, batch*4, h, w) latents at the end"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        zs = []
        for i in range(steps):
            u = unit_index(rank, i, world)
            zs.append(torch.stack([fake_latent(u * batch + j) for j in range(batch)]))
        zloc = torch.stack(zs)
        zall = gather_latents(zloc.reshape(steps, -1, 8, 8), world * steps).reshape(world * steps, batch, 4, 8, 8)
        ref = torch.stack([torch.stack([fake_latent(u * batch + j) for j in range(batch)]) for u in range(world * steps)])
        q.put((rank, bool(torch.equal(zall, ref)), [unit_index(rank, i, world) for i in range(steps)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("steps,batch", [(3, 1), (2, 2), (1, 1)])
def test_bench_unit_mapping_and_gather_order(steps, batch):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, steps, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    for rank, _, units in res:
        assert units == shard_indices(world * steps, rank, world)      # bench's mapping IS the sharding rule
    assert sorted(u for _, _, units in res for u in units) == list(range(world * steps))


def test_shard_indices_properties():
    for world in (1, 2, 4, 8):
        for total in (0, 1, 7, 8, 16):
            allidx = [i for r in range(world) for i in shard_indices(total, r, world)]
            assert sorted(allidx) == list(range(total))
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def test_single_process_passthrough():
    z = torch.randn(3, 4, 8, 8)
    assert gather_latents(z, 3) is z


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` needs no external launcher: the parent (which never touches a device) starts one child per GPU with
    the torch.distributed environment set.  The children only report that environment here (SDEO_BENCH_ECHO_ENV)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = str(tmp_path / "env")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["SDEO_BENCH_ECHO_ENV"] = base
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, timeout=300)
    assert r.returncode == 0
    got = [json.load(open(f"{base}.{i}")) for i in range(2)]
    assert [g["RANK"] for g in got] == ["0", "1"] and [g["LOCAL_RANK"] for g in got] == ["0", "1"]
    assert all(g["WORLD_SIZE"] == "2" and g["MASTER_ADDR"] == "127.0.0.1" for g in got)
    assert got[0]["MASTER_PORT"] == got[1]["MASTER_PORT"] and int(got[0]["MASTER_PORT"]) > 0
    # under an external launcher (RANK already set) nothing is spawned: the process reports its own environment
    env2 = dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", SDEO_BENCH_ECHO_ENV=base + "x")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env2, timeout=300)
    assert r.returncode == 0 and json.load(open(base + "x.1"))["MASTER_PORT"] == "29511" and not os.path.exists(base + "x.0")
