"""CPU, world_size 2 over gloo: the multi-GPU data paths -- (a) tile shards into zero-initialised films,
reduce(sum) to rank 0; (b) tile-major slabs gathered to rank 0 and scattered into the film (what bench.py
does by default) -- with the oracle standing in for the per-rank render."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_film


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ps, w, h, spp, depth, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    film = torch.from_numpy(oracle.render_shard(ps, (w, h), spp, depth, rank, world, threads=2))
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)  # exact: every pixel is x + 0
    if rank == 0:
        np.save(out_path, film.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_shards_reduce_to_the_full_film(tmp_path, oracle):
    ref, ps, spp, depth = load_film("cbox_committed_ragged_45x37_s8_d3")
    h, w, _ = ref.shape
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, _free_port(), ps, w, h, spp, depth, out), nprocs=2, join=True)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _slab_of(film, w, h, rank, world):
    """Host statement of what pine_gpu_plan_launch_packed writes: this rank's tiles, tile-major."""
    import pine_amd
    from pine_amd import _lib
    n = int(_lib.lib.pine_gpu_packed_slab_floats(w, h, world))
    slab = np.zeros((n // 4, 4), dtype=np.float32)
    for y in range(h):
        for x in range(w):
            r, o = pine_amd.packed_offset((w, h), world, x, y)
            if r == rank:
                slab[o] = film[y, x]
    return slab.reshape(-1)


def _gather_worker(rank, world, port, ps, w, h, spp, depth, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    import pine_amd
    shard = oracle.render_shard(ps, (w, h), spp, depth, rank, world, threads=2)
    slab = torch.from_numpy(_slab_of(shard, w, h, rank, world))
    slabs = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
    dist.gather(slab, slabs, dst=0)
    if rank == 0:
        film = np.zeros((h, w, 4), dtype=np.float32)
        for y in range(h):
            for x in range(w):
                r, o = pine_amd.packed_offset((w, h), world, x, y)
                film[y, x] = slabs[r].numpy().reshape(-1, 4)[o]
        np.save(out_path, film)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slabs_gather_to_the_full_film(tmp_path, oracle):
    ref, ps, spp, depth = load_film("cbox_committed_ragged_45x37_s8_d3")
    h, w, _ = ref.shape
    out = str(tmp_path / "film_gather.npy")
    mp.spawn(_gather_worker, args=(2, _free_port(), ps, w, h, spp, depth, out), nprocs=2, join=True)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_slab_layout_covers_every_pixel_once():
    import pine_amd
    from pine_amd import _lib
    for (w, h, world) in [(45, 37, 1), (45, 37, 2), (64, 64, 3), (17, 9, 8)]:
        n4 = int(_lib.lib.pine_gpu_packed_slab_floats(w, h, world)) // 4
        seen = set()
        for y in range(h):
            for x in range(w):
                r, o = pine_amd.packed_offset((w, h), world, x, y)
                assert 0 <= r < world and 0 <= o < n4
                assert r == _lib.lib.pine_gpu_shard_of_pixel(w, x, y, world)
                seen.add((r, o))
        assert len(seen) == w * h
