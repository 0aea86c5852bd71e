"""CPU, world_size 2 over gloo: the multi-GPU data path (tile shards into zero-initialised films,
reduce(sum) to rank 0) with the oracle standing in for the per-rank render."""
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
