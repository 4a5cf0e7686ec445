"""The N>1 path of bench.py on CPU: two gloo ranks own disjoint row shards, process them
independently and meet only in the final count/time reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
import benchgen as bg
import oracle_lib as orc

ROWS, BLOCKS, NS = 300, 2, 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_output(rank):
    cfg = bg.make_cfg("c4", n_samples=NS)
    hdr = bg.header(cfg)
    outs = []
    for first in bench.rank_blocks(rank, BLOCKS, ROWS):
        rc, out, _, n = orc.run(hdr + bg.rows_host(cfg, first, ROWS))
        assert rc == 0 and n == ROWS
        outs.append(out)
    return b"".join(outs)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _rank_output(rank)
    elapsed, total = bench.reduce_over_ranks(1.0 + rank, ROWS * BLOCKS, "cpu", world)
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put((elapsed, total, b"".join(gathered)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_rows_without_overlap():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT, q)) for r in range(world)]
    for p in procs:
        p.start()
    elapsed, total, sharded = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert elapsed == 2.0            # max over ranks
    assert total == world * ROWS * BLOCKS  # sum over ranks
    # the union of the shards is the single-process result over the same rows
    cfg = bg.make_cfg("c4", n_samples=NS)
    rc, single, _, n = orc.run(bg.header(cfg) + bg.rows_host(cfg, 0, world * BLOCKS * ROWS))
    assert rc == 0 and n == world * BLOCKS * ROWS
    assert sharded == single


PORT = _free_port()


def test_rank_blocks_are_disjoint_and_contiguous():
    seen = []
    for r in range(8):
        seen += bench.rank_blocks(r, 4, 1000)
    assert seen == list(range(0, 8 * 4 * 1000, 1000))
