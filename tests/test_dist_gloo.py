"""N>1 path on CPU: world_size-2 gloo.  Blocks shard round-robin (block i -> rank i mod N),
every rank encodes only its own blocks (here with the oracle as the stand-in encoder, since the HIP
path needs a GPU), rank 0 gathers (blockId, bits, bytes) and assembles the .knz with the product's
host-only container code.  The result must equal the single-process stream: no data-path collective."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import kanzi_amd as kz
    import oracle
    import datagen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bs, nblocks = 16384, 7
    data = datagen.stream(nblocks, bs).tobytes()[:-100]           # ragged last block
    chain, ent = "BWT+RANK+ZRLT", "ANS0"
    mine = kz.shard_blocks(nblocks, world, rank)
    local = []
    for i in mine:
        s, w, _, _ = oracle.encode_block(chain, ent, data[i * bs:(i + 1) * bs])
        local.append((i, w, s))
    # timing reduce as bench.py does it: MAX over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    if rank == 0:
        allb = sorted(x for part in gathered for x in part)
        assert [i for i, _, _ in allb] == list(range(nblocks))      # every block exactly once
        knz = kz.knz_assemble(chain, ent, bs, len(data), [s for _, _, s in allb], [w for _, w, _ in allb])
        ref = oracle.compress(chain, ent, bs, data, jobs=1)
        q.put((knz == ref, float(t.item()), len(knz)))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_two_ranks_gloo():
    import kanzi_amd as kz
    assert kz.shard_blocks(7, 2, 0) == [0, 2, 4, 6] and kz.shard_blocks(7, 2, 1) == [1, 3, 5]
    assert sorted(kz.shard_blocks(239, 8, r)[k] for r in range(8) for k in range(len(kz.shard_blocks(239, 8, r)))) == list(range(239))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, tmax, n = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same and tmax == 2.0 and n > 0


def _worker_stream(rank, world, port, q):
    """4 ranks, 239 blocks (enwik9's count: shares of 60 / 60 / 60 / 59), ragged tail.  Rank 0 does not gather everything first: round
    r brings blocks r * N .. r * N + N - 1, which go straight into the streaming writer (kz_knz_writer_*)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import kanzi_amd as kz
    import oracle
    import datagen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bs, nblocks = 2048, 239
    data = datagen.stream(nblocks, bs).tobytes()[:-777]
    chain, ent = "BWT+RANK+ZRLT", "ANS0"
    mine = kz.shard_blocks(nblocks, world, rank)
    assert len(mine) == (60 if rank < 3 else 59)
    rounds = (nblocks + world - 1) // world
    writer = kz.KnzWriter(chain, ent, bs, len(data), kz.load_library().kz_compress_bound(len(data), bs)) if rank == 0 else None
    held = 0
    for r in range(rounds):
        i = r * world + rank
        item = None
        if i < nblocks:
            s, w, _, _ = oracle.encode_block(chain, ent, data[i * bs:(i + 1) * bs])
            item = (i, w, s)
        got = [None] * world if rank == 0 else None
        dist.gather_object(item, got, dst=0)
        if rank == 0:
            held = max(held, sum(1 for x in got if x is not None))
            for k, x in enumerate(got):
                if x is not None:
                    assert x[0] == r * world + k                      # block-id order without sorting
                    writer.add(x[2], x[1])
    if rank == 0:
        knz = writer.close()
        ref = oracle.compress(chain, ent, bs, data, jobs=2)
        q.put((knz == ref, held, len(kz.knz_index(knz)["blocks"])))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_four_ranks_streaming_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stream, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    same, held, nb = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same and held == 4 and nb == 239
