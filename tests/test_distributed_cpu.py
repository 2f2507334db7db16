"""CPU, world_size 2, gloo: the multi-GPU path of the hot path = shard utterances, gather hypotheses."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from masr_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, Tp, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = parallel.shard_range(n_items, rank, world)
    per = -(-n_items // world)
    tokens = torch.full((per, Tp), -1, dtype=torch.int32)
    ntok = torch.zeros(per, dtype=torch.int32)
    score = torch.zeros(per, dtype=torch.float32)
    for j, item in enumerate(range(lo, hi)):          # fake local "hypotheses": item i -> tokens [i, i+1, ..]
        n = 1 + item % (Tp - 1)
        tokens[j, :n] = torch.arange(item, item + n, dtype=torch.int32)
        ntok[j] = n
        score[j] = 0.5 + item
    t, n, s = parallel.gather_hypotheses(tokens, ntok, score)
    if rank == 0:
        q.put((t.numpy(), n.numpy(), s.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 255, 256):
        for w in (1, 2, 3, 8):
            r = [parallel.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert [parallel.sticky_stream_owner(s, 8) for s in (0, 7, 8, 129)] == [0, 7, 0, 1]


def test_gather_hypotheses_world2_gloo():
    world, n_items, Tp = 2, 7, 12
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, Tp, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, n, s = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    per = -(-n_items // world)
    assert t.shape == (world * per, Tp)
    seen = []
    for r in range(world):
        lo, hi = parallel.shard_range(n_items, r, world)
        for j, item in enumerate(range(lo, hi)):
            row = r * per + j
            k = 1 + item % (Tp - 1)
            assert n[row] == k and list(t[row, :k]) == list(range(item, item + k)) and (t[row, k:] == -1).all()
            assert s[row] == 0.5 + item
            seen.append(item)
    assert seen == list(range(n_items))
