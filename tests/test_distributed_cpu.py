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


# ---- the product's sharding helpers and bench.py's step / gather / timing code on 2 gloo ranks ----------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(target, world, *args, timeout=300):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=timeout)
        assert p.exitcode == 0
    return dict(out)


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    return parallel.init_from_env('gloo')


def _fake_tokens(item, length):
    return [3 + (item * 7 + k) % 50 for k in range(1 + length % 9)]


def _sharded_worker(rank, world, port, q, lengths):
    _init(rank, world, port)
    shards = parallel.length_balanced_shards(lengths, world)
    mine = shards[rank]
    rows = [_fake_tokens(i, lengths[i]) for i in mine]
    tmax = max(len(r) for r in rows)
    tok = torch.full((len(mine), tmax), -1, dtype=torch.int32)
    nt = torch.zeros(len(mine), dtype=torch.int32)
    sc = torch.zeros(len(mine), dtype=torch.float32)
    for j, r in enumerate(rows):
        tok[j, :len(r)] = torch.tensor(r, dtype=torch.int32)
        nt[j], sc[j] = len(r), 0.25 * mine[j]
    t, n, s = parallel.gather_sharded_results(tok, nt, sc, shards, len(lengths))
    q.put((rank, (t.tolist(), n.tolist(), s.tolist())))
    dist.barrier()
    dist.destroy_process_group()


def test_length_balanced_shards_and_gather_world2():
    """what MASRPredictor.predict_batch / evaluate do under a process group: deal the utterances out length-balanced, decode
    the local shard, ONE all-gather, every rank ends with all hypotheses in input order"""
    lengths = [160000, 32000, 96000, 320000, 48000, 200000, 64000]
    shards = parallel.length_balanced_shards(lengths, 2)
    assert sorted(shards[0] + shards[1]) == list(range(7)) and abs(len(shards[0]) - len(shards[1])) <= 1
    for s in shards:                                         # each shard is itself sorted longest-first
        assert [lengths[i] for i in s] == sorted((lengths[i] for i in s), reverse=True)
    tot = [sum(lengths[i] for i in s) for s in shards]
    assert abs(tot[0] - tot[1]) <= max(lengths)
    res = _spawn(_sharded_worker, 2, lengths)
    assert res[0] == res[1]                                  # every rank holds the same gathered result
    toks, n, sc = res[0]
    for i, L in enumerate(lengths):
        want = _fake_tokens(i, L)
        assert n[i] == len(want) and toks[i][:n[i]] == want and all(v == -1 for v in toks[i][n[i]:])
        assert sc[i] == 0.25 * i


def test_length_balanced_shards_and_gather_world4_uneven():
    """30 utterances over FOUR gloo ranks (shards of 8 / 8 / 7 / 7: an uneven deal, the gather pads the short shards): every rank
    ends with all 30 hypotheses in input order -- the 4-GPU point of the scaling sweep, on CPU"""
    rng = __import__('numpy').random.default_rng(5)
    lengths = [int(v) for v in rng.integers(32000, 320001, 30)]
    shards = parallel.length_balanced_shards(lengths, 4)
    assert sorted(i for s in shards for i in s) == list(range(30))
    assert sorted(len(s) for s in shards) == [7, 7, 8, 8]
    tot = [sum(lengths[i] for i in s) for s in shards]
    assert max(tot) - min(tot) <= max(lengths)               # length-balanced, not just count-balanced
    res = _spawn(_sharded_worker, 4, lengths)
    assert res[0] == res[1] == res[2] == res[3]
    toks, n, sc = res[0]
    for i, L in enumerate(lengths):
        want = _fake_tokens(i, L)
        assert n[i] == len(want) and toks[i][:n[i]] == want and all(v == -1 for v in toks[i][n[i]:])
        assert sc[i] == 0.25 * i


class _Pool:
    """StreamPool interface stand-in: a session's tokens are the byte counts of the chunks it has been fed"""
    vocab = [str(i) for i in range(100)]

    def __init__(self):
        self.s, self.fed = {}, {}

    def open(self):
        h = len(self.s)
        self.s[h] = []
        return h

    def close(self, h):
        self.s.pop(h)

    def reset(self, h):
        self.s[h] = []

    def feed(self, h, data, is_end=False, **kw):
        self.s[h].append(len(data) % 100)
        self.fed[h] = is_end

    def last_tokens(self, h):
        return self.s[h]

    def step(self):
        fed, self.fed = self.fed, {}
        return {h: ({'text': ''.join(self.vocab[t] for t in self.s[h]), 'score': 0.5 * len(self.s[h]) + 1e-9}
                    if len(self.s[h]) >= 2 else None) for h in fed}


def _stream_worker(rank, world, port, q):
    _init(rank, world, port)
    pool = parallel.ShardedStreamPool(_Pool())
    gids = [pool.open() for _ in range(5)]
    assert pool.local_ids() == [g for g in gids if g % world == rank]      # sticky: stream_id % world
    log = []
    for c in range(3):
        for g in gids:
            if not (c == 1 and g == 4):                                      # stream 4 skips the second round
                pool.feed(g, b'x' * (10 * g + c + 1))
        log.append(pool.step(gather=True))
    pool.close(gids[0])
    q.put((rank, log))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stream_pool_world2():
    res = _spawn(_stream_worker, 2)
    assert res[0] == res[1]
    log = res[0]
    assert all(v is None for v in log[0].values()) and len(log[0]) == 5       # one chunk each: no partial result yet
    assert log[1][4] is None and log[1][3] == {'text': '3132', 'score': 1.0 + 1e-9}
    assert log[2][4] == {'text': '4143', 'score': 1.0 + 1e-9} and log[2][0]['text'] == '123'


def _contract_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ['MASR_BENCH_ENGINE_FACTORY'] = 'tests.fake_engine:make'
    _init(rank, world, port)
    import bench
    from masr_amd.utils import synthetic
    eng = bench.make_engine('conformer', rank)
    cs = bench.ContractStep(eng, rank, world, synthetic.synthetic_vocab(bench.VOCAB))
    dt = parallel.timed_region(lambda i: cs.step(i, 'full'), 2, 1, flush=cs.flush)
    assert cs.pending is None and dt > 0 and eng.calls == 3
    q.put((rank, (cs.texts, cs.n_texts, dt)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_contract_step_world2():
    """bench.py's own step (transcribe -> all-gather -> D2H -> text, pipelined) and timing bracket on two gloo ranks"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from masr_amd.utils import synthetic
    from tests import fake_engine
    res = _spawn(_contract_worker, 2)
    vocab = synthetic.synthetic_vocab(bench.VOCAB)
    want = []
    for r in range(2):                                       # what each rank's stand-in engine emits for ITS batch
        eng = fake_engine.make('conformer', r, bench.VOCAB)
        pcm = torch.from_numpy(synthetic.synthetic_pcm(bench.BATCH, bench.N_SAMPLES, seed=1234 + r))
        Tp = eng.out_frames(1 + (bench.N_SAMPLES - 400) // 160)
        out = (torch.empty(bench.BATCH, Tp, dtype=torch.int32), torch.empty(bench.BATCH, dtype=torch.int32),
               torch.empty(bench.BATCH, dtype=torch.float32))
        eng.transcribe_batch(pcm, None, out=out)
        want.append(parallel.tokens_to_text(out[0], out[1], vocab))
    texts0, n0, dt0 = res[0]
    texts1, n1, dt1 = res[1]
    assert texts0 == want[0] + want[1]                       # rank 0 holds the transcripts of the whole job, in rank order
    assert texts1 == want[1]                                 # the other ranks build their own shard
    assert n0 == 3 * 64 and n1 == 3 * 32 and dt0 == dt1      # max-over-ranks time is the same number everywhere


def test_bench_cli_launches_ranks_and_rejects_mismatch():
    """--gpus 2 outside a launcher starts 2 ranks itself (the CLI the driver calls); a WORLD_SIZE that disagrees is an error"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASR_BENCH_ENGINE_FACTORY='tests.fake_engine:make', PYTHONPATH=ROOT)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-extra', '--no-cpu-baseline']
    p = subprocess.run(cmd + ['--gpus', '2'], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(line) == 1                                    # ONE JSON line, from rank 0
    r = json.loads(line[0])
    assert r['n_gpus'] == 2 and r['config']['world_size_observed'] == 2 and r['config']['backend'] == 'gloo'
    assert r['config']['global_batch'] == 64 and r['steps'] == 2 and r['value'] > 0 and 'STAND-IN' in r['data']
    assert set(r['timing']) == {'device_only', 'host_to_host'}
    # the exchange of the timed region's last step delivered every rank's rows; both ranks report their own time
    assert r['config']['rccl_ranks_seen'] == [0, 1]
    assert 0 < r['config']['ms_per_step_per_rank']['min'] <= r['config']['ms_per_step_per_rank']['max'] <= r['ms_per_step'] * 1.001
    assert 'use_db_normalization = 2' in r['config']['normalisation'] and 'mode 2' in r['config']['timed_region']
    p1 = subprocess.run(cmd + ['--gpus', '1'], env=env, capture_output=True, text=True, timeout=600)
    r1 = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith('{')][0])
    assert r1['n_gpus'] == 1 and r1['config']['global_batch'] == 32
    bad = subprocess.run(cmd + ['--gpus', '2'], env=dict(env, WORLD_SIZE='4', RANK='0'), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE=4' in (bad.stderr + bad.stdout)
    # configs[3] as its own workload on 2 ranks: 256 utterances sharded 128 / 128, gathered, 256 transcripts on rank 0
    p3 = subprocess.run(cmd + ['--gpus', '2', '--workload', 'efficient_b256'], env=env, capture_output=True, text=True, timeout=600)
    assert p3.returncode == 0, p3.stderr[-2000:]
    r3 = json.loads([ln for ln in p3.stdout.splitlines() if ln.startswith('{')][0])
    assert r3['n_gpus'] == 2 and r3['transcripts'] == 256 and r3['scaling'] == 'strong' and '128 on rank 0' in r3['workload']


def test_hypothesis_rows_pack_and_unpack_round_trip():
    """the one-payload form of a batch of hypotheses ([B, T'+2] int32: tokens | count | score bits) -- what travels through the
    all-gather and the single copy to the host -- unpacks to the same tokens, counts and float32 scores (tensor and numpy)"""
    import numpy as np
    import torch
    from masr_amd import parallel
    g = torch.Generator().manual_seed(5)
    tok = torch.randint(0, 4233, (7, 13), generator=g, dtype=torch.int32)
    nt = torch.randint(0, 14, (7,), generator=g, dtype=torch.int32)
    sc = torch.randn(7, generator=g) * 50
    rows = parallel.gather_hypothesis_rows(tok, nt, sc)          # no process group: the packed rows themselves
    assert rows.shape == (7, 15) and rows.dtype == torch.int32
    for unpacked in (parallel.unpack_hypothesis_rows(rows), parallel.unpack_hypothesis_rows(rows.numpy())):
        t, n, s = (torch.as_tensor(np.asarray(u)) for u in unpacked)
        assert torch.equal(t, tok) and torch.equal(n, nt) and torch.equal(s, sc)
    t, n, s = parallel.gather_hypotheses(tok, nt, sc)
    assert torch.equal(t, tok) and torch.equal(n, nt) and torch.equal(s, sc)
