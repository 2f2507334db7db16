"""Data-parallel sharding of the hot path across the GPUs of one node (BASELINE configs[3,4]).

One process per GPU (``torch.distributed``; on ROCm the "nccl" backend IS RCCL over xGMI, "gloo" runs the same code on CPU).
Utterances and streams are independent, weights are replicated, so there is NO collective on the data path: every rank runs
fbank -> encoder -> CTC decode on its own shard and the only exchange step is ONE all-gather of fixed-shape hypotheses per
batch (token ids int32 [B_local, T'] padded with -1 | token count | score bits = tens of KB: latency-, not bandwidth-bound, no
ring / all-reduce ever appears).  Streams are sticky to ``stream_id % world`` -- their key/value and conv caches never move.

Used by the product (``MASRPredictor.predict_batch`` / ``evaluate`` when a process group is initialised, ``ShardedStreamPool``)
and by ``bench.py`` (``timed_region``, ``gather_hypotheses``); the CPU tests drive exactly these functions on gloo.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist


# ---- process group ---------------------------------------------------------------------------------------------------
def world_info(group=None):
    """(rank, world) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (``torch.distributed.run`` sets them); returns
    (rank, world, local_rank).  No-op without WORLD_SIZE > 1.  Backend: RCCL ("nccl") when a GPU is visible, else gloo."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world <= 1 and os.environ.get('MASR_FORCE_DIST') != '1':
        return 0, 1, local
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def collectives_on(group=None):
    """True when the exchange steps really run: more than one rank, or MASR_FORCE_DIST=1 with an initialised group (a
    single-GPU box then goes through RCCL init, all-gather, all-reduce and barriers exactly like a multi-GPU job)"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get('MASR_FORCE_DIST') == '1'


def comm_device():
    """where collective payloads live: the current GPU under RCCL, the host under gloo"""
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


# ---- sharding --------------------------------------------------------------------------------------------------------
def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for this rank (first n_items % world ranks get +1)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def length_balanced_shards(lengths, world):
    """Utterance indices per rank: sort by length (longest first) and deal them out round-robin, so every rank holds the same
    mix of lengths (equal work, and each rank's shard is itself length-sorted: its padded sub-batches stay tight).
    Deterministic on every rank; ties keep the original order."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[r::world] for r in range(world)]


def sticky_stream_owner(stream_id, world):
    """Streams never migrate: their KV / conv caches stay on the owning GPU."""
    return stream_id % world


# ---- the exchange step -----------------------------------------------------------------------------------------------
def gather_hypothesis_rows(tokens, ntok, scores, group=None):
    """Per-rank hypotheses packed as ONE int32 payload [B, T' + 2] (tokens | count | score bits) and all-gathered over the
    ranks -> [world * B, T' + 2] on every rank (world 1: the packed rows themselves).  One tensor means one collective and one
    copy to the host; ``unpack_hypothesis_rows`` splits it again."""
    payload = torch.cat([tokens, ntok.view(-1, 1), scores.view(-1, 1).view(torch.int32)], dim=1).contiguous()
    return gather_rows(payload, group)


def gather_rows(rows, group=None):
    """packed hypothesis rows [B, T' + 2] int32 of this rank (what ``masr_transcribe_rows`` writes) -> the rows of all ranks
    [world * B, T' + 2] on every rank: ONE all-gather (world 1: the rows themselves)"""
    if not collectives_on(group):
        return rows
    w = dist.get_world_size(group)
    out = torch.empty((w * rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, rows.contiguous(), group=group)
    return out


def unpack_hypothesis_rows(rows):
    """[R, T' + 2] int32 rows (tensor or numpy array) -> tokens [R, T'], counts [R], scores f32 [R] (views where possible)"""
    Tp = rows.shape[1] - 2
    if torch.is_tensor(rows):
        return rows[:, :Tp], rows[:, Tp], rows[:, Tp + 1].contiguous().view(torch.float32)
    return rows[:, :Tp], rows[:, Tp], np.ascontiguousarray(rows[:, Tp + 1]).view(np.float32)


def gather_hypotheses(tokens, ntok, scores, group=None):
    """All-gather per-rank hypotheses.  tokens int32 [B, T'], ntok int32 [B], scores f32 [B] with the
    SAME shapes on every rank (pad shards with empty utterances).  Returns the concatenation over ranks."""
    if not collectives_on(group):
        return tokens, ntok, scores
    tok, nt, sc = unpack_hypothesis_rows(gather_hypothesis_rows(tokens, ntok, scores, group))
    return tok.contiguous(), nt.contiguous(), sc


def tokens_to_text(tokens, ntok, vocab):
    """token ids [B, T'] + counts [B] -> texts (``vocab``: list of tokens, or the same as a numpy object array)"""
    toks = tokens.cpu().numpy() if torch.is_tensor(tokens) else np.asarray(tokens)
    n = ntok.cpu().numpy() if torch.is_tensor(ntok) else np.asarray(ntok)
    va = vocab if isinstance(vocab, np.ndarray) else np.array(vocab, dtype=object)
    return [''.join(va[toks[i, :n[i]]]).replace('<space>', ' ') for i in range(toks.shape[0])]


def gather_sharded_results(local_tokens, local_ntok, local_scores, shards, n_items, group=None):
    """Rank-local hypotheses of ``shards[rank]`` (token rows in the order of that index list) -> hypotheses of all n_items
    utterances in their original order, on every rank.  Shards are padded to a common [per, T'] shape for the one all-gather
    (T' = max over ranks, agreed with a tiny all-reduce)."""
    rank, world = world_info(group)
    dev = comm_device()
    per = max(len(s) for s in shards) if shards else 0
    tp = torch.tensor([local_tokens.shape[1] if local_tokens.numel() else 0], dtype=torch.int32, device=dev)
    if collectives_on(group):
        dist.all_reduce(tp, op=dist.ReduceOp.MAX, group=group)
    Tp = max(int(tp.item()), 1)
    tok = torch.full((per, Tp), -1, dtype=torch.int32, device=dev)
    nt = torch.zeros(per, dtype=torch.int32, device=dev)
    sc = torch.zeros(per, dtype=torch.float32, device=dev)
    k = len(shards[rank])
    if k:
        tok[:k, :local_tokens.shape[1]] = local_tokens.to(dev)
        nt[:k] = local_ntok.to(dev)
        sc[:k] = local_scores.to(dev)
    tok, nt, sc = gather_hypotheses(tok, nt, sc, group)
    tok, nt, sc = tok.cpu().numpy(), nt.cpu().numpy(), sc.cpu().numpy()
    out_tok = np.full((n_items, Tp), -1, np.int32)
    out_nt = np.zeros(n_items, np.int32)
    out_sc = np.zeros(n_items, np.float32)
    for r, idx in enumerate(shards):
        for j, item in enumerate(idx):
            out_tok[item], out_nt[item], out_sc[item] = tok[r * per + j], nt[r * per + j], sc[r * per + j]
    return out_tok, out_nt, out_sc


# ---- timing contract of bench.py ---------------------------------------------------------------------------------------
def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_region(step, steps, warmup, group=None, after_warmup=None, flush=None, return_local=False):
    """``warmup`` untimed calls of ``step(i)``, then EXACTLY ``steps`` timed calls bracketed by a barrier + device synchronise
    on both sides; returns the MAX over ranks of the wall time of the timed region in seconds (identical on every rank).
    ``flush``: called after the last step INSIDE the timed region (a pipelined step finishes its last host-side stage there).
    ``return_local``: -> (max over ranks, this rank's own time up to its device synchronise, before the closing barrier)."""
    rank, world = world_info(group)
    for i in range(warmup):
        step(i)
    if flush is not None:
        flush()
    _sync()
    if after_warmup is not None:
        after_warmup()
    coll = collectives_on(group)
    if coll:
        dist.barrier(group=group)
    _sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    if flush is not None:
        flush()
    _sync()
    dt_local = time.perf_counter() - t0
    if coll:
        dist.barrier(group=group)
    dt = time.perf_counter() - t0
    if coll:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dt = float(t.item())
    return (dt, dt_local) if return_local else dt


def gather_floats(values, group=None):
    """concatenate a rank-local list of floats over ranks (latency samples), on every rank"""
    rank, world = world_info(group)
    if not collectives_on(group):
        return list(values)
    dev = comm_device()
    n = torch.tensor([len(values)], dtype=torch.int64, device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    buf = torch.full((int(n.item()),), float('nan'), dtype=torch.float64, device=dev)
    buf[:len(values)] = torch.tensor(list(values), dtype=torch.float64)
    out = torch.empty(world * buf.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.cpu().numpy()
    return [float(v) for v in out[~np.isnan(out)]]


# ---- sticky stream routing (BASELINE configs[4]) -----------------------------------------------------------------------
class ShardedStreamPool:
    """Concurrent ``predict_stream`` sessions spread over the ranks of a node.  Every rank constructs it around ITS OWN
    ``StreamPool`` (one engine per GPU); sessions carry global ids, session g lives on rank ``g % world`` for its whole life
    (caches never move).  ``feed`` is a no-op on ranks that do not own the session, ``step()`` advances the local sessions
    and returns the local partial results; ``step(gather=True)`` additionally all-gathers the partial results of all ranks
    (token ids, -1 padded) so that every rank sees {global id: {'text', 'score'} or None} -- the only exchange step.
    ``pool`` needs ``open() -> handle``, ``feed(handle, audio, is_end, **kw)``, ``step() -> {handle: result}``,
    ``close(handle)``, ``reset(handle)`` and ``vocab`` / ``last_tokens(handle)`` for the gather."""

    def __init__(self, pool, group=None):
        self.pool = pool
        self.group = group
        self.rank, self.world = world_info(group)
        self._local = {}          # global id -> local handle
        self._next = 0
        self._open = set()

    def owner(self, gid):
        return sticky_stream_owner(gid, self.world)

    def open(self):
        """collective: every rank calls it in the same order; returns the new global id"""
        gid = self._next
        self._next += 1
        self._open.add(gid)
        if self.owner(gid) == self.rank:
            self._local[gid] = self.pool.open()
        return gid

    def close(self, gid):
        self._open.discard(gid)
        if gid in self._local:
            self.pool.close(self._local.pop(gid))

    def reset(self, gid):
        if gid in self._local:
            self.pool.reset(self._local[gid])

    def feed(self, gid, audio_data, is_end=False, **kw):
        if gid in self._local:
            self.pool.feed(self._local[gid], audio_data, is_end, **kw)

    def local_ids(self):
        return sorted(self._local)

    def step(self, gather=False):
        res = self.pool.step()
        back = {h: g for g, h in self._local.items()}
        local = {back[h]: r for h, r in res.items()}
        if not gather or not collectives_on(self.group):
            return local
        # exchange: one int32 row per open session, [ntok (-1 = no result in this step) | score f64 bits (2) | tokens ...], rows
        # of a FIXED width (the pool's frame capacity bounds a transcript), so the step needs no size agreement: ONE all-gather.
        # A greedy StreamPool hands over the packed rows it already has on the device; nothing is rebuilt on the host.
        gids = sorted(self._open)
        per = max(sum(1 for g in gids if self.owner(g) == r) for r in range(self.world))
        mine = [g for g in gids if self.owner(g) == self.rank]
        width = int(getattr(self.pool, 'max_frames_out', 0) or 1024)
        dev = comm_device()
        packed = getattr(self.pool, 'last_packed', None)
        if packed is not None and packed[0].device.type == dev.type:
            rows_dev, tmax, sids = packed
            slot = {g: j for j, g in enumerate(mine)}
            at = torch.tensor([slot[back[h]] for h in sids], dtype=torch.long, device=rows_dev.device)
            ntok = rows_dev[:, tmax]
            score = rows_dev[:, tmax + 1].contiguous().view(torch.float32).double() * 100.0
            score = torch.where(ntok > 0, score, torch.zeros_like(score))
            mine_t = torch.full((per, 3 + width), -1, dtype=torch.int32, device=rows_dev.device)
            mine_t[at, 0] = ntok
            mine_t[at, 1:3] = score.view(torch.int32).view(-1, 2)
            mine_t[at, 3:3 + min(tmax, width)] = rows_dev[:, :min(tmax, width)]
        else:
            payload = np.full((per, 3 + width), -1, np.int32)
            for j, g in enumerate(mine):
                r = local.get(g)
                if r is not None:
                    toks = np.asarray(self.pool.last_tokens(self._local[g]), np.int32)[:width]
                    payload[j, 0] = len(toks)
                    payload[j, 1:3] = np.array([r['score']], np.float64).view(np.int32)
                    payload[j, 3:3 + len(toks)] = toks
            mine_t = torch.from_numpy(payload).to(dev)
        allp = torch.empty((self.world * per, 3 + width), dtype=torch.int32, device=mine_t.device)
        dist.all_gather_into_tensor(allp, mine_t, group=self.group)
        allp = allp.cpu().numpy()
        out = {}
        for r in range(self.world):
            for j, g in enumerate([g for g in gids if self.owner(g) == r]):
                row = allp[r * per + j]
                if row[0] < 0:
                    out[g] = None
                else:
                    text = ''.join(self.pool.vocab[k] for k in row[3:3 + row[0]]).replace('<space>', ' ')
                    out[g] = {'text': text, 'score': float(np.ascontiguousarray(row[1:3]).view(np.float64)[0])}
        return out
