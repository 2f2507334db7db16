"""Data-parallel sharding of the hot path across the GPUs of one node.

Utterances (and streams) are independent: every rank owns a contiguous shard, runs
fbank -> encoder -> greedy locally, and the ONLY exchange is an all-gather of fixed-shape hypothesis
tensors (token ids int32 [B_local, T'] padded with -1, token counts, scores).  On ROCm the "nccl"
backend is RCCL over xGMI; the payload is tens of KB, so the step is latency- not bandwidth-bound
and no ring/all-reduce ever appears on the path.  The same code runs on "gloo" for CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for this rank (first n_items % world ranks get +1)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def sticky_stream_owner(stream_id, world):
    """Streams never migrate: their KV / conv caches stay on the owning GPU."""
    return stream_id % world


def gather_hypotheses(tokens, ntok, scores, group=None):
    """All-gather per-rank hypotheses.  tokens int32 [B, T'], ntok int32 [B], scores f32 [B] with the
    SAME shapes on every rank (pad shards with empty utterances).  Returns the concatenation over ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tokens, ntok, scores
    w = dist.get_world_size(group)
    # one payload: [B, T' + 2] int32 rows = tokens | ntok | score bits
    payload = torch.cat([tokens, ntok.view(-1, 1), scores.view(-1, 1).view(torch.int32)], dim=1).contiguous()
    out = torch.empty((w * payload.shape[0], payload.shape[1]), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(out, payload, group=group)
    Tp = tokens.shape[1]
    return out[:, :Tp].contiguous(), out[:, Tp].contiguous(), out[:, Tp + 1].contiguous().view(torch.float32)


def tokens_to_text(tokens, ntok, vocab):
    toks = tokens.cpu().numpy()
    n = ntok.cpu().numpy()
    return [''.join(vocab[j] for j in toks[i, :n[i]]).replace('<space>', ' ') for i in range(toks.shape[0])]
