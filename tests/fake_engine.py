"""Stand-in for HipEngine in the CPU tests of bench.py's launch / shard / gather / timing code (selected with
MASR_BENCH_ENGINE_FACTORY=tests.fake_engine:make).  It computes nothing of the hot path: ``transcribe_batch`` derives
deterministic token ids from the PCM so that sharding and gather mistakes are visible."""
import torch


class FakeEngine:
    def __init__(self, kind, device, vocab):
        self.kind, self.vocab_size = kind, vocab
        self.device = torch.device('cpu')
        self.calls = 0

    def out_frames(self, T):
        Tp = ((T - 1) // 2 - 1) // 2
        return (Tp + 1) // 2 if self.kind == 'efficient_conformer' else Tp

    def transcribe_batch(self, pcm, n_samples, out=None, **kw):
        B, n_max = pcm.shape
        Tp = self.out_frames(1 + (n_max - 400) // 160)
        tokens, ntok, score = out
        pos = torch.arange(Tp) * 640
        ids = (pcm[:, pos].to(torch.int64).abs() % (self.vocab_size - 3) + 3).to(torch.int32)
        tokens.copy_(ids)
        ntok.fill_(Tp // 2)
        score.copy_(pcm[:, 0].float())
        self.calls += 1
        return out

    def mean_square(self, pcm, n_samples, out=None):
        ms = (pcm.float() / 32768.0).pow(2).mean(dim=1)
        if out is not None:
            out.copy_(ms)
            return out
        return ms

    def transcribe_rows(self, pcm, n_samples, use_db_normalization=True, target_db=-20.0, gain_in=None, out=None, **kw):
        """the packed form [B, T' + 2] = tokens | count | score bits (what masr_transcribe_rows writes)"""
        assert gain_in is not None and gain_in.shape == (pcm.shape[0],) and bool((gain_in > 0).all())
        B, n_max = pcm.shape
        Tp = self.out_frames(1 + (n_max - 400) // 160)
        trip = (torch.empty(B, Tp, dtype=torch.int32), torch.empty(B, dtype=torch.int32), torch.empty(B, dtype=torch.float32))
        self.transcribe_batch(pcm, n_samples, out=trip)
        rows = torch.cat([trip[0], trip[1].view(-1, 1), trip[2].view(-1, 1).view(torch.int32)], dim=1)
        if out is not None:
            out.copy_(rows)
            return out
        return rows

    def profile_select(self, kind):
        pass

    def profile_read(self, reset=True):
        return 0.0, 0, 0.0

    def close(self):
        pass


def make(kind, device, vocab):
    return FakeEngine(kind, device, vocab)
