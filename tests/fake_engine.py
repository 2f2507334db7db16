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

    def profile_select(self, kind):
        pass

    def profile_read(self, reset=True):
        return 0.0, 0, 0.0

    def close(self):
        pass


def make(kind, device, vocab):
    return FakeEngine(kind, device, vocab)
