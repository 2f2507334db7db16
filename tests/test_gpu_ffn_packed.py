"""The full (non-split) FFN launches read PACKED weight copies straight into MFMA operand registers (ffn_pc.hip VAR == 2,
round 3) instead of staging them through wave-private LDS slabs: the same operand values in the same MFMA order, so the encoder
output must be BIT-identical with the switch (masr_debug_set key 23) on and off -- for every family that uses the kernel, with
the QKV tail and conv-module head stages riding on it."""
import numpy as np
import pytest
import torch

from conftest import needs_experiments

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kind,streaming', [('conformer', True), ('conformer', False), ('squeezeformer', False),
                                            ('efficient_conformer', True)])
def test_packed_weights_are_bit_identical_to_the_slab_pipeline(kind, streaming):
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    V = 512
    sd = getattr(synthetic, kind + '_state_dict')(0, V)
    eng = HipEngine(sd, vocab_size=V, use_model=kind, streaming=streaming)
    rng = np.random.default_rng(3)
    lens = rng.integers(60000, 160001, 32).astype(np.int32)
    pcm = synthetic.synthetic_pcm(32, 160000, seed=9)
    for i, l in enumerate(lens):
        pcm[i, l:] = 0
    feats, frames = eng.fbank_batch(torch.from_numpy(pcm).cuda(), torch.from_numpy(lens).cuda())
    out = {}
    try:
        for v in (1, 0, 1):
            eng.lib.masr_debug_set(eng.h, 23, v)
            out[v] = eng.encode_full(feats, frames, -1).clone()
            probs = eng.ctc_probs(out[v])
            out[('p', v)] = probs.clone()
    finally:
        eng.lib.masr_debug_set(eng.h, 23, 1)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]) and torch.equal(out[('p', 0)], out[('p', 1)])
    assert float(out[1].abs().max()) > 0
    eng.close()


@needs_experiments()
@pytest.mark.parametrize('kind,streaming', [('conformer', True), ('conformer', False), ('squeezeformer', False),
                                            ('efficient_conformer', True)])
def test_two_chain_ffn_is_bit_identical_to_the_single_chain_kernel(kind, streaming):
    """ffn_dual.hip (two independent accumulator chains per wave over chunks of 256 hidden units, three in the QKV tail stage)
    sums every accumulator's k in the same ascending order as ffn_pc.hip: encoder output and probabilities BIT-identical with
    masr_debug_set key 24 on and off; ragged batch with a partial last row block."""
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    V = 512
    sd = getattr(synthetic, kind + '_state_dict')(0, V)
    eng = HipEngine(sd, vocab_size=V, use_model=kind, streaming=streaming)
    rng = np.random.default_rng(5)
    lens = rng.integers(50000, 160001, 31).astype(np.int32)
    lens[0] = 160000
    pcm = synthetic.synthetic_pcm(31, 160000, seed=11)
    for i, l in enumerate(lens):
        pcm[i, l:] = 0
    feats, frames = eng.fbank_batch(torch.from_numpy(pcm).cuda(), torch.from_numpy(lens).cuda())
    out = {}
    try:
        for v in (1, 0, 1):
            eng.lib.masr_debug_set(eng.h, 24, v)
            out[v] = eng.encode_full(feats, frames, -1).clone()
            out[('p', v)] = eng.ctc_probs(out[v]).clone()
    finally:
        eng.lib.masr_debug_set(eng.h, 24, 1)
    torch.cuda.synchronize()
    assert torch.isfinite(out[1]).all() and float(out[1].abs().max()) > 0
    assert torch.equal(out[0], out[1]), float((out[0] - out[1]).abs().max())
    assert torch.equal(out[('p', 0)], out[('p', 1)])
    eng.close()


@pytest.mark.parametrize('kind,streaming', [('conformer', True), ('conformer', False), ('squeezeformer', False), ('squeezeformer', True),
                                            ('efficient_conformer', True)])
def test_packed_row_block_projections_are_bit_identical_to_the_slab_pipeline(kind, streaming):
    """rowgemm.hip: every full row-block launch (LN -> QKV, out-projection, pointwise convolutions, the out-proj + pw1 chain, the
    fused CTC head) reads a packed copy of its weights with buffer loads (masr_debug_set key 25, default on) instead of staging
    them through wave-private LDS slabs: same operands in the same MFMA order, so encoder output, frame argmax and frame
    probability are BIT-identical with the switch on and off -- for every family, ragged batch, partial last row block."""
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    V = 600                                    # not a multiple of 256: the packed CTC weights are zero-padded to 768 rows
    sd = getattr(synthetic, kind + '_state_dict')(0, V, **({'streaming': streaming} if kind == 'squeezeformer' else {}))
    eng = HipEngine(sd, vocab_size=V, use_model=kind, streaming=streaming)
    rng = np.random.default_rng(7)
    lens = rng.integers(50000, 160001, 31).astype(np.int32)
    lens[0] = 160000
    pcm = synthetic.synthetic_pcm(31, 160000, seed=13)
    for i, l in enumerate(lens):
        pcm[i, l:] = 0
    feats, frames = eng.fbank_batch(torch.from_numpy(pcm).cuda(), torch.from_numpy(lens).cuda())
    out = {}
    try:
        for v in (1, 0, 1):
            eng.lib.masr_debug_set(eng.h, 25, v)
            enc = eng.encode_full(feats, frames, -1).clone()
            idx, mp = eng.ctc_greedy_frames(enc)
            out[v] = (enc, idx.clone(), mp.clone())
    finally:
        eng.lib.masr_debug_set(eng.h, 25, 1)
    torch.cuda.synchronize()
    assert torch.isfinite(out[1][0]).all() and float(out[1][0].abs().max()) > 0
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    eng.close()


@needs_experiments()
@pytest.mark.parametrize('streaming,chunk', [(True, -1), (False, -1), (True, 16)])
def test_attention_chain_kernel_is_bit_identical_to_the_two_launches(streaming, chunk):
    """Round 4: attention + [out-projection + residual -> LN -> pointwise_conv1 -> GLU] as ONE launch (attn_chain_kernel: 32 queries
    x all four heads per workgroup, the context rows stay in LDS) performs the operations of attention_kernel<1> and of the
    EPI_CHAIN rowgemm in the same order: encoder output and probabilities BIT-identical with masr_debug_set key 34 on and off --
    ragged batch (pad masks, a partial last query block per sequence), causal and symmetric conv builds, chunk-masked attention."""
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    V = 512
    sd = synthetic.conformer_state_dict(0, V)
    eng = HipEngine(sd, vocab_size=V, streaming=streaming)
    rng = np.random.default_rng(7)
    lens = rng.integers(60000, 160001, 32).astype(np.int32)
    lens[0] = 160000
    pcm = synthetic.synthetic_pcm(32, 160000, seed=13)
    for i, l in enumerate(lens):
        pcm[i, l:] = 0
    feats, frames = eng.fbank_batch(torch.from_numpy(pcm).cuda(), torch.from_numpy(lens).cuda())
    out = {}
    try:
        for v in (1, 0, 1):
            eng.lib.masr_debug_set(eng.h, 34, v)
            out[v] = eng.encode_full(feats, frames, chunk).clone()
            out[('p', v)] = eng.ctc_probs(out[v]).clone()
    finally:
        eng.lib.masr_debug_set(eng.h, 34, 0)               # (the default: measured no faster than the two launches)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]) and torch.equal(out[('p', 0)], out[('p', 1)])
    assert float(out[1].abs().max()) > 0 and bool(torch.isfinite(out[1]).all())
    eng.close()
