"""bench.py -- audio-seconds/sec of the offline hot path (BASELINE config 2) on N MI355X GPUs.

A "step" is one pass of PCM -> fbank -> Conformer encoder -> CTC greedy over one batch of
32 x 10 s synthetic 16 kHz utterances per GPU, inputs resident in HBM (weak scaling: every rank
owns its own batch; for N > 1 the hypotheses are all-gathered over RCCL inside the step).
Prints ONE JSON line on rank 0 (contract in the task statement; extra keys: roofline, cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 32
N_SAMPLES = 160000          # 10 s @ 16 kHz
VOCAB = 4233
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def committed_traffic():
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes of this build
    (profiles/r01_hbm_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); None if absent."""
    try:
        k = json.load(open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')))['kernels']
        return k['masr::ffn_pc_kernel<0, 0, 0, 0>']['hbm_bytes']
    except Exception:
        return None


def host_cores():
    """Threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return min(n, 64)


def cpu_baseline(sample_utts=4, reps=2):
    """The CPU oracle (restatement of the reference's PyTorch CPU path, pinned bit-identical to the
    reference modules) timed on this host: featurize + get_encoder_out + greedy on a bounded sample."""
    from masr_amd.utils import synthetic
    from oracle import conformer as oc, decoders as od, fbank as ofb
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f'cpu_baseline: {cores} threads')
    sd = synthetic.conformer_state_dict(0, VOCAB)
    vocab = synthetic.synthetic_vocab(VOCAB)
    pcm = synthetic.synthetic_pcm(sample_utts, N_SAMPLES, seed=1234)
    times = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        feats = np.stack([ofb.featurize_pcm16(pcm[i])[0] for i in range(sample_utts)])
        with torch.no_grad():
            probs = oc.get_encoder_out(sd, torch.from_numpy(feats), torch.full((sample_utts,), feats.shape[1])).numpy()
        od.greedy_decoder_batch(list(probs), vocab)
        dt = time.perf_counter() - t0
        log(f'cpu_baseline rep {r}: {dt:.2f} s for {sample_utts} x 10 s')
        if r > 0:
            times.append(dt)
        if dt > 60 and r >= 1:
            break
    med = float(np.median(times))
    return {'value': sample_utts * 10.0 / med, 'unit': 'audio-seconds/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{sample_utts} x 10 s utterances (batched get_encoder_out path, trainer.py:632), '
                      f'median of {reps}, torch threads={cores}'}


def other_workload(args, device):
    """Secondary workloads (parity-test configurations of BASELINE.json timed for DESIGN.md; NOT the contract line)."""
    import ctypes as C
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    torch.cuda.set_device(device)
    rng = np.random.default_rng(1234)
    if args.workload == 'efficient_b32':
        eng = HipEngine(synthetic.efficient_conformer_state_dict(0, VOCAB), vocab_size=VOCAB, streaming=True,
                        use_model='efficient_conformer', device=device)
        pcm = torch.from_numpy(synthetic.synthetic_pcm(BATCH, N_SAMPLES, seed=1234)).cuda()
        n = torch.full((BATCH,), N_SAMPLES, dtype=torch.int32, device='cuda')
        step = lambda: eng.transcribe_batch(pcm, n)
        audio = BATCH * 10.0
        desc = 'configs[3] shard: efficient_conformer.yml, 32 x 10 s per GPU, ctc_greedy'
    elif args.workload == 'squeezeformer_b64_beam':
        from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
        eng = HipEngine(synthetic.squeezeformer_state_dict(0, VOCAB), vocab_size=VOCAB, streaming=False,
                        use_model='squeezeformer', device=device)
        lens = rng.integers(32000, 320001, 64).astype(np.int32)
        order = np.argsort(-lens)                       # longest first (one padded batch; sort limits nothing here)
        lens = lens[order]
        pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
        for i, l in enumerate(lens):
            pcm_h[i, l:] = 0
        # length buckets (padded-to-max computes 2.55 TFLOP for 1.43 TFLOP of audio): the batch is sorted by length
        # and cut into sub-batches (2 of 32 by default), each padded to ITS longest utterance; all of them resident in HBM before the step
        nb = int(os.environ.get('MASR_BENCH_BUCKETS', '2'))
        parts = []
        for b0 in range(0, 64, 64 // nb):
            b1 = b0 + 64 // nb
            fr = 1 + (lens[b0:b1].astype(np.int64) - 400) // 160                    # feature frames, encoder frames (host side:
            parts.append((torch.from_numpy(np.ascontiguousarray(pcm_h[b0:b1, :int(lens[b0])])).cuda(),   # no sync in the step)
                          torch.from_numpy(lens[b0:b1]).cuda(), (((fr - 1) // 2 - 1) // 2).clip(min=0).tolist()))
        dec = BeamSearchDecoder(alpha=0, beta=0, beam_size=300, cutoff_prob=0.99, cutoff_top_n=40,
                                vocab_list=synthetic.synthetic_vocab(VOCAB), num_processes=min(host_cores(), 32))

        side = torch.cuda.Stream()

        def step():
            # longest bucket first; the prefix search of a bucket (one workgroup per utterance, ~21 us per frame with these flat
            # synthetic posteriors) runs on a side stream under the encoder of the next, shorter one
            main = torch.cuda.current_stream()
            pend, seqs = [], []
            for k, (pcm_b, n_b, nenc) in enumerate(parts):
                feats, frames = eng.fbank_batch(pcm_b, n_b)
                enc = eng.encode_full(feats, frames, -1)
                probs = eng.ctc_probs(enc)
                seqs += [probs[i, :nenc[i]] for i in range(probs.shape[0])]
                if nb <= 2 or k % 2 == 1:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        pend.append((dec._batch(seqs, defer=True), seqs))
                    seqs = []
            out = []
            for p_, _ in pend:
                out += dec._batch_collect(p_)
            main.wait_stream(side)
            return out
        audio = float(lens.sum()) / 16000.0
        desc = f'configs[2]: squeezeformer.yml non-streaming, 64 utterances 2-20 s ({audio:.1f} audio-s) in {nb} length buckets ' \
               f'of {64 // nb}, ctc_beam_search (LM-free, beam 300, cutoff_top_n 40; pruning and prefix search on the GPU, on a ' \
               f'side stream under the next buckets\' encoder)'
    elif args.workload == 'stream16':
        eng = HipEngine(synthetic.conformer_state_dict(0, VOCAB), vocab_size=VOCAB, device=device)
        ns = 16
        pcm = torch.from_numpy(synthetic.synthetic_pcm(ns, N_SAMPLES, seed=1234)).cuda()
        n = torch.full((ns,), N_SAMPLES, dtype=torch.int32, device='cuda')
        feats, _ = eng.fbank_batch(pcm, n)                       # [16, 998, 80]
        sids = [eng.stream_open(300) for _ in range(ns)]
        lat = []

        def step():
            for sid in sids:
                eng.stream_reset(sid)
            for cur in range(0, feats.shape[1] - 67 + 1, 64):     # 15 chunk steps of 0.64 s each, 16 streams in lock-step
                t0 = time.perf_counter()
                _, idx, mp = eng.encode_chunk(sids, feats[:, cur:cur + 67].contiguous(), want_probs=False, want_argmax=True)
                idx.cpu()
                lat.append(time.perf_counter() - t0)
        audio = ns * 15 * 0.64
        desc = 'configs[4] shard: conformer.yml streaming, 16 concurrent streams per GPU in lock-step, 67-frame windows / 64 stride'
    elif args.workload in ('deepspeech2_b1', 'deepspeech2_b32'):
        # configs[0] is CPU plumbing in BASELINE.json; timed here on the GPU for DESIGN.md (bi-directional, one 8.39 s
        # utterance the length of dataset/test.wav) and at batch 32 x 10 s
        B = 1 if args.workload == 'deepspeech2_b1' else BATCH
        ns = 134240 if B == 1 else N_SAMPLES
        eng = HipEngine(synthetic.deepspeech2_state_dict(0, VOCAB, bidirectional=True), vocab_size=VOCAB, streaming=False,
                        encoder_conf={'num_rnn_layers': 5, 'rnn_size': 1024}, use_model='deepspeech2', device=device)
        pcm = torch.from_numpy(synthetic.synthetic_pcm(B, ns, seed=1234)).cuda()
        n = torch.full((B,), ns, dtype=torch.int32, device='cuda')
        step = lambda: eng.transcribe_batch(pcm, n)
        audio = B * ns / 16000.0
        desc = f'configs[0] on the GPU: deepspeech2.yml non-streaming (5 x bi-LSTM-1024), batch {B} x {ns / 16000:.2f} s, ctc_greedy'
    else:
        raise SystemExit(f'unknown workload {args.workload}')
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.workload == 'stream16':
        lat.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {'workload': desc, 'value': round(audio * args.steps / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1,
           'steps': args.steps, 'ms_per_step': round(dt * 1e3 / args.steps, 3), 'dtype': 'f32', 'data': 'synthetic'}
    if args.workload == 'stream16':
        res['chunk_call_latency_ms'] = {'p50': round(float(np.percentile(lat, 50)) * 1e3, 3),
                                        'p95': round(float(np.percentile(lat, 95)) * 1e3, 3), 'calls': len(lat)}
    eng.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='conformer_b32',
                    help='conformer_b32 (BASELINE configs[1], the contract line) | squeezeformer_b64_beam (configs[2]) | '
                         'efficient_b32 (configs[3] per-GPU shard) | stream16 (configs[4] per-GPU shard) | deepspeech2_b1 | deepspeech2_b32')
    ap.add_argument('--profile-kind', type=int, default=2, help='kernel class timed with HIP events (2 = fused FFN)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    # MASR_BENCH_FORCE_DIST=1: take the RCCL path (init, all-gather, barriers, max over ranks) even with one rank --
    # lets a 1-GPU box exercise the code the multi-GPU launch runs
    force_dist = os.environ.get('MASR_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))   # RCCL
    n_gpus = world if world > 1 else 1
    use_dist = world > 1 or force_dist

    from masr_amd.engine import HipEngine, subsampled_len
    from masr_amd.utils import synthetic
    if args.workload != 'conformer_b32':
        if rank == 0:
            print(json.dumps(other_workload(args, local)), flush=True)
        return
    sd = synthetic.conformer_state_dict(0, VOCAB)
    eng = HipEngine(sd, vocab_size=VOCAB, device=local)
    pcm = torch.from_numpy(synthetic.synthetic_pcm(BATCH, N_SAMPLES, seed=1234 + rank)).cuda()
    n = torch.full((BATCH,), N_SAMPLES, dtype=torch.int32, device='cuda')
    Tp = subsampled_len(1 + (N_SAMPLES - 400) // 160)
    out = (torch.empty(BATCH, Tp, dtype=torch.int32, device='cuda'), torch.empty(BATCH, dtype=torch.int32, device='cuda'),
           torch.empty(BATCH, dtype=torch.float32, device='cuda'))
    gathered = torch.empty(world * BATCH, Tp, dtype=torch.int32, device='cuda') if use_dist else None

    def step():
        eng.transcribe_batch(pcm, n, out=out)
        if use_dist:   # the only exchange step: hypotheses (token ids, -1 padded), ~32 KB per rank
            dist.all_gather_into_tensor(gathered, out[0])

    log(f'rank {rank}: engine ready, warmup {args.warmup}')
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    eng.profile_select(args.profile_kind)
    eng.profile_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    log(f'rank {rank}: timed region done: {dt * 1e3 / args.steps:.2f} ms/step')
    prof_ms, prof_n, prof_flops = eng.profile_read(reset=True)
    eng.profile_select(0)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        audio_s = n_gpus * BATCH * (N_SAMPLES / 16000.0) * args.steps
        roofline = None
        if prof_n > 0 and prof_ms > 0:
            achieved = prof_flops / (prof_ms * 1e-3) / 1e12
            roofline = {'bound': 'mfma', 'kernel': 'ffn_pc_kernel (LN + [B*T\',256]x[256,2048] + SiLU + x[2048,256] + residual)',
                        'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': committed_traffic(),
                        'traffic_note': 'HBM bytes per launch from committed PMC passes (profiles/r01_hbm_traffic.json); '
                                        'algorithmic bytes per launch = 20.4 MB (x in/out + W1 + W2); measured = x in/out 16.3 MB + the 4.2 MB of weights '
                                        'fetched once by each of the 8 XCD L2s (Infinity Cache hits after the first)',
                        'launches': int(prof_n), 'avg_us': round(prof_ms * 1e3 / prof_n, 2),
                        'flops_per_launch': prof_flops / prof_n}
        res = {'metric': 'audio-seconds/sec (RTF^-1), conformer_streaming_fbank b32x10s, PCM->fbank->encoder->ctc_greedy',
               'value': round(audio_s / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': n_gpus, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(dt * 1e3 / args.steps, 3), 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'configs[1]: conformer.yml streaming fbank, batch=32 synthetic 16 kHz 10 s '
                                      'utterances per GPU, ctc_greedy, random-init weights V=4233',
                          'global_batch': n_gpus * BATCH, 'audio_seconds_per_step': n_gpus * BATCH * 10.0,
                          'parallelism': f'dp{n_gpus}', 'algorithmic_gflop_per_step_per_gpu': 742.0},
               'roofline': roofline}
        if n_gpus == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
