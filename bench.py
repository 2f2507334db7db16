"""bench.py -- audio-seconds/sec of the offline hot path (BASELINE configs[1]) on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of PCM -> fbank -> Conformer encoder -> CTC greedy -> hypotheses as TEXT ON THE HOST over one batch of
32 x 10 s synthetic 16 kHz utterances per GPU (weak scaling: every rank owns its own batch; for N > 1 the hypotheses are
all-gathered over RCCL inside the step -- the path's only exchange).  The int16 PCM is resident in HBM when the timed region
starts; the token ids come back through a pinned buffer and step k's text is built while step k+1 runs on the GPU.

``--gpus N`` with N > 1 launches the N ranks itself (``python -m torch.distributed.run``, one process per GPU) unless it is
already running under such a launcher (WORLD_SIZE set, as the driver does); a WORLD_SIZE that disagrees with ``--gpus`` is an
error.  Rank 0 prints ONE JSON line (contract in the task statement) with these additions:
  roofline      the fused FFN kernel: algorithmic FLOPs / HIP-event time of its launches inside the timed region
  value_host_to_host  SURVEY 8(d) / BASELINE.md's definition of the metric (pinned host PCM -> ... -> text on the host); ``value``
                follows the task contract (inputs resident in HBM when the timed region starts, the PCIe-inclusive rate never
                the headline) -- both are in the line, 1-3 % apart
  timing        the same step measured two more ways: ``device_only`` (no D2H, no text: round 1's number) and
                ``host_to_host`` (pinned host PCM -> H2D -> ... -> text, SURVEY 8(d)'s "PCM-on-host to hypotheses-on-host")
  cpu_baseline  the reference's CPU path on this node's host cores, bounded sample (N = 1 only)
  extra         BASELINE configs[2,3,4] timed on the same engine build (short runs; ``--no-extra`` skips them)
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 32
N_SAMPLES = 160000          # 10 s @ 16 kHz
VOCAB = 4233
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
GFLOP_PER_STEP = 742.0        # 23.18 GFLOP per 10 s utterance x 32
GFLOP_PER_UTT_EFFICIENT = 16.99   # Efficient-Conformer, 10 s utterance, V = 4233
GFLOP_SQUEEZEFORMER_B64 = 1430.0  # configs[2]'s 64 utterances (727.6 audio-s), useful (unpadded) work
GFLOP_PER_UTT_DS2_BI = 61.0       # DeepSpeech2 bi-directional (deepspeech2.yml, streaming: False), 10 s utterance, V = 4233
GFLOP_DS2_BI_TESTWAV = 51.1       # the same model on dataset/test.wav (837 frames -> T' = 208)
PEAK_HBM_TBS = 8.0                # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E


def conformer_gflop(T, t2=None, vocab=VOCAB, d=256, dff=2048, layers=12, k=15, pos_per_utt=False):
    """SURVEY 8(d)'s ALGORITHMIC FLOPs (2 * MAC, GEMM / conv terms only) of one Conformer pass over T feature frames; ``t2`` =
    attention key length (None: full context T'; streaming chunk step: cache + 16, with the positional projection counted per
    stream as the survey's 1.45 / 1.90 / 3.31 GFLOP figures do)"""
    t1 = (T - 1) // 2
    tp = (t1 - 1) // 2
    t2 = tp if t2 is None else t2
    front = 2 * 9 * d * t1 * 39 + 2 * 9 * d * d * tp * 19 + 2 * 19 * d * d * tp
    layer = (8 * d * dff * tp + 8 * d * d * tp + (2 * d * d * t2 if pos_per_utt else 0) + 4 * d * tp * t2 + 2 * d * tp * t2 +
             4 * d * d * tp + 2 * k * d * tp + 2 * d * d * tp)
    return (front + layers * layer + 2 * d * vocab * tp) / 1e9


def workload_roofline(gflop, ms, note):
    """whole-workload roofline block of a secondary config: algorithmic GFLOP of one step / its wall time, against the fp32 MFMA
    peak (the per-kernel evidence of these workloads is under profiles/r06_*_kernel_stats.txt)"""
    ach = gflop / ms                                               # GFLOP / ms = TFLOP/s
    return {'bound': 'mfma', 'scope': 'whole step (wall time, host framing included)', 'achieved': round(ach, 2),
            'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            'algorithmic_gflop_per_step': round(gflop, 1), 'note': note}


def flush_c_stdio():
    """RCCL prints a version banner through C stdio when its first communicator is created; with stdout redirected that buffer
    is written at process exit -- i.e. BEHIND the JSON line of rank 0.  Flushing the C streams once the collectives are warm
    (and again just before the line) keeps the contract line the last thing on stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                                   # noqa: BLE001  (cosmetic)
        pass


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


# ---- launch ------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(n, argv):
    """--gpus N outside a launcher: start the N ranks ourselves (one process per GPU) and relay rank 0's JSON line"""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    log('launching ' + ' '.join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def make_engine(kind, device, vocab=VOCAB):
    """the HIP engine of one rank; MASR_BENCH_ENGINE_FACTORY=module:callable swaps in a stand-in (CPU tests of the
    launch / shard / gather / timing code -- the JSON line then says so in ``data``)"""
    hook = os.environ.get('MASR_BENCH_ENGINE_FACTORY')
    if hook:
        mod, fn = hook.split(':')
        return getattr(importlib.import_module(mod), fn)(kind, device, vocab)
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    if kind == 'conformer':
        return HipEngine(synthetic.conformer_state_dict(0, vocab), vocab_size=vocab, device=device)
    if kind == 'efficient_conformer':
        return HipEngine(synthetic.efficient_conformer_state_dict(0, vocab), vocab_size=vocab, streaming=True,
                         use_model='efficient_conformer', device=device)
    if kind == 'squeezeformer':
        return HipEngine(synthetic.squeezeformer_state_dict(0, vocab), vocab_size=vocab, streaming=False,
                         use_model='squeezeformer', device=device)
    if kind in ('deepspeech2', 'deepspeech2_streaming'):
        bi = kind == 'deepspeech2'
        return HipEngine(synthetic.deepspeech2_state_dict(0, vocab, bidirectional=bi), vocab_size=vocab, streaming=not bi,
                         use_model='deepspeech2', device=device)
    raise SystemExit(f'unknown engine kind {kind}')


def data_tag():
    return 'synthetic' if not os.environ.get('MASR_BENCH_ENGINE_FACTORY') else 'synthetic, STAND-IN ENGINE (test hook, not a measurement)'


# ---- evidence helpers ------------------------------------------------------------------------------------------------------
# kernels timed with HIP events inside the engine (masr_profile_*): (profile kind, rocprof kernel-name fragment, description)
PROF_STRIDE = 5          # the timed region brackets every 5th launch of the roofline kernel with HIP events

ROOFLINE_KERNELS = [
    (6, 'ffn_pc_kernel<0, 0, 2, 1, 0>', "ffn_pc_kernel TAIL: LN + [B*T',256]x[256,2048] + SiLU + x[2048,256] + 1/2 residual, then LN + fused "
        "QKV projection [256,768] on the same rows (16.64 + 3.12 GFLOP per launch)"),
    (7, 'ffn_pc_kernel<0, 0, 2, 0, 15>', 'ffn_pc_kernel HEAD: depthwise conv + LN + SiLU + pointwise_conv2 [256,256] + residual, then '
        'LN + FFN + 1/2 residual (1.04 + 16.64 GFLOP per launch)'),
    (3, 'gemm_f32_kernel<128, 128, 2, 4, 1, 0>', 'Conv2d(256,256,3,stride 2) of the subsampling front-end as an implicit GEMM (177.9 GFLOP)'),
    (4, 'attention_kernel', 'rel-pos multi-head self-attention (algorithmic 6 * d * T\'^2 * B = 3.0 GFLOP per launch: the reference\'s '
        'two score terms + the value product; the kernel folds the positional keys into the keys and issues 2.0 GFLOP of MFMAs)'),
]


def committed_traffic(fragment):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this build (profiles/rNN_hbm_traffic.json:
    FETCH_SIZE with the gfx950 correction + WRITE_SIZE); (bytes, file) or (None, None)."""
    for name in ('r06_hbm_traffic.json', 'r05_hbm_traffic.json', 'r04_hbm_traffic.json', 'r03_hbm_traffic.json', 'r02_hbm_traffic.json', 'r01_hbm_traffic.json'):
        try:
            k = json.load(open(os.path.join(ROOT, 'profiles', name)))['kernels']
            for key, v in k.items():
                # (the third template argument is the weight path of the kernel: 0 LDS slabs until round 3, 2 packed copies since)
                alt = fragment.replace('<0, 0, 2, ', '<0, 0, 0, ')
                if fragment in key or alt in key or (name.startswith('r01') and 'ffn_pc_kernel<0, 0, 0, 1>' in key and alt == 'ffn_pc_kernel<0, 0, 0, 1, 0>'):
                    return v['hbm_bytes'], name
        except Exception:
            continue
    return None, None


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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(budget_s=30.0, force_port=False):
    """MASR's own CPU predict path on this node's host cores, on a bounded sample of the contract workload.  Two legs
    (SURVEY 8(d)): (i) the batched ``get_encoder_out`` path (trainer.py:632) on the config's own 32 x 10 s batch, (ii) the per-utterance
    ``MASRPredictor.predict`` loop (featurize + encoder + greedy, B = 1 calls, predict.py:167-192) -- ``value`` is leg (ii),
    the way MASR users run it.  kind = "reference" when /root/reference is importable (the unmodified reference modules via
    oracle/shims), else "port" (oracle/, pinned bit-identical to those modules).  >= 5 timed repetitions per leg, median."""
    from masr_amd.utils import synthetic
    from oracle import conformer as oc, decoders as od, fbank as ofb, shims
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synthetic.conformer_state_dict(0, VOCAB)
    vocab = synthetic.synthetic_vocab(VOCAB)
    pcm = synthetic.synthetic_pcm(BATCH, N_SAMPLES, seed=1234)       # configs[1]'s batch; leg (ii) takes its first 4 utterances
    kind = 'port'
    encode = lambda f, l: oc.get_encoder_out(sd, f, l)
    if shims.reference_available() and not force_port:
        try:
            from oracle import make_golden
            shims.install()
            model = make_golden.build_reference_conformer(sd, VOCAB, tempfile.mkdtemp(prefix='masr_ref_'))[0]
            encode = lambda f, l: model.get_encoder_out(f, l)      # the reference's ConformerModel with these weights
            kind = 'reference'
        except Exception as exc:                                        # noqa: BLE001
            log(f'cpu_baseline: reference modules not usable here ({exc}); timing the port')
    log(f'cpu_baseline: {kind}, {cores} threads, {cpu_model()}')

    def leg_batched():
        feats = np.stack([ofb.featurize_pcm16(pcm[i])[0] for i in range(BATCH)])
        with torch.no_grad():
            probs = encode(torch.from_numpy(feats), torch.full((BATCH,), feats.shape[1])).numpy()
        od.greedy_decoder_batch(list(probs), vocab)
        return BATCH * 10.0

    def leg_predict_loop():
        for i in range(4):
            feat = ofb.featurize_pcm16(pcm[i])[0][None]
            with torch.no_grad():
                probs = encode(torch.from_numpy(feat), torch.tensor([feat.shape[1]])).numpy()[0]
            od.greedy_decoder(probs, vocab)
        return 40.0

    out = {}
    for name, leg in (('batched', leg_batched), ('predict_loop', leg_predict_loop)):
        leg()                                                           # warm-up
        times, t_leg = [], time.perf_counter()
        # (the batched leg is configs[1]'s whole padded batch, ~2.5 s per repetition: 5 repetitions; the B = 1 loop up to 9)
        while len(times) < 5 or (name != 'batched' and time.perf_counter() - t_leg < budget_s / 2 and len(times) < 9):
            t0 = time.perf_counter()
            audio = leg()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_leg > budget_s and len(times) >= 5:
                break
        out[name] = {'value': round(audio / float(np.median(times)), 2), 'reps': len(times),
                     'median_s': round(float(np.median(times)), 3)}
        log(f'cpu_baseline {name}: {out[name]}')
    return {'value': out['predict_loop']['value'], 'unit': 'audio-seconds/sec', 'cores': cores, 'kind': kind,
            'cpu_model': cpu_model(),
            'sample': f'leg (ii) per-utterance predict loop: 4 x 10 s utterances, B = 1 calls (featurize + get_encoder_out + '
                      f'greedy), median of {out["predict_loop"]["reps"]} reps, torch threads = {cores}',
            'batched': {'value': out['batched']['value'], 'reps': out['batched']['reps'],
                        'sample': f'leg (i) batched get_encoder_out path (trainer.py:632) on configs[1]\'s padded batch: {BATCH} x 10 s '
                                  '(featurize each utterance, one encoder call, batch greedy decode)'}}


# ---- the contract workload ----------------------------------------------------------------------------------------------
class ContractStep:
    """one rank's step of configs[1] on the BIT-EXACT normalisation route (``masr_transcribe_rows``, use_db_normalization = 2):
    the mean squares of a batch come from the device in numpy's summation order, the gain is the reference's own scalar numpy
    expression evaluated on this host (``engine.reference_gains``; audio.py:287-304,256-264,519-529), so the int16 samples the
    fbank kernel sees are the reference's, bit for bit.  The round trip (mean squares -> host -> gains) of step k + 1 is issued
    under step k's kernels on a side stream and is recomputed for every step.

    ``mode``: 'full' (HBM-resident PCM -> text on host, pipelined), 'device' (-> packed hypothesis rows on the device, no
    per-step wait on the results), 'host' (pinned host PCM -> H2D -> ... -> text on host: SURVEY 8(d)'s metric).

    The exchange of a step -- RCCL all-gather of the packed rows (N > 1), copy to the host -- runs on a side stream behind
    an event of the compute stream, so the next step's kernels start while the collective of this one is in flight; the
    device outputs and the pinned host buffers are double-buffered (slot = step parity)."""

    TARGET_DB = -20.0

    def __init__(self, eng, rank, world, vocab):
        from masr_amd import parallel
        from masr_amd.engine import reference_gains
        from masr_amd.utils import synthetic
        self.parallel, self.eng, self.rank, self.world = parallel, eng, rank, world
        self.reference_gains = reference_gains
        self.dev = eng.device
        self.vocab = np.array(vocab, dtype=object)
        pcm = torch.from_numpy(synthetic.synthetic_pcm(BATCH, N_SAMPLES, seed=1234 + rank))
        gpu = self.dev.type == 'cuda'
        self.pcm_host = pcm.pin_memory() if gpu else pcm
        self.pcm = pcm.to(self.dev)
        self.n = torch.full((BATCH,), N_SAMPLES, dtype=torch.int32, device=self.dev)
        self.Tp = eng.out_frames(1 + (N_SAMPLES - 400) // 160)
        # hypotheses are ONE int32 payload [B, T' + 2] (tokens | count | score bits), written by the collapse kernel itself
        self.out = [torch.empty(BATCH, self.Tp + 2, dtype=torch.int32, device=self.dev) for _ in range(2)]
        rows = world * BATCH
        self.host = [torch.empty(rows, self.Tp + 2, dtype=torch.int32, pin_memory=gpu) for _ in range(2)]
        # the step's helper streams are the library's (masr_side_stream: chosen by probing so that none shares the compute stream's
        # hardware queue, whatever the process created before -- on a rank of a multi-GPU job RCCL comes first and shifts the
        # runtime's round-robin deal of queues); the stand-in engine of the CPU tests has none
        lib_streams = gpu and hasattr(eng, 'side_stream')
        self.side = (eng.side_stream(1) if lib_streams else torch.cuda.Stream(device=self.dev)) if gpu else None
        self.computed = [torch.cuda.Event() if gpu else None for _ in range(2)]      # compute stream: outputs of the slot written
        self.events = [torch.cuda.Event() if gpu else None for _ in range(2)]        # side stream: slot gathered (and on the host)
        self.used = [False, False]
        # the gain round trip: mean squares (device) -> pinned host -> reference_gains -> pinned host -> device, two slots
        self.prep_stream = (eng.side_stream(2) if lib_streams else torch.cuda.Stream(device=self.dev)) if gpu else None
        self.ms_dev = [torch.empty(BATCH, dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.ms_host = [torch.empty(BATCH, dtype=torch.float32, pin_memory=gpu) for _ in range(2)]
        self.ms_ready = [torch.cuda.Event() if gpu else None for _ in range(2)]
        self.gain_host = [torch.empty(BATCH, dtype=torch.float32, pin_memory=gpu) for _ in range(2)]
        self.gain_dev = [torch.empty(BATCH, dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.gain_up = [torch.cuda.Event() if gpu else None for _ in range(2)]       # compute stream: the slot's gains were copied up
        self.gain_used = [False, False]
        self.prepared = [None, None]                                                  # data_ptr of the PCM the slot's mean squares belong to
        self.turn = 0
        self.h2d = None
        self.prefetch_pcm_for = None
        self.pending = None
        self.texts = None
        self.n_texts = 0
        self.last_gains = None

    def _finish(self, slot):
        """hypotheses of a finished step as text on the host (rank 0: of all ranks' utterances; others: their own shard)"""
        if self.events[slot] is not None:
            self.events[slot].synchronize()
        tok, nt, _ = self.parallel.unpack_hypothesis_rows(self.host[slot].numpy())
        lo, hi = (0, self.world * BATCH) if self.rank == 0 else (self.rank * BATCH, (self.rank + 1) * BATCH)
        self.texts = self.parallel.tokens_to_text(tok[lo:hi], nt[lo:hi], self.vocab)
        self.n_texts += len(self.texts)

    def flush(self):
        if self.pending is not None:
            self._finish(self.pending)
            self.pending = None

    # ---- gains -------------------------------------------------------------------------------------------------------
    def prepare(self, pcm, stream=None, after_compute=False):
        """enqueue the mean squares of ``pcm`` (the NEXT step's batch) and their copy to the host on ``stream`` (default: the
        preparation stream, which does NOT wait for the compute stream: the batch is resident) -- runs under the current
        step's kernels; ``_gains`` picks the result up"""
        k = self.turn
        self.turn ^= 1
        if self.prep_stream is None:
            self.eng.mean_square(pcm, self.n, out=self.ms_dev[k])
            self.ms_host[k].copy_(self.ms_dev[k])
        else:
            st = stream if stream is not None else self.prep_stream
            if after_compute:
                st.wait_stream(torch.cuda.current_stream())               # a batch nobody announced: it may still be on its way
            with torch.cuda.stream(st):
                self.eng.mean_square(pcm, self.n, out=self.ms_dev[k])
                self.ms_host[k].copy_(self.ms_dev[k], non_blocking=True)
                self.ms_ready[k].record()
        self.prepared[k] = pcm.data_ptr()

    def _gains(self, pcm):
        """linear gains [B] (device) of ``pcm`` for the step about to be enqueued: the prepared mean squares (or, for a batch
        nobody announced, mean squares computed now) -> this host's numpy -> device"""
        ptr = pcm.data_ptr()
        k = next((j for j in (0, 1) if self.prepared[j] == ptr), None)
        if k is None:
            self.prepare(pcm, after_compute=True)
            k = self.turn ^ 1
        if self.ms_ready[k] is not None:
            self.ms_ready[k].synchronize()
        self.prepared[k] = None
        if self.gain_up[k] is not None and self.gain_used[k]:
            self.gain_up[k].synchronize()                                   # the slot's previous upload has left the pinned buffer
        g = self.reference_gains(self.ms_host[k].numpy(), self.TARGET_DB)
        self.last_gains = g
        self.gain_host[k].numpy()[:] = g
        self.gain_dev[k].copy_(self.gain_host[k], non_blocking=True)
        if self.gain_up[k] is not None:
            self.gain_up[k].record()
            self.gain_used[k] = True
        return self.gain_dev[k]

    # ---- PCM over PCIe ('host' mode) ------------------------------------------------------------------------------------
    def _copy_pcm(self, k):
        """enqueue the host -> device copy of step k's PCM on the copy stream (buffer k & 1), and behind it, on the same
        stream, the mean squares of that batch"""
        h = self.h2d
        with torch.cuda.stream(h['stream']):
            if self.used[k & 1]:
                h['stream'].wait_event(self.computed[k & 1])           # the kernels that read this buffer two steps ago are done
            h['buf'][k & 1].copy_(self.pcm_host, non_blocking=True)
            h['ready'][k & 1].record()
        self.prepare(h['buf'][k & 1], stream=h['stream'])
        h['issued'] = k

    def _pcm_from_host(self, i):
        """'host' mode: this step's PCM from pinned host memory.  On the GPU the copy of step i + 1 is issued on a copy stream
        as soon as step i is enqueued (two device buffers), so PCIe time hides under the previous step's kernels; the very
        first call copies its own batch."""
        if self.side is None:
            return self.pcm_host.to(self.dev)
        if self.h2d is None:
            self.h2d = {'stream': self.eng.side_stream(0) if hasattr(self.eng, 'side_stream') else torch.cuda.Stream(device=self.dev),
                        'buf': [torch.empty_like(self.pcm) for _ in range(2)],
                        'ready': [torch.cuda.Event() for _ in range(2)], 'issued': None}
        if self.h2d['issued'] != i:
            self._copy_pcm(i)
        torch.cuda.current_stream().wait_event(self.h2d['ready'][i & 1])
        self.prefetch_pcm_for = i + 1                                    # issued by step() right after this step's kernels
        return self.h2d['buf'][i & 1]

    def _exchange(self, slot, mode):
        rows = self.parallel.gather_rows(self.out[slot])               # RCCL all-gather of [32, T'+2] int32 per rank (N > 1)
        if mode != 'device':
            self.host[slot].copy_(rows, non_blocking=True)
            self.last_host_slot = slot

    def ranks_seen(self):
        """ranks whose hypothesis rows arrived through the LAST step's all-gather: every rank's batch is seeded 1234 + rank, so
        rank r's block of the gathered payload must equal what rank r alone produces -- here checked the cheap way: each rank
        all-gathers the checksum of its own rows and compares it with the checksum of the block it received"""
        if not self.parallel.collectives_on():
            return [self.rank]
        import torch.distributed as dist
        slot = getattr(self, 'last_host_slot', None)
        if slot is None:
            return []
        if self.events[slot] is not None:
            self.events[slot].synchronize()
        got = self.host[slot].numpy().astype(np.int64)
        mine = torch.tensor([int(got[self.rank * BATCH:(self.rank + 1) * BATCH].sum())], dtype=torch.int64,
                            device=self.parallel.comm_device())
        sums = torch.empty(dist.get_world_size(), dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(sums, mine)
        sums = sums.cpu().numpy()
        return [r for r in range(len(sums)) if int(got[r * BATCH:(r + 1) * BATCH].sum()) == int(sums[r])]

    def step(self, i, mode='full', next_pcm=None):
        """``next_pcm``: the HBM-resident batch of the following step when it is not ``self.pcm`` again (tests).
        ``mode='lanes'``: the 'full' step with consecutive steps on the engine's two lanes (masr_select_lane: two workspace sets,
        two streams), so the kernels of step k + 1 fill what step k's leave idle -- reported beside the contract line, whose
        timed region stays on one lane (a kernel bracketed by HIP events must not share the CUs with another step's kernels)."""
        if mode == 'lanes':
            lane = i & 1
            if getattr(self, 'lane_streams', None) is None:
                self.lane_streams = [torch.cuda.current_stream(self.dev), self.eng.side_stream(4)]
                self.lane_streams[1].wait_stream(self.lane_streams[0])
            self.eng.select_lane(lane)
            try:
                with torch.cuda.stream(self.lane_streams[lane]):
                    return self.step(i, 'full', next_pcm)
            finally:
                self.eng.select_lane(0)
        slot = i & 1
        pcm = self.pcm
        if mode == 'host':
            pcm = self._pcm_from_host(i)                                # 10.2 MB over PCIe per step
        gains = self._gains(pcm)
        if self.side is not None and self.used[slot]:
            torch.cuda.current_stream().wait_event(self.events[slot])   # the slot's previous exchange has read its outputs
        self.eng.transcribe_rows(pcm, self.n, True, self.TARGET_DB, gain_in=gains, out=self.out[slot])
        if self.side is None:
            if mode != 'host':
                self.prepare(next_pcm if next_pcm is not None else self.pcm)
            self._exchange(slot, mode)
        else:
            self.computed[slot].record()
            if self.prefetch_pcm_for is not None:                       # 'host' mode: the next step's PCM starts its way over PCIe
                self._copy_pcm(self.prefetch_pcm_for)                   # (+ its mean squares behind the copy)
                self.prefetch_pcm_for = None
            elif mode != 'host':
                self.prepare(next_pcm if next_pcm is not None else self.pcm)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.computed[slot])
                self._exchange(slot, mode)
                self.events[slot].record()
            self.used[slot] = True
        if mode == 'device':
            return
        self.flush()                                                    # text of the previous step, under this step's kernels
        self.pending = slot


def run_contract(args, rank, world, local):
    from masr_amd import parallel
    from masr_amd.utils import synthetic
    eng = make_engine('conformer', local)
    cs = ContractStep(eng, rank, world, synthetic.synthetic_vocab(VOCAB))
    log(f'rank {rank}/{world}: engine ready on device {local}, warmup {args.warmup}')

    def prof_on():
        # the roofline kernel is timed live over the timed region, on a sample of its launches: every PROF_STRIDE-th one (the
        # stride is coprime to the 12 launches of a step, so every layer is sampled) -- two HIP event records per timed launch
        # cost ~6 us of stream time, 0.07 ms per step when every launch is bracketed
        if hasattr(eng, 'lib'):
            eng.lib.masr_debug_set(eng.h, 16, PROF_STRIDE)
        eng.profile_select(args.profile_kind)
        eng.profile_read(reset=True)

    dt, dt_local = parallel.timed_region(lambda i: cs.step(i, 'full'), args.steps, args.warmup, after_warmup=prof_on,
                                         flush=cs.flush, return_local=True)
    # evidence that the step's exchange really spanned the job: the ranks whose hypothesis rows arrived through the all-gather of
    # the timed region's last step (ContractStep.ranks_seen), and every rank's own time for the region
    per_rank_ms = parallel.gather_floats([dt_local * 1e3 / args.steps])
    ranks_seen = cs.ranks_seen()
    prof_ms, prof_n, prof_flops = eng.profile_read(reset=True)
    eng.profile_select(0)
    if hasattr(eng, 'lib'):
        eng.lib.masr_debug_set(eng.h, 16, 1)
    n_texts, sample_text = cs.n_texts, (cs.texts[0] if cs.texts else '')
    log(f'rank {rank}: timed region done: {dt * 1e3 / args.steps:.3f} ms/step')
    flush_c_stdio()
    # (three untimed steps each: the host mode creates its copy stream, device buffers and events in its first steps -- with one
    #  warm-up step ~4 ms of one-time work sat in its 20 timed steps: 6.70 against 6.45-6.47 ms per step, tools/h2h_ab.py)
    dt_dev = parallel.timed_region(lambda i: cs.step(i, 'device'), args.steps, 3)
    dt_host = parallel.timed_region(lambda i: cs.step(i, 'host'), args.steps, 3, flush=cs.flush)
    others = []
    # the other heavy kernels, each timed the same way over a few more steps (every rank runs the steps -- they contain the
    # all-gather -- rank 0 keeps the numbers)
    for kind, frag, desc in ROOFLINE_KERNELS:
        if kind == args.profile_kind:
            continue
        eng.profile_select(kind)
        eng.profile_read(reset=True)
        for i in range(3):
            cs.step(i, 'device')
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        ms, n, fl = eng.profile_read(reset=True)
        if rank == 0 and n > 0 and ms > 0:
            ach = fl / (ms * 1e-3) / 1e12
            others.append({'kernel': desc, 'rocprof_name': frag, 'achieved': round(ach, 2), 'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                           'launches': int(n), 'avg_us': round(ms * 1e3 / n, 2), 'flops_per_launch': fl / n})
    eng.profile_select(0)
    res = None
    if rank == 0:
        audio_step = world * BATCH * (N_SAMPLES / 16000.0)
        roofline = None
        if prof_n > 0 and prof_ms > 0:
            achieved = prof_flops / (prof_ms * 1e-3) / 1e12
            kind, frag, desc = next((k for k in ROOFLINE_KERNELS if k[0] == args.profile_kind), (args.profile_kind, '', f'profile kind {args.profile_kind}'))
            traffic, tfile = committed_traffic(frag)
            roofline = {'bound': 'mfma', 'kernel': desc, 'rocprof_name': frag,
                        'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic,
                        'traffic_note': f'HBM bytes per launch from the committed PMC passes (profiles/{tfile}); algorithmic bytes '
                                        'per launch = x in/out 16.3 MB + qkv out 24.4 MB + weights 5.0 MB; the weights are fetched '
                                        'once by each of the 8 XCD L2s (Infinity Cache hits after the first)',
                        'launches': int(prof_n), 'sampling': f'every {PROF_STRIDE}th launch of the timed region bracketed by HIP events',
                        'avg_us': round(prof_ms * 1e3 / prof_n, 2),
                        'flops_per_launch': prof_flops / prof_n, 'other_kernels': others}
        per = lambda t: {'value': round(audio_step * args.steps / t, 1), 'ms_per_step': round(t * 1e3 / args.steps, 3)}
        res = {'metric': 'audio-seconds/sec (RTF^-1), conformer_streaming_fbank b32x10s, PCM->fbank->encoder->ctc_greedy->text',
               'value': per(dt)['value'], 'unit': 'audio-seconds/sec', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': per(dt)['ms_per_step'], 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': data_tag(),
               'config': {'workload': 'configs[1]: conformer.yml streaming fbank, batch=32 synthetic 16 kHz 10 s '
                                      'utterances per GPU, ctc_greedy, random-init weights V=4233',
                          'global_batch': world * BATCH, 'audio_seconds_per_step': audio_step,
                          'parallelism': f'dp{world}', 'world_size_observed': parallel.world_info()[1],
                          'backend': torch.distributed.get_backend() if torch.distributed.is_initialized() else 'none',
                          'rccl_ranks_seen': ranks_seen,
                          'ms_per_step_per_rank': {'min': round(min(per_rank_ms), 3), 'max': round(max(per_rank_ms), 3)},
                          'algorithmic_gflop_per_step_per_gpu': GFLOP_PER_STEP,
                          'normalisation': 'use_db_normalization = 2 (masr_transcribe_rows): mean squares on the device in numpy\'s '
                                           'summation order, gain = the reference\'s scalar numpy expressions on this host -- the '
                                           'int16 samples are the reference\'s bit for bit; the round trip of step k+1 runs under '
                                           'step k and is recomputed every step',
                          'timed_region': 'int16 PCM resident in HBM -> mean squares -> host gains -> masr_transcribe_rows (mode 2) '
                                          '-> packed hypothesis rows -> (all-gather) -> D2H -> text on host; '
                                          f'{n_texts} transcripts built inside it, e.g. {sample_text[:12]!r}'},
               'value_definition': 'task contract: inputs resident in HBM when the timed region starts; SURVEY 8(d)\'s host-to-host '
                                   'rate of the same route is value_host_to_host',
               'value_host_to_host': per(dt_host)['value'],
               'roofline': roofline,
               'timing': {'device_only': dict(per(dt_dev), note='PCM in HBM -> packed hypothesis rows in HBM; the host waits only for the (earlier) mean squares'),
                          'host_to_host': dict(per(dt_host), note='pinned host int16 PCM -> H2D (copy stream, under the previous step) -> ... -> D2H -> text on host')}}
    return eng, res


# ---- secondary workloads (BASELINE configs[2,3,4]) ----------------------------------------------------------------------
SHARP_HEAD_GAIN = 4.0


def extra_contract_two_lanes(args, rank, world, local):
    """configs[1], the contract step ('full': HBM-resident PCM -> text on host) with CONSECUTIVE STEPS ON THE ENGINE'S TWO LANES
    (masr_select_lane: two workspace sets over one set of weights, two streams): the kernels of step k + 1 fill the CUs that
    step k's leave idle.  A line of its own, not the contract line: a kernel bracketed by HIP events shares its CUs with the
    other step's kernels here (reported: what the roofline kernel then reads), so ``value`` / ``roofline`` of the contract line
    and the committed rocprofv3 trace of its command stay on one lane."""
    from masr_amd import parallel
    from masr_amd.utils import synthetic
    eng = make_engine('conformer', local)
    if not (hasattr(eng, 'side_stream') and torch.cuda.is_available()):
        eng.close()
        return {'workload': 'configs[1] on two lanes: needs the GPU engine', 'value': None}
    cs = ContractStep(eng, rank, world, synthetic.synthetic_vocab(VOCAB))

    def prof_on():
        eng.lib.masr_debug_set(eng.h, 16, PROF_STRIDE)
        eng.profile_select(args.profile_kind)
        eng.profile_read(reset=True)
    steps = max(args.steps, 10)
    dt = parallel.timed_region(lambda i: cs.step(i, 'lanes'), steps, 4, flush=cs.flush, after_warmup=prof_on)
    ms, n, _ = eng.profile_read(reset=True)
    eng.profile_select(0)
    eng.lib.masr_debug_set(eng.h, 16, 1)
    dt1 = parallel.timed_region(lambda i: cs.step(i, 'full'), steps, 3, flush=cs.flush)
    eng.close()
    audio_step = world * BATCH * (N_SAMPLES / 16000.0)
    return {'workload': 'configs[1]: conformer.yml streaming fbank, batch=32 x 10 s per GPU per step, HBM-resident PCM -> text on host, '
                        'consecutive steps on the two lanes of the engine (steps k and k + 1 side by side)',
            'value': round(audio_step * steps / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': world, 'steps': steps,
            'ms_per_step': round(dt * 1e3 / steps, 3), 'one_lane_ms_per_step_same_process': round(dt1 * 1e3 / steps, 3),
            'roofline_kernel_avg_us_beside_the_other_lane': round(ms * 1e3 / n, 2) if n else None,
            'roofline': workload_roofline(GFLOP_PER_STEP, dt * 1e3 / steps, 'SURVEY 8(d): 742 GFLOP per batch-32 step')}


def facade(use_model, decoder, device, streaming=True, vocab=VOCAB, beam_conf=None, head_gain=None):
    """a MASRPredictor on synthetic weights (the drop-in surface; StreamPool / predict_batch hang off it)"""
    from masr_amd.predict import MASRPredictor
    from masr_amd.utils import synthetic
    d = tempfile.mkdtemp(prefix='masr_bench_')
    vpath = os.path.join(d, 'vocabulary.txt')
    with open(vpath, 'w', encoding='utf-8') as f:
        for t in synthetic.synthetic_vocab(vocab):
            f.write(f'{t}\t1\n')
    sd = {'conformer': synthetic.conformer_state_dict, 'efficient_conformer': synthetic.efficient_conformer_state_dict,
          'squeezeformer': synthetic.squeezeformer_state_dict}[use_model](0, vocab)
    if head_gain:                # a sharper CTC head: the logits of the random-init projection scaled up (weights AND bias)
        sd = dict(sd)
        for k in ('ctc.ctc_lo.weight', 'ctc.ctc_lo.bias'):
            sd[k] = sd[k] * head_gain
    cfg = {'encoder_conf': {}, 'preprocess_conf': {'feature_method': 'fbank', 'n_mels': 80, 'n_mfcc': 40, 'sample_rate': 16000,
                                                   'use_dB_normalization': True, 'target_dB': -20},
           'dataset_conf': {'dataset_vocab': vpath}, 'use_model': use_model, 'streaming': streaming, 'decoder': decoder,
           'metrics_type': 'cer'}
    if beam_conf:
        cfg['ctc_beam_search_decoder_conf'] = beam_conf
    torch.cuda.set_device(device)
    return MASRPredictor(configs=cfg, use_gpu=True, state_dict=sd)


def extra_efficient_b256(args, rank, world, local):
    """configs[3]: efficient_conformer.yml streaming fbank, 256 x 10 s utterances sharded DP over the ranks (256 / N per GPU in
    device passes of 32), RCCL all-gather of the hypotheses; strong scaling by construction of the config"""
    from masr_amd import parallel
    eng = make_engine('efficient_conformer', local)
    from masr_amd.utils import synthetic
    total = 256
    lo, hi = parallel.shard_range(total, rank, world)
    # device passes of 64 utterances where the rank's share allows it: behind the stride layer this encoder runs at HALF the frame
    # rate, and 32 x 10 s are then only 124 row blocks of 32 for 256 CUs; 64 x 10 s fill the chip there (248 row blocks) and are
    # two full rounds in the full-rate layers (MASR_BENCH_EFFICIENT_PASS overrides; at 8 GPUs the share itself is 32)
    per_pass = int(os.environ.get('MASR_BENCH_EFFICIENT_PASS', '64'))
    passes = [(torch.from_numpy(synthetic.synthetic_pcm(min(per_pass, hi - b0), N_SAMPLES, seed=77 + b0)).to(eng.device),
               torch.full((min(per_pass, hi - b0),), N_SAMPLES, dtype=torch.int32, device=eng.device))
              for b0 in range(lo, hi, per_pass)]
    per = -(-total // world)
    Tp = eng.out_frames(1 + (N_SAMPLES - 400) // 160)
    tok = torch.full((per, Tp), -1, dtype=torch.int32, device=eng.device)
    nt = torch.zeros(per, dtype=torch.int32, device=eng.device)
    sc = torch.zeros(per, dtype=torch.float32, device=eng.device)
    vocab = np.array(synthetic.synthetic_vocab(VOCAB), dtype=object)
    texts = []

    # consecutive passes on the engine's two lanes (masr_select_lane: two workspace sets, two streams): the kernels of one pass
    # fill the CUs the other's leave idle (MASR_BENCH_EFFICIENT_LANES=1: one after the other, rounds 1-5)
    lanes = max(1, min(2, int(os.environ.get('MASR_BENCH_EFFICIENT_LANES', '2')))) if len(passes) > 1 and hasattr(eng, 'side_stream') and torch.cuda.is_available() else 1
    main = torch.cuda.current_stream(eng.device) if torch.cuda.is_available() else None
    lane_streams = [main] + ([eng.side_stream(4)] if lanes > 1 else [])

    def step(i):
        k = 0
        for st in lane_streams[1:]:
            st.wait_stream(main)                      # (the previous step's gather has read the outputs)
        for j, (pcm, n) in enumerate(passes):
            b = pcm.shape[0]
            if lanes > 1:
                eng.select_lane(j % lanes)
                with torch.cuda.stream(lane_streams[j % lanes]):
                    eng.transcribe_batch(pcm, n, out=(tok[k:k + b], nt[k:k + b], sc[k:k + b]))
            else:
                eng.transcribe_batch(pcm, n, out=(tok[k:k + b], nt[k:k + b], sc[k:k + b]))
            k += b
        if lanes > 1:
            eng.select_lane(0)
            for st in lane_streams[1:]:
                main.wait_stream(st)
        t, c, _ = parallel.gather_hypotheses(tok, nt, sc)
        if rank == 0:
            texts[:] = parallel.tokens_to_text(t.cpu().numpy(), c.cpu().numpy(), vocab)
    steps = max(10, args.steps // 2)
    dt = parallel.timed_region(step, steps, 1)
    eng.close()
    return {'workload': f'configs[3]: efficient_conformer.yml streaming fbank, 256 x 10 s utterances sharded over {world} GPU(s) '
                        f'({hi - lo} on rank 0, device passes of {min(per_pass, hi - lo)}' + (', consecutive passes on the two lanes of the engine' if lanes > 1 else '')
                        + '), ctc_greedy, all-gather of hypotheses, text on host',
            'value': round(total * 10.0 * steps / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': world, 'steps': steps,
            'ms_per_step': round(dt * 1e3 / steps, 3), 'scaling': 'strong', 'transcripts': len(texts),
            'roofline': workload_roofline(GFLOP_PER_UTT_EFFICIENT * (hi - lo), dt * 1e3 / steps,
                                          'SURVEY 8(d): 16.99 GFLOP per 10 s utterance x the utterances of rank 0')}


def extra_stream128(args, rank, world, local, n_streams=None, reps=None):
    """configs[4]: conformer.yml streaming, 128 concurrent synthetic streams (128 / N sticky per GPU) fed 0.5 s (8000-sample)
    int16 PCM chunks in lock-step through the REAL stream framing (StreamPool.feed / step: ragged fbank of the pending
    samples, 67-frame windows, greedy history collapse); per-call latency p50 / p95 and aggregate audio-s/s"""
    from masr_amd import parallel
    from masr_amd.serving import StreamPool
    from masr_amd.utils import synthetic
    pred = facade('conformer', 'ctc_greedy', local)
    pool = parallel.ShardedStreamPool(StreamPool(pred, max_frames_out=320))
    n_streams = n_streams or int(os.environ.get('MASR_BENCH_STREAMS', '128'))
    chunk, n_chunks = 8000, 20
    reps = reps or (10 if n_streams <= 32 else 4)              # 20 calls per utterance: >= 80 timed calls (stream128), 200 (stream16)
    gids = [pool.open() for _ in range(n_streams)]
    mine = pool.local_ids()
    pcm = synthetic.synthetic_pcm(len(mine), chunk * n_chunks, seed=4321 + rank)
    # the wire format: every 0.5 s chunk of every stream as the bytes object a websocket hands over (sliced before the clock starts)
    wire = [[pcm[j, c * chunk:(c + 1) * chunk].tobytes() for j in range(len(mine))] for c in range(n_chunks)]
    lat = []

    def utterance(record):
        for g in mine:
            pool.reset(g)
        for c in range(n_chunks):
            t0 = time.perf_counter()
            for j, g in enumerate(mine):
                pool.feed(g, wire[c][j], is_end=(c == n_chunks - 1))
            pool.step()
            if record:
                lat.append(time.perf_counter() - t0)

    utterance(False)
    dt = parallel.timed_region(lambda i: utterance(True), reps, 0)
    lat_all = parallel.gather_floats(lat)
    pred.predictor.engine.close()
    # algorithmic work of one 10 s stream: its 67-frame windows (stride 64) against a cache that grows by 16 keys per window
    n_win = ((1 + (chunk * n_chunks - 400) // 160) - 67) // 64 + 1
    gflop_stream = sum(conformer_gflop(67, t2=16 * (w + 1), pos_per_utt=True) for w in range(n_win))
    return {'workload': f'configs[4]: conformer.yml streaming chunk = 0.5 s online, {n_streams} concurrent synthetic streams over {world} '
                        f'GPU(s) ({len(mine)} per GPU, sticky), real predict_stream framing, ctc_greedy partials every call',
            'value': round(n_streams * n_chunks * 0.5 * reps / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': world,
            'steps': reps * n_chunks,
            'call_latency_ms': {'p50': round(float(np.percentile(lat_all, 50)) * 1e3, 3),
                                'p95': round(float(np.percentile(lat_all, 95)) * 1e3, 3), 'calls': len(lat_all),
                                'note': 'one call = feed + step of all streams of a GPU for one 0.5 s chunk (python framing included)'},
            'roofline': workload_roofline(gflop_stream * len(mine), dt * 1e3 / reps,
                                          f'one step = one 10 s utterance of every stream of rank 0 ({n_win} chunk steps of 16 '
                                          f'encoder frames, {gflop_stream:.1f} GFLOP per stream by SURVEY 8(d)); the chunk step is '
                                          'latency-, not MFMA-bound (DESIGN 9)')}


def extra_squeezeformer_beam(args, rank, world, local, lm=True, sharp=False, word_lm=False):
    """configs[2]: squeezeformer.yml non-streaming fbank, 64 utterances of 2-20 s (seed 1234) padded per length bucket,
    ctc_beam_search (beam 300, cutoff_top_n 40, alpha 2.2 / beta 4.3 with a synthetic character n-gram LM when ``lm``)"""
    from masr_amd.utils import synthetic
    rng = np.random.default_rng(1234)
    lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
    pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
    audio = [pcm_h[i, :lens[i]] for i in range(64)]
    conf = {'alpha': 2.2 if lm else 0, 'beta': 4.3 if lm else 0, 'beam_size': 300, 'cutoff_prob': 0.99, 'cutoff_top_n': 40,
            'num_processes': 10}
    if lm and word_lm:
        # a WORD-based scorer (an LM word longer than one character: the reference's English configurations,
        # beam_search_decoder.py:29-30 is_character_based()): spelling dictionary + scoring at <space>; that search runs on host
        # threads (num_processes), the vocabulary pruning stays on the GPU
        from masr_amd.decoders.lm_scorer import write_synthetic_word_arpa
        rng_w = np.random.default_rng(9)
        chars = [t for t in synthetic.synthetic_vocab(VOCAB) if len(t) == 1][:400]
        words = sorted({''.join(chars[int(i)] for i in rng_w.integers(0, len(chars), int(rng_w.integers(2, 5)))) for _ in range(3000)})
        d = tempfile.mkdtemp(prefix='masr_lm_')
        conf['language_model_path'] = write_synthetic_word_arpa(os.path.join(d, 'wlm.arpa'), words, seed=5)
    elif lm:
        from masr_amd.decoders.lm_scorer import write_synthetic_arpa
        d = tempfile.mkdtemp(prefix='masr_lm_')
        conf['language_model_path'] = write_synthetic_arpa(os.path.join(d, 'lm.arpa'), synthetic.synthetic_vocab(VOCAB), seed=5)
    else:
        conf['language_model_path'] = None           # explicit scorer-free search (not a reference configuration)
    pred = facade('squeezeformer', 'ctc_beam_search', local, streaming=False, beam_conf=conf,
                  head_gain=SHARP_HEAD_GAIN if sharp else None)
    steps = 6 if word_lm else 10
    # device passes of 32 utterances, longest first: the prefix search of a pass (one workgroup per utterance, frames in sequence --
    # the longest utterance is the call's critical path) runs on a side stream under the encoder of the next pass.  Measured with
    # MASR_BENCH_BEAM_PASS: passes of 16 are faster with the sharpened head (27.2 vs 30.7 ms per call: the longest utterance's search
    # starts earlier) and slower with flat posteriors (56.6 vs 46.6 ms: four long searches share two side streams); 32 for both lines
    # Passes: with the prefix search on the GPU, fixed passes of 32 (the longest utterance's search -- ~30 us per frame x 494 frames,
    # the call's critical path -- starts behind ONE encoder pass and the second pass runs underneath it; round 5 timelines,
    # tools/beam_batch_profile.py: 45.7 ms flat / 29.1 ms sharp against 55 / 39 ms with three passes of equal padded size).  With
    # the search on host threads (word LM) the encoder is what counts and passes of EQUAL PADDED SIZE win
    # (predict_batch(batch_size='balanced'): count x longest utterance <= 32 x 10 s, one full round of 248 row-block workgroups
    # per pass: 16 / 19 / 29 utterances in three rounds where 2 x 32 take four, the second half empty).  MASR_BENCH_BEAM_PASS
    # overrides (a number, or 'balanced').
    per_pass = os.environ.get('MASR_BENCH_BEAM_PASS', 'balanced' if (lm and word_lm) else '32')
    per_pass = per_pass if per_pass == 'balanced' else ([int(v) for v in per_pass.split(',')] if ',' in per_pass else int(per_pass))
    pass_padded = float(os.environ['MASR_BENCH_BEAM_PADDED']) * 160000 if os.environ.get('MASR_BENCH_BEAM_PADDED') else None      # (x 10 s of audio)
    # three untimed calls: the caching allocator's per-stream pools (probabilities of a pass: 134 MB, allocated on the main
    # stream, restacked on a side stream) reach their steady state only with the third call -- with one warm-up call the first
    # timed calls still paid device allocations (round 5: 62.7 ms mean against 45.8 ms per call in the steady state, same box)
    for _ in range(3):
        pred.predict_batch(audio, batch_size=per_pass, pass_padded=pass_padded)
    torch.cuda.synchronize()
    calls = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        res = pred.predict_batch(audio, batch_size=per_pass, pass_padded=pass_padded)        # (synchronous: returns with the transcripts on the host)
        calls.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = float(lens.sum()) / 16000.0
    # how many candidates per frame the search really saw (vocabulary pruning at cutoff_prob / cutoff_top_n), on 4 utterances
    eng = pred.predictor.engine
    k = [0, 21, 42, 63]
    nmax = int(max(lens[i] for i in k))
    x = np.zeros((4, nmax), np.int16)
    for j, i in enumerate(k):
        x[j, :lens[i]] = audio[i]
    feats, frames = eng.fbank_batch(torch.from_numpy(x).to(eng.device), torch.tensor([int(lens[i]) for i in k], dtype=torch.int32,
                                                                                    device=eng.device))
    probs = eng.ctc_probs(eng.encode_full(feats, frames, -1))
    nenc = eng.enc_frames(frames).tolist()
    cnt = torch.cat([pred.beam_search_decoder._candidates(probs[j, :nenc[j]], to_host=False)[2] for j in range(4)])
    cand_mean = float(cnt.float().mean())
    pred.predictor.engine.close()
    return {'workload': f'configs[2]: squeezeformer.yml non-streaming fbank, 64 utterances 2-20 s ({total:.1f} audio-s), '
                        + ('length-sorted passes of equal padded size (16 / 19 / 29 utterances)' if per_pass == 'balanced' else
                           f'length-sorted passes of {per_pass} utterances (longest first)' if isinstance(per_pass, list) else
                           f'{-(-64 // per_pass)} length buckets of {per_pass}') + ', ctc_beam_search beam 300 / top-n 40, '
                        + ('alpha 2.2 beta 4.3 with a synthetic 3-gram WORD LM: prefix search on %d host threads, each pass searched when it is '
                           'collected, under the encoders (two lanes) of the passes launched behind it' % conf['num_processes']
                           if lm and word_lm else
                           'alpha 2.2 beta 4.3 with a synthetic 3-gram character LM scored on the GPU' if lm else 'LM-free'),
            'posteriors': ('SHARPENED CTC head (random-init logits x %g: 1-3 candidates survive cutoff_prob 0.99 per frame, what a '
                           'trained model gives the search)' % SHARP_HEAD_GAIN) if sharp else
                          'FLAT random-init posteriors: cutoff_top_n = 40 candidates survive in EVERY frame -- the search\'s worst case, '
                          'which no trained model produces',
            'candidates_per_frame_mean': round(cand_mean, 2),
            'value': round(total * steps / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1, 'steps': steps,
            'ms_per_step': round(dt * 1e3 / steps, 3), 'transcripts': len(res),
            'call_ms': {'p50': round(float(np.percentile(calls, 50)) * 1e3, 3), 'min': round(min(calls) * 1e3, 3),
                        'max': round(max(calls) * 1e3, 3)},
            'roofline': workload_roofline(GFLOP_SQUEEZEFORMER_B64, dt * 1e3 / steps,
                                          'SURVEY 8(d): 1.43 TFLOP of useful encoder work in the 64 utterances; with a sharpened head the call is its two '
                                          'encoder passes + the second search (3.5 ms); with flat posteriors the first search (DESIGN 5, 9)')}


def extra_squeezeformer_greedy(args, rank, world, local):
    """configs[2]'s 64 utterances through the ENCODER path only (ctc_greedy instead of the prefix search): what the Squeezeformer
    layers cost, with the two fused stage kernels of a layer (csrc/sqz_layer.hip) timed by HIP events in the same run"""
    from masr_amd.utils import synthetic
    rng = np.random.default_rng(1234)
    lens = np.sort(rng.integers(32000, 320001, 64).astype(np.int32))[::-1].copy()
    pcm_h = synthetic.synthetic_pcm(64, int(lens.max()), seed=1234)
    audio = [pcm_h[i, :lens[i]] for i in range(64)]
    pred = facade('squeezeformer', 'ctc_greedy', local, streaming=False)
    eng = pred.predictor.engine
    steps = 10
    out = {}
    for mode in ('balanced', 32):
        for _ in range(2):
            pred.predict_batch(audio, batch_size=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = pred.predict_batch(audio, batch_size=mode)
        torch.cuda.synchronize()
        out[mode] = (time.perf_counter() - t0) / steps
    # A/B of the fused-layer threshold (masr_debug_set key 36: row blocks from which a layer takes the fused stage kernels;
    # 0 = the twelve separate launches of round 4) on the same call
    ab = {}
    if os.environ.get('MASR_BENCH_SQZ_AB', '1') == '1' and hasattr(eng, 'lib'):
        for blocks in (0, 96, 128, 192):
            eng.lib.masr_debug_set(eng.h, 36, blocks)
            pred.predict_batch(audio, batch_size='balanced')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                pred.predict_batch(audio, batch_size='balanced')
            torch.cuda.synchronize()
            ab[f'fused_from_{blocks}_row_blocks' if blocks else 'separate_launches'] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        eng.lib.masr_debug_set(eng.h, 36, 128)
    kernels = []
    lanes_env = os.environ.get('MASR_LANES')
    os.environ['MASR_LANES'] = '1'          # a kernel bracketed by HIP events must not share the CUs with the other lane's pass
    for kind, name in ((6, 'sqz_stage_kernel<0, 1>: [out-proj + LN1] + FFN1 + LN2 + [pw1 + GLU]'),
                       (7, 'sqz_stage_kernel<1, 31>: [dwconv + BN + SiLU + pw2 + LN3] + FFN2 + LN4 + [next QKV]')):
        eng.profile_select(kind)
        eng.profile_read(reset=True)
        pred.predict_batch(audio, batch_size='balanced')
        torch.cuda.synchronize()
        ms, n, fl = eng.profile_read(reset=True)
        if n > 0 and ms > 0:
            ach = fl / (ms * 1e-3) / 1e12
            kernels.append({'kernel': name, 'launches': int(n), 'avg_us': round(ms * 1e3 / n, 2), 'achieved': round(ach, 2),
                            'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4), 'unit': 'TFLOP/s',
                            'note': 'HIP events around every launch of one predict_batch call on ONE lane (launches of full-rate and half-rate '
                                    'layers, all passes); half-rate layers below 128 row blocks run the separate d_ff-split launches instead'})
    eng.profile_select(0)
    if lanes_env is None:
        os.environ.pop('MASR_LANES', None)
    else:
        os.environ['MASR_LANES'] = lanes_env
    total = float(lens.sum()) / 16000.0
    best = min(out, key=out.get)          # both policies are measured in this run; the line is the faster one and says which
    dt = out[best]
    eng.close()
    roof = workload_roofline(GFLOP_SQUEEZEFORMER_B64, dt * 1e3, 'SURVEY 8(d): 1.43 TFLOP of useful (unpadded) encoder work in the 64 utterances')
    roof['kernels'] = kernels
    return {'workload': f'configs[2] with ctc_greedy: squeezeformer.yml non-streaming fbank, 64 utterances 2-20 s ({total:.1f} audio-s) -> text, '
                        + ('length-sorted passes of equal padded size (16 / 19 / 29 utterances)' if best == 'balanced' else
                           'two length-sorted passes of 32 side by side on the two lanes of the engine (masr_select_lane; the row blocks of padded '
                           'frames are not computed: masr_debug_set key 38)')
                        + '; encoder-bound line of the Squeezeformer',
            'value': round(total / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1, 'steps': steps, 'ms_per_step': round(dt * 1e3, 3),
            'fixed_passes_of_32_ms_per_step': round(out[32] * 1e3, 3), 'balanced_passes_ms_per_step': round(out['balanced'] * 1e3, 3),
            'fused_layer_ab_ms_per_step': ab, 'transcripts': len(res),
            'roofline': roof}


def extra_deepspeech2(args, rank, world, local):
    """DeepSpeech2 (deepspeech2.yml, streaming: False = bi-directional LSTMs), BASELINE configs[0]'s model on the GPU:
    ``ds2_testwav_b1``  one test.wav-sized utterance (8.39 s, T' = 208) PCM -> text, p50 of 50 calls;
    ``ds2_b32x10s``     32 x 10 s per pass.
    The recurrence is one launch per timestep and layer; at B = 1 its bound is the recurrent weight stream (W_hh: 16.8 MB per
    direction re-read every timestep from L2 / Infinity Cache), reported against the HBM peak; at B = 32 the matrix pipe."""
    from masr_amd.utils import synthetic
    eng = make_engine('deepspeech2', local)
    golden = os.path.join(ROOT, 'tests', 'golden', 'testwav.npz')
    wav = np.load(golden)['pcm'] if os.path.exists(golden) else synthetic.synthetic_pcm(1, 134240, seed=7)[0]
    out = {}
    xs = torch.from_numpy(np.ascontiguousarray(wav[None])).to(eng.device)
    ns = torch.tensor([len(wav)], dtype=torch.int32, device=eng.device)

    def one():
        rows = eng.transcribe_rows(xs, ns, True, -20.0, gain_in=eng.host_gains(xs, ns, -20.0))
        return eng.to_host(rows)
    for _ in range(3):
        one()
    lat = []
    for _ in range(50):
        t0 = time.perf_counter()
        one()
        lat.append(time.perf_counter() - t0)
    p50 = float(np.percentile(lat, 50)) * 1e3
    Tq = eng.out_frames(1 + (len(wav) - 400) // 160)
    whh_bytes = 2 * 4 * 1024 * 1024 * 4.0 * Tq * 5                 # directions x [4 x 1024, 1024] f32 x timesteps x layers
    out['ds2_testwav_b1'] = {
        'workload': f'deepspeech2.yml streaming: False (bi-LSTM x 5), one {len(wav) / 16000.0:.2f} s utterance (dataset/test.wav, T\' = {Tq}): '
                    'int16 PCM in HBM -> mean squares -> host gains -> masr_transcribe_rows -> packed row on the host, 50 calls',
        'latency_ms': {'p50': round(p50, 3), 'p95': round(float(np.percentile(lat, 95)) * 1e3, 3), 'calls': len(lat)},
        'value': round(len(wav) / 16000.0 / (p50 * 1e-3), 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1,
        'roofline': {'bound': 'hbm', 'scope': 'whole call (wall time); the dominant stream is W_hh re-read per timestep (served by L2 / '
                                              'Infinity Cache after the first touch: the HBM peak is the stated roof, not a bound it can exceed)',
                     'achieved': round(whh_bytes / (p50 * 1e-3) / 1e12, 3), 'peak': PEAK_HBM_TBS, 'unit': 'TB/s',
                     'frac': round(whh_bytes / (p50 * 1e-3) / 1e12 / PEAK_HBM_TBS, 4),
                     'algorithmic_bytes_per_call': whh_bytes,
                     'mfma_view': workload_roofline(GFLOP_DS2_BI_TESTWAV, p50, 'SURVEY 8(d): 51.1 GFLOP for test.wav (bi)')}}
    B = 32
    pcm = torch.from_numpy(synthetic.synthetic_pcm(B, N_SAMPLES, seed=1234 + rank)).to(eng.device)
    n = torch.full((B,), N_SAMPLES, dtype=torch.int32, device=eng.device)

    def batch():
        return eng.transcribe_rows(pcm, n, True, -20.0, gain_in=eng.host_gains(pcm, n, -20.0))
    for _ in range(2):
        batch()
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        rows = batch()
    eng.to_host(rows)
    dt = (time.perf_counter() - t0) / steps
    out['ds2_b32x10s'] = {
        'workload': 'deepspeech2.yml streaming: False (bi-LSTM x 5), 32 x 10 s per pass: int16 PCM in HBM -> packed hypothesis rows',
        'value': round(B * 10.0 / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1, 'steps': steps, 'ms_per_step': round(dt * 1e3, 3),
        'roofline': workload_roofline(GFLOP_PER_UTT_DS2_BI * B, dt * 1e3,
                                      'SURVEY 8(d): 61.0 GFLOP per 10 s utterance (bi); the recurrence is 248 dependent launches per '
                                      'layer and direction pair')}
    eng.close()
    return out


def extra_facade(args, rank, world, local):
    """The drop-in surface itself (the headline is measured one layer below it, on ``masr_transcribe_batch``):
    ``facade_b32``  MASRPredictor.predict_batch on configs[1]'s batch handed over as 32 host int16 ndarrays -> 32 transcripts
                    (staging, upload, the bit-exact normalisation route with its mean-square read-back, one device pass, text);
    ``predict_b1``  MASRPredictor.predict on one test.wav-sized utterance (134 240 samples; masr/predict.py:167-192, the call
                    docs/infer.md times at 101 ms): p50 / p95 of the call and the GPU time of its device pass."""
    from masr_amd.utils import synthetic
    pred = facade('conformer', 'ctc_greedy', local)
    eng = pred.predictor.engine
    audio = list(synthetic.synthetic_pcm(BATCH, N_SAMPLES, seed=1234 + rank))
    steps = max(10, args.steps)
    for _ in range(3):
        res = pred.predict_batch(audio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = pred.predict_batch(audio)
    dt = (time.perf_counter() - t0) / steps
    out = {'facade_b32': {
        'workload': 'configs[1] through the facade: MASRPredictor.predict_batch(32 host int16 ndarrays of 10 s) -> 32 transcripts, '
                    'one call after the other (nothing overlaps between calls); use_dB_normalization on, gains by the bit-exact '
                    'route (device mean square -> host numpy -> device)',
        'value': round(BATCH * 10.0 / dt, 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1, 'steps': steps,
        'ms_per_step': round(dt * 1e3, 3), 'transcripts': len(res),
        'roofline': workload_roofline(GFLOP_PER_STEP, dt * 1e3, 'SURVEY 8(d): 742 GFLOP per 32 x 10 s')}}
    # the same batches through predict_batch_deferred, as the server worker issues them: batch k + 1 staged, uploaded and launched (on
    # the engine's other lane) before batch k is collected -- host numpy arrays in, text out, everything inside the timed region
    if hasattr(pred, 'predict_batch_deferred'):
        def pipelined(n):
            prev, last = None, None
            for _ in range(n):
                cur = pred.predict_batch_deferred(audio)
                if prev is not None:
                    last = prev()
                prev = cur
            return prev()
        pipelined(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res2 = pipelined(steps)
        dt2 = (time.perf_counter() - t0) / steps
        out['facade_b32_pipelined'] = {
            'workload': 'configs[1] through the facade, pipelined: MASRPredictor.predict_batch_deferred(32 host int16 ndarrays of 10 s) for '
                        'batch k + 1 before the results of batch k are fetched (two in flight, consecutive batches on the two lanes of the '
                        'engine) -- what server.EngineWorker does with the batches it forms; host arrays in, text out, bit-exact gain route',
            'value': round(BATCH * 10.0 / dt2, 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1, 'steps': steps,
            'ms_per_step': round(dt2 * 1e3, 3), 'transcripts': len(res2), 'same_transcripts_as_facade_b32': res2 == res,
            'roofline': workload_roofline(GFLOP_PER_STEP, dt2 * 1e3, 'SURVEY 8(d): 742 GFLOP per 32 x 10 s')}
    golden = os.path.join(ROOT, 'tests', 'golden', 'testwav.npz')
    wav = np.load(golden)['pcm'] if os.path.exists(golden) else synthetic.synthetic_pcm(1, 134240, seed=7)[0]
    for _ in range(5):
        one = pred.predict(wav)
    lat = []
    for _ in range(200):
        t0 = time.perf_counter()
        one = pred.predict(wav)
        lat.append(time.perf_counter() - t0)
    # GPU time of the same device pass: the call's kernels enqueued back to back, nothing synchronised in between
    xs = torch.from_numpy(np.ascontiguousarray(wav[None])).to(eng.device)
    ns = torch.tensor([len(wav)], dtype=torch.int32, device=eng.device)
    gain = eng.host_gains(xs, ns, -20.0)
    for _ in range(3):
        eng.transcribe_rows(xs, ns, True, -20.0, gain_in=gain)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.transcribe_rows(xs, ns, True, -20.0, gain_in=gain)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / 50 * 1e3
    p50 = float(np.percentile(lat, 50)) * 1e3
    secs = len(wav) / 16000.0
    out['predict_b1'] = {
        'workload': f'MASRPredictor.predict(one {secs:.2f} s utterance as a host int16 ndarray: dataset/test.wav) -> text, 200 calls',
        'latency_ms': {'p50': round(p50, 3), 'p95': round(float(np.percentile(lat, 95)) * 1e3, 3), 'calls': len(lat)},
        'gpu_ms_of_the_pass': round(gpu_ms, 3), 'gpu_share_of_p50': round(gpu_ms / p50, 3),
        'value': round(secs / (p50 * 1e-3), 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1, 'text_chars': len(one['text']),
        'reference_doc_figure': 'docs/infer.md:93: 101 ms for this file (the reference on its own GPU, not comparable hardware)',
        'roofline': workload_roofline(conformer_gflop(1 + (len(wav) - 400) // 160), p50,
                                      'B = 1: 7 row blocks of 32 on 256 CUs -- latency-bound, listed for completeness')}
    eng.close()
    return out


def extra_bf16x3(args, rank, world, local):
    """EXPLORATORY, never the contract line: configs[1]'s step with conv2, the embed projection and the fused FFN blocks as
    split-bf16 products on the bf16 matrix pipe (masr_debug_set key 20 = 3; csrc/gemm_bf16x3.hip, csrc/ffn_x3.hip:
    a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulation).  Reported with the distance to the exact-fp32 kernels on the same batch; dtype "bf16x3"."""
    from masr_amd.utils import synthetic
    eng = make_engine('conformer', local)
    pcm = torch.from_numpy(synthetic.synthetic_pcm(BATCH, N_SAMPLES, seed=1234 + rank)).to(eng.device)
    n = torch.full((BATCH,), N_SAMPLES, dtype=torch.int32, device=eng.device)
    feats, frames = eng.fbank_batch(pcm, n)
    enc32 = eng.encode_full(feats, frames, -1).clone()
    tok32, nt32, _ = [t.clone() for t in eng.transcribe_batch(pcm, n)]
    steps = max(10, args.steps)
    out = {}
    for mode in (0, 1):
        eng.lib.masr_debug_set(eng.h, 20, 3 if mode else 0)
        for _ in range(3):
            eng.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.transcribe_batch(pcm, n)
        torch.cuda.synchronize()
        out[mode] = (time.perf_counter() - t0) / steps
    eng.lib.masr_debug_set(eng.h, 20, 3)
    enc3 = eng.encode_full(feats, frames, -1).clone()
    tok3, nt3, _ = [t.clone() for t in eng.transcribe_batch(pcm, n)]
    eng.lib.masr_debug_set(eng.h, 20, 0)
    torch.cuda.synchronize()
    same = int(sum(int(nt32[i]) == int(nt3[i]) and torch.equal(tok32[i, :int(nt32[i])], tok3[i, :int(nt3[i])]) for i in range(BATCH)))
    eng.close()
    audio = BATCH * N_SAMPLES / 16000.0
    return {'workload': 'configs[1] in the EXPLORATORY split-bf16 mode (device-only step: PCM in HBM -> token ids in HBM): conv2, embed '
                        'projection and the fused FFN blocks on the bf16 matrix pipe, everything else as in the contract line',
            'value': round(audio / out[1], 1), 'unit': 'audio-seconds/sec', 'n_gpus': 1, 'steps': steps, 'dtype': 'bf16x3',
            'ms_per_step': round(out[1] * 1e3, 3), 'fp32_ms_per_step_same_loop': round(out[0] * 1e3, 3),
            'max_abs_diff_encoder_output_vs_fp32_kernels': float((enc32 - enc3).abs().max()),
            'utterances_with_identical_greedy_tokens': f'{same}/{BATCH}',
            'note': 'not the reference arithmetic (fp32): never the headline; tests/test_gpu_bf16x3.py holds it to the 1e-3 bars'}


class ExtrasWatchdog:
    """The secondary workloads run AFTER the contract measurement but before its line is printed.  Under several ranks they use
    collectives (hypothesis all-gathers); should one of them ever stall (a rank lost, an interconnect fault), this thread prints
    the already measured contract line -- with whatever extras finished and a note -- and ends the process, on every rank, so
    that a stalled extra can cost the extras but never the measurement.  (Blocking device synchronisation releases the GIL, so
    the thread runs while the main thread waits.)  MASR_BENCH_EXTRA_TIMEOUT seconds, default 600; 0 disables."""

    def __init__(self, rank, res, out):
        import threading
        self.limit = float(os.environ.get('MASR_BENCH_EXTRA_TIMEOUT', '600'))
        self.done = threading.Event()
        self.rank, self.res, self.out = rank, res, out
        if self.limit > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def _watch(self):
        if self.done.wait(self.limit):
            return
        log(f'rank {self.rank}: the secondary workloads exceeded {self.limit:.0f} s -- printing the contract line without them')
        if self.rank == 0:
            line = dict(self.res)
            line['extra'] = dict(self.out, _note=f'extras aborted by the watchdog after {self.limit:.0f} s')
            print(json.dumps(line, ensure_ascii=False), flush=True)
        os._exit(0)


def run_extras(args, rank, world, local, out=None):
    out = {} if out is None else out
    jobs = [('conformer_b32_two_lanes', extra_contract_two_lanes), ('efficient_b256', extra_efficient_b256), ('stream128', extra_stream128)]
    if world == 1:
        # the per-GPU share of configs[4] on an 8-GPU node (128 streams / 8), measured on this one GPU
        jobs.append(('stream16', lambda a, r, w, l: extra_stream128(a, r, w, l, n_streams=16)))
        # (rounds 3-5 measured these two lines in a process of their own: the prefix search ran on torch side streams whose hardware
        #  queue depended on what the process had created before.  The library owns its side streams now -- masr_side_stream, a
        #  queue pool of their own -- and the lines are measured here, in the long bench process, like everything else.)
        jobs.append(('squeezeformer_b64_beam', extra_squeezeformer_beam))
        jobs.append(('squeezeformer_b64_beam_sharp', lambda a, r, w, l: extra_squeezeformer_beam(a, r, w, l, sharp=True)))
        jobs.append(('squeezeformer_b64_beam_wordlm_host', lambda a, r, w, l: extra_squeezeformer_beam(a, r, w, l, sharp=True, word_lm=True)))
        jobs.append(('squeezeformer_b64_greedy', extra_squeezeformer_greedy))
        jobs.append(('facade', extra_facade))
        jobs.append(('deepspeech2', extra_deepspeech2))
        from masr_amd import build as _build
        if _build.has_experiments():             # (the exploratory split-bf16 mode exists only in MASR_BUILD_EXPERIMENTS=1 libraries)
            jobs.append(('conformer_b32_bf16x3_exploratory', extra_bf16x3))
    for name, fn in jobs:
        try:
            t0 = time.perf_counter()
            res = fn(args, rank, world, local)
            if name in ('facade', 'deepspeech2'):  # several lines from one predictor / engine
                out.update(res)
            else:
                out[name] = res
            log(f'rank {rank}: extra {name} done in {time.perf_counter() - t0:.1f} s')
        except Exception as exc:                                        # noqa: BLE001  (the contract line must still print)
            # the ranks run the same code on the same shapes, so a failure here is a failure on every rank at the same point
            # (no rank is left waiting in a collective); it is recorded and the already measured contract line still prints
            out[name] = {'error': f'{type(exc).__name__}: {exc}'}
            log(f'rank {rank}: extra {name} FAILED: {exc}')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true')
    ap.add_argument('--workload', default='conformer_b32',
                    help='conformer_b32 (BASELINE configs[1], the contract line, default) | efficient_b256 | stream128 | '
                         'squeezeformer_b64_beam | squeezeformer_b64_beam_nolm  (one secondary workload only, own JSON line)')
    ap.add_argument('--profile-kind', type=int, default=6,
                    help='kernel timed with HIP events for the roofline block (6 = fused FFN + QKV tail, the heaviest kernel by total '
                         'time; 7 = conv-module head + FFN, 3 = conv2 implicit GEMM, 4 = attention)')
    args = ap.parse_args()

    env_world = int(os.environ.get('WORLD_SIZE', '0'))
    if env_world == 0 and args.gpus > 1:
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))
    if env_world and env_world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks')
    from masr_amd import parallel
    rank, world, local = parallel.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)

    line = None
    if args.workload != 'conformer_b32':
        fn = {'efficient_b256': extra_efficient_b256, 'stream128': extra_stream128, 'bf16x3': extra_bf16x3,
              'conformer_b32_two_lanes': extra_contract_two_lanes,
              'squeezeformer_b64_beam': extra_squeezeformer_beam,
              'squeezeformer_b64_beam_nolm': lambda a, r, w, l: extra_squeezeformer_beam(a, r, w, l, lm=False),
              'squeezeformer_b64_beam_sharp': lambda a, r, w, l: extra_squeezeformer_beam(a, r, w, l, sharp=True),
              'squeezeformer_b64_beam_wordlm_host': lambda a, r, w, l: extra_squeezeformer_beam(a, r, w, l, sharp=True, word_lm=True),
              'squeezeformer_b64_greedy': extra_squeezeformer_greedy, 'deepspeech2': extra_deepspeech2,
              'facade': extra_facade}[args.workload]
        res = fn(args, rank, world, local)
        if rank == 0:
            line = json.dumps(dict({'dtype': 'f32'}, **res, data=data_tag()), ensure_ascii=False)
    else:
        eng, res = run_contract(args, rank, world, local)
        eng.close()
        extra = None
        if not args.no_extra:
            partial = {}
            dog = ExtrasWatchdog(rank, res, partial)
            extra = run_extras(args, rank, world, local, partial)
            dog.done.set()
        if rank == 0:
            if extra is not None:
                res['extra'] = extra
                two = extra.get('conformer_b32_two_lanes') or {}
                if two.get('value'):
                    # the same step with consecutive steps side by side on the engine's two lanes: beside `value`, never instead of it
                    # (`value` / `roofline` are measured on ONE lane: a kernel bracketed by HIP events must have the CUs to itself)
                    res['value_two_lanes'] = two['value']
            if world == 1 and not args.no_cpu_baseline:
                try:
                    res['cpu_baseline'] = cpu_baseline()
                except Exception as exc:                                # noqa: BLE001  (the measured line must still print)
                    res['cpu_baseline'] = {'error': f'{type(exc).__name__}: {exc}'}
            line = json.dumps(res, ensure_ascii=False)
    # the JSON line is the LAST thing on stdout: every rank empties its C buffers (RCCL's banner), all ranks meet, rank 0 prints
    flush_c_stdio()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    if line is not None:
        print(line, flush=True)
    try:                                     # whatever a library still writes to stdout on its way out goes to stderr
        sys.stdout.flush()
        os.dup2(2, 1)
    except OSError:
        pass
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
