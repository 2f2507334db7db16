"""GPU-side time line of the LAST predict_batch call in a rocprofv3 kernel trace of `bench.py --workload squeezeformer_b64_beam*`:
the prefix-search and pruning launches (start / end, per queue) against the envelope of every other kernel (the encoder passes).
usage: python tools/beam_gpu_timeline.py path/to/results.db [passes per call (2)] [which call from the end (1 = last; the bench
workload runs a short 4-utterance probe AFTER its timed calls, so the last timed call is 1)]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
qcol = 'queue_id' if 'queue_id' in cols else ('queue' if 'queue' in cols else None)
sel = "select name, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
rows = c.execute(sel).fetchall()
beams = [r for r in rows if 'beam_search_kernel' in r[0]]
if not beams:
    sys.exit('no beam_search_kernel in the trace')
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 2
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hi = len(beams) - (back - 1) * npass
last = beams[hi - npass:hi]
# the call starts with the first kernel after the previous call's last search ended and ends with its own last search
prev_end = max([b[2] for b in beams[:hi - npass]], default=rows[0][1])
call_end = max(b[2] for b in last)
call = [r for r in rows if r[1] >= prev_end - 1 and r[1] <= call_end]
t0 = call[0][1]
print(f'call of {(max(r[2] for r in call) - t0) / 1e6:.2f} ms on the GPU, {len(call)} launches')
for r in call:
    if 'beam_search_kernel' in r[0] or 'topk' in r[0]:
        print(f'  {r[0][:60]:60s} queue {r[3]}: {(r[1] - t0) / 1e6:7.2f} -> {(r[2] - t0) / 1e6:7.2f} ms  ({(r[2] - r[1]) / 1e3:9.1f} us)')
others = [r for r in call if 'beam_search_kernel' not in r[0]]
# envelope of the other kernels per queue, split where a gap of > 0.5 ms opens
byq = {}
for r in others:
    byq.setdefault(r[3], []).append(r)
for q, rs in byq.items():
    seg_s, seg_e, busy, nk = rs[0][1], rs[0][2], 0, 0
    for r in rs:
        if r[1] - seg_e > 2e5:
            print(f'  other kernels, queue {q}: {(seg_s - t0) / 1e6:7.2f} -> {(seg_e - t0) / 1e6:7.2f} ms  ({nk} launches, {busy / 1e6:.2f} ms of kernel time)')
            seg_s, busy, nk = r[1], 0, 0
        seg_e = max(seg_e, r[2])
        busy += r[2] - r[1]
        nk += 1
    print(f'  other kernels, queue {q}: {(seg_s - t0) / 1e6:7.2f} -> {(seg_e - t0) / 1e6:7.2f} ms  ({nk} launches, {busy / 1e6:.2f} ms of kernel time)')
