"""Study (round 6): memory copies > 1 MB of the LAST predict_batch call in a rocprofv3 --kernel-trace --memory-copy-trace of
bench.py --workload squeezeformer_b64_beam_sharp, against the call's first kernel.  usage: python tools/studies/copy_trace_dump.py results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
MIN_BYTES = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 20)
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
cand = [t for t in tables if 'memory_cop' in t.lower() or 'memcpy' in t.lower()]
print('copy tables / views:', cand)
ker = c.execute('select name, start, end from kernels order by start').fetchall()
beams = [r for r in ker if 'beam_search_kernel' in r[0]]
prev_end = max([b[2] for b in beams[:-2]], default=ker[0][1])      # the last call: behind the previous call's last search
call_end = max(b[2] for b in beams[-2:])
last = [r for r in ker if r[1] >= prev_end - 1 and r[1] <= call_end]
t0 = last[0][1]
print(f'call: {len(last)} kernels, {(last[-1][2] - t0) / 1e6:.2f} ms')
for t in cand:
    cols = [r[1] for r in c.execute(f'pragma table_info({t})').fetchall()]
    print(t, cols)
    if not {'start', 'end'} <= set(cols):
        continue
    size = 'size' if 'size' in cols else ('bytes' if 'bytes' in cols else None)
    rows = c.execute(f'select * from {t} where start >= ? and start <= ? order by start', (t0 - 3_000_000, last[-1][2])).fetchall()
    for r in rows:
        d = dict(zip(cols, r))
        if size and d[size] < MIN_BYTES:
            continue
        print(f"  {(d['start'] - t0) / 1e6:8.3f} -> {(d['end'] - t0) / 1e6:8.3f} ms  {d.get(size, '?')} B  " +
              ' '.join(f'{k}={d[k]}' for k in cols if k not in ('start', 'end', size, 'id', 'guid') and d[k] is not None)[:160])
    break
