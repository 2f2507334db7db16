"""Both ``kind`` legs of bench.cpu_baseline() on ONE host: the oracle's restatement ("port", what the GPU box can run) next to the
unmodified reference modules through oracle/shims ("reference", needs /root/reference) -- the same inputs, weights, thread count
and repetition scheme.  Writes profiles/r05_cpu_baseline_port_vs_reference.json.  VERDICT r4 item 7: the port is pinned
bit-identical to the reference; this shows whether it is also equal in SPEED, i.e. whether the GPU box's "port" baseline may
stand in for MASR's own CPU path."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    out = {'host': {'cpu_model': bench.cpu_model(), 'threads': bench.host_cores()}, 'runs': []}
    # interleaved: port, reference, port, reference -- so that a drifting host shows up as disagreement between repeats
    for rep in range(2):
        for port in (True, False):
            r = bench.cpu_baseline(budget_s=40.0, force_port=port)
            out['runs'].append({'kind': r['kind'], 'predict_loop_audio_s_per_s': r['value'], 'batched_audio_s_per_s': r['batched']['value'],
                                'sample': r['sample']})
            print(out['runs'][-1], flush=True)
    by = lambda k, f: [x[f] for x in out['runs'] if x['kind'] == k]
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    out['summary'] = {}
    for f in ('predict_loop_audio_s_per_s', 'batched_audio_s_per_s'):
        p, r = by('port', f), by('reference', f)
        if p and r:
            out['summary'][f] = {'port': p, 'reference': r, 'port_over_reference': round((sum(p) / len(p)) / (sum(r) / len(r)), 4)}
    path = os.path.join(ROOT, 'profiles', 'r05_cpu_baseline_port_vs_reference.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print('wrote', path)
    print(json.dumps(out['summary'], indent=1))


if __name__ == '__main__':
    main()
