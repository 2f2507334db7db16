"""Copy the summaries of one measurement pass (tools/measure_round.sh <dir>) into profiles/<round>_*: text files keep their
one-line header (what the file is, which command made it) with the pass's name in it, JSON files are copied as they are.
usage: python profiles/collect_pass.py gpurun_out/r06m5 r06 ["note on the box"]"""
import json
import os
import re
import shutil
import sys

src, rnd = sys.argv[1].rstrip('/'), sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else "one box of the pool's normal kind"
here = os.path.dirname(os.path.abspath(__file__))
TEXT = {'b1_kernel_stats.txt': 'b1_kernel_stats.txt', 'beam_history.txt': 'beam_history.txt', 'beam_profile.txt': 'beam_profile.txt',
        'beam_timeline_flat.txt': 'beam_timeline_squeezeformer_b64_beam.txt',
        'beam_timeline_sharp.txt': 'beam_timeline_squeezeformer_b64_beam_sharp.txt',
        'deepspeech2_kernel_stats.txt': 'deepspeech2_kernel_stats.txt', 'efficient_kernel_stats.txt': 'efficient_kernel_stats.txt',
        'kernel_stats.txt': 'kernel_stats.txt', 'squeezeformer_beam_kernel_stats.txt': 'squeezeformer_beam_kernel_stats.txt',
        'squeezeformer_greedy_kernel_stats.txt': 'squeezeformer_greedy_kernel_stats.txt',
        'stream128_kernel_stats.txt': 'stream128_kernel_stats.txt', 'stream16_kernel_stats.txt': 'stream16_kernel_stats.txt',
        'sqz_skip_ab.txt': 'sqz_skip_ab.txt'}
JSON = ['efficient_mfma_util.json', 'hbm_traffic.json', 'mfma_util.json', 'rccl_single_rank.json', 'serving.json',
        'squeezeformer_greedy_hbm_traffic.json', 'squeezeformer_greedy_mfma_util.json', 'squeezeformer_mfma_util.json',
        'stream128_mfma_util.json']
HOST = [('tools/facade_profile.py', 'facade_profile.txt'), ('tools/stream_host_profile.py', 'pool_host.txt'),
        ('tools/b1_ab.py', 'b1_ab.txt'), ('tools/chunk_lat.py build', 'chunk_lat.txt')]
lead = f'# round {rnd[1:].lstrip("0")} measurement pass of the FINAL build (tools/measure_round.sh {src}; {note}): '


def header_of(path, fallback):
    if os.path.exists(path):
        first = open(path, encoding='utf-8').readline().rstrip('\n')
        m = re.match(r'# round \d+ measurement pass[^:]*\(tools/measure_round\.sh [^)]*\): (.*)', first)
        if m:
            return lead + m.group(1)
    return lead + fallback


for dst, name in TEXT.items():
    out = os.path.join(here, f'{rnd}_{dst}')
    body = open(os.path.join(src, name), encoding='utf-8').read()
    head = header_of(out, f'{rnd}_{dst}')
    open(out, 'w', encoding='utf-8').write(head + '\n' + body)
for name in JSON:
    shutil.copy(os.path.join(src, name), os.path.join(here, f'{rnd}_{name}'))
line = [ln for ln in open(os.path.join(src, 'bench.json'), encoding='utf-8').read().splitlines() if ln.startswith('{')][-1]
json.loads(line)
open(os.path.join(here, f'{rnd}_bench_line.json'), 'w', encoding='utf-8').write(line + '\n')
with open(os.path.join(here, f'{rnd}_host_profiles.txt'), 'w', encoding='utf-8') as f:
    f.write(lead + 'host-side profiles\n')
    for k, (tool, name) in enumerate(HOST):
        f.write(('\n' if k else '') + f'# {tool}\n' + open(os.path.join(src, name), encoding='utf-8').read())
print('collected', len(TEXT) + len(JSON) + 2, 'files from', src)
