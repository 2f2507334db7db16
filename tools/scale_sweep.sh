#!/bin/bash
# The 1 / 2 / 4 / 8-GPU scaling curve of the contract workload in one command, for the first box with more than one GPU
# (the driver runs the same commands for SCALE_rNN.json):
#     bash tools/scale_sweep.sh [steps] [warmup]        ->  gpurun_out/scale/scale_N.json + scale_summary.json
# Every N is launched exactly as the driver launches it (python -m torch.distributed.run, one rank per GPU over RCCL).  Checks:
# the N = 1 value equals the plain `python bench.py` line within 3 %, every line saw all its ranks in the timed region's
# all-gather (config.rccl_ranks_seen), and the per-rank step times (reported next to the max-over-ranks value) lie within 5 % of
# each other.  Exit code 2 when a check fails; gpurun_out/scale/scale_summary.json holds the SCALE-shaped rows either way.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
STEPS=${1:-20}; WARM=${2:-5}
OUT=gpurun_out/scale; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NGPU=${MASR_SWEEP_GPUS:-$(python -c "import torch; print(torch.cuda.device_count())")}
echo "GPUs visible: $NGPU"
python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-extra --no-cpu-baseline > $OUT/plain_1.json 2> $OUT/plain_1.log || exit 1
PORT=29611
for N in 1 2 4 8; do
    [ "$N" -le "$NGPU" ] || { echo "skip N=$N (only $NGPU GPUs)"; continue; }
    PORT=$((PORT + 1))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --steps $STEPS --warmup $WARM --no-extra --no-cpu-baseline > $OUT/scale_$N.json 2> $OUT/scale_$N.log \
        || { echo "N=$N FAILED"; tail -5 $OUT/scale_$N.log; exit 1; }
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
line = lambda p: json.loads([l for l in open(p) if l.startswith('{')][-1])
plain = line(os.path.join(out, 'plain_1.json'))
rows, ok = [], True
for p in sorted(glob.glob(os.path.join(out, 'scale_*.json')), key=lambda p: int(p.split('_')[-1].split('.')[0])):
    r = line(p)
    n = r['n_gpus']
    seen = r['config'].get('rccl_ranks_seen')
    rows.append({'n_gpus': n, 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'per_rank_ms': r['config'].get('ms_per_step_per_rank'),
                 'rccl_ranks_seen': seen, 'backend': r['config'].get('backend'),
                 'efficiency_vs_n1': None})
    if seen != list(range(n)):
        ok = False
        print(f'N={n}: the timed all-gather saw ranks {seen}, expected {list(range(n))}')
    per = r['config'].get('ms_per_step_per_rank') or []
    if len(per) == n and n > 1 and (max(per) - min(per)) > 0.05 * max(per):
        ok = False
        print(f'N={n}: per-rank step times spread by more than 5 %: {per}')
base = next((r['value'] for r in rows if r['n_gpus'] == 1), None)
for r in rows:
    if base:
        r['efficiency_vs_n1'] = round(r['value'] / (base * r['n_gpus']), 4)
if base and abs(base / plain['value'] - 1) > 0.03:
    ok = False
    print(f'N=1 under the launcher ({base}) and the plain run ({plain["value"]}) differ by more than 3 %')
json.dump({'plain_n1': plain['value'], 'rows': rows, 'checks_passed': ok}, open(os.path.join(out, 'scale_summary.json'), 'w'), indent=1)
print(json.dumps(rows, indent=1))
sys.exit(0 if ok else 2)
PY
