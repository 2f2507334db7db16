mkdir -p gpurun_out/r06s; O=gpurun_out/r06s
timeout 900 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for L in 2 1 2 1; do echo "efficient lanes=$L" >> $O/eff.txt; MASR_BENCH_EFFICIENT_LANES=$L python bench.py --workload efficient_b256 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); print('   ', {k: j[k] for k in j if k in ('value', 'ms_per_step')})
" >> $O/eff.txt; done
cat $O/eff.txt
python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
j = json.loads(open('gpurun_out/r06s/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['timing'])
for k, v in (j.get('extra') or {}).items():
    if isinstance(v, dict): print(k, v.get('value'), v.get('ms_per_step'))
PY
