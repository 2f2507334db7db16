mkdir -p gpurun_out/r06q; O=gpurun_out/r06q
timeout 900 python -m pytest tests/test_gpu_facade.py -x -q -k "two_lanes or config2 or equals_single or evaluate or rccl or squeezeformer_beam" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/studies/sqz_two_lane_probe.py 32,32 > $O/two_lane.txt 2>&1
for L in 2 1; do echo "MASR_LANES=$L" >> $O/head.txt; MASR_LANES=$L python tools/studies/predict_batch_head.py greedy >> $O/head.txt 2>&1; done
for L in 2 1 2 1; do
  for W in squeezeformer_b64_greedy squeezeformer_b64_beam_sharp squeezeformer_b64_beam squeezeformer_b64_beam_wordlm_host; do
    echo "MASR_LANES=$L $W" >> $O/bench.txt
    MASR_LANES=$L MASR_BENCH_SQZ_AB=0 python bench.py --workload $W 2>>$O/bench.err | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); print({k: j[k] for k in j if k in ('value', 'ms_per_step', 'fixed_passes_of_32_ms_per_step', 'balanced_passes_ms_per_step', 'latency_ms')})
" >> $O/bench.txt
  done
done
cat $O/two_lane.txt $O/head.txt $O/bench.txt
