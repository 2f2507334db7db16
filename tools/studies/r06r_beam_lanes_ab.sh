# Study (round 6): configs[2] GPU-search lines against lanes, pass sizes and the number of hardware queues
mkdir -p gpurun_out/r06r; O=gpurun_out/r06r; : > $O/bench.txt
timeout 600 python -m pytest tests/test_gpu_facade.py -x -q -k "two_lanes or config2" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
run() {   # lanes pass queues workload
  echo "lanes=$1 pass=$2 hwq=$3 $4" >> $O/bench.txt
  ( [ "$3" != "-" ] && export GPU_MAX_HW_QUEUES=$3; MASR_LANES=$1 MASR_BENCH_BEAM_PASS=$2 python bench.py --workload $4 2>>$O/bench.err ) | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); print('   ', {k: j[k] for k in j if k in ('value', 'ms_per_step')})
" >> $O/bench.txt
}
for W in squeezeformer_b64_beam_sharp squeezeformer_b64_beam; do
  run 1 32 - $W
  run 2 32 - $W
  run 2 32 8 $W
  run 2 16 - $W
  run 2 16 8 $W
  run 1 16 - $W
  run 2 16,16,32 8 $W
  run 2 12,20,32 8 $W
  run 2 24,24,16 8 $W
done
cat $O/bench.txt
