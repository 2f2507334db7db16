#!/bin/bash
# One measurement pass of a build on the GPU box (rounds 4 - 6): GPU test suite, the default bench line, rocprofv3 kernel traces of
# EVERY BASELINE config's workload (+ the B = 1 pass), PMC passes (FETCH_SIZE / WRITE_SIZE for the contract step; matrix-pipe busy
# for the contract step, the Efficient-Conformer passes, the Squeezeformer + beam search call and the 128-stream pool), summaries
# next to them.      usage (from the repo root, through gpurun):  bash tools/measure_round.sh gpurun_out/r05m
D=${1:-gpurun_out/measure}; mkdir -p $D
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $D/gputests_tail.txt; grep -E "passed|failed|error" $D/gputests_tail.txt | tail -3 > $D/gputests.txt; cat $D/gputests.txt
python bench.py > $D/bench.json 2> $D/bench.err; tail -1 $D/bench.err
export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
MF="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$D/kt -o r2 -- $B > $R/$D/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$D/pf -o f -- $B2 > $R/$D/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$D/pw -o w -- $B2 > $R/$D/pw.log 2>&1
rocprofv3 --pmc $MF --kernel-trace -d $R/$D/pm -o m -- $B2 > $R/$D/pm.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/kte -o e -- python $R/bench.py --workload efficient_b256 --steps 6 > $R/$D/kte.log 2>&1
rocprofv3 --pmc $MF --kernel-trace -d $R/$D/pme -o m -- python $R/bench.py --workload efficient_b256 --steps 2 > $R/$D/pme.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/ktq -o q -- python $R/bench.py --workload squeezeformer_b64_beam > $R/$D/ktq.log 2>&1
rocprofv3 --pmc $MF --kernel-trace -d $R/$D/pmq -o m -- python $R/bench.py --workload squeezeformer_b64_beam_sharp > $R/$D/pmq.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/ktg -o g -- env MASR_BENCH_SQZ_AB=0 python $R/bench.py --workload squeezeformer_b64_greedy > $R/$D/ktg.log 2>&1
rocprofv3 --pmc $MF --kernel-trace -d $R/$D/pmg -o m -- env MASR_BENCH_SQZ_AB=0 python $R/bench.py --workload squeezeformer_b64_greedy > $R/$D/pmg.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$D/pfg -o f -- env MASR_BENCH_SQZ_AB=0 python $R/bench.py --workload squeezeformer_b64_greedy > $R/$D/pfg.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$D/pwg -o w -- env MASR_BENCH_SQZ_AB=0 python $R/bench.py --workload squeezeformer_b64_greedy > $R/$D/pwg.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/ktd -o d -- python $R/bench.py --workload deepspeech2 > $R/$D/ktd.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/kts -o s -- env MASR_BENCH_STREAMS=16 python $R/bench.py --workload stream128 > $R/$D/kts.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/ktS -o S -- python $R/bench.py --workload stream128 > $R/$D/ktS.log 2>&1
rocprofv3 --pmc $MF --kernel-trace -d $R/$D/pmS -o m -- python $R/bench.py --workload stream128 > $R/$D/pmS.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/ktb -o b -- python $R/tools/b1_ab.py trace > $R/$D/ktb.log 2>&1
cd $R
python tools/serve_bench.py > $D/serving.json 2> $D/serving.err
# the prefix search alone: flat posteriors (40 candidates per frame: the wide step) and a sharpened head (3.7: the narrow step),
# without / with the scorer; the narrow step off (masr_debug_set key 37 through BEAM_PROFILE_NARROW=0); 64 searches side by side
{ for sc in 1 14; do for o in 0 3 5; do python tools/beam_profile.py 498 4233 300 $o 1 1 $sc 2>&1 | grep -v amdgpu | tail -4; done; done
  echo "== narrow step off (every frame on the wide step), sharpened head"
  for o in 0 3; do BEAM_PROFILE_NARROW=0 python tools/beam_profile.py 498 4233 300 $o 1 1 14 2>&1 | grep -v amdgpu | tail -2; done
  echo "== 64 utterances side by side, sharpened head, 3-gram"
  BEAM_PROFILE_B=64 python tools/beam_profile.py 498 4233 300 3 1 1 14 2>&1 | grep -v amdgpu | tail -1; } > $D/beam_profile.txt 2>&1
# the beam call after a server-like process history against a cold process (tests/test_gpu_bench.py asserts within 10 %)
{ python tools/beam_history_probe.py 0 2>/dev/null | grep RESULT; python tools/beam_history_probe.py 12 2>/dev/null | grep RESULT; } > $D/beam_history.txt
# GPU-side time lines of one configs[2] call (rocprofv3 kernel trace): searches against the encoder passes
for w in squeezeformer_b64_beam_sharp squeezeformer_b64_beam; do
  ( cd /tmp; rocprofv3 --kernel-trace -d $R/$D/ktl -o t -- python $R/bench.py --workload $w --no-cpu-baseline --steps 4 > $R/$D/ktl_$w.log 2>&1 )
  python tools/beam_gpu_timeline.py $(find $D/ktl -name "*.db") 2 2 > $D/beam_timeline_$w.txt 2>&1; rm -rf $D/ktl
done
python tools/sqz_skip_ab.py 2>&1 | grep -v amdgpu | grep "skip padded" > $D/sqz_skip_ab.txt
python tools/chunk_lat.py build 2>&1 | tail -1 > $D/chunk_lat.txt
python tools/stream_host_profile.py 128 3 2>&1 | grep -E "streams:|masr_pool_step" > $D/pool_host.txt
python tools/stream_host_profile.py 16 3 2>&1 | grep -E "streams:|masr_pool_step" >> $D/pool_host.txt
python tools/facade_profile.py 2>&1 | grep -v amdgpu > $D/facade_profile.txt
python tools/b1_ab.py 2>&1 | grep -v amdgpu > $D/b1_ab.txt
MASR_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $D/rccl_single_rank.json 2> $D/rccl_single_rank.err
python profiles/summarize_rocpd.py $(find $D/ktg -name "*.db") > $D/squeezeformer_greedy_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/ktd -name "*.db") > $D/deepspeech2_kernel_stats.txt
python profiles/summarize_mfma.py $(find $D/pmg -name "*.db") $D/squeezeformer_greedy_mfma_util.json > $D/squeezeformer_greedy_mfma.txt 2>&1
python profiles/summarize_pmc.py $(find $D/pfg -name "*.db") $(find $D/pwg -name "*.db") $D/squeezeformer_greedy_hbm_traffic.json > $D/squeezeformer_greedy_hbm.txt 2>&1
python profiles/summarize_rocpd.py $(find $D/kt -name "*.db") 48 > $D/kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/kte -name "*.db") > $D/efficient_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/ktq -name "*.db") > $D/squeezeformer_beam_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/kts -name "*.db") > $D/stream16_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/ktS -name "*.db") > $D/stream128_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/ktb -name "*.db") > $D/b1_kernel_stats.txt
python profiles/summarize_pmc.py $(find $D/pf -name "*.db") $(find $D/pw -name "*.db") $D/hbm_traffic.json > $D/hbm.txt 2>&1
python profiles/summarize_mfma.py $(find $D/pm -name "*.db") $D/mfma_util.json > $D/mfma.txt 2>&1
python profiles/summarize_mfma.py $(find $D/pme -name "*.db") $D/efficient_mfma_util.json > $D/efficient_mfma.txt 2>&1
python profiles/summarize_mfma.py $(find $D/pmq -name "*.db") $D/squeezeformer_mfma_util.json > $D/squeezeformer_mfma.txt 2>&1
python profiles/summarize_mfma.py $(find $D/pmS -name "*.db") $D/stream128_mfma_util.json > $D/stream128_mfma.txt 2>&1
head -14 $D/kernel_stats.txt
# the experimental kernels: rebuild with them, run their tests, restore the product library
MASR_BUILD_EXPERIMENTS=1 python -c "import masr_amd.build as b; b.build()" > $D/experiments_build.log 2>&1
MASR_BUILD_EXPERIMENTS=1 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_few_rows.py tests/test_gpu_ffn_packed.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -2 > $D/experiments_tests.txt
python -c "import masr_amd.build as b; b.build()" >> $D/experiments_build.log 2>&1
cat $D/experiments_tests.txt
find $D -name "*.db" -delete
