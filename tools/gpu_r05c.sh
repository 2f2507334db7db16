#!/bin/bash
# round 5, GPU pass C: XCD-resident layer skeleton, the bit-exact route test with diagnostics, beam timelines with grouped searches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/_xcd_layer.bin > $O/xcd_layer_study.txt 2>&1
cat $O/xcd_layer_study.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "contract_route or transcribe_batch_against or baseline_size" 2>&1 | tail -60 > $O/pytest_route.txt
grep -E "passed|failed|differ|Error" $O/pytest_route.txt | head -20
for cfg in "balanced 0" "32 0" "balanced 1" "32 1"; do
  set -- $cfg
  echo "=== passes $1 sharp $2 (searches after the group's encoders, 3 side streams)" >> $O/beam_timeline.txt
  MASR_BENCH_BEAM_PASS=$1 MASR_PROFILE_SHARP=$2 timeout 300 python tools/beam_batch_profile.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -12 >> $O/beam_timeline.txt
done
echo "=== passes 32 sharp 1, group 1 (search right behind its own encoder, next encoder behind the search)" >> $O/beam_timeline.txt
MASR_BENCH_BEAM_PASS=32 MASR_PROFILE_SHARP=1 MASR_BEAM_GROUP=1 timeout 300 python tools/beam_batch_profile.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -8 >> $O/beam_timeline.txt
cat $O/beam_timeline.txt
