#!/bin/bash
# round 5, GPU pass B: fbank A/B, beam-search pass timelines, full GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/fbank_ab.py > $O/fbank_ab.txt 2>&1
cat $O/fbank_ab.txt | tail -8
for cfg in "balanced 2 0" "balanced 3 0" "32 2 0" "balanced 3 1" "32 2 1" "16 4 1"; do
  set -- $cfg
  echo "=== passes $1 sides $2 sharp $3" >> $O/beam_timeline.txt
  MASR_BENCH_BEAM_PASS=$1 MASR_BEAM_SIDES=$2 MASR_PROFILE_SHARP=$3 timeout 300 python tools/beam_batch_profile.py 2>&1 | grep -v Warning | tail -14 >> $O/beam_timeline.txt
done
cat $O/beam_timeline.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.txt
tail -12 $O/pytest.txt
