#!/bin/bash
# round 5, GPU pass D: numpy summation-order probe on the GPU box's host, Silero odd windows, bit-exact route test
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/mean_square_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/mean_square_probe.txt
cat $O/mean_square_probe.txt
timeout 600 python -m pytest tests/test_silero.py tests/test_gpu_identity.py -m gpu -q 2>&1 | tail -15 > $O/pytest_small.txt
tail -6 $O/pytest_small.txt
