#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests/test_gpu_identity.py tests/test_gpu_parity.py tests/test_gpu_bench.py tests/test_silero.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r05f/pytest.txt
tail -30 gpurun_out/r05f/pytest.txt
