#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05e
timeout 300 python tools/mean_square_locate.py 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/r05e/locate.txt
cat gpurun_out/r05e/locate.txt
