#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05i
mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
