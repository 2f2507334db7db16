timeout 900 python -m pytest tests/test_beam_search.py -m gpu -x -q 2>&1 | tail -3
for o in 0 3 5; do timeout 120 python tools/beam_profile.py 498 4233 300 $o 1 1 14 2>&1 | grep -v amdgpu | tail -3; done
timeout 120 python tools/beam_profile.py 498 4233 300 3 1 1 1 2>&1 | grep -v amdgpu | tail -1
