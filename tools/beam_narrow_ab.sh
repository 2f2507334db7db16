#!/bin/bash
# A/B of the narrow step's wave count (beam_gpu.hip BS_NARROW_WAVES) on the GPU box: rebuilds beam_gpu.o per variant, relinks, runs
# tools/beam_profile.py at sharp posteriors without / with a 3-gram.   usage: bash tools/beam_narrow_ab.sh gpurun_out/r06c "4 8 16"
D=${1:-gpurun_out/narrow_ab}; mkdir -p $D
L=masr_amd/lib
OBJS=$(ls $L/*.o | grep -v beam_gpu)
for W in ${2:-4 8 16}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DBS_NARROW_WAVES=$W -c masr_amd/csrc/beam_gpu.hip -o /tmp/beam_gpu_w$W.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $L/libmasr_hip.so $OBJS /tmp/beam_gpu_w$W.o || exit 1
  echo "== BS_NARROW_WAVES=$W"
  for o in 0 3; do timeout 120 python tools/beam_profile.py 498 4233 300 $o 1 1 14 2>&1 | grep -v amdgpu | tail -5; done
done > $D/narrow_ab.txt 2>&1
cat $D/narrow_ab.txt
