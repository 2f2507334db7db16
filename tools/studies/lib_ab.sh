#!/bin/bash
# A/B of library builds on ONE box: installs each of the given .so files in turn (round-robin, REPS rounds) and prints the
# kernel times of the batch-32 step.  usage (through gpurun): bash tools/lib_ab.sh ab/libA.so ab/libB.so ...
REPS=${REPS:-3}
for r in $(seq $REPS); do
  for lib in "$@"; do
    cp $lib masr_amd/lib/libmasr_hip.so
    python tools/kernel_times.py $(basename $lib .so) 2>&1 | tail -1
  done
done
