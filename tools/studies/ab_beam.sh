# A/B of library builds on the GPU beam search (phase cycles of workgroup 0, us per frame): bash tools/ab_beam.sh ab/libA.so ab/libB.so
for r in 1 2; do
for lib in "$@"; do
  cp $lib masr_amd/lib/libmasr_hip.so
  echo "== $lib"
  python tools/beam_profile.py 498 4233 300 3 2>&1 | tail -2
  python tools/beam_profile.py 498 4233 300 5 2>&1 | tail -2
done
done
