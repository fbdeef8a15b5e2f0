#!/bin/bash
# Per-call cost per library variant under gpurun_variants/: tools/call_overhead.py (4, 16 and 128 spp per call of the
# benchmark frame) once per variant; prints the wall times and the finishing launch's share.
export MIPT_DEBUG=1
for lib in gpurun_variants/*.so; do
  echo "== $lib"
  MIPT_LIB=$PWD/$lib timeout -k 10 200 python -u tools/call_overhead.py 2>/dev/null | grep -E "spp (plain|events)" | sed -e 's/launches per call: //' -e 's/intersect.*tail/tail/' || exit 1
done
