#!/bin/bash
# runs bench.py once per library variant under gpurun_variants/ (kernel tuning)
for lib in gpurun_variants/*.so; do
  MIPT_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; print('$lib', round(j['value'],1), 'intersect %.1f shade %.1f shadow %.1f' % (k['intersect'],k['shade'],k['shadow']))" || exit 1
done
