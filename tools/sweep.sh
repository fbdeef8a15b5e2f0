#!/bin/bash
# Kernel tuning: runs bench.py once per library variant under gpurun_variants/ and per
# value of the environment knob given as KNOB=v1,v2,... (default: none).
# usage: tools/sweep.sh [KNOB=v1,v2,...] [bench args...]
export MIPT_DEBUG=1  # the library reads its tuning knobs only then
knob=""; vals="-"
if [[ "$1" == *=* ]]; then knob="${1%%=*}"; vals="${1#*=}"; shift; fi
for lib in gpurun_variants/*.so; do
  for v in ${vals//,/ }; do
    [ -n "$knob" ] && export "$knob=$v"
    MIPT_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity "$@" 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; c=j['counters']
print('$lib $knob=$v', round(j['value'],1), 'intersect %.1f shade %.1f shadow %.1f sum %.1f' % (k['intersect'],k['shade'],k['shadow'],k['shadow_sum']), 'nodes c/s %d/%d prims c/s %d/%d M' % (c['node_visits']/1e6-c['shadow_node_visits']/1e6, c['shadow_node_visits']/1e6, c['prim_tests']/1e6-c['shadow_prim_tests']/1e6, c['shadow_prim_tests']/1e6), j['config']['bvh_nodes'])" || exit 1
  done
done
