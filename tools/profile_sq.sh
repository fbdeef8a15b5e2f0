#!/bin/bash
# one SQ counter pass over a short bench run: tools/profile_sq.sh <outdir>
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/sq1" -- python3 bench.py --steps 1 --warmup 0 --spp-per-step 2 --no-cpu-baseline "$@" > "$OUT/sq1.json" 2> "$OUT/sq1.err" || exit 1
python3 tools/pmc_summary.py "$OUT"/sq1 > "$OUT/summary.json"
python3 - "$OUT/summary.json" <<'PY'
import json,sys
s=json.load(open(sys.argv[1]))
for k in sorted(s):
    if not k.startswith('k_'): continue
    c={n:v['sum'] for n,v in s[k].items()}
    if c.get('SQ_WAVES',0)==0 or c.get('SQ_ACTIVE_INST_VALU',0)==0: continue
    wc=c['SQ_WAVE_CYCLES']
    print('%-28s waves %.3g valu %.3g salu %.3g valu/wave %.0f active_valu/wc %.3f lane_util %.3f wave_cycles %.3g' % (k, c['SQ_WAVES'], c['SQ_INSTS_VALU'], c['SQ_INSTS_SALU'], c['SQ_INSTS_VALU']/c['SQ_WAVES'], c['SQ_ACTIVE_INST_VALU']/wc, c['SQ_THREAD_CYCLES_VALU']/(c['SQ_ACTIVE_INST_VALU']*64), wc))
PY
