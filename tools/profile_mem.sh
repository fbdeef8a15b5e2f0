#!/bin/bash
# Texture-path (TA/TCP/TD) counter passes over one bench step; one pass per profiler run.
# usage: tools/profile_mem.sh <outdir> [bench args...]      (run on the GPU box)
set -e
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
ARGS="--steps 1 --warmup 0 --no-cpu-baseline $*"
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
}
pass m1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
pass m2 TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass m3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_avr TA_TA_BUSY_max GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py "$OUT"/m1 "$OUT"/m2 "$OUT"/m3 > "$OUT/mem_summary.json"
