#!/bin/bash
# One round's profiling evidence for one bench.py workload (run on the GPU box):
#   1. rocprofv3 --kernel-trace --stats over the bench command (its JSON line is kept beside it),
#   2. separate counter passes (FETCH_SIZE; WRITE_SIZE + L2 hits; two SQ mixes; two texture-path
#      mixes) over ONE step of it, each in a profiler run of its own,
#   3. tools/make_counters.py -> profiles/counters.json (stamped with the kernel-source hash).
# usage: tools/profile_round.sh <outdir> <tag> <workload> [bench args...]
export MIPT_DEBUG=1  # the library reads its tuning knobs only then
OUT=$1; TAG=$2; WL=$3; shift 3
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --workload "$WL" --no-cpu-baseline --no-parity "$@" > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || { echo "stats pass failed"; tail -5 "$OUT/stats.err"; exit 1; }
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_${TAG}_${WL}.csv" \;
ARGS="--workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-parity $*"
pass() {
  name=$1; shift
  timeout -s KILL 500 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -5 "$OUT/$name.err"; exit 1; }
  echo "pass $name done"
}
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY &&
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 &&
pass m1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE &&
pass m3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_avr TA_TA_BUSY_max GRBM_GUI_ACTIVE &&
cp profiles/counters.json "$OUT/counters.json" 2>/dev/null
python3 tools/make_counters.py "$OUT" "$WL" "$TAG" "$OUT/counters.json" && echo "counters written"
