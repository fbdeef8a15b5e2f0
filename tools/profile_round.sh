#!/bin/bash
# One round's profiling evidence for bench.py's default workload:
#   1. rocprofv3 --kernel-trace --stats over the default bench command,
#   2. separate counter passes (FETCH_SIZE, WRITE_SIZE, SQ mix) over one step of it.
# usage: tools/profile_round.sh <outdir> <tag> [bench args...]      (run on the GPU box)
set -e
OUT=$1; TAG=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/stats_bench.json" 2> "$OUT/stats.err"
ARGS="--steps 1 --warmup 0 --no-cpu-baseline $*"
pass() {
  name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64
python3 tools/pmc_summary.py "$OUT"/fetch "$OUT"/write "$OUT"/sq1 "$OUT"/sq2 > "$OUT/pmc_${TAG}.json"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_${TAG}.csv" \;
