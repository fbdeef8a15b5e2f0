#!/bin/bash
# rocprofv3 counter passes over a short bench run; one pass per invocation of the profiler.
# usage: tools/profile_pmc.sh <outdir> [bench args...]
set -e
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
ARGS="--steps 1 --warmup 0 --spp-per-step 2 --no-cpu-baseline $*"
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass misc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LEVEL_WAVES SQ_CYCLES
python3 tools/pmc_summary.py "$OUT"/sq1 "$OUT"/sq2 "$OUT"/fetch "$OUT"/write "$OUT"/misc > "$OUT/summary.json"
