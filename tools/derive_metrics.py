#!/usr/bin/env python3
"""Derived utilisation figures of the traversal kernels from the counter summaries of
tools/profile_round.sh and tools/profile_mem.sh (one bench step each, under rocprofv3 --pmc).

usage: derive_metrics.py <pmc_summary.json> <mem_summary.json> <sq1 bench json> <m1 bench json> > profiles/r01_derived.json

  valu_busy        VALU-active wave-quad-cycles x 4 / (kernel cycles x 1024 SIMDs), kernel cycles
                   from the library's HIP-event time of the same (profiled) run at 2.4 GHz
  lane_utilisation SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): share of VALU lanes doing work
  waves_per_simd   SQ_WAVE_CYCLES x 4 / (kernel cycles x 1024): average resident waves per SIMD (of 5 compiled)
  ta_busy, tcp_busy  TA_TA_BUSY_sum, TCP_GATE_EN1_sum / (256 CUs x kernel cycles)
  l1_miss_rate     TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum
  l2_hit_rate      TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import json, sys

pmc, mem = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
b1 = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
b2 = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
CLOCK, SIMDS, CUS = 2.4e9, 1024, 256
out = {"clock_hz_assumed": CLOCK, "note": "one bench step under rocprofv3 --pmc; kernels run a few % slower than untraced"}
for kern, key in (("k_shadow_any<false>", "shadow"), ("k_intersect<false>", "intersect")):
    v = {c: x["sum"] for c, x in pmc[kern].items()}
    m = {c: x["sum"] for c, x in mem[kern].items()}
    cyc1 = b1["kernel_ms_per_step"][key] * 1e-3 * CLOCK
    cyc2 = b2["kernel_ms_per_step"][key] * 1e-3 * CLOCK
    out[kern.split("<")[0]] = {
        "valu_busy": v["SQ_ACTIVE_INST_VALU"] * 4 / (cyc1 * SIMDS),
        "lane_utilisation": v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]),
        "waves_per_simd": v["SQ_WAVE_CYCLES"] * 4 / (cyc1 * SIMDS),
        "waves_per_simd_compiled": 5,
        "wave_time_waiting_on_memory": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
        "valu_instructions": v["SQ_INSTS_VALU"],
        "salu_instructions": v["SQ_INSTS_SALU"],
        "fp64_valu_instructions": v["SQ_INSTS_VALU_FMA_F64"] + v["SQ_INSTS_VALU_ADD_F64"] + v["SQ_INSTS_VALU_MUL_F64"] + v["SQ_INSTS_VALU_TRANS_F64"],
        "ta_busy": m["TA_TA_BUSY_sum"] / (CUS * cyc2),
        "tcp_busy": m["TCP_GATE_EN1_sum"] / (CUS * cyc2),
        "l1_miss_rate": m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"],
        "l2_hit_rate": v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]),
    }
json.dump(out, sys.stdout, indent=1)
