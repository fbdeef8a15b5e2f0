#!/usr/bin/env python3
"""Static census of a kernel's vector instructions by issue class.

On gfx950 a wave64 vector instruction occupies its SIMD's issue port for ~2.2 cycles, and most opcodes
additionally hold a second, narrower pipe for ~4.1 cycles; only a handful (the fp32 fma / add / mul, moves and
the plain bitwise / add / sub / right-shift integer ops) do not.  Measured with tools/micro/valu_issue.hip
(profiles/r03_valu_issue.json); the sets below are that table.  A loop of N vector instructions of which S are
of the slow class therefore takes at least max(2.2 N, 4.1 S) cycles per SIMD per wave-iteration, however many
waves are resident.

usage: tools/valu_mix.py [kernel-name-regex ...]     (compiles csrc/kernels.hip to assembly, prints JSON)
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_fma_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32"}
QUARTER = {"v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64"}  # ~16 cycles (v_rcp_f64, v_sqrt_f64 measured)
HALF_TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}  # ~8 cycles (v_rcp_f32 measured)
CYC_ISSUE, CYC_SLOW = 2.2, 4.13


def base(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def census(lines):
    ops = collections.Counter()
    for ln in lines:
        m = re.match(r"\s+(v_\w+)", ln)
        if m:
            ops[base(m.group(1))] += 1
    n = sum(ops.values())
    fast = sum(c for o, c in ops.items() if o in FAST)
    slow_cycles = sum(c * (16.0 if o in QUARTER else 8.0 if o in HALF_TRANS else CYC_SLOW) for o, c in ops.items() if o not in FAST)
    return {"valu": n, "fast_class": fast, "slow_class": n - fast, "slow_share": (n - fast) / n if n else 0.0,
            "min_cycles_per_pass": max(CYC_ISSUE * n, slow_cycles),
            "cycles_per_instruction": max(CYC_ISSUE * n, slow_cycles) / n if n else 0.0,
            "top": sorted(ops.items(), key=lambda x: -x[1])[:12]}


def kernel_bodies(asm):
    for f in re.split(r"\n(?=_Z\w+:)", asm):
        name = f.split(":", 1)[0]
        if name.startswith("_Z") and "s_endpgm" in f:
            yield name, f.split("\n")


def hot_loop(lines):
    """The largest backward-branch loop of the kernel (for the persistent traversal kernels: the loop that takes
    rays, walks and tests) and the innermost loop that holds the node fetch (the descend loop)."""
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, ln in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", ln) or re.search(r"s_branch (\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    if not loops:
        return None, None
    outer = max(loops, key=lambda ab: ab[1] - ab[0])
    with_fetch = [ab for ab in loops if any("v_cvt_f32_ubyte" in l for l in lines[ab[0]:ab[1] + 1])]
    inner = min(with_fetch, key=lambda ab: ab[1] - ab[0]) if with_fetch else None
    return outer, inner


def compile_asm():
    """The kernels' gfx950 assembly, compiled as csrc/Makefile compiles them."""
    src = os.path.join(ROOT, "path_tracer_amd", "csrc", "kernels.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DMIPT_TRAV_WAVES=5", "-DMIPT_LEAN_WAVES=6",
                               "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
        return open(out).read()


def main():
    pats = sys.argv[1:] or [r"k_shadow_anyILb0ELb0ELb1ELb1ELb0E", r"k_intersectILb0ELb0ELb0E", r"k_shadow_anyILb0ELb0ELb0ELb0ELb1E"]
    asm = compile_asm()
    sys.path.insert(0, ROOT)
    from path_tracer_amd.buildinfo import kernel_source_hash

    res = {"kernel_source_hash": kernel_source_hash(), "cycles_per_issue": CYC_ISSUE, "cycles_slow_class": CYC_SLOW,
           "fast_class_opcodes": sorted(FAST), "kernels": {}}
    for name, lines in kernel_bodies(asm):
        if not any(re.search(p, name) for p in pats):
            continue
        outer, inner = hot_loop(lines)
        entry = {"whole_kernel": census(lines)}
        if outer:
            entry["persistent_loop"] = census(lines[outer[0]:outer[1] + 1])
        if inner:
            entry["descend_loop"] = census(lines[inner[0]:inner[1] + 1])
        res["kernels"][name] = entry
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
