#!/usr/bin/env python3
"""Issue cost of the vector instructions of each kernel, from its ISA listing and the measured per-instruction rates.

   The PMC counters split a kernel's vector instructions into FP64 add / mul / fma, transcendentals and "the rest"; on gfx950 the rest
   does NOT issue at one rate (profiles/ubench/valu_rate.hip, measured with 8 waves per SIMD, cycles per wave-instruction on a SIMD):
       2   v_xor / v_and / v_or / v_not / v_bitop3 / v_mov / v_add_u32 / v_sub_u32 / v_add_f32 / v_mul_f32 / v_fma(c)_f32
           on vector registers, inline constants or a literal
       4   the same with a scalar-register operand (SGPR, VCC, EXEC), any DPP / SDWA form, and every other vector instruction:
           v_bcnt, v_min / v_max, shifts, v_and_or, v_or3, v_add3, v_lshl_add, v_mad / v_mul (int), v_cvt, v_cmp, v_cndmask,
           v_readlane, v_bfe, v_perm, v_pk_*, and the FP64 min / max / cmp / ldexp
       4   v_add_f64 / v_mul_f64 / v_fma_f64 (4.2 - 4.8 at the nominal clock)
       8   FP32 transcendentals        16   FP64 transcendentals
   This script takes hmmufotu_amd/csrc/hu_engine.s (`make asm`), finds each kernel's innermost loops (a label and a backward branch
   to it with no other backward branch between) and prices the "rest" instructions in them: the mean goes into
   profiles/<tag>_isa_costs.json and make_pmc_summary.py uses it in place of a flat 2 cycles when it turns the typed instruction
   counts into issue cycles (`valu_issue_frac`).  Static mean over the loop bodies, not a dynamic count: the kernels in question spend
   their time in unrolled inner loops whose instruction mix is uniform.

   Usage: python3 profiles/isa_cost.py hmmufotu_amd/csrc/hu_engine.s profiles/r02_isa_costs.json"""
import json, re, subprocess, sys

FAST = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_bitop3_b32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32"}
F64 = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64"}
T32 = {"v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}
T64 = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}
SCALAR = re.compile(r"(^|[\s,\[])(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|scc)($|[\s,\]])")


def classify(line):
    """-> (class, cycles) for a vector instruction, None for anything else"""
    t = line.split(";")[0].strip()
    if not t.startswith("v_"):
        return None
    op = t.split()[0]
    mod = "dpp" in op or "sdwa" in op or " quad_perm" in t or " row_" in t or "_sel:" in t
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if base in F64:
        return "f64", 4
    if base in T64:
        return "t64", 16
    if base in T32:
        return "t32", 8
    args = t[len(op):]
    if base in FAST and not mod and not SCALAR.search(args):
        return "rest", 2
    return "rest", 4


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield name, body
                name = None


def innermost_loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    back = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i:
                back.append((tgt, i))
    return [(a, b) for a, b in back if not any((c, d) != (a, b) and a <= c and d <= b for c, d in back)]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    names = []
    rows = []
    for name, body in kernels(src):
        loops = innermost_loops(body)
        lines = [l for a, b in loops for l in body[a:b + 1]] if loops else body
        cls = [c for c in map(classify, lines) if c]
        rest = [cy for k, cy in cls if k == "rest"]
        if not cls:
            continue
        names.append(name)
        rows.append(dict(loops=len(loops), vector_instructions_in_loops=len(cls), rest=len(rest), rest_mean_cycles=(sum(rest) / len(rest) if rest else None),
                         rest_at_2_cycles=sum(1 for x in rest if x == 2), fp64=sum(1 for k, _ in cls if k == "f64"),
                         trans=sum(1 for k, _ in cls if k in ("t32", "t64"))))
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for n, d, r in zip(names, dem, rows):
        key = d.split("(")[0].replace("void ", "").strip()
        if key.startswith("k_"):
            out[key] = r
    json.dump(dict(note="static issue cost of the non-FP64, non-transcendental vector instructions in each kernel's innermost loops; rates from "
                        "profiles/ubench/valu_rate.hip; see profiles/isa_cost.py", kernels=out), open(dst, "w"), indent=1)
    for k in sorted(out):
        r = out[k]
        if r["rest_mean_cycles"]:
            print("%-60s loops %2d  rest %5d  mean %.2f cycles" % (k[:60], r["loops"], r["rest"], r["rest_mean_cycles"]))


if __name__ == "__main__":
    main()
