"""tools/valu_issue output -> profiles/rNN_valu_issue.json: the issue cost (cycles per wave-instruction per
SIMD at saturation = 8 waves per SIMD) of the two classes of VALU instructions bench.py's roofline_valu uses.

usage: make_valu_issue_profile.py <valu_issue.txt> <out.json>"""
import json
import re
import sys

FAST = ("v_add_f32 (2 banks)", "v_sub_f32", "v_mul_f32", "v_fmac_f32")
SLOW = ("v_med3_f32 (3 banks)", "v_min_f32", "v_max_f32", "v_and_or_b32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cndmask_b32 (sgpr pair)")
rows = {}
for line in open(sys.argv[1]):
    m = re.findall(r"(\d)w:\s+([\d.]+) cyc \(([\d.]+) ns, ([\d.]+) GHz\)", line)
    if m:
        rows[line[:28].strip()] = {int(w): {"cycles": float(c), "ns": float(ns), "GHz": float(g)} for w, c, ns, g in m}
fast = [rows[n][8]["cycles"] for n in FAST if n in rows]
slow = [rows[n][8]["cycles"] for n in SLOW if n in rows]
clk = [v[8]["GHz"] for v in rows.values()]
out = {
    "_note": "tools/valu_issue.hip on MI355X: cycles per wave-instruction per SIMD, 8 independent instructions per loop trip, "
             "fixed physical registers, in-kernel clock from s_memtime / s_memrealtime. fast class = v_add/sub/mul/fmac_f32, "
             "v_and/or/xor_b32, v_add/sub_u32 (and a three-VGPR v_fma_f32 at ~3.2); slow class = v_min/max/med3 (f32, i32, u32), "
             "compares, v_cndmask, shifts, every other three-operand form, packed fp32 and all fp64. Class figures = mean over "
             "the listed opcodes at 8 waves per SIMD (the lowest, i.e. most generous, cost).",
    "fast_class_cycles": round(sum(fast) / len(fast), 3),
    "slow_class_cycles": round(sum(slow) / len(slow), 3),
    "clock_GHz": round(sum(clk) / len(clk), 3),
    "fast_class_opcodes": list(FAST), "slow_class_opcodes": list(SLOW),
    "cycles_by_opcode_and_waves_per_simd": {n: {str(w): v["cycles"] for w, v in r.items()} for n, r in rows.items()},
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fast_class_cycles", "slow_class_cycles", "clock_GHz")}))
