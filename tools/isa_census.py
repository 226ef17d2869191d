#!/usr/bin/env python3
"""Diagnostic: instruction census per basic block of one kernel in a hipcc -S listing.
usage: isa_census.py file.s <kernel-substring> [min_block_size]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(l.split(":")[0][0:0] or l.rstrip()[-1]) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(m):
    if m.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_cmp_", "v_div_", "v_rcp_f64", "v_ldexp_f64", "v_frexp", "v_fract_f64", "v_trunc_f64", "v_floor_f64", "v_rndne_f64", "v_cvt_f64", "v_cvt_i32_f64", "v_sqrt_f64", "v_rsq_f64")):
        if m.startswith("v_cmp_") and "f64" not in m: return "valu32"
        return "fp64"
    if m.startswith("v_accvgpr"): return "agpr_mov"
    if "dpp" in m: return "dpp"
    if m.startswith("v_cndmask"): return "cndmask"
    if m.startswith(("v_mov", "v_readlane", "v_writelane", "v_readfirstlane")): return m.split("_e")[0]
    if m.startswith("v_"): return "valu32"
    if m.startswith("s_waitcnt"): return "waitcnt"
    if m.startswith("s_nop"): return "nop"
    if m.startswith("s_cbranch") or m.startswith("s_branch"): return "branch"
    if m.startswith("s_"): return "salu"
    if m.startswith(("ds_",)): return "lds"
    if m.startswith(("global_", "buffer_", "scratch_", "flat_")): return m.split("_")[0]
    return "other"
blocks, cur, name = [], collections.Counter(), "entry"
full = collections.defaultdict(list)
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not t.startswith(".LBB"): continue
    if t.startswith(".LBB") and t.split()[0].endswith(":"):
        blocks.append((name, cur)); cur = collections.Counter(); name = t.split(":")[0]; continue
    m = t.split()[0]
    if "dpp" in t or "quad_perm" in t or "row_" in t: m = m + "_dpp"
    cur[cls(m)] += 1; full[name].append(t)
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks:
    s = sum(c.values()); tot.update(c)
    if s >= minsz:
        print(f"{n:12s} {s:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
print("TOTAL", sum(tot.values()), dict(tot))
if len(sys.argv) > 4:
    for t in full[sys.argv[4]]: print(t)
