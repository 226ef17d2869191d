#!/usr/bin/env python3
"""Per-dispatch means of the counters tools/pmc_placement.sh collected, for the 16-lane evaluation kernel of each arithmetic build."""
import csv, glob, sys, collections
tag, root = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "eval_quad_kernel" not in k:
            continue
        arith = "fma" if "<0, 1, true>" in k or "ILi0ELi1ELb1" in k else ("strict" if "<0, 0, true>" in k or "ILi0ELi0ELb1" in k else k[-30:])
        acc[arith][r["Counter_Name"]].append(float(r["Counter_Value"]))
for arith, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    waves = m.get("SQ_WAVES", 0) or 1
    line = [f"{tag:14s} {arith:6s}"]
    for c in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS",
              "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES"):
        if c in m:
            line.append(f"{c[3:] if c.startswith('SQ_') else c}={m[c] / waves:.0f}")
    print(" ".join(line), f"(per wave; waves {waves:.0f})")
