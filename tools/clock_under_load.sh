#!/bin/bash
# The clock the chip holds: (1) a pure FP64 FMA loop at 1 / 2 / 4 / 8 waves per SIMD (tools/ubench/fp64_sustained),
# (2) GRBM_GUI_ACTIVE / 8 / kernel time of the evaluation kernel at 262 144 chains (dispatches of >= 10 ms, where that
# quotient is a good estimate: MI355X_MICROARCH.md, DVFS give-back) and at 4096 chains (short: reads high).
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 120 tools/ubench/fp64_sustained > gpurun_out/fp64_sustained.log 2>&1 || exit 1
cat gpurun_out/fp64_sustained.log
cd /tmp
for S in 262144 4096; do
  O=$R/gpurun_out/clock_$S
  rm -rf $O; mkdir -p $O
  SWEEP_SIZES=$S timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -- python3 $R/tools/sweep_batch.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for S in (262144, 4096):
    d = f"gpurun_out/clock_{S}"
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in cc:
        for r in csv.DictReader(open(f)):
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = collections.defaultdict(list)
    for k, (name, ns) in dur.items():
        if "sepaihrd_eval" in name and k in agg and ns > 0:
            rows[name[:60]].append((ns, agg[k].get("GRBM_GUI_ACTIVE", 0), agg[k].get("SQ_WAVE_CYCLES", 0), agg[k].get("SQ_BUSY_CYCLES", 0)))
    for name, v in rows.items():
        v = v[len(v) // 2:]
        ns = sum(x[0] for x in v) / len(v); g = sum(x[1] for x in v) / len(v)
        print(f"chains={S} {name}: n={len(v)} avg {ns/1e6:.3f} ms, GRBM_GUI_ACTIVE/8/time = {g/8/ns:.3f} GHz, SQ_WAVE_CYCLES {sum(x[2] for x in v)/len(v):.4g}")
PY
