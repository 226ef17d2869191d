"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 csv output) into profiles/<tag>_*.{csv,json}."""
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
for f in sorted(glob.glob(f"{src}/stats/*/*kernel_stats.csv"), key=os.path.getmtime)[-1:]:
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
bench = None
try:
    for line in open(f"{src}/bench.json"):
        if line.startswith("{"):
            bench = json.loads(line)
except FileNotFoundError:
    pass
# the bench also launches the OTHER arithmetic's kernel a few times: count only the measured one
# (third template argument: 1 = fma, 0 = strict)
arith_flag = "1" if (bench or {}).get("config", {}).get("arith", "fma") == "fma" else "0"
import re
# ... or, for tolerance-mode batches of up to 4096 chains of a 4-age problem, the 16-lane form (fma only)
pat = re.compile(r"sepaihrd_eval_kernel<\d+, \d+, " + arith_flag + "," + r"|sepaihrd_eval_quad_kernel<\d+, " + arith_flag + "[,>]")
if (bench or {}).get("dtype") == "f32":
    pat = re.compile(r"sepaihrd_eval_f32_kernel<")
counters = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_f64", "pmc_f32"):
    files = sorted(glob.glob(f"{src}/{d}/*/*counter_collection.csv"), key=os.path.getmtime)
    for f in files[-1:]:  # newest run only: gpurun merges into an existing directory
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            counters[k] = sum(v) / len(v)
out = {"tag": tag, "per_launch_counter_averages": counters,
       "units": "SQ_* cycle counters in quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB (gfx950: FETCH_SIZE may "
                "under-report wide loads by 2x, MI355X_MICROARCH.md HBM section)",
       "bench_line": bench}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    out["hbm_bytes_per_launch"] = {"fetch_reported": counters["FETCH_SIZE"] * 1024, "write": counters["WRITE_SIZE"] * 1024,
                                   "total_with_2x_fetch_correction": (2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
