"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 csv output) into profiles/<tag>_*.{csv,json}."""
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
for f in glob.glob(f"{src}/stats/*/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
counters = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_f64"):
    for f in glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "sepaihrd_eval_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            counters[k] = sum(v) / len(v)
bench = None
try:
    for line in open(f"{src}/bench.json"):
        if line.startswith("{"):
            bench = json.loads(line)
except FileNotFoundError:
    pass
out = {"tag": tag, "per_launch_counter_averages": counters,
       "units": "SQ_* cycle counters in quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB (gfx950: FETCH_SIZE may "
                "under-report wide loads by 2x, MI355X_MICROARCH.md HBM section)",
       "bench_line": bench}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    out["hbm_bytes_per_launch"] = {"fetch_reported": counters["FETCH_SIZE"] * 1024, "write": counters["WRITE_SIZE"] * 1024,
                                   "total_with_2x_fetch_correction": (2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
