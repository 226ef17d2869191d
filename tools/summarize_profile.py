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
# derived per-launch figures the documents quote
c = counters
if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
    f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    out["derived"] = {"valu_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"], "salu_per_wave": c.get("SQ_INSTS_SALU", 0.0) / c["SQ_WAVES"],
                      "fp64_share_of_valu": f64 / c["SQ_INSTS_VALU"] if f64 else None,
                      # wave64 instruction counts x 64 lanes; an FMA is two flops
                      "fp64_flops_issued": 64 * (c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + 2 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0)) if f64 else None}
    if bench and out["derived"]["fp64_flops_issued"]:
        alg = bench["roofline"]["algorithmic_flops_per_eval"] * bench["config"]["chains_per_gpu"]
        out["derived"]["fp64_flops_algorithmic"] = alg
        out["derived"]["issued_over_algorithmic"] = out["derived"]["fp64_flops_issued"] / alg
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
# the replayed HBM traffic bench.py prints (roofline.traffic): profiles/pmc_traffic.json, one key per workload / solver / arithmetic / batch
if bench and "hbm_bytes_per_launch" in out:
    cfg = bench["config"]
    wl = cfg["workload"].split()[1].rstrip(":")
    solver = "cashkarp" if "cashkarp" in cfg["workload"] else "dopri5"
    arith = "f32" if bench.get("dtype") == "f32" else cfg["arith"]
    key = f"{wl}_{solver}_{arith}_B{cfg['chains_per_gpu']}"
    path = "profiles/pmc_traffic.json"
    table = json.load(open(path)) if os.path.exists(path) else {}
    alg = bench["roofline"]["hbm"]["algorithmic_bytes_per_eval"] * cfg["chains_per_gpu"]
    ws = bench["roofline"]["hbm"]["likelihood_workspace_bytes_per_eval"] * cfg["chains_per_gpu"]
    table[key] = {"hbm_bytes_per_launch": out["hbm_bytes_per_launch"]["total_with_2x_fetch_correction"],
                  "fetch_size_reported_bytes": out["hbm_bytes_per_launch"]["fetch_reported"], "write_size_bytes": out["hbm_bytes_per_launch"]["write"],
                  "algorithmic_bytes_per_launch": alg, "over_algorithmic": out["hbm_bytes_per_launch"]["total_with_2x_fetch_correction"] / alg,
                  "likelihood_workspace_bytes_per_launch": ws, "kernel": bench["roofline"]["kernel"],
                  "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, per-launch average of the measured integrator kernel only)",
                  "correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section: gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported"}
    json.dump(table, open(path, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
