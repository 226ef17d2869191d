#!/bin/bash
# Runs on the GPU box (gpurun): a subset of the GPU suite (-k expression in $1, default: everything), then the default bench line.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/gpu_check.sh "libm or checkpoints" tag'
K=${1:-""}
TAG=${2:-check}
O=gpurun_out
mkdir -p $O
if [ -n "$K" ]; then
  python -m pytest tests -m gpu -x -q -k "$K" > $O/${TAG}_tests.log 2>&1
else
  python -m pytest tests -m gpu -x -q > $O/${TAG}_tests.log 2>&1
fi
RC=$?
tail -8 $O/${TAG}_tests.log
[ $RC -ne 0 ] && exit $RC
python bench.py ${BENCH_ARGS:-} > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -c 800 $O/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${TAG}_bench.json"))
r = d["roofline"]
print("c1 value %.4g strict %.4g ms/step %.4f kernel_ms %.4f (its pass %.4f) frac %.4f phase_pass %s" % (d["value"], d["value_strict"] or 0, d["ms_per_step"], r["kernel_ms"], r["kernel_pass_step_ms"], r["frac"], d["kernel_info"]["phase_pass_applied"]))
for k, v in (d.get("other_workloads") or {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"]); continue
    print("%-7s %.4g evals/s  ms/step %.3f  kernel %.3f + ll %.3f  frac %.3f  of step %.3f  vgprs %d  %s" % (k, v["evals_per_s"], v["ms_per_step"], v["kernel_ms"], v["likelihood_pass_ms"], v["roofline"]["frac"], v["roofline"]["frac_of_step"], v["kernel_info"]["vgprs"], v["likelihood_form"]))
    if v.get("sampler_iteration"):
        print("        sampler at this size:", v["sampler_iteration"])
sp = d.get("sampler_pipeline") or {}
print("sampler", {k: sp.get(k) for k in ("ms_per_iteration", "proposals_per_s", "accept_trace_mismatches_vs_strict", "error")}, (sp.get("long_run") or {}).get("ms_per_iteration"))
print("cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
PY
