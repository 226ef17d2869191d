#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of the sampler loop: the period of the evaluation kernel and what fills the time
between one evaluation's end and the next one's start.  usage: sampler_timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [r for r in rows if "sepaihrd_eval" in r["Kernel_Name"]]
ev = ev[len(ev) // 4:]  # steady state
period = [(int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a, b in zip(ev, ev[1:])]
dur = [(int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a in ev]
gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(ev, ev[1:])]
med = lambda v: sorted(v)[len(v) // 2]
print(f"evaluations {len(ev)}: period median {med(period):.1f} us, kernel {med(dur):.1f} us, end-to-next-start {med(gap):.1f} us")
# what ran in a typical gap
a, b = ev[len(ev) // 2], ev[len(ev) // 2 + 1]
t0 = int(a["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0 and e <= int(b["Start_Timestamp"]):
        print(f"  +{(s - t0) / 1e3:7.1f} us  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:70]}")
import statistics
steady = period[len(period) // 8:]
print(f"period: mean {statistics.mean(steady):.1f} us, median {statistics.median(steady):.1f}, p90 {sorted(steady)[int(0.9 * len(steady))]:.1f}, "
      f"over 700 us: {sum(p > 700 for p in steady)} of {len(steady)}")
