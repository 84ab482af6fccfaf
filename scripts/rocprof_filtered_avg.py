"""Per-kernel average duration from a rocprofv3 kernel trace with the early-exit dispatches left out
(PCG launches enqueued after a solve converged return at once: < 20 us against > 100 us of work).
Usage: python scripts/rocprof_filtered_avg.py <kernel_trace.csv> <out.json>"""
import csv, json, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    head = name.split("(")[0]
    # the PCG's own SpMV: MODE 0 on FP64 blocks -- <CH, NT, 0, double[, K, DIAGK]>; the one-system kernel is
    # K = 1 (since round 4 one template serves one and several right-hand sides)
    if "k_spmv_span" not in head or not (", 0>" in head or ", 0, double>" in head or ", 0, double, 1, false>" in head):
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(name.split("(")[0], dict(all_n=0, all_us=0.0, work_n=0, work_us=0.0))
    a["all_n"] += 1; a["all_us"] += d
    if d >= 20.0:
        a["work_n"] += 1; a["work_us"] += d
out = {k: dict(dispatches=v["all_n"], avg_us_all=v["all_us"] / max(v["all_n"], 1), working_dispatches=v["work_n"],
               avg_us_working=v["work_us"] / max(v["work_n"], 1)) for k, v in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
