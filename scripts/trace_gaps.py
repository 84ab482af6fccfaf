"""Kernel time and idle gaps of the last part of a rocprofv3 kernel trace (csv)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
def short(n):
    m = re.match(r"(?:void )?(?:sim3opt::|sim3opt_bundle::)?(\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:30]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("span %.1f ms, kernel time %.1f ms (%.0f %%), %d launches" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(rows)))
agg = {}
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    if prev is not None: a[2] += max(0, s - prev) / 1e3
    prev = e
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-36s %6d x  kernel %8.1f us (avg %6.1f)  gap before it %8.1f us (avg %5.1f)" % (k, n, d, d / n, g, g / n))
