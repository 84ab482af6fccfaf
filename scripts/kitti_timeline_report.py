"""Launch sequence of the last LM iterations from a rocprofv3 kernel trace (csv): name, duration, gap before."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tail = rows[-n:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%7.1f us  gap %6.1f  %s" % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r["Kernel_Name"][:80]))
    prev = e
