"""Per-kernel mean durations and GPU busy fraction over a window of a rocprofv3 kernel trace (argv: csv, from, to as fractions)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
a, b = float(sys.argv[2]), float(sys.argv[3])
lo, hi = t0 + int(a * (t1 - t0)), t0 + int(b * (t1 - t0))
ev = [e for e in ev if lo <= e[0] <= hi]
per, cnt = collections.Counter(), collections.Counter()
for s, e, n in ev:
    per[n] += e - s
    cnt[n] += 1
span = max(e[1] for e in ev) - ev[0][0]
busy, cs, ce = 0, None, None
for s, e, _ in ev:
    if ce is None or s > ce:
        if ce is not None: busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("window %.1f ms, busy %.1f %%, kernel time / busy %.2f" % (span / 1e6, 100.0 * busy / span, sum(per.values()) / busy))
for n, t in per.most_common(8):
    print("  %-52s n %5d  mean %7.1f us  total %7.2f ms" % (n[:52], cnt[n], t / cnt[n] / 1e3, t / 1e6))
