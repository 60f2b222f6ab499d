"""Reads a rocprofv3 kernel trace CSV: GPU busy fraction (union of kernel intervals), mean concurrency, per-kernel totals
over a window of the trace (fractions given as argv[2], argv[3]; default the last 60 %)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
a, b = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.4, 1.0)
lo, hi = t0 + int(a * (t1 - t0)), t0 + int(b * (t1 - t0))
ev = [e for e in ev if lo <= e[0] <= hi]
span = max(e[1] for e in ev) - ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in ev)
print("window %.2f ms: GPU busy %.1f %%, mean kernels in flight while busy %.2f" % (span / 1e6, 100.0 * busy / span, tot / busy))
per = collections.Counter()
for s, e, n in ev: per[n] += e - s
for n, t in per.most_common(8): print("  %-46s %.3f ms (%.1f %% of kernel time)" % (n, t / 1e6, 100.0 * t / tot))
