"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / total / avg, GPU busy vs wall span."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
agg = collections.OrderedDict()
busy = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = r["Kernel_Name"]
    import re
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0][:90]
    a = agg.setdefault(n, [0, 0, []])
    a[0] += 1; a[1] += d; a[2].append(d)
    busy += d
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels %d  busy %.3f ms  span %.3f ms  (idle %.1f%%)" % (len(rows), busy / 1e6, span / 1e6, 100 * (1 - busy / span)))
# median next to the average: the first launch of a kernel under the profiler can take tens of ms (code load / first touch) and skews the average
for n, (c, t, ds) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%6d  %10.1f us total  %8.1f us avg  %8.1f us median  %5.1f%%  %s" % (c, t / 1e3, t / 1e3 / c, sorted(ds)[len(ds) // 2] / 1e3, 100 * t / busy, n))
