cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b1prof -o t -- python bench.py --episodes-per-step 1 --no-cpu-baseline --no-kernel-events --no-fp16-leg --steps 30 --warmup 5 > gpurun_out/b1prof.log 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/b1prof/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last 20 steps' worth: find the repeating marker = im2col kernel
idx = [i for i, r in enumerate(rows) if "im2col_p16" in r["Kernel_Name"]]
# two im2col per step (support, target)
starts = idx[::2]
a, b = starts[-21], starts[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = []
for x, y in zip(seg[:-1], seg[1:]):
    gaps.append(int(y["Start_Timestamp"]) - int(x["End_Timestamp"]))
print("20 steps: wall %.3f ms/step, kernel busy %.3f ms/step, launches/step %d, mean gap %.2f us, gaps>3us: %d/step (sum %.3f ms/step)" % (
    (t1 - t0) / 20e6, busy / 20e6, len(seg) / 20, sum(gaps) / len(gaps) / 1e3, sum(1 for g in gaps if g > 3000) / 20, sum(g for g in gaps if g > 3000) / 20e6))
d = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"][:90]; d[k][0] += 1; d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%5.1f/step %8.1f us/step %7.1f us avg  %s" % (n / 20, t / 20e3, t / n / 1e3, k))
PY
rm -rf gpurun_out/b1prof
