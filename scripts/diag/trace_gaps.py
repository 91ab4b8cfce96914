"""Kernel trace (rocprofv3 --kernel-trace csv) of bench.py --inner -> per step: span, union of the kernel intervals (time with at
least one kernel running), the uncovered remainder, and the largest uncovered gaps with the kernels either side.  Steps are cut at
the optimiser kernel (adam_dev_kernel)."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r[2]]
print("%d kernels, %d steps" % (len(rows), len(ends)))
stats = []
for a, b in zip(ends[:-1], ends[1:]):
    step = rows[a + 1:b + 1]
    if len(step) < 20:
        continue
    t0, t1 = rows[a][1], step[-1][1]   # from the previous step's optimiser end to this one's
    cover, cur_s, cur_e, gaps = 0, None, None, []
    last_name = rows[a][2]
    cur_e = t0
    for s, e, name in step:
        if s > cur_e:
            gaps.append((s - cur_e, last_name, name))
            cover += 0
            cur_e = e
            cover += e - s
        else:
            if e > cur_e:
                cover += e - cur_e
                cur_e = e
        if e >= cur_e:
            last_name = name
    stats.append((t1 - t0, cover, len(step), gaps))
stats = stats[len(stats) // 2:]  # the timed half (after warm-up)
n = len(stats)
span = sum(s[0] for s in stats) / n
cover = sum(s[1] for s in stats) / n
print("steps used: %d; kernels per step %.0f; span %.3f ms; covered by >= 1 kernel %.3f ms; uncovered %.3f ms (%.1f %%)"
      % (n, sum(s[2] for s in stats) / n, span / 1e6, cover / 1e6, (span - cover) / 1e6, 100.0 * (span - cover) / span))
g = sorted(stats[-1][3], reverse=True)
print("gaps of the last step: %d, sum %.1f us; largest:" % (len(g), sum(x[0] for x in g) / 1e3))
for d, a, b in g[:25]:
    print("  %6.1f us   after %-50s before %s" % (d / 1e3, a, b))
