"""Kernel timeline of the bench step: GPU busy vs idle, idle time attributed to the kernel that follows it.
Runs rocprofv3 --kernel-trace on bench.py and prints a summary (run on the GPU box)."""
import collections, csv, glob, os, subprocess, sys
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/gap_trace"
steps, warm = 10, 5
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--",
       sys.executable, os.path.join(repo, "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline"]
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print(r.stderr[-2000:]); sys.exit(1)
rows = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"]) for x in csv.DictReader(open(f[0]))]
rows.sort()
# one step = the span between consecutive launches of the first kernel of render (sample_coarse)
marks = [i for i, x in enumerate(rows) if "sample_coarse" in x[2]]
lo, hi = marks[warm + 2], marks[warm + 2 + 5]     # 5 timed steps
seg = rows[lo:hi]
span = seg[-1][1] - seg[0][0]
busy = 0; cur_end = seg[0][0]
gap_by = collections.defaultdict(lambda: [0, 0]); dur_by = collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    k = k.split("(")[0].replace("void ", "")[:70]
    if s > cur_end:
        gap_by[k][0] += s - cur_end; gap_by[k][1] += 1
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    dur_by[k][0] += e - s; dur_by[k][1] += 1
n = 5.0
print("per step: span %.3f ms, busy %.3f ms, idle %.3f ms, kernels %d" % (span / n / 1e6, busy / n / 1e6, (span - busy) / n / 1e6, len(seg) / n))
print("-- top kernels by time (ms/step, launches/step)")
for k, (t, c) in sorted(dur_by.items(), key=lambda kv: -kv[1][0])[:28]:
    print("  %-70s %.4f %5.1f" % (k, t / n / 1e6, c / n))
print("-- idle attributed to the kernel that follows (ms/step, count/step)")
for k, (t, c) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:16]:
    print("  %-70s %.4f %5.1f" % (k, t / n / 1e6, c / n))
