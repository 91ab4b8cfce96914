"""Run bench.py (no CPU baseline) against library variants and print the per-kernel times side by side.
usage: python scripts/bench_kernels.py tag1 tag2 ...   ("" or "default" = the product library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
tags = sys.argv[1:] or ["default"]
for tag in tags:
    env = dict(os.environ)
    if tag not in ("", "default"):
        env["NEUCONW_HIP_LIB"] = os.path.join(ROOT, "neuralrecon-w_amd", "libneuconw_hip_%s.so" % tag)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, "FAILED", r.stderr[-500:]); continue
    d = json.loads(line[-1])
    rows[tag] = dict(d["roofline"]["per_step_kernel_ms"], __step_ms=d["ms_per_step"], __Mrs=d["value"] / 1e6)
keys = ["__step_ms", "__Mrs"] + sorted({k for r in rows.values() for k in r if not k.startswith("__")},
                                       key=lambda k: -max(r.get(k, 0) for r in rows.values()))[:10]
print("%-22s" % "kernel" + "".join("%12s" % t for t in rows))
for k in keys:
    print("%-22s" % k + "".join("%12.4f" % rows[t].get(k, float("nan")) for t in rows))
