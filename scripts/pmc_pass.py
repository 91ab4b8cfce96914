"""Collect hardware counters for the bench step, one rocprofv3 pass per counter group, and print the
per-kernel mean per launch.  Usage: python scripts/pmc_pass.py OUT.json "CTR_A CTR_B" "CTR_C ..." """
import collections, csv, glob, json, os, subprocess, sys
out_path, groups = sys.argv[1], sys.argv[2:]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(dict)
for gi, grp in enumerate(groups):
    d = "/tmp/pmc_pass_%d" % gi
    cmd = ["rocprofv3", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(repo, "bench.py"), "--inner", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-pmc", "--no-parity-mode"] + os.environ.get("NCW_PMC_BENCH_ARGS", "").split()  # e.g. "--config shipped"
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print("group failed:", grp, r.stderr[-600:])
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(files[0])):
        # keep anonymous-namespace kernels and template arguments apart (round 1 collapsed them into one "" entry)
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0] if "(" in k else k
        a = acc[(k, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in acc.items():
        res[k][c] = s / n
        res[k]["launches"] = n
json.dump(res, open(out_path, "w"), indent=1)
keep = [k for k in res if any(t in k for t in ("sdf_", "color_", "nerf_", "wgrad", "pack_", "unpack_"))]
for k in sorted(keep):
    print(k[:60], {c: round(v) for c, v in res[k].items()})
