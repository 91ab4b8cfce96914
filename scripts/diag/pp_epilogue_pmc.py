"""GPU: SQ counters of the three epilogue variants of the plain fp16 sdf_inferC (scripts/diag/pp_epilogue.py): VALU / MFMA
instruction counts, MFMA-busy and wave-wait cycles per launch.  One rocprofv3 --pmc pass per variant and counter group."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GROUPS = ["SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES",
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"]
res = {}
for epi in ("f32", "pk16", "poly16"):
    res[epi] = collections.defaultdict(dict)
    for gi, grp in enumerate(GROUPS):
        d = "/tmp/pp_pmc_%s_%d" % (epi, gi)
        cmd = ["rocprofv3", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                                                      os.path.join(ROOT, "scripts", "diag", "pp_epilogue.py"), "--one"]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", NCW_PP_EPI=epi), capture_output=True, text=True)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(files[0])):
            if "sdf_inferC" not in row["Kernel_Name"] or int(row.get("Grid_Size", 0) or 0) != 1024 * 512:
                continue  # the 131,072-point launches of the fp16 build (1024 workgroups x 512 threads)
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
        for c, (s, n) in acc.items():
            res[epi]["counters"][c] = s / n
            res[epi]["counters"]["launches"] = n
    c = res[epi]["counters"]
    if c.get("SQ_INSTS_MFMA"):
        res[epi]["valu_per_mfma"] = round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2)
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs' matrix pipes; SQ_BUSY_CYCLES over the 8 XCDs
        res[epi]["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["SQ_BUSY_CYCLES"] / 8.0), 3)
    if c.get("SQ_WAVE_CYCLES"):
        res[epi]["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
        res[epi]["wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
    print(epi, json.dumps(res[epi]), flush=True)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/pp_pmc.json", "w"), indent=1)
