import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith("{"):
        d=json.loads(line); r=d["roofline"]; print(round(d["ms_per_step"],3), {k:v for k,v in r["per_step_kernel_ms"].items() if v>0.1})
