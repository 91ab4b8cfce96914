"""Stash-byte ledger of one bench step: every stash operand of the three networks -- who writes it, who reads
it, blocks (32 features x 2 B = 64 B per point per block) -- summed per kernel and compared with the HBM traffic the PMC
passes measured (gpurun_out/r03/pmc_traffic_v1.json: 2 x FETCH_SIZE + WRITE_SIZE, KB per launch).

    python scripts/stash_ledger.py [pmc_traffic.json]

Shapes: BASELINE configs[1] (1024 rays x 128 inside samples, 132 background samples; SDF 8 x 256, colour 4 x 256 with a 2-layer
128-wide head, background NeRF 8 x 256 with a 4-layer 128-wide head), 16-bit stash.  No GPU needed."""
import json
import sys

R, S, O = 1024, 128, 4
N_IN, N_BG = R * S, R * (S + O)
B = 64  # bytes per block per point (32 features x 2 B)

# (operand, blocks, writer, [readers]) -- readers may repeat a kernel (read twice)
sdf = [("gamma", 2, "sdf_fwd", ["sdf_bwd", "wgrad", "wgrad"])]                     # W_0 and the skip layer's gamma columns
sdf += [("h_%d" % l, 8, "sdf_fwd", ["sdf_fwd", "sdf_bwd", "wgrad"] + (["wgrad", "wgrad"] if l == 8 else [])) for l in range(1, 9)]
sdf += [("feat", 8, "sdf_fwd", ["color_fwd", "wgrad"])]                            # colour net input; xyz_encoding_final's wgrad
sdf += [("t_%d" % l, 8, "sdf_fwd", ["sdf_bwd", "wgrad"] + (["wgrad"] if l == 4 else [])) for l in range(0, 8)]
sdf += [("dfeat", 8, "color_bwd", ["sdf_bwd", "wgrad"])]
sdf += [("zbar2_%d (in zbar_%d)" % (l, l), 8, "sdf_bwd", ["sdf_bwd"]) for l in range(0, 8)]   # pass (1) -> pass (2)
sdf += [("zbar_%d" % l, 8, "sdf_bwd", ["wgrad"] + (["wgrad"] if l == 4 else [])) for l in range(0, 8)]
sdf += [("qbar_0", 2, "sdf_bwd", ["wgrad", "wgrad"])] + [("qbar_%d" % l, 8, "sdf_bwd", ["wgrad"]) for l in range(1, 9)]
sdf += [("zsdf", 1, "sdf_bwd", ["wgrad"]), ("one", 1, "sdf_bwd", ["wgrad"])]

col = [("aux1", 3, "color_fwd", ["wgrad"]), ("aux2", 1, "color_fwd", ["wgrad"]), ("f", 8, "color_fwd", ["color_bwd", "wgrad"])]
col += [("e_%d" % i, 4, "color_fwd", ["color_bwd", "wgrad"]) for i in range(2)]
col += [("x_%d" % i, 8, "color_fwd", ["color_bwd", "wgrad"]) for i in range(4)]
col += [("zf", 8, "color_bwd", ["wgrad"]), ("zo", 1, "color_bwd", ["wgrad"])]
col += [("ze_%d" % i, 4, "color_bwd", ["wgrad"] + (["wgrad"] if i == 0 else [])) for i in range(2)]
col += [("zx_%d" % i, 8, "color_bwd", ["wgrad"] + (["wgrad"] if i == 0 else [])) for i in range(4)]

bg = [("gp", 3, "nerf_fwd", ["wgrad", "wgrad"]), ("aux1", 3, "nerf_fwd", ["nerf_fwd", "wgrad"])]
bg += [("h_%d" % i, 8, "nerf_fwd", ["nerf_bwd", "wgrad"] + (["wgrad"] if i == 8 else [])) for i in range(1, 9)]
bg += [("featn", 8, "nerf_fwd", ["wgrad"])]
bg += [("e_%d" % i, 4, "nerf_fwd", ["nerf_bwd", "wgrad"]) for i in range(4)]
bg += [("zp_%d" % i, 8, "nerf_bwd", ["wgrad"] + (["wgrad"] if i == 5 else [])) for i in range(8)]
bg += [("zalpha", 1, "nerf_bwd", ["wgrad"]), ("zfeat", 8, "nerf_bwd", ["wgrad"]), ("zrgb", 1, "nerf_bwd", ["wgrad"])]
bg += [("ze_%d" % i, 4, "nerf_bwd", ["wgrad"] + (["wgrad"] if i == 0 else [])) for i in range(4)]

kern = {}
store = 0


def add(k, kind, nbytes):
    kern.setdefault(k, {"r": 0.0, "w": 0.0})[kind] += nbytes


rows = []
for name, ops, n in (("SDF", sdf, N_IN), ("colour", col, N_IN), ("background", bg, N_BG)):
    for op, blocks, writer, readers in ops:
        by = blocks * B * n
        store += 0 if "zbar2" in op else by
        add(writer, "w", by)
        for rd in readers:
            add(rd, "r", by)
        rows.append((name, op, blocks, writer, readers, by))

pmc = None
if len(sys.argv) > 1:
    raw = json.load(open(sys.argv[1]))
    pmc = {}
    for k, v in raw.items():
        for short in ("wgrad", "sdf_bwd", "sdf_fwd", "nerf_bwd", "nerf_fwd", "color_bwd", "color_fwd"):
            if short in k:
                pmc[short] = (2 * v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024)

print("| net | operand | blocks | B / sample | written by | read by | GB / step (x readers) |")
print("|---|---|---|---|---|---|---|")
seen = {}
for name, op, blocks, writer, readers, by in rows:
    key = (name, op.split("_")[0] if op[-1].isdigit() or "zbar2" in op else op, blocks, writer, tuple(sorted(readers)))
    seen.setdefault(key, []).append((op, by))
for (name, base, blocks, writer, readers), items in seen.items():
    ops = [o for o, _ in items]
    label = ops[0] if len(ops) == 1 else "%s .. %s" % (ops[0], ops[-1].split("_")[-1] if "zbar2" not in ops[-1] else ops[-1])
    by = sum(b for _, b in items)
    print("| %s | %s | %d x %d | %d | %s | %s | %.3f w + %.3f r |" % (name, label, len(ops), blocks, len(ops) * blocks * B, writer,
                                                                  ", ".join(readers), by / 1e9, by * len(readers) / 1e9))
print()
print("| kernel | stash read GB | stash write GB | total | PMC read (2 x FETCH) | PMC write | PMC total |")
print("|---|---|---|---|---|---|---|")
tot = tot_p = 0.0
for k in ("wgrad", "sdf_bwd", "sdf_fwd", "nerf_bwd", "nerf_fwd", "color_bwd", "color_fwd"):
    r, w = kern[k]["r"] / 1e9, kern[k]["w"] / 1e9
    tot += r + w
    if pmc and k in pmc:
        pr, pw = pmc[k][0] / 1e9, pmc[k][1] / 1e9
        tot_p += pr + pw
        print("| %s | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f |" % (k, r, w, r + w, pr, pw, pr + pw))
    else:
        print("| %s | %.3f | %.3f | %.3f | | | |" % (k, r, w, r + w))
print("| **sum** | | | **%.3f** | | | **%.3f** |" % (tot, tot_p))
print()
print("stash held in HBM: %.2f GB (%.1f KB per inside sample incl. its share of the background stash)" % (store / 1e9, store / N_IN / 1e3))
