#!/bin/bash
# Round-5 GPU session 4: (a) timing probe: the weight-gradient launch with its split-K flush as plain stores instead of f32 atomics
# (probe library `wst`, wrong results, timing only) beside the product; (b) SQ counters of sdf_inferC with one / two output blocks per wave.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05d; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
for i in 1 2; do
  $T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/headline_$i.json 2>/dev/null; echo "headline $i rc $?" >> $OUT/status
  NEUCONW_HIP_LIB=neuralrecon-w_amd/libneuconw_hip_wst.so $T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/headline_wst_$i.json 2>/dev/null; echo "headline wst $i rc $?" >> $OUT/status
done
export TMPDIR=/tmp
for NB in 1 2; do
  for G in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
    D=/tmp/pmc_nb${NB}_$(echo $G | cut -c1-12 | tr ' ' '_')
    (cd /tmp && NCW_PP_NB=$NB NEUCONW_HIP_LIB=$GRAFT_REPO_ROOT/neuralrecon-w_amd/libneuconw_hip_nb2.so $T 200 rocprofv3 --pmc $G --output-format csv -d $D -o p -- python $GRAFT_REPO_ROOT/scripts/diag/pp_nb2.py --one > /dev/null 2>&1)
    F=$(find $D -name "*counter_collection.csv" | head -1)
    echo "== NB=$NB counters: $G" >> $OUT/pp_nb_pmc.log
    [ -n "$F" ] && python - "$F" >> $OUT/pp_nb_pmc.log <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "sdf_inferC" not in row["Kernel_Name"]:
        continue
    a = acc[(row["Counter_Name"], row.get("Grid_Size", ""))]
    a[0] += float(row["Counter_Value"]); a[1] += 1
for (c, g), (s, n) in sorted(acc.items()):
    print("  %-28s grid %-10s mean per launch %.4g  (%d launches)" % (c, g, s / n, n))
PY
  done
done
echo "pmc done" >> $OUT/status
cat $OUT/status; cat $OUT/pp_nb_pmc.log | head -60
