#!/bin/bash
# Round-5 GPU sessions (the 18 gpurun calls of that round, kept as a record of how profiles/r05/ was produced), one function per call:
#     gpurun --timeout T -- 'bash scripts/r05/gpu_sessions.sh <N>'      (N = 1 .. 18)
set -u
cd "$(dirname "$0")/../.."

# Round-5 GPU session 1: the whole -m gpu suite on the new build, smoke, the default bench line, the new secondary rows
# (render / voxel / grid512 with parity), the shipped shape three times with and without the elimination (VERDICT weak 5), the
# atomics probe and the RCCL world-1 call sequence under NCCL_DEBUG=INFO.  Every stage under `timeout`, own log.
s1() {
OUT=gpurun_out/r05a; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 -x > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
$T 900 python -m pytest tests/test_gpu_render_only.py tests/test_gpu_voxel.py tests/test_gpu_grid.py tests/test_gpu_fullsize.py tests/test_gpu_trainer.py -m gpu -q --timeout 500 -s > $OUT/new_tests.log 2>&1; echo "new tests rc $?" >> $OUT/status
$T 120 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/status
$T 400 python bench.py --save-trained-state $OUT/trained_state_bench.pt > $OUT/bench_v1.json 2> $OUT/bench_v1.err; echo "bench rc $?" >> $OUT/status
$T 300 python bench.py --config render > $OUT/bench_render.json 2> $OUT/bench_render.err; echo "render rc $?" >> $OUT/status
$T 300 python bench.py --config render --bg-eliminate --no-pmc > $OUT/bench_render_elim.json 2> $OUT/bench_render_elim.err; echo "render elim rc $?" >> $OUT/status
$T 300 python bench.py --config voxel --no-pmc > $OUT/bench_voxel.json 2> $OUT/bench_voxel.err; echo "voxel rc $?" >> $OUT/status
$T 300 python bench.py --config grid512 --prec f16 > $OUT/bench_grid512_f16.json 2> $OUT/bench_grid512.err; echo "grid512 rc $?" >> $OUT/status
for i in 1 2 3; do
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/shipped_dense_$i.json 2>/dev/null; echo "shipped dense $i rc $?" >> $OUT/status
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_$i.json 2>/dev/null; echo "shipped elim $i rc $?" >> $OUT/status
done
$T 120 scripts/probes/atomic_probe > $OUT/atomic_probe.log 2>&1; echo "atomic probe rc $?" >> $OUT/status
PORT=$((20000 + RANDOM % 20000))
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL \
  $T 200 python tests/_rccl_world1_worker.py > $OUT/rccl_world1_nccl_debug.log 2>&1; echo "rccl world1 rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log $OUT/new_tests.log | head; tail -3 $OUT/smoke.log
python scripts/show_bench.py < $OUT/bench_v1.json 2>/dev/null | head -40
}

# Round-5 GPU session 2: adjoint sweep with hi + lo weights (sdf_fwdSA), adaptive selection share in the weight-gradient plan,
# the tests that failed in session 1, shipped-shape sweep of the split-K target.
s2() {
OUT=gpurun_out/r05b; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 600 python -m pytest tests/test_gpu_render_only.py tests/test_gpu_voxel.py tests/test_gpu_fullsize.py tests/test_gpu_sdf.py tests/test_gpu_sdf_train.py tests/test_gpu_bg_select.py tests/test_gpu_trainer.py -m gpu -q --timeout 500 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
$T 400 python bench.py --no-pmc > $OUT/bench_v2.json 2> $OUT/bench_v2.err; echo "bench rc $?" >> $OUT/status
NEUCONW_SDF_ADJ_SPLIT=0 $T 300 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_v2_adj_off.json 2>/dev/null; echo "bench adj off rc $?" >> $OUT/status
for i in 1 2; do
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/shipped_dense_$i.json 2>/dev/null; echo "shipped dense $i rc $?" >> $OUT/status
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_$i.json 2>/dev/null; echo "shipped elim $i rc $?" >> $OUT/status
done
for W in 256 384 512 640 1024; do
  NCW_WGRAD_TARGET_WGS=$W $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_wgs$W.json 2>/dev/null; echo "shipped elim wgs $W rc $?" >> $OUT/status
  NCW_WGRAD_TARGET_WGS=$W $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/shipped_dense_wgs$W.json 2>/dev/null; echo "shipped dense wgs $W rc $?" >> $OUT/status
done
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/headline_elim.json 2>/dev/null; echo "headline elim rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
}

# Round-5 GPU session 3: the NB = 2 / AGPR-weights probe of sdf_inferC, the tests that failed on the probe allocation under capture,
# over-plan sweep of the selection share (headline + shipped, elimination on).
s3() {
OUT=gpurun_out/r05c; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
NEUCONW_HIP_LIB=neuralrecon-w_amd/libneuconw_hip_nb2.so $T 300 python scripts/diag/pp_nb2.py > $OUT/pp_nb2.log 2>&1; echo "pp_nb2 rc $?" >> $OUT/status
$T 900 python -m pytest tests -m gpu -q --timeout 500 > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
for OP in 1.0 1.3 1.6 2.0 2.5; do
  NCW_SEL_OVERPLAN=$OP $T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/headline_elim_op$OP.json 2>/dev/null; echo "headline elim op $OP rc $?" >> $OUT/status
  NCW_SEL_OVERPLAN=$OP $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_op$OP.json 2>/dev/null; echo "shipped elim op $OP rc $?" >> $OUT/status
done
for i in 1 2 3; do
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --graph > $OUT/shipped_dense_graph_$i.json 2>/dev/null; echo "shipped dense graph $i rc $?" >> $OUT/status
done
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log | head; cat $OUT/pp_nb2.log
}

# Round-5 GPU session 4: (a) timing probe: the weight-gradient launch with its split-K flush as plain stores instead of f32 atomics
# (probe library `wst`, wrong results, timing only) beside the product; (b) SQ counters of sdf_inferC with one / two output blocks per wave.
s4() {
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
}

# Round-5 GPU session 5: the W = 512 adjoint sweep with hi + lo weights (sdf_fwdS16<., true>): full suite, shipped-shape bench
# with its parity object (dense + elimination), beside NEUCONW_SDF_ADJ_SPLIT=0.
s5() {
OUT=gpurun_out/r05e; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
$T 400 python bench.py --config shipped --no-pmc > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err; echo "shipped rc $?" >> $OUT/status
NEUCONW_SDF_ADJ_SPLIT=0 $T 300 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_shipped_adj_off.json 2>/dev/null; echo "shipped adj off rc $?" >> $OUT/status
$T 300 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_shipped_elim.json 2>/dev/null; echo "shipped elim rc $?" >> $OUT/status
$T 300 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_shipped_2.json 2>/dev/null; echo "shipped 2 rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log | head
}

# Round-5 GPU session 6: full suite on the tree with the large-ray per-ray kernels (> 512 samples per ray) and the coherent-error normals test;
# the default bench line again (the per-ray kernels moved into a namespace: same code).
s6() {
OUT=gpurun_out/r05g; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
$T 300 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_edges.py -m gpu -q -s -k "adjoint or 512" > $OUT/new_tests.log 2>&1; echo "new tests rc $?" >> $OUT/status
$T 100 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/status
$T 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log | head; grep -E "normals vs fp64|samples per ray" $OUT/new_tests.log
}

# Round-5 GPU session 7: the background NeRF's split-precision refinement (ncw_nerf_refine): unit test, the suites it touches, the bench
# line over four batch seeds (parity of the trained point) beside NEUCONW_NERF_REFINE=0.
s7() {
OUT=gpurun_out/r05k; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 600 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_color_nerf.py tests/test_gpu_fullsize.py tests/test_gpu_render_only.py tests/test_gpu_trainer.py tests/test_gpu_voxel.py -m gpu -q --timeout 500 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for S in 1000 2000 3000 4000; do
  $T 200 python bench.py --seed $S --no-pmc --no-parity-mode > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?" >> $OUT/status
done
NEUCONW_NERF_REFINE=0 $T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_refine_off.json 2>/dev/null; echo "refine off rc $?" >> $OUT/status
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_elim.json 2>/dev/null; echo "elim rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head; grep -E "background NeRF at the" $OUT/tests.log
}

# Round-5 GPU session 8: the colour network's lin0 with [points | normals] as hi + lo pairs (third ring pass) on top of the background
# refinement: the suites it touches, the bench line over four batch seeds, the kernel table of the headline command.
s8() {
OUT=gpurun_out/r05l; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 700 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_color_nerf.py tests/test_gpu_fullsize.py tests/test_gpu_render_only.py tests/test_gpu_voxel.py tests/test_gpu_grid.py tests/test_gpu_render.py -m gpu -q --timeout 600 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for S in 1000 2000 3000 4000; do
  $T 200 python bench.py --seed $S --no-pmc --no-parity-mode > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?" >> $OUT/status
done
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_elim.json 2>/dev/null; echo "elim rc $?" >> $OUT/status
export TMPDIR=/tmp
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python bench.py --inner --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc $?" >> $OUT/status
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/prof
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head; grep -E "background NeRF at the" $OUT/tests.log
head -25 $OUT/kernel_stats.csv | cut -c1-200
}

# Round-5 GPU session 9: the three networks' pack / weight-norm backward as ONE launch each (packing.pack_many / unpack_many):
# the suites that train, and the headline line three times.
s9() {
OUT=gpurun_out/r05m; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 700 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fullsize.py tests/test_gpu_repro.py tests/test_gpu_train_driver.py tests/test_gpu_ddp.py tests/test_gpu_rccl_world1.py tests/test_gpu_render_only.py -m gpu -q --timeout 600 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for I in 1 2 3; do
  $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_$I.json 2>/dev/null; echo "bench $I rc $?" >> $OUT/status
done
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_elim.json 2>/dev/null; echo "elim rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05m/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']['per_step_kernel_ms']
    print(f, round(d['ms_per_step'],3), {k:r[k] for k in ('ncw_pack_weights','ncw_unpack_grads','ncw_nerf_refine','ncw_wgrad_tiled') if k in r}, d['roofline'].get('sum_kernel_ms_per_step'))
P
}

# Round-5 GPU session 10: where the step's wall time is not covered by kernels: kernel trace of the inner loop (union of the
# kernel intervals per step against the step's span), and the same step replayed as a HIP graph.
s10() {
OUT=gpurun_out/r05n; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
export TMPDIR=/tmp
$T 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python bench.py --inner --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc $?" >> $OUT/status
find $OUT/prof -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/prof
python scripts/diag/trace_gaps.py $OUT/kernel_trace.csv > $OUT/trace_gaps.log 2>&1; echo "gaps rc $?" >> $OUT/status
rm -f $OUT/kernel_trace.csv
$T 200 python bench.py --graph --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_graph.json 2>$OUT/graph.err; echo "graph rc $?" >> $OUT/status
$T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_eager.json 2>/dev/null; echo "eager rc $?" >> $OUT/status
cat $OUT/status; cat $OUT/trace_gaps.log | tail -60
python - <<'P'
import json
for f in ('bench_graph','bench_eager'):
    try:
        d=json.loads(open('gpurun_out/r05n/%s.json'%f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3))
    except Exception as e: print(f,'ERR',e)
P
tail -3 $OUT/graph.err
}

# Round-5 GPU session 11: the whole -m gpu suite with its printed measurements (tolerances tightened to the north-star bar where the
# round's numerics allow).
s11() {
OUT=gpurun_out/r05o; mkdir -p $OUT; rm -f $OUT/status
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -s > $OUT/tests_printed.log 2>&1; echo "tests rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests_printed.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests_printed.log | head; grep -E "single ray|508 \+ 4|samples per ray:" $OUT/tests_printed.log | cut -c1-250
}

# Round-5 GPU session 12: nerf_refineS with its trunk weights in rotating register buffers loaded a whole layer ahead (QU units per chunk x
# NB buffers): unit test, then the headline line per variant (probe libraries q44 = 4 x 4, q28 = 2 x 8, old = two halves) -- twice, alternating.
s12() {
OUT=gpurun_out/r05p; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 400 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_render_only.py -m gpu -q --timeout 300 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for I in 1 2; do
for V in main q44 q28 old; do
  if [ $V = main ]; then unset NEUCONW_HIP_LIB; else export NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_$V.so; fi
  $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_${V}_$I.json 2>/dev/null; echo "bench $V $I rc $?" >> $OUT/status
done
done
unset NEUCONW_HIP_LIB
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head; grep "background NeRF at" $OUT/tests.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05p/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']['per_step_kernel_ms']
    print(f, round(d['ms_per_step'],3), 'refine', r.get('ncw_nerf_refine'), 'trained colour', d['parity']['trained_40_steps_inv_s_403']['colour'] if d.get('parity') else None)
P
}

# Round-5 GPU session 13: nerf_refineS as a grid-stride kernel (at most one workgroup per CU): main = that + three rotating weight buffers;
# g10 = the same kernel launched with up to 10 workgroups per CU (the former full grid at the headline shape); old = grid-stride with the
s13() {
OUT=gpurun_out/r05q; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 400 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_render_only.py -m gpu -q --timeout 300 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for I in 1 2; do
for V in main g10 old; do
  if [ $V = main ]; then unset NEUCONW_HIP_LIB; else export NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_$V.so; fi
  $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_${V}_$I.json 2>/dev/null; echo "bench $V $I rc $?" >> $OUT/status
done
done
unset NEUCONW_HIP_LIB
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head; grep "background NeRF at" $OUT/tests.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05q/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']['per_step_kernel_ms']
    print(f, round(d['ms_per_step'],3), 'refine', r.get('ncw_nerf_refine'), 'trained colour', d['parity']['trained_40_steps_inv_s_403']['colour'] if d.get('parity') else None)
P
}

# Round-5 GPU session 14: SQ counters of nerf_refineS_kernel (what bounds its 0.09 ms: 26 us of MFMAs, 20 us of LDS reads on paper).
s14() {
OUT=gpurun_out/r05r; mkdir -p $OUT
timeout -k 10 600 python scripts/pmc_pass.py $OUT/pmc_refine.json "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT" > $OUT/pmc_refine.log 2>&1
grep -E "refine|nerf_fwdB|sdf_inferS2" $OUT/pmc_refine.log | cut -c1-900
}

# Round-5 GPU session 15: the merge of primary and outside depths (z_feed) launched on the background stream; the suites that render, the
# headline line three times.
s15() {
OUT=gpurun_out/r05s; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 700 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_render.py tests/test_gpu_render_only.py tests/test_gpu_fullsize.py tests/test_gpu_repro.py tests/test_gpu_trainer.py tests/test_gpu_voxel.py tests/test_gpu_edges.py -m gpu -q --timeout 600 > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for I in 1 2 3; do
  $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_$I.json 2>/dev/null; echo "bench $I rc $?" >> $OUT/status
done
NEUCONW_BG_STREAM=0 $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_one_stream.json 2>/dev/null; echo "one stream rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05s/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), d['roofline'].get('sum_kernel_ms_per_step'))
P
}

# Round-5 GPU session 16: does the SAMPLER need the split-precision SDF (0.275 ms per step against 0.12 in plain fp16), or only the final
# evaluation?  scripts/diag/sampler_split.py: the composed step against the fp64 oracle with the sampler's queries in plain fp16.
s16() {
OUT=gpurun_out/r05t; mkdir -p $OUT
timeout -k 10 600 python scripts/diag/sampler_split.py > $OUT/sampler_split.log 2>&1; echo "rc $?"
grep -E "^variance" $OUT/sampler_split.log
}

# Round-5 GPU session 17: six more ray batches (bench.py --seed) through the final kernels: the trained-weights parity point of each.
s17() {
OUT=gpurun_out/r05u; mkdir -p $OUT; rm -f $OUT/status
for S in 5000 6000 7000 8000 9000 10000; do
  timeout -k 10 200 python bench.py --seed $S --no-pmc --no-parity-mode > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?" >> $OUT/status
done
cat $OUT/status
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05u/bench_seed*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d['parity']; t=p['trained_40_steps_inv_s_403']; f32=p['f32_mode_trained_40_steps_inv_s_403']
    print(f.split('seed')[1][:-5], round(d['ms_per_step'],3), 'init %.2e/%.2e/%.2e'%(p['colour'],p['depth'],p['weights_sum']), 'trained colour %.2e p99 %.2e above %.4f depth %.2e ws %.2e | f32 mode colour %.2e depth %.2e | fixed_z colour %.2e'%(t['colour'],t['colour_p99'],t['colour_rays_above_1e-4'],t['depth'],t['weights_sum'],f32['colour'],f32['depth'],t['fixed_z']['colour']))
P
}

# Round-5 GPU session 18: the trained-weights parity point's state_dict of ray batches 5000 and 10000 (the two worst of ten) for the CPU emulation.
s18() {
OUT=gpurun_out/r05v; mkdir -p $OUT
for S in 5000 10000; do
  timeout -k 10 200 python bench.py --seed $S --no-pmc --no-parity-mode --save-trained-state $OUT/trained_seed$S.pt > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?"
done
ls -la $OUT
}

"s${1:?session number 1..18}"
