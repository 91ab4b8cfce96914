#!/bin/bash
# Round-6 GPU sessions, one function per gpurun call:   gpurun --timeout T -- 'bash scripts/r06/gpu_sessions.sh <name>'
# Every session writes under gpurun_out/r06_<name>/ ; what is judged is copied into profiles/r06/ afterwards.
set -u
cd "$(dirname "$0")/../.."
NAME=${1:?session name}
OUT=gpurun_out/r06_$NAME; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-parity-mode --no-cpu-baseline"

s1() {  # trained-weights parity point of the SHIPPED shape for the CPU emulation; config 4's whole batch on one GPU; training equivalence
  for S in 1000 2000 3000; do
    timeout -k 10 300 $B --config shipped --seed $S --save-trained-state $OUT/trained_shipped_seed$S.pt > $OUT/bench_shipped_seed$S.json 2>$OUT/bench_shipped_seed$S.err; echo "shipped seed $S rc $?"
  done
  timeout -k 10 400 python bench.py --rays 8192 --no-pmc > $OUT/bench_rays8192.json 2>$OUT/bench_rays8192.err; echo "rays 8192 rc $?"
  for N in 1500 3000; do
    timeout -k 10 600 python scripts/diag/train_equivalence.py --steps $N --precs f32,f32b,f16,f16_noextras --out $OUT/train_equivalence_$N.json > $OUT/train_equivalence_$N.log 2>&1; echo "train_equivalence $N rc $?"
  done
}

s2() {  # (s1 skipped the parity legs with --no-cpu-baseline: the trained state is written by the parity leg)
  for S in 1000 2000 3000; do
    timeout -k 10 400 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S --save-trained-state $OUT/trained_shipped_seed$S.pt > $OUT/bench_shipped_seed$S.json 2>$OUT/bench_shipped_seed$S.err; echo "shipped seed $S rc $?"
  done
}

s3() {  # s2 again on a consistent build + the new voxel / kaolin-boundary tests
  s2
  timeout -k 10 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_octree_refresh.py tests/test_gpu_mesh.py tests/test_gpu_trainer.py tests/test_compat_kaolin.py tests/test_octree_from_sfm.py -x -q -s -m "gpu or not gpu" > $OUT/voxel_tests.log 2>&1; echo "voxel tests rc $?"
  tail -5 $OUT/voxel_tests.log
}

s4() {  # W = 512: adjoint sweep with both operands as pairs + colour activations as pairs: unit tests, the shipped shape's parity / cost
  timeout -k 10 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_sdf.py tests/test_gpu_color_nerf.py -x -q -s > $OUT/unit_tests.log 2>&1; echo "unit tests rc $?"; tail -3 $OUT/unit_tests.log
  timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_render_only.py -x -q -s > $OUT/fullsize_tests.log 2>&1; echo "fullsize tests rc $?"; tail -3 $OUT/fullsize_tests.log
  for S in 1000 3000 4000 5000; do
    timeout -k 10 400 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S > $OUT/bench_shipped_seed$S.json 2>$OUT/bench_shipped_seed$S.err; echo "shipped seed $S rc $?"
  done
  NEUCONW_SDF_ADJ_SPLIT=0 NEUCONW_COLOR_ASPLIT=0 timeout -k 10 400 $B --config shipped > $OUT/bench_shipped_r5_kernels.json 2>/dev/null; echo "shipped, round-5 kernels rc $?"
  NEUCONW_SDF_ADJ_SPLIT=2 NEUCONW_COLOR_ASPLIT=0 timeout -k 10 400 $B --config shipped > $OUT/bench_shipped_adj2_only.json 2>/dev/null; echo "shipped, adj 2 only rc $?"
  NEUCONW_SDF_ADJ_SPLIT=0 NEUCONW_COLOR_ASPLIT=1 timeout -k 10 400 $B --config shipped > $OUT/bench_shipped_asplit_only.json 2>/dev/null; echo "shipped, act_split only rc $?"
  NEUCONW_COLOR_ASPLIT=1 timeout -k 10 400 python bench.py --no-pmc --no-parity-mode > $OUT/bench_headline_asplit.json 2>/dev/null; echo "headline + act_split rc $?"
  timeout -k 10 400 $B > $OUT/bench_headline.json 2>/dev/null; echo "headline rc $?"
}

s5() {  # t-units in the value-only SDF kernels: correctness (every sdf / sampler test), time and counters; the fixed tests of s4
  timeout -k 10 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_sdf.py tests/test_gpu_grid.py tests/test_gpu_rays.py tests/test_gpu_octree_refresh.py -x -q -s > $OUT/unit_tests.log 2>&1; echo "unit tests rc $?"; tail -3 $OUT/unit_tests.log
  timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_render.py tests/test_gpu_f16.py tests/test_gpu_edges.py -x -q -s > $OUT/fullsize_tests.log 2>&1; echo "fullsize tests rc $?"; tail -3 $OUT/fullsize_tests.log
  timeout -k 10 600 python scripts/diag/sdf_infer_units.py --pmc > $OUT/sdf_infer_units.log 2>&1; echo "sdf_infer_units rc $?"; cat $OUT/sdf_infer_units.log
  timeout -k 10 400 $B > $OUT/bench_headline.json 2>/dev/null; echo "headline rc $?"
  timeout -k 10 400 $B --config grid512 > $OUT/bench_grid512.json 2>/dev/null; echo "grid512 rc $?"
}

s6() {  # t-units: infer vs the training forward on the same points; the test files (no -x: every failure at once)
  timeout -k 10 300 python scripts/diag/tunits_check.py > $OUT/tunits_check.log 2>&1; echo "tunits_check rc $?"; cat $OUT/tunits_check.log
  timeout -k 10 1200 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_sdf.py tests/test_gpu_grid.py tests/test_gpu_rays.py tests/test_gpu_octree_refresh.py tests/test_gpu_fullsize.py tests/test_gpu_render.py tests/test_gpu_f16.py tests/test_gpu_edges.py tests/test_gpu_color_nerf.py -q -s > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -12 $OUT/tests.log
}

s7() {
  timeout -k 10 300 python scripts/diag/tunits_bisect.py > $OUT/tunits_bisect.log 2>&1; echo "tunits_bisect rc $?"; cat $OUT/tunits_bisect.log
}

s8() {  # after pinning the hi / lo conversions (ncw_split8): the bisect again, the t-units check, every affected test file, timings
  s7
  timeout -k 10 300 python scripts/diag/tunits_check.py > $OUT/tunits_check.log 2>&1; echo "tunits_check rc $?"; grep "f16 split" $OUT/tunits_check.log
  timeout -k 10 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -8 $OUT/tests.log
  timeout -k 10 600 python scripts/diag/sdf_infer_units.py --pmc > $OUT/sdf_infer_units.log 2>&1; echo "sdf_infer_units rc $?"
  timeout -k 10 400 python bench.py --no-pmc --no-parity-mode > $OUT/bench_headline.json 2>/dev/null; echo "headline rc $?"
  timeout -k 10 400 python bench.py --no-pmc --no-parity-mode --config shipped > $OUT/bench_shipped.json 2>/dev/null; echo "shipped rc $?"
}

s9() {  # pair-wise pinned conversions, DDA exit depths from the voxel index: whole GPU suite, timings, benches
  timeout -k 10 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -6 $OUT/tests.log
  timeout -k 10 300 python scripts/diag/tunits_check.py > $OUT/tunits_check.log 2>&1; echo "tunits_check rc $?"; grep "f16 split" $OUT/tunits_check.log
  timeout -k 10 600 python scripts/diag/sdf_infer_units.py --pmc > $OUT/sdf_infer_units.log 2>&1; echo "sdf_infer_units rc $?"; cat $OUT/sdf_infer_units.log
  timeout -k 10 400 python bench.py --no-pmc --no-parity-mode > $OUT/bench_headline.json 2>/dev/null; echo "headline rc $?"
  timeout -k 10 400 python bench.py --no-pmc --no-parity-mode --config shipped > $OUT/bench_shipped.json 2>/dev/null; echo "shipped rc $?"
}

s10() {  # item 3: two blocks per wave with hand-written loads into AGPRs (probe library `asm`); item 5: the timed program's parity under pytest
  NEUCONW_HIP_LIB=neuralrecon-w_amd/libneuconw_hip_asm.so timeout -k 10 400 python scripts/diag/pp_nb2.py > $OUT/pp_asm.log 2>&1; echo "pp_nb2 (asm lib) rc $?"; cat $OUT/pp_asm.log
  NCW_PP_NB=2 NEUCONW_HIP_LIB=neuralrecon-w_amd/libneuconw_hip_asm.so timeout -k 10 600 python scripts/diag/sdf_infer_units.py --pmc > $OUT/sdf_infer_units_asm_nb2.log 2>&1; echo "units asm nb2 rc $?"; grep -A1 "per launch" $OUT/sdf_infer_units_asm_nb2.log
  timeout -k 10 900 python -m pytest tests/test_gpu_timed_program_parity.py -q -s -p no:cacheprovider > $OUT/parity_test.log 2>&1; echo "parity test rc $?"; grep "|" $OUT/parity_test.log | cut -c1-330; tail -3 $OUT/parity_test.log
}

s11() {  # one code path for the prefetch segment (no hoisted v_exp burst): product timings + counters; the asm probe's second (last) session
  timeout -k 10 600 python scripts/diag/sdf_infer_units.py --pmc > $OUT/sdf_infer_units.log 2>&1; echo "sdf_infer_units rc $?"; cat $OUT/sdf_infer_units.log
  NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_asm.so timeout -k 10 400 python scripts/diag/pp_nb2.py > $OUT/pp_asm.log 2>&1; echo "pp_nb2 (asm lib) rc $?"; cat $OUT/pp_asm.log
  NCW_PP_NB=2 NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_asm.so timeout -k 10 600 python scripts/diag/sdf_infer_units.py --pmc > $OUT/sdf_infer_units_asm_nb2.log 2>&1; echo "units asm nb2 rc $?"; grep -A1 "per launch" $OUT/sdf_infer_units_asm_nb2.log
  timeout -k 10 900 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_rays.py tests/test_gpu_grid.py -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log
  timeout -k 10 400 $B > $OUT/bench_headline.json 2>/dev/null; echo "headline rc $?"
  timeout -k 10 400 $B --config grid512 --grid-width 256 > $OUT/bench_grid512_w256.json 2>/dev/null; echo "grid512 w256 rc $?"
}

s12() {  # colour activations as pairs in two ring passes instead of three: tests, cost at the shipped shape and at the headline's
  timeout -k 10 900 python -m pytest tests/test_gpu_color_nerf.py tests/test_gpu_fullsize.py tests/test_gpu_render_only.py tests/test_gpu_timed_program_parity.py -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log
  for S in 1000 3000; do
    timeout -k 10 400 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S > $OUT/bench_shipped_seed$S.json 2>/dev/null; echo "shipped seed $S rc $?"
  done
  NEUCONW_COLOR_ASPLIT=1 timeout -k 10 400 $B > $OUT/bench_headline_asplit.json 2>/dev/null; echo "headline + act_split rc $?"
  timeout -k 10 400 $B > $OUT/bench_headline.json 2>/dev/null; echo "headline rc $?"
}

s13() {  # where do the W = 512 kernels' weights come from?  L2 hit / miss and memory-side requests of the shipped shape's kernels
  (cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -E "TCC_|TCP_" | cut -c1-160 | sort -u | head -150) > $OUT/counters_avail.log 2>&1
  NCW_PMC_BENCH_ARGS="--config shipped" timeout -k 10 900 python scripts/pmc_pass.py $OUT/pmc_shipped_l2.json "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_TAG_STALL_sum TCC_BUBBLE_sum" > $OUT/pmc_shipped_l2.log 2>&1; echo "pmc shipped rc $?"
  grep -E "sdf_fwdS16|sdf_inferS16|sdf_bwd16|group failed" $OUT/pmc_shipped_l2.log | cut -c1-400
}

s14() {  # W = 512 split kernels: does pinning the weight ring's loads / a deeper ring help?  (probe libraries pin, d8, pind8)
  for rep in 1 2; do
    for L in "" pin d8 pind8; do
      if [ -z "$L" ]; then timeout -k 10 200 python scripts/diag/w512_probe.py; else NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_$L.so timeout -k 10 200 python scripts/diag/w512_probe.py; fi
    done
  done 2>&1 | grep -v amdgpu.ids | tee $OUT/w512_probe.log
}

s15() {  # headline shape, ten ray batches: trained-weights colour and step time with the colour activations as pairs (act_split) and without
  for S in 1000 2000 3000 4000 5000 6000 7000 8000 9000 10000; do
    for A in 0 1; do
      NEUCONW_COLOR_ASPLIT=$A timeout -k 10 300 python bench.py --no-pmc --no-parity-mode --seed $S > $OUT/bench_seed${S}_asplit$A.json 2>/dev/null; echo "seed $S asplit $A rc $?"
    done
  done
}

s16() {  # W = 512 split value chain: k-units walked in an order rotated per workgroup (probe `rot`): is the L2 limited by every CU asking for the same lines at once?
  for rep in 1 2; do
    timeout -k 10 200 python scripts/diag/w512_probe.py
    NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_rot.so timeout -k 10 200 python scripts/diag/w512_probe.py
  done 2>&1 | grep -v amdgpu.ids | tee $OUT/w512_rot.log
}

s17() {  # nerf_bwdB with its ReLU masks prefetched a layer ahead: tests, bench (per-kernel times)
  timeout -k 10 900 python -m pytest tests/test_gpu_color_nerf.py tests/test_gpu_bg_select.py tests/test_gpu_render.py -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log
  for i in 1 2; do timeout -k 10 400 $B > $OUT/bench_headline_$i.json 2>/dev/null; echo "headline rc $?"; done
  timeout -k 10 400 $B --bg-eliminate > $OUT/bench_elim.json 2>/dev/null; echo "elim rc $?"
}

s18() {  # kernel time (rocprofv3) of the value-only kernels, to set beside round 5's 0.157 ms / 131,072 points
  cd /tmp && rm -rf /tmp/units_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/units_stats -o p -- python $GRAFT_REPO_ROOT/scripts/diag/sdf_infer_units.py > $GRAFT_REPO_ROOT/$OUT/units_under_rocprof.log 2>&1; head -1 $(find /tmp/units_stats -name '*kernel_trace.csv' | head -1) > $GRAFT_REPO_ROOT/$OUT/trace_header.txt
  cd $GRAFT_REPO_ROOT; f=$(find /tmp/units_stats -name '*kernel_trace.csv' | head -1); python - "$f" > $OUT/sdf_infer_kernel_times.log <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "sdf_infer" in k:
        g = r.get("Grid_Size") or r.get("Grid_Size_X") or r.get("Grid_Size_x") or "0"
        import re
        m = re.search(r"(sdf_infer\w*?_kernel(?:<[^>]*>)?)", k)
        acc[(m.group(1) if m else k[:40], int(g))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for (k, g), v in sorted(acc.items()):
    v.sort()
    print("%-42s grid %9d  launches %3d  median %.4f ms  min %.4f ms" % (k, g, len(v), v[len(v) // 2], v[0]))
PY
  cat $OUT/sdf_infer_kernel_times.log
}

s19() {  # the shipped shape's trained-weights parity point over ten ray batches (final kernels); the whole GPU suite once more
  for S in 1000 2000 3000 4000 5000 6000 7000 8000 9000 10000; do
    timeout -k 10 300 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S > $OUT/bench_shipped_seed$S.json 2>/dev/null; echo "shipped seed $S rc $?"
  done
  timeout -k 10 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log
}

s20() {  # trained states of the shipped shape's outlier batches for the CPU emulation
  for S in 7000 10000 6000; do
    timeout -k 10 300 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S --save-trained-state $OUT/trained_shipped_seed$S.pt > $OUT/bench_shipped_seed$S.json 2>/dev/null; echo "shipped seed $S rc $?"
  done
}

s21() {  # phi' of the W = 512 adjoint sweep from h as a hi + lo pair (residual stash): tests, the outlier ray batches, cost
  timeout -k 10 900 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_sdf_train.py tests/test_gpu_fullsize.py tests/test_gpu_render_only.py tests/test_gpu_grid.py -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log; grep "W=512 normals" $OUT/tests.log
  for S in 7000 1000 5000 10000; do
    timeout -k 10 300 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S > $OUT/bench_shipped_seed$S.json 2>/dev/null; echo "shipped seed $S rc $?"
  done
}

s22() {  # whole GPU suite on the residual-stash build; shipped ten ray batches again
  timeout -k 10 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log
  for S in 2000 3000 4000 6000 8000 9000; do
    timeout -k 10 300 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S > $OUT/bench_shipped_seed$S.json 2>/dev/null; echo "shipped seed $S rc $?"
  done
}

s23() {  # training equivalence at the SHIPPED shape (W = 512, 8 + 16, 2048 rays per step): fp32 twice, fp16 with / without the forward-only refinements
  for N in 1500 3000; do
    timeout -k 10 1200 python scripts/diag/train_equivalence.py --shipped --rays 2048 --steps $N --precs f32,f32b,f16,f16_noextras --out $OUT/train_equivalence_shipped_$N.json > $OUT/train_equivalence_shipped_$N.log 2>&1; echo "train_equivalence shipped $N rc $?"; tail -4 $OUT/train_equivalence_shipped_$N.log
  done
}

"$NAME"
ls -la $OUT
