#!/bin/bash
# Round profile set (run on the GPU box through gpurun): rocprofv3 kernel stats of the bench command, the
# PMC passes (HBM traffic in separate passes, SQ counters) and the plain bench line; round 6: the same kernel trace with ONE stream
# (bench.py --one-stream: no overlap, the table reproduces `per_step_kernel_ms`).  Outputs: gpurun_out/<NCW_PROFILE_ROUND> (default r06)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${NCW_PROFILE_ROUND:-r06}; mkdir -p "$OUT"
TAG=${1:-v1}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- \
    python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-parity-mode > "$OUT/bench_${TAG}_under_rocprof.json" 2> /tmp/prof_stats.err
f=$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_${TAG}_kernel_stats.csv"
f=$(find /tmp/prof_stats -name '*domain_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_${TAG}_domain_stats.csv"
rm -rf /tmp/prof_stats1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats1 -o p -- \
    python "$REPO/bench.py" --one-stream --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-parity-mode > "$OUT/bench_onestream_${TAG}_under_rocprof.json" 2> /tmp/prof_stats1.err
f=$(find /tmp/prof_stats1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_onestream_${TAG}_kernel_stats.csv"
(cd "$REPO" && python bench.py --one-stream --no-cpu-baseline --no-pmc --no-parity-mode > "$OUT/bench_onestream_${TAG}.json" 2> /dev/null)
python "$REPO/scripts/pmc_pass.py" "$OUT/pmc_traffic_${TAG}.json" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/pmc_traffic_${TAG}.log" 2>&1
python "$REPO/scripts/pmc_pass.py" "$OUT/pmc_${TAG}.json" \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
    "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
    "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" > "$OUT/pmc_${TAG}.log" 2>&1
cd "$REPO" && python bench.py > "$OUT/bench_${TAG}.json" 2> /dev/null
tail -c 400 "$OUT/bench_${TAG}.json"; echo; ls -la "$OUT"
