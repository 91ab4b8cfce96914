#!/bin/bash
# Round-5 GPU session 18: the trained-weights parity point's state_dict of ray batches 5000 and 10000 (the two worst of ten) for the CPU emulation.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05v; mkdir -p $OUT
for S in 5000 10000; do
  timeout -k 10 200 python bench.py --seed $S --no-pmc --no-parity-mode --save-trained-state $OUT/trained_seed$S.pt > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?"
done
ls -la $OUT
