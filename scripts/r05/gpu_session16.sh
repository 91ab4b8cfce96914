#!/bin/bash
# Round-5 GPU session 16: does the SAMPLER need the split-precision SDF (0.275 ms per step against 0.12 in plain fp16), or only the final
# evaluation?  scripts/diag/sampler_split.py: the composed step against the fp64 oracle with the sampler's queries in plain fp16.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05t; mkdir -p $OUT
timeout -k 10 600 python scripts/diag/sampler_split.py > $OUT/sampler_split.log 2>&1; echo "rc $?"
grep -E "^variance" $OUT/sampler_split.log
