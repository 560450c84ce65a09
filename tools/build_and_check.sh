#!/bin/bash
# Build the library and succeed only if it really built (used to gate GPU runs).
set -o pipefail
OUT=$(/root/repo/deep-statistical-solver-for-distribution-system-state-estimation_amd/csrc/build.sh 2>&1)
if echo "$OUT" | grep -q "^built " && ! echo "$OUT" | grep -q "error"; then echo "BUILD OK"; exit 0; fi
echo "$OUT" | grep -E "error" | head -10; echo "BUILD FAILED"; exit 1
