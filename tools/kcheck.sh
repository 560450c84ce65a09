#!/bin/bash
# Compile ONE csrc/*.hip translation unit for gfx950 into /tmp and print every kernel's register / spill / LDS / scratch notes.
#   tools/kcheck.sh dss2_wgrad16h [extra hipcc flags]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$ROOT/deep-statistical-solver-for-distribution-system-state-estimation_amd/csrc"
src="$1"; shift || true
out="/tmp/kcheck_$src"; rm -rf "$out"; mkdir -p "$out"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I"$ROOT/include" -I"$CS" -Wall -Wno-unused-function \
  -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c "$CS/$src.hip" -o "$out/x.o" 2> >(grep -v "is not a recognized feature" >&2)
(cd "$out" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null)
co="$(ls "$out"/x.o.*gfx950* | head -1)"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | python3 "$ROOT/tools/kcheck_notes.py"
echo "code object: $co"
