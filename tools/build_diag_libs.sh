#!/bin/bash
# Rebuilds the three DIAGNOSTIC builds of libdss2_hip.so (phase stamps compiled in) from the sources as they are NOW:
#   tools/diag_lib/libdss2_cstamps.so  -DDSS2_CHAIN_STAMPS   (layer chains: tools/stamps.py chain)
#   tools/diag_lib/libdss2_hstamps.so  -DDSS2_STAMPS         (weight gradient / gemm_prop: tools/stamps.py wgradh, gemm, ...)
#   tools/diag_lib/libdss2_sstamps.so  -DDSS2_STACK_STAMPS   (whole-stack kernels: tools/stamps.py stack)
# They are not product code (git-ignored, never loaded unless DSS2_LIB points at one).  The hash of the compiled sources is left in
# tools/diag_lib/SOURCES.sha256; tools/stamps.py refuses a diagnostic library whose sources have changed since (stamp
# profiles must come from HEAD); `DSS2_BUILD_DIAG=1 python
# __graft_entry__.py` runs this script after the product build.  ~1 minute each on 8 cores.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CSRC="$ROOT/deep-statistical-solver-for-distribution-system-state-estimation_amd/csrc"
mkdir -p "$ROOT/tools/diag_lib"
for spec in "cstamps:-DDSS2_CHAIN_STAMPS" "hstamps:-DDSS2_STAMPS" "sstamps:-DDSS2_STACK_STAMPS"; do
  name="${spec%%:*}"; flag="${spec##*:}"
  DSS2_OUT="$ROOT/tools/diag_lib/libdss2_$name.so" DSS2_OBJ="/tmp/dss2_obj_$name" bash "$CSRC/build.sh" "$flag"
done
python3 - "$ROOT" <<'PY'
import glob, hashlib, os, sys
root = sys.argv[1]
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "deep-*", "csrc", "*.h*")) + glob.glob(os.path.join(root, "include", "*.h"))):
    h.update(open(f, "rb").read())
open(os.path.join(root, "tools", "diag_lib", "SOURCES.sha256"), "w").write(h.hexdigest() + "\n")
PY
echo "diagnostic libraries rebuilt"
