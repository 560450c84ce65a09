#!/bin/bash
# Diagnostic builds of the f16x3 weight gradient (csrc/dss2_wgrad16h.hip) with one phase taken out each -- how much of the launch is the
# split of X (what "X as ready-made planes" would save at most), the reads of X (what "X read once" is bounded by), the propagation hops, the
# matrix instructions.  Builds tools/diag_lib/libdss2_wg_<name>.so from the product's object files with that one translation unit recompiled;
# run on the GPU box:  for l in tools/diag_lib/libdss2_wg_*.so; do PMC_TIME=1 DSS2_LIB=$l python tools/pmc_one.py wgrad3 400; done
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CSRC="$ROOT/deep-statistical-solver-for-distribution-system-state-estimation_amd/csrc"
bash "$CSRC/build.sh" > /dev/null
for spec in "xsplit:-DDSS2_ABLATE_XSPLIT" "xload:-DDSS2_ABLATE_XLOAD" "xboth:-DDSS2_ABLATE_XSPLIT -DDSS2_ABLATE_XLOAD" "hops:-DDSS2_ABLATE_HOPS" "mfma:-DDSS2_ABLATE_MFMA"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  obj="/tmp/dss2_obj_wg_$name"; rm -rf "$obj"; cp -r "$CSRC/obj" "$obj"; rm -f "$obj/dss2_wgrad16h.o"
  DSS2_OUT="$ROOT/tools/diag_lib/libdss2_wg_$name.so" DSS2_OBJ="$obj" bash "$CSRC/build.sh" $flags > /dev/null
  echo "built libdss2_wg_$name.so"
done
