"""GPU: the environment switches of DESIGN.md section 4.6 select other kernels / schedules for the same mathematics.  Every
non-default setting that ships is exercised here by running parity tests in a child interpreter with the switches set (they
are read at import or on first use, so a child process is the honest way to flip them).  Each variant runs the golden cases
of the reference plus the tests whose kernels it changes -- not the whole suites (round 3 spent 650 s of the 1 200 s GPU test
budget re-running everything six times)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARITY, ROUND2 = os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_round2.py")
C1, C2, C3, C5S = "grids0-64-32-1", "grids1-4096-128-4", "grids2-1024-128-4", "grids3-512-256-8"

# name -> (switches, [(test file, -k expression)])
FALLBACKS = {
    # fp32 MFMA kernels instead of the bf16x6 ones (tile GEMM of the layer chain, weight gradient, edge MLP)
    "fp32-mfma": (dict(DSS2_CHAIN_BF16="0", DSS2_WGRAD_BF16="0", DSS2_EDGE_BF16="0"),
                  [(PARITY, f"golden or tagconv or {C1} or {C2} or {C3}")]),
    # one autograd node per block, separate dx GEMMs, no K split, one launch per layer instead of the layer chain,
    # one weight-gradient launch per layer, un-folded second edge-MLP layer
    "unfused": (dict(DSS2_STACK_NODE="0", DSS2_DX_MERGE="0", DSS2_WGRAD_KSPLIT="0", DSS2_CHAIN="0", DSS2_WGRAD_BATCH="0",
                     DSS2_FOLD_W2="0"),
                [(PARITY, f"golden or hipgraph or runner or {C1} or {C2}"), (ROUND2, "")]),
    # the fp32-tile form of the bf16x6 chain instead of the split-plane form, the narrow head's data gradient as its own launch
    "chain-variants-a": (dict(DSS2_CHAIN_SP="0", DSS2_CHAIN_HEAD="0"),
                         [(PARITY, f"golden or {C2} or {C3} or {C5S}")]),
    # the head's FORWARD as a launch of its own (inside the chained launch since round 5); ReLU gates read from the activations instead
    # of the forward chain's bit words
    "chain-variants-b": (dict(DSS2_CHAIN_HEAD_FWD="0", DSS2_CHAIN_GATE_BITS="0"),
                         [(PARITY, f"golden or {C2} or {C3} or tall_tiles")]),
    # the generic narrow kernels, scalar-VALU edge MLP, edges through the global CSR instead of tile-local lists; the 64-row chain's tile
    # GEMM on 32x32x16 MFMAs instead of 16x16x32
    "generic-narrow-edge": (dict(DSS2_NARROW_STREAM="0", DSS2_EDGE_MFMA="0", DSS2_EDGE_TILE="0", DSS2_CHAIN_MFMA16="0"),
                            [(PARITY, f"golden or edge_aggregation or tagconv or {C2}"), (ROUND2, "propagate or message or known_answers")]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FALLBACKS))
def test_parity_holds_with_the_fallback_switches(name):
    switches, runs = FALLBACKS[name]
    env = dict(os.environ, **switches)
    for path, expr in runs:
        cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", path]
        if expr:
            cmd += ["-k", expr]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, f"{name}: {switches} on {os.path.basename(path)} -k '{expr}'\n{r.stdout[-3000:]}\n{r.stderr[-1500:]}"
        assert " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-500:]
