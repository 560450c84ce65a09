"""GPU: the environment switches of DESIGN.md section 4.6 select other kernels / schedules for the same mathematics.  Every
non-default setting is exercised here by running the parity tests in a child interpreter with the switches set (they are
read at import or on first use, so a child process is the honest way to flip them)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FALLBACKS = {
    # fp32 MFMA kernels instead of the bf16x6 ones (tile GEMM of the layer chain, weight gradient)
    "fp32-mfma": dict(DSS2_CHAIN_BF16="0", DSS2_WGRAD_BF16="0"),
    # one autograd node per block, separate dx GEMMs, no K split, one launch per layer instead of the layer chain,
    # one weight-gradient launch per layer, un-folded second edge-MLP layer
    "unfused": dict(DSS2_STACK_NODE="0", DSS2_DX_MERGE="0", DSS2_WGRAD_KSPLIT="0", DSS2_CHAIN="0", DSS2_WGRAD_BATCH="0",
                    DSS2_FOLD_W2="0"),
    # round-3 chain variants: the fp32-tile form of the bf16x6 chain instead of the split-plane form, the narrow head's data
    # gradient as its own launch; and the head's FORWARD inside the chained launch (off by default)
    "chain-variants-a": dict(DSS2_CHAIN_SP="0", DSS2_CHAIN_HEAD="0"),
    # tall tiles: the generic chain kernel instead of the per-direction ones, the four-wave weight gradient at 192 rows
    "chain-variants-c": dict(DSS2_CHAIN_SP6_DIR="0", DSS2_WGRAD_W8="0"),
    # ... plus: 96-row tiles on the one-workgroup-per-CU chain (dss2_gemm_chain_sp3.hip), tall-tile ReLU gates read from the
    # activations instead of the forward chain's bit words
    "chain-variants-b": dict(DSS2_CHAIN_HEAD_FWD="1", DSS2_CHAIN_SP3B="0", DSS2_CHAIN_GATE_BITS="0"),
    # the generic narrow kernels, scalar-VALU edge MLP, edges through the global CSR instead of tile-local lists
    "generic-narrow-edge": dict(DSS2_NARROW_STREAM="0", DSS2_EDGE_MFMA="0", DSS2_EDGE_TILE="0"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FALLBACKS))
def test_parity_holds_with_the_fallback_switches(name):
    env = dict(os.environ, **FALLBACKS[name])
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_round2.py")],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, f"{name}: {FALLBACKS[name]}\n{r.stdout[-3000:]}\n{r.stderr[-1500:]}"
