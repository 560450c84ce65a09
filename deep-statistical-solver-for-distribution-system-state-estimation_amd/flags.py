"""Module-level switches of the hot path, in ONE place (environment defaults).

``ops.py``, ``plans.py`` and ``networks.py`` read them at call time as ``flags.<NAME>``, so a test or a tool flips a code path
inside one process with ``pkg.flags.<NAME> = value`` (and restores it afterwards)."""
import os as _os

EDGE_TILE_KERNELS = _os.environ.get("DSS2_EDGE_TILE", "1") == "1"       # 0 = row-per-wave CSR kernels
CHAIN_LAYERS = _os.environ.get("DSS2_CHAIN", "1") == "1"               # hid->hid layers of a block: one chained launch
WGRAD_BF16 = _os.environ.get("DSS2_WGRAD_BF16", "1") == "1"            # weight gradients as bf16x6 where the kernel covers the shape
CHAIN_GATE_BITS = _os.environ.get("DSS2_CHAIN_GATE_BITS", "1") == "1"  # split-plane chains (64-, 96-, 192-row tiles): the backward chain's ReLU gates as bit words written by the forward chain
CHAIN_BF16 = _os.environ.get("DSS2_CHAIN_BF16", "1") == "1"            # its tile GEMM as bf16x6 on the bf16 matrix pipe (fp32-accurate)
# ... as f16x3 where the split-plane chain of 64-row tiles has the form (round 5: two fp16 pieces per operand after an exact power-of-two
# scale, three MFMAs per product instead of six; csrc/dss2_gemm_chain_sp.hip MS = 2).  0 = bf16x6.
CHAIN_F16 = _os.environ.get("DSS2_CHAIN_F16", "1") == "1"
CHAIN_HEAD = _os.environ.get("DSS2_CHAIN_HEAD", "1") == "1"            # the narrow head TAGConv's data gradient inside the chained launch of the data gradients
# ... and the head's forward inside the forward chain (round 2: break-even, off; with the 16x16x32 chain of round 4 the fused step is
# 4 us shorter; round 5: on, and bench.py counts the head's FLOPs in the launch it rides in).  DSS2_CHAIN_HEAD_FWD=0: its own launch.
CHAIN_HEAD_WGRAD = _os.environ.get("DSS2_CHAIN_HEAD_WGRAD", "1") == "1"   # ... and the head's weight gradient in the same staging (round 5; 0 = its own launch)
CHAIN_HEAD_FWD = _os.environ.get("DSS2_CHAIN_HEAD_FWD", "1") == "1"
WGRAD_BATCH = _os.environ.get("DSS2_WGRAD_BATCH", "1") == "1"
STACK_NODE = _os.environ.get("DSS2_STACK_NODE", "1") == "1"              # PFN / SkipPFN as ONE autograd node (_PFNFn)
DX_MERGE = _os.environ.get("DSS2_DX_MERGE", "1") == "1"                  # dx of the edge MLP as ONE K = 2 hid GEMM
WGRAD_JOIN_FOLDED = None    # None / True: the folded conv 0 rides in the batched weight-gradient launch of the plain layers; False: its own launch
FOLD_W2 = _os.environ.get("DSS2_FOLD_W2", "1") == "1"   # 0 = run the edge MLP's second Linear as its own GEMMs
# ... and on 32-row tiles as f16x3: two fp16 pieces per operand after a power-of-two scale, three MFMAs per product instead of six
# (csrc/dss2_wgrad16h.hip, round 5; errors of the size of fp32 arithmetic itself).  0 = bf16x6 there too.
WGRAD_F16 = _os.environ.get("DSS2_WGRAD_F16", "1") == "1"
# 1 = the loss's batch sums are finished by the last workgroup of the partials launch at every batch size (no finish launch; bitwise the
# same sums).  Off above 16 workgroups: at C2 the 240 arrivals on one counter word (~12 ns each) and the last workgroup's round trip cost
# what the 4.65 us finish launch costs -- A/B on one box 0.4047 / 0.4017 ms fused against 0.4022 / 0.3979 (end of round 5)
WLS_FUSED_FINISH = _os.environ.get("DSS2_WLS_FUSED_FINISH", "0") == "1"
# 0 = the two-launch finish also up to 16 workgroups (where the fused finish is the default; tests compare the two)
WLS_FUSED_FINISH_SMALL = _os.environ.get("DSS2_WLS_FUSED_FINISH_SMALL", "1") == "1"
WGRAD_TM32 = _os.environ.get("DSS2_WGRAD_TM32", "1") == "1"      # bf16x6 weight gradient on 32-row tiles, two workgroups per CU (wgrad16b_kernel)
WGRAD_TM32_MAX_BYTES = 64 << 20      # ... as bf16x6 (DSS2_WGRAD_F16=0) only while one layer's input (N * hin * 4 bytes) stays well inside the Infinity Cache; the f16x3 kernel has no such limit
WGRAD_PER_CU = int(_os.environ.get("DSS2_WGRAD_PER_CU", "2"))   # cap on persistent wgrad workgroups per CU (= slabs / 256)
CHAIN_MAX = 8      # layers per dss2_gemm_prop_chain launch (csrc/dss2_gemm_chain.hip)
