"""MI355X-native (gfx950) implementation of the DSS2 message-passing + WLS-loss hot path.

Drop-in modules ``networks`` (EdgeAggregation, TAGConv, MPN, SkipMPN, PFN, SkipPFN) and ``data``
(gsp_wls_edge, get_pflow) mirror /root/reference/networks.py and /root/reference/data.py for the
path BASELINE.json names; everything below them is hand-written HIP in libdss2_hip.so.
"""
from . import _lib, flags, synthetic, topology  # noqa: F401
from . import ops, plans  # noqa: F401
from . import networks, data, parallel, graphs, optim, dataset  # noqa: F401
from .optim import FusedAdamax  # noqa: F401
from . import runner, multi  # noqa: F401
from .multi import MaskEmbdMPN, MultiMPN, MaskEmbdMultiMPN, MaskEmbdMultiMPN_NoMP, EdgeAggregationGeneral  # noqa: F401
from .networks import EdgeAggregation, TAGConv, MPN, SkipMPN, PFN, SkipPFN, MessagePassing  # noqa: F401
from .data import gsp_wls_edge, get_pflow  # noqa: F401
from .dataset import data_from_pickles, DataLoader, DeviceDataset, MixedDataset  # noqa: F401

__all__ = ["MaskEmbdMPN", "MultiMPN", "MaskEmbdMultiMPN", "MaskEmbdMultiMPN_NoMP", "EdgeAggregationGeneral", "EdgeAggregation", "TAGConv", "MPN", "SkipMPN", "PFN", "SkipPFN", "MessagePassing",
           "gsp_wls_edge", "get_pflow", "data_from_pickles", "DataLoader", "DeviceDataset", "MixedDataset", "FusedAdamax", "dataset", "networks", "data", "parallel", "graphs", "optim", "synthetic",
           "topology", "flags", "ops", "plans"]
