import inspect

import torch
import torch.nn as nn

from ..utils import scatter, degree


class MessagePassing(nn.Module):
    """aggr='add', flow='source_to_target', node_dim=0 only."""

    def __init__(self, aggr="add", **kwargs):
        super().__init__()
        assert aggr == "add"
        self.aggr = aggr

    def propagate(self, edge_index, size=None, **kwargs):
        params = list(inspect.signature(self.message).parameters)
        args = {}
        for name in params:
            if name.endswith("_j"):
                args[name] = kwargs[name[:-2]][edge_index[0]]
            elif name.endswith("_i"):
                args[name] = kwargs[name[:-2]][edge_index[1]]
            else:
                args[name] = kwargs[name]  # kwargs not named by message() are ignored
        msg = self.message(**args)
        ref = kwargs["x"]
        return scatter(msg, edge_index[1], dim=0, dim_size=ref.size(0), reduce="sum")

    def message(self, x_j):
        return x_j


class TAGConv(MessagePassing):
    def __init__(self, in_channels, out_channels, K=3, bias=True, normalize=True, **kwargs):
        super().__init__(aggr="add")
        self.in_channels, self.out_channels, self.K, self.normalize = in_channels, out_channels, K, normalize
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(K + 1)])
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, edge_index, edge_weight=None):
        if self.normalize:  # gcn_norm(improved=False, add_self_loops=False)
            row, col = edge_index[0], edge_index[1]
            ew = torch.ones(edge_index.size(1), dtype=x.dtype, device=x.device)
            deg = scatter(ew, col, dim=0, dim_size=x.size(0), reduce="sum")
            dis = deg.pow(-0.5)
            dis.masked_fill_(dis == float("inf"), 0)
            edge_weight = dis[row] * ew * dis[col]
        out = self.lins[0](x)
        for lin in self.lins[1:]:
            x = self.propagate(edge_index, x=x, edge_weight=edge_weight)
            out = out + lin(x)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, edge_weight):
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j


def _placeholder(name):
    class _P(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} is not part of the stand-in (out of scope)")
    _P.__name__ = name
    return _P


GCN2Conv = _placeholder("GCN2Conv")
FAConv = _placeholder("FAConv")
GINEConv = _placeholder("GINEConv")
GCNConv = _placeholder("GCNConv")
ChebConv = _placeholder("ChebConv")
GATv2Conv = _placeholder("GATv2Conv")
