import torch


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    assert dim == 0 and reduce in ("sum", "add")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((n,), dtype=dtype, device=index.device)
    return out.scatter_add_(0, index, torch.ones((index.size(0),), dtype=out.dtype, device=index.device))


def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None, num_nodes=None):
    assert normalization is None
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    row = edge_index[0]
    deg = scatter(edge_weight, row, 0, dim_size=n, reduce="sum")
    loop = torch.arange(n, device=edge_index.device)
    ei = torch.cat([edge_index, torch.stack([loop, loop])], dim=1)
    ew = torch.cat([-edge_weight, deg], dim=0)
    return ei, ew
