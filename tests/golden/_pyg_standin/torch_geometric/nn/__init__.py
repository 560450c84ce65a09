from . import conv  # noqa: F401


class Sequential:  # placeholder: only gnn_dsse (out of scope) uses it
    def __init__(self, *a, **k):
        raise NotImplementedError("torch_geometric.nn.Sequential is not part of the stand-in")
