import torch

from ..data import Data


def collate(data_list):
    """DataLoader collation: concat x / edge_attr / y, offset edge_index."""
    xs, eis, eas, ys, batch, off = [], [], [], [], [], 0
    for g, d in enumerate(data_list):
        xs.append(d.x); eas.append(d.edge_attr); ys.append(d.y)
        eis.append(d.edge_index + off)
        batch.append(torch.full((d.x.size(0),), g, dtype=torch.long))
        off += d.x.size(0)
    return Data(x=torch.cat(xs), edge_index=torch.cat(eis, 1), edge_attr=torch.cat(eas), y=torch.cat(ys),
                batch=torch.cat(batch))
