class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kw):
        self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
        for k, v in kw.items():
            setattr(self, k, v)

    def validate(self, raise_on_error=True):
        ok = self.edge_index is None or int(self.edge_index.max()) < self.x.size(0)
        if not ok and raise_on_error:
            raise ValueError("edge_index out of range")
        return ok


class InMemoryDataset:  # placeholder (imported, never used on the path)
    pass


def download_url(*a, **k):
    raise NotImplementedError
