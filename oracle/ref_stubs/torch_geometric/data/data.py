class Data:
    """Attribute bag standing in for torch_geometric.data.data.Data (utils.py:798 subclasses it)."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return self.x.shape[0]

    def to(self, device):
        import torch
        for k, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self
