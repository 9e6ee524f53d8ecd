"""Optimisers of the path (trainer_node_classification.py:293-296,310): `--optfun` names map to
classes with torch.optim's constructor signature and update rule."""
import torch


def resolve(name):
    if name == 'torch.optim.Adam':
        return torch.optim.Adam
    if name == 'torch.optim.SGD':
        return torch.optim.SGD
    raise ValueError(f'unknown --optfun {name}')
