"""Coarse-to-fine schedule for multi-resolution encodings: which levels are active at a training iteration.

Counterpart of the reference's nr3d_lib/models/grid_encodings/multires_annealer.py (MultiresAnnealer :19-80).  It is what
drives the hot path's ``max_level`` argument (levels above it get no forward work and, on the binned gradient path, no
stage-A blocks or work items) and the optional per-feature window the encoder output is multiplied with.

  hardmask: max_level climbs linearly from ``start_level`` to the last level between ``start_it`` and ``stop_it`` (in
            steps of ``update_every`` iterations); no window.
  cosine:   additionally a window that fades every level in over one level's share of the schedule
            (0.5 * (1 - cos(pi * clip(progress - level + 1, 0, 1)))), repeated per feature of the level.
The reference builds the per-level repeat counts in the OUTPUT dtype, which ``repeat_interleave`` rejects (its cosine mode
raises as shipped); integer repeats -- the evident intent -- are used here.
"""
from math import pi
from typing import List, Tuple, Union

import torch
import torch.nn as nn

__all__ = ['MultiresAnnealer']


class MultiresAnnealer(nn.Module):
    def __init__(self, level_n_feats: List[int], type: str, stop_it: int, start_it: int = 0, update_every: int = 1,
                 start_level: int = 0, dtype=torch.float, device=None) -> None:
        super().__init__()
        if type not in ('hardmask', 'cosine'):
            raise RuntimeError(f'Invalid anneal_type={type}')
        self.type = type
        self.num_levels = len(level_n_feats)
        self.register_buffer('level_n_feats', torch.as_tensor(list(level_n_feats), dtype=torch.long, device=device), persistent=False)
        self.register_buffer('level_arange', torch.arange(self.num_levels, dtype=dtype, device=device), persistent=False)
        self.start_it, self.stop_it, self.update_every = int(start_it), int(stop_it), int(update_every)
        self.it = self.stop_it                                   # fully annealed unless an iteration is set
        self.total_stages = (self.stop_it - self.start_it) // self.update_every
        # -1: no level active at the start; 0: the first level
        self.start_level = max(min(int(start_level), self.num_levels - 1), -1)

    def set_iter(self, it: int):
        self.it = it

    def progress(self, it: int = None) -> float:
        it = self.it if it is None else it
        stage = (it - self.start_it) // self.update_every
        return min(1.0, max(0.0, stage / self.total_stages))

    def forward(self, it: int = None) -> Tuple[int, Union[None, torch.Tensor]]:
        """-> (max_level, window | None)"""
        alpha = self.progress(it)
        span = (self.num_levels - 1) - self.start_level
        if self.type == 'hardmask':
            return self.start_level + min(int(alpha * span), span), None
        raw = self.start_level + alpha * span - self.level_arange + 1
        window = 0.5 * (1 + torch.cos(pi * torch.clip(raw, 0.0, 1.0) + pi))
        return int((raw > 0).sum().item()) - 1, window.repeat_interleave(self.level_n_feats)
