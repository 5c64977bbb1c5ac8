"""Return records of the ray marchers -- counterpart of nr3d_lib/graphics/raymarch/__init__.py:10-80.
All sample-level tensors are packed; ``pack_infos`` describes one pack per hit ray."""
from dataclasses import dataclass, fields
from typing import Optional

import torch


@dataclass
class RaymarchRetBase:
    num_hit_rays: int
    ridx_hit: Optional[torch.Tensor]        # [num_hit_rays]      indices of the input rays that got >= 1 sample
    samples: Optional[torch.Tensor]         # [num_samples, 3]    sample positions
    depth_samples: Optional[torch.Tensor]   # [num_samples]       sample depths
    deltas: Optional[torch.Tensor]          # [num_samples]       interval lengths
    ridx: Optional[torch.Tensor]            # [num_samples]       ray index of every sample
    pack_infos: Optional[torch.Tensor]      # [num_hit_rays, 2]   (first sample, #samples) of every hit ray

    def __iter__(self):
        return iter(tuple(getattr(self, f.name) for f in fields(self)))

    def __getitem__(self, name: str):
        return getattr(self, name)


@dataclass
class RaymarchRetSingle(RaymarchRetBase):
    gidx: Optional[torch.Tensor]            # [num_samples]       flat voxel index of every sample
    gidx_pack_infos: Optional[torch.Tensor]  # [num_voxel_packs, 2]


@dataclass
class RaymarchRetBatched(RaymarchRetBase):
    bidx: Optional[torch.Tensor]            # [num_samples]       batch index of every sample
    gidx: Optional[torch.Tensor]
    gidx_pack_infos: Optional[torch.Tensor]


@dataclass
class RaymarchRetDynamic(RaymarchRetBase):
    ts: Optional[torch.Tensor]
    gidx: Optional[torch.Tensor]
    gidx_pack_infos: Optional[torch.Tensor]


@dataclass
class RaymarchRetBatchedDynamic(RaymarchRetBase):
    bidx: Optional[torch.Tensor]
    ts: Optional[torch.Tensor]
    gidx: Optional[torch.Tensor]
    gidx_pack_infos: Optional[torch.Tensor]


@dataclass
class RaymarchRetForest(RaymarchRetBase):
    blidx: Optional[torch.Tensor]
    blidx_pack_infos: Optional[torch.Tensor]
    gidx: Optional[torch.Tensor]
    gidx_pack_infos: Optional[torch.Tensor]
