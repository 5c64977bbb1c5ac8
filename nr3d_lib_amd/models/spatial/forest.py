"""Forest of equally sized blocks addressed through a byte octree -- counterpart of ``ForestBlockSpace``
(nr3d_lib/models/spatial/forest.py:34-420), the owner of the ``ForestMeta`` that the forest LoTD encoder and the
forest marcher consume.

The reference builds and queries the octree with kaolin (``unbatched_points_to_octree``, ``unbatched_query``, the SPC
pyramids); kaolin is not a dependency here: the octree is built from the block coordinates with a few sorts
(``octree_from_corners``) in exactly the SPC layout -- nodes level by level, children in child-index order
(x<<2 | y<<1 | z), one occupancy byte per non-leaf node, ``exsum`` the exclusive prefix sum of the bytes' popcounts,
``pyramid`` the per-level node counts and first indices -- and point queries run the ``identify`` kernel
(``bindings._forest.forest_identify``).  Same method names and return conventions: ``populate`` modes, ``reset``,
``normalize_coords[_01]`` / ``unnormalize_coords`` (block index -1 for points outside every block),
``sample_pts_uniform``, ``ray_test`` (ray / block boxes, segments sorted by entry depth per ray), ``ray_step_coarse``.
"""
from numbers import Number
from typing import Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from nr3d_lib_amd.bindings._forest import ForestMeta, forest_identify
from nr3d_lib_amd.graphics.pack_ops import (get_pack_infos_from_first, interleave_sample_step_wrt_depth_in_packed_segments,
                                            mark_pack_boundaries, packed_sort_inplace)

__all__ = ['ForestBlockSpace', 'octree_from_corners', 'ray_box_intersection']


def _morton(k: torch.Tensor, level: int) -> torch.Tensor:
    """sort key of integer coordinates [n,3] that reproduces the octree's breadth-first child order"""
    k = k.long()
    code = torch.zeros(k.shape[0], dtype=torch.long, device=k.device)
    for b in range(level):
        code |= (((k[:, 0] >> b) & 1) << (3 * b + 2)) | (((k[:, 1] >> b) & 1) << (3 * b + 1)) | (((k[:, 2] >> b) & 1) << (3 * b))
    return code


@torch.no_grad()
def octree_from_corners(corners: torch.Tensor, level: int):
    """integer coordinates [n,3] of the occupied cells on `level` -> (octree uint8 [n_nodes], exsum int32 [n_nodes+1],
    point_hierarchy int16 [n_points,3], pyramid int64 [2, level+2]: row 0 the node count per level, row 1 its
    exclusive prefix sum)"""
    dev = corners.device
    cur = torch.unique(corners.reshape(-1, 3).long(), dim=0)
    assert cur.numel() > 0 and int(cur.min()) >= 0 and int(cur.max()) < (1 << level), "corners outside the 2^level cube"
    per_level, octree_bytes = [None] * (level + 1), [None] * level
    cur = cur[torch.argsort(_morton(cur, level))]
    per_level[level] = cur
    for l in range(level - 1, -1, -1):
        child = per_level[l + 1]
        parent = child >> 1
        first = torch.ones(child.shape[0], dtype=torch.bool, device=dev)
        first[1:] = (parent[1:] != parent[:-1]).any(dim=1)          # children are Morton sorted: equal parents are adjacent
        pid = torch.cumsum(first.long(), 0) - 1
        bit = 1 << (((child[:, 0] & 1) << 2) | ((child[:, 1] & 1) << 1) | (child[:, 2] & 1))
        octree_bytes[l] = torch.zeros(int(pid[-1]) + 1, dtype=torch.long, device=dev).index_add_(0, pid, bit)
        per_level[l] = parent[first]
    octree = torch.cat(octree_bytes).to(torch.uint8) if level > 0 else torch.zeros(0, dtype=torch.uint8, device=dev)
    pop = torch.zeros_like(octree, dtype=torch.int32)
    for b in range(8):
        pop += ((octree >> b) & 1).int()
    exsum = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), torch.cumsum(pop, 0).int()])
    counts = torch.tensor([p.shape[0] for p in per_level] + [0], dtype=torch.long)
    pyramid = torch.stack([counts, torch.cumsum(counts, 0) - counts])
    return octree, exsum, torch.cat(per_level).to(torch.int16).contiguous(), pyramid


def ray_box_intersection(rays_o, rays_d, *, aabb_min=-0.5, aabb_max=0.5, t_min_cons=None, t_max_cons=None):
    """slab test (nr3d_lib/graphics/raytest.py:88-133) -> (t_near, t_far, hit)"""
    t_min, t_max = (aabb_min - rays_o) / rays_d, (aabb_max - rays_o) / rays_d
    t_near = torch.minimum(t_min, t_max).max(dim=-1).values
    t_far = torch.maximum(t_min, t_max).min(dim=-1).values
    if t_min_cons is not None:
        t_near = torch.maximum(t_near, torch.as_tensor(t_min_cons, dtype=t_near.dtype, device=t_near.device))
    if t_max_cons is not None:
        t_far = torch.minimum(t_far, torch.as_tensor(t_max_cons, dtype=t_far.dtype, device=t_far.device))
    return t_near, t_far, (t_far > t_near) & (t_far > 0)


class ForestBlockSpace(nn.Module):
    def __init__(self, continuity_enabled=True, dtype=torch.float, device=None) -> None:
        super().__init__()
        self.dtype = dtype
        self.continuity_enabled = continuity_enabled
        for name in ('world_origin', 'world_origin0'):          # block [0,0,0]'s position in the world
            self.register_buffer(name, torch.zeros(3, dtype=dtype, device=device), persistent=True)
        for name in ('world_block_size', 'world_block_size0'):
            self.register_buffer(name, torch.ones(3, dtype=dtype, device=device), persistent=True)
        self.register_buffer('_octree', torch.empty([0], dtype=torch.uint8, device=device), persistent=True)
        self.register_buffer('_max_length', torch.tensor([0], dtype=torch.long, device=device), persistent=True)
        self.register_buffer('_level', torch.tensor([0], dtype=torch.long, device=device), persistent=True)
        self.meta: ForestMeta = None
        self._register_load_state_dict_pre_hook(self._before_load_state_dict)

    @property
    def device(self) -> torch.device:
        return self.world_origin.device

    def _before_load_state_dict(self, state_dict, prefix, *unused):
        octree = state_dict[prefix + '_octree'].to(self.device)
        level = int(state_dict[prefix + '_level'].item()) if prefix + '_level' in state_dict else None
        self.reset(octree, level=level, max_length=int(state_dict[prefix + '_max_length'].item()),
                   world_origin=state_dict[prefix + 'world_origin'], world_block_size=state_dict[prefix + 'world_block_size'])

    # ---- the forest_meta's fields ------------------------------------------------------------------------------------
    n_trees = property(lambda self: self.meta.n_trees)
    level = property(lambda self: self.meta.level)
    block_ks = property(lambda self: self.meta.block_ks)

    def get_aabb(self) -> torch.Tensor:
        lo = self.block_ks.min(dim=0).values.to(self.dtype) * self.world_block_size + self.world_origin
        hi = (self.block_ks.max(dim=0).values.to(self.dtype) + 1) * self.world_block_size + self.world_origin
        return torch.stack([lo, hi], 0)

    def set_enable_continuity(self, enabled=True):
        self.continuity_enabled = self.meta.continuity_enabled = enabled

    def _vec3(self, v):
        if isinstance(v, Number):
            v = [v] * 3
        if isinstance(v, (list, tuple, np.ndarray)):
            return torch.tensor(np.asarray(v, dtype=np.float64), dtype=self.dtype, device=self.device)
        if isinstance(v, torch.Tensor):
            return v.to(dtype=self.dtype, device=self.device)
        raise RuntimeError(f"Invalid type {type(v)}")

    @torch.no_grad()
    def reset(self, octree: torch.Tensor, level: int = None, max_length: int = None, world_origin=None, world_block_size=None):
        """rebuild the meta from an octree byte string (forest.py:113-158); the hierarchy is re-derived by walking it"""
        if world_origin is not None:
            self.world_origin = self._vec3(world_origin)
        if world_block_size is not None:
            self.world_block_size = self._vec3(world_block_size)
        octree = octree.to(device=self.device, dtype=torch.uint8).reshape(-1).contiguous()
        points, counts = _walk_octree(octree, level)
        level = len(counts) - 1
        if max_length is None:
            max_length = 2 ** level
        pyr_first = np.concatenate([[0], np.cumsum(counts)])
        pop = torch.zeros_like(octree, dtype=torch.int32)
        for b in range(8):
            pop += ((octree >> b) & 1).int()
        meta = ForestMeta()
        meta.block_ks = points[int(pyr_first[level]):int(pyr_first[level + 1])].contiguous()
        meta.n_trees = meta.block_ks.shape[0]
        meta.level, meta.level_poffset = level, int(pyr_first[level])
        meta.world_block_size, meta.world_origin = self.world_block_size.tolist(), self.world_origin.tolist()
        meta.resolution = [max_length] * 3
        meta.octree = octree
        meta.exsum = torch.cat([torch.zeros(1, dtype=torch.int32, device=self.device), torch.cumsum(pop, 0).int()])
        meta.continuity_enabled = self.continuity_enabled
        self.meta = meta
        self.point_hierarchies, self.pyramids = points, torch.tensor(np.stack([np.append(counts, 0), pyr_first]))
        self._octree = octree.clone()
        self._max_length[:] = max_length
        self._level[:] = level

    @torch.no_grad()
    def populate(self, mode: str, **kwargs):
        try:
            fn = dict(single_block=self.populate_single_block, dense=self.populate_dense,
                      from_corners=self.populate_from_corners)[mode]
        except KeyError:
            raise RuntimeError(f"Invalid mode={mode}")
        fn(**kwargs)
        # the coordinate scale at population time stays fixed for the life of the model (forest.py:172-180)
        self.world_block_size0, self.world_origin0 = self.world_block_size.clone(), self.world_origin.clone()

    @torch.no_grad()
    def populate_single_block(self, world_origin=None, world_block_size=None):
        self.reset(torch.zeros(0, dtype=torch.uint8, device=self.device), level=0, world_origin=world_origin,
                   world_block_size=world_block_size)

    @torch.no_grad()
    def populate_dense(self, level, world_origin=None, world_block_size=None):
        n = sum(8 ** l for l in range(level))
        self.reset(torch.full((n,), 255, dtype=torch.uint8, device=self.device), level=level, world_origin=world_origin,
                   world_block_size=world_block_size)

    @torch.no_grad()
    def populate_from_corners(self, corners, *, level: int = None, world_origin=None, world_block_size=None):
        corners = torch.as_tensor(corners, device=self.device).to(torch.int16).contiguous()
        if level is None:
            level = int(np.log2(corners.max().item())) + 1
            max_length = corners.max().item() + 1
        else:
            max_length = 2 ** level
        octree = octree_from_corners(corners, level)[0]
        self.reset(octree, level=level, max_length=max_length, world_origin=world_origin, world_block_size=world_block_size)

    # ---- coordinates ---------------------------------------------------------------------------------------------------
    def pidx2blidx(self, pidx: torch.Tensor):
        return torch.where(pidx == -1, pidx, pidx - self.meta.level_poffset)

    def blidx2pidx_unsafe(self, blidx: torch.Tensor):
        return blidx + self.meta.level_poffset

    def normalize_coords_01(self, world_coords: torch.Tensor, block_inds: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (coordinates in the point's block in [0,1], block index; -1 = outside the forest: do not use those rows)"""
        dtype = world_coords.dtype
        coords_ = (world_coords - self.world_origin.to(dtype)) / self.world_block_size.to(dtype)
        if block_inds is None:
            # .int() truncates toward zero like the reference's query (negative fractions land in block 0 there too)
            blidx = forest_identify(self.meta, coords_.int()).long()
        else:
            blidx = block_inds
        return coords_ - self.block_ks[blidx].to(dtype), blidx

    def normalize_coords(self, world_coords: torch.Tensor, block_inds: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        x01, blidx = self.normalize_coords_01(world_coords, block_inds)
        return x01 * 2. - 1., blidx

    def unnormalize_coords(self, coords_in_block: torch.Tensor, block_inds: Union[torch.Tensor, int]) -> torch.Tensor:
        dtype = coords_in_block.dtype
        coords_ = (coords_in_block + 1.) / 2. + self.block_ks[block_inds].to(dtype)
        return coords_ * self.world_block_size.to(dtype) + self.world_origin.to(dtype)

    def sample_pts_uniform(self, num_pts: int = None, num_pts_per_block: int = None):
        if num_pts_per_block is None:
            num_pts_per_block = int(num_pts // self.n_trees) + 1
        block_x = torch.empty([self.n_trees, num_pts_per_block, 3], dtype=self.dtype, device=self.device).uniform_(-1, 1)
        blidx = torch.arange(self.n_trees, device=self.device).unsqueeze_(-1).expand(-1, num_pts_per_block).contiguous()
        return block_x, blidx

    # ---- rays ----------------------------------------------------------------------------------------------------------
    def ray_test(self, rays_o: torch.Tensor, rays_d: torch.Tensor, near=None, far=None, return_rays=True, **extra_ray_data):
        """ray / block-box intersections of every (ray, block) pair, kept as per-ray packs of (block, entry, exit)
        segments sorted by entry depth (forest.py:353-394)"""
        wo, wb = self.world_origin.to(rays_o.dtype), self.world_block_size.to(rays_o.dtype)
        with torch.no_grad():
            bmin = self.block_ks.to(rays_o.dtype) * wb + wo
            t_near, t_far, check = ray_box_intersection(
                rays_o.unsqueeze(1), rays_d.unsqueeze(1), aabb_min=bmin.unsqueeze(0), aabb_max=(bmin + wb).unsqueeze(0),
                t_min_cons=0. if near is None else (near.unsqueeze(-1) if isinstance(near, torch.Tensor) else near),
                t_max_cons=far.unsqueeze(-1) if isinstance(far, torch.Tensor) else far)
            ridx, blidx = check.nonzero(as_tuple=True)
            if ridx.numel() == 0:
                ret = dict(num_rays=0, rays_inds=None)
                if return_rays:
                    ret.update(rays_o=rays_o[:0], rays_d=rays_d[:0])
                return ret
            boundary = mark_pack_boundaries(ridx)
            first_inds = boundary.nonzero().long()[..., 0]
            ridx_hit = ridx[first_inds]
            num_rays = first_inds.numel()
            seg_pack_infos = get_pack_infos_from_first(first_inds, boundary.numel())
            seg_entries, seg_exits = t_near[ridx, blidx].contiguous(), t_far[ridx, blidx].contiguous()
            order = packed_sort_inplace(seg_entries, seg_pack_infos)
            seg_exits, blidx = seg_exits[order], blidx[order]
            if near is None:
                near = seg_entries[first_inds].contiguous()
            else:
                near = near[ridx_hit] if isinstance(near, torch.Tensor) else rays_o.new_full([num_rays], near)
            if far is None:
                far = seg_exits[seg_pack_infos.sum(-1).sub_(1)].contiguous()
            else:
                far = far[ridx_hit] if isinstance(far, torch.Tensor) else rays_o.new_full([num_rays], far)
        ret = dict(num_rays=num_rays, rays_inds=ridx_hit.long().contiguous(), near=near, far=far,
                   seg_pack_infos=seg_pack_infos.long().contiguous(), seg_block_inds=blidx.long().contiguous(),
                   seg_entries=seg_entries.contiguous(), seg_exits=seg_exits.contiguous())
        ret.update({k: v[ridx_hit] if isinstance(v, torch.Tensor) else v for k, v in extra_ray_data.items()})
        if return_rays:
            ret.update(rays_o=rays_o.index_select(0, ridx_hit), rays_d=rays_d.index_select(0, ridx_hit))
        return ret

    def ray_step_coarse(self, rays_o, rays_d, near, far, seg_block_inds, seg_entries, seg_exits, seg_pack_infos, *,
                        step_mode: str = 'depth', **step_kwargs):
        """depth-proportional stepping inside the block segments (forest.py:396-419, 'depth' mode) ->
        dict(num_hit_rays, ridx_hit, samples, depth_samples, deltas, ridx, pack_infos, blidx, blidx_pack_infos)"""
        if step_mode != 'depth':
            raise RuntimeError(f"Invalid step_mode={step_mode}")
        depths, deltas, ridx, pack_infos, sidx, blidx_pack_infos = interleave_sample_step_wrt_depth_in_packed_segments(
            near, far, seg_entries, seg_exits, seg_pack_infos, **step_kwargs)
        ridx_hit = ridx[pack_infos[:, 0]]
        return dict(num_hit_rays=ridx_hit.numel(), ridx_hit=ridx_hit,
                    samples=torch.addcmul(rays_o[ridx], rays_d[ridx], depths.unsqueeze(-1)), depth_samples=depths,
                    deltas=deltas, ridx=ridx, pack_infos=pack_infos, blidx=seg_block_inds[sidx].long().contiguous(),
                    blidx_pack_infos=blidx_pack_infos)

    def extra_repr(self) -> str:
        if self.meta is None:
            return "unpopulated"
        return f"level={self.level}, n_trees={self.n_trees}, world_origin={self.world_origin.tolist()}, " \
               f"world_block_size={self.world_block_size.tolist()}, continuity_enabled={self.continuity_enabled}"


def _walk_octree(octree: torch.Tensor, level: int = None):
    """byte octree -> (point hierarchy int16 [n_points,3] in breadth-first order, node count per level); one vectorised
    expansion per level.  `level` = depth to walk to (None: until the bytes run out)."""
    dev = octree.device
    pts = torch.zeros(1, 3, dtype=torch.long, device=dev)
    out, counts, used = [pts], [1], 0
    child_off = torch.tensor([[(c >> 2) & 1, (c >> 1) & 1, c & 1] for c in range(8)], dtype=torch.long, device=dev)
    l = 0
    while (level is None and used < octree.numel()) or (level is not None and l < level):
        n = pts.shape[0]
        assert used + n <= octree.numel(), "octree byte string shorter than its own hierarchy"
        bits = octree[used:used + n].long()
        used += n
        has = ((bits.unsqueeze(1) >> torch.arange(8, device=dev)) & 1).bool()        # [n, 8]
        pts = (pts.unsqueeze(1) * 2 + child_off.unsqueeze(0))[has]                   # row-major: parent order, child order
        out.append(pts)
        counts.append(pts.shape[0])
        l += 1
    return torch.cat(out).to(torch.int16), np.array(counts, dtype=np.int64)
