from .aabb import *  # noqa: F401,F403
from .forest import *  # noqa: F401,F403
from .batched import *  # noqa: F401,F403


def get_space(space_cfg=None):
    """name or {'type': name, **kwargs} -> space module (nr3d_lib/models/spatial/__init__.py:10-32); the time-varying
    spaces ('aabb_dynamic', 'batched_dynamic') are not provided"""
    if space_cfg is None:
        return None
    cfg = {'type': space_cfg} if isinstance(space_cfg, str) else dict(space_cfg)
    kind = cfg.pop('type').lower()
    if kind == 'aabb':
        return AABBSpace(**cfg)            # noqa: F405
    if kind == 'batched':
        return BatchedBlockSpace(**cfg)    # noqa: F405
    if kind == 'forest':
        return ForestBlockSpace(**cfg)     # noqa: F405
    if kind in ('unbounded', 'none'):
        return None
    if kind in ('aabb_dynamic', 'batched_dynamic'):
        raise NotImplementedError(f"nr3d_lib_amd: space type {kind!r} (time-varying) is not provided")
    raise RuntimeError(f"Invalid space_type={kind}")


def create_dense_grid(level: int, device=None):
    """[8^level, 3] int16 integer coordinates of the full grid at ``level`` (spatial/utils.py:24-30)"""
    import torch
    ax = [torch.arange(2 ** level, device=device, dtype=torch.short) for _ in range(3)]
    return torch.stack(torch.meshgrid(ax, indexing='ij'), dim=-1).reshape(-1, 3)


def create_octree_dense(level: int, device=None):
    """byte octree with every node of ``level`` levels present (spatial/utils.py:32-36; kaolin's
    unbatched_points_to_octree there, the package's own builder here)"""
    assert level > 0, "level must be > 0 during creation of octree."
    return octree_from_corners(create_dense_grid(level, device=device), level)[0]    # noqa: F405


def create_octree_root_only(device=None):
    import torch
    return torch.tensor([255], device=device, dtype=torch.uint8)
