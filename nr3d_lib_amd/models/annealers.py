"""Scalar hyper-parameter schedules (e.g. the `inv_s` the occupancy grid converts SDF values with).

Counterpart of the reference's nr3d_lib/models/annealers.py: the functional forms ``get_anneal_val`` (:13-48; 'linear',
'logspace', 'milestones') and the object forms behind ``get_annealer`` (:50-205; plus 'constant' and 'partitions').  Pure
host arithmetic, pinned against the reference's outputs (tests/golden/ref_scalar_annealers.json).

A schedule advances in stages of ``update_every`` iterations between ``start_it`` and ``stop_it``;
progress = clip(stage / number of stages, 0, 1).  linear: value interpolated between start and stop value; logspace: the
same on the logarithms; milestones: piecewise constant, ``vals[i]`` from milestone i-1 (inclusive) to milestone i.
"""
from bisect import bisect_right
from math import exp, log
from typing import Any, List

__all__ = ['get_anneal_val', 'get_anneal_val_linear', 'get_anneal_val_logspace', 'get_anneal_val_milestones', 'get_annealer']


def _progress(it, start_it, stop_it, update_every):
    return min(1.0, max(0.0, ((it - start_it) // update_every) / ((stop_it - start_it) // update_every)))


def get_anneal_val_linear(it: int, *, stop_it: int, start_it: int = 0, start_val: float = 0.0, stop_val: float = 1.0,
                          update_every: int = 1) -> float:
    a = _progress(it, start_it, stop_it, update_every)
    return (1 - a) * start_val + a * stop_val


def get_anneal_val_logspace(it: int, *, stop_it: int, start_it: int = 0, start_val: float = 1.0, stop_val: float = 10.0,
                            update_every: int = 1) -> float:
    a = _progress(it, start_it, stop_it, update_every)
    return exp(log(start_val) * (1 - a) + log(stop_val) * a)


def get_anneal_val_milestones(it: int, milestones: List[int], vals: List[Any]):
    assert (len(milestones) + 1) == len(vals), '`vals` should have one more element than `milestones`'
    return vals[bisect_right(milestones, it)]


_FUNCTIONS = dict(linear=get_anneal_val_linear, logspace=get_anneal_val_logspace, milestones=get_anneal_val_milestones)


def get_anneal_val(type: str, **params) -> float:
    if type not in _FUNCTIONS:
        raise RuntimeError(f"Invalid type={type}")
    return _FUNCTIONS[type](**params)


class _Annealer:
    """object form: ``a(it)``, or ``a.set_iter(it); a.get_val()``; ``set_val`` pins the value"""
    type = None

    def __init__(self):
        self._it, self._bypass_val = None, None

    def value_at(self, it: int):
        raise NotImplementedError

    def __call__(self, it: int):
        return self._bypass_val if self._bypass_val is not None else self.value_at(it)

    def set_iter(self, it: int):
        self._it = it

    def set_val(self, val):
        self._bypass_val = val

    def get_val(self):
        assert self._it is not None, "Please call `set_iter` first"
        return self(self._it)


class AnnealerConstant(_Annealer):
    def __init__(self, val: float):
        super().__init__()
        self.val = val

    def __call__(self, it: int = None):
        return self.val

    def set_val(self, val):
        self.val = val

    def get_val(self):
        return self.val


class AnnealerMilestones(_Annealer):
    def __init__(self, milestones: List[int], vals: List[Any]):
        super().__init__()
        assert (len(milestones) + 1) == len(vals), '`vals` should have one more element than `milestones`'
        self.milestones, self.vals = milestones, vals

    def value_at(self, it):
        return self.vals[bisect_right(self.milestones, it)]

    def get_val(self):
        return self(self._it)


class _Ramp(_Annealer):
    def __init__(self, stop_it: int, start_it: int = 0, start_val: float = None, stop_val: float = None, update_every: int = 1):
        super().__init__()
        self.start_it, self.stop_it, self.update_every = int(start_it), int(stop_it), int(update_every)
        self.start_val, self.stop_val = start_val, stop_val

    def value_at(self, it):
        if it < self.start_it:
            return self.start_val
        if it >= self.stop_it:
            return self.stop_val
        return self.blend(_progress(it, self.start_it, self.stop_it, self.update_every))


class AnnealerLinear(_Ramp):
    def __init__(self, stop_it: int, start_it: int = 0, start_val: float = 0.0, stop_val: float = 1.0, update_every: int = 1):
        super().__init__(stop_it, start_it, start_val, stop_val, update_every)

    def blend(self, a):
        return (1 - a) * self.start_val + a * self.stop_val


class AnnealerLogSpace(_Ramp):
    def __init__(self, stop_it: int, start_it: int = 0, start_val: float = 1.0, stop_val: float = 10.0, update_every: int = 1):
        assert start_val > 0, "Invalid log(start_val)"
        super().__init__(stop_it, start_it, start_val, stop_val, update_every)

    def blend(self, a):
        return exp(log(self.start_val) * (1 - a) + log(self.stop_val) * a)


class AnnealerPartitions(_Annealer):
    """consecutive schedules: partition i runs until its ``stop_it`` and starts (by default) where the previous one stopped"""

    def __init__(self, partition_cfgs: List[dict]):
        super().__init__()
        self.partitions, self.partition_stops = [], []
        prev_stop = 0
        for cfg in partition_cfgs:
            cfg.setdefault('start_it', prev_stop)
            self.partitions.append(get_annealer(**cfg))
            prev_stop = cfg['stop_it']
            self.partition_stops.append(prev_stop)

    def value_at(self, it):
        p = self.partitions[bisect_right(self.partition_stops, it)]
        p.set_iter(it)
        return p.get_val()

    def get_val(self):
        return self(self._it)


_CLASSES = dict(linear=AnnealerLinear, logspace=AnnealerLogSpace, constant=AnnealerConstant, milestones=AnnealerMilestones,
                partitions=AnnealerPartitions)


def get_annealer(type: str = None, **params):
    if type not in _CLASSES:
        raise RuntimeError(f"Invalid type={type}")
    a = _CLASSES[type](**params)
    a.type = type
    return a
