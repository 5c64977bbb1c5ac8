"""Stand-in for the hooks of the reference's tree profiler (nr3d_lib/profile.py:490-562): the same
``profile`` object usable as decorator and as ``with profile("name")`` context, costing nothing.
Device-side timing on this platform comes from rocprofv3 (see profiles/)."""
import functools


class _Profile:
    def __call__(self, arg=None):
        if callable(arg):           # used as a bare decorator
            return arg
        return self                 # used as profile("name")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


profile = _Profile()
