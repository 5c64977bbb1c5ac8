"""nr3d_lib_amd -- MI355X-native (gfx950) implementation of the nr3d_lib neural-rendering hot path:
LoTD grid/hash encoder, occupancy-grid ray marching and packed volume-rendering reductions.

Module layout mirrors the reference so that ``nr3d_lib.X`` -> ``nr3d_lib_amd.X`` is a drop-in for:
    nr3d_lib_amd.bindings._lotd / ._pack_ops / ._occ_grid      (reference: pybind extensions)
    nr3d_lib_amd.models.grid_encodings.lotd                    (LoTDFunction*, LoTD, lotd_encoding*)
    nr3d_lib_amd.graphics.pack_ops                             (packed_* ops with autograd)
    nr3d_lib_amd.graphics.raymarch                             (occgrid_raymarch, RaymarchRet*)
Kernels live in nr3d_lib_amd/csrc (HIP, C ABI in include/nr3d_hip.h) and are loaded by ``_hip``.
"""
__version__ = "0.1.0"
