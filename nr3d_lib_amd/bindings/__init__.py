"""Python-visible twins of the reference's compiled extensions (nr3d_lib.bindings._lotd, ._pack_ops,
._occ_grid), implemented over the C ABI of libnr3d_hip.so."""
