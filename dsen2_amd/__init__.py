"""dsen2_amd — MI355X (gfx950) native DSen2 / VDSen2 Sentinel-2 super-resolution inference path.

Drop-in surface (same names as the reference's testing/supres.py, utils/DSen2Net.py, utils/patches.py):
    from dsen2_amd.supres import DSen2_20, DSen2_60
Everything computes in hand-written HIP kernels behind the C ABI in include/dsen2_hip.h; importing
this package does not load the shared library, calling into it does and fails loudly if it is missing.
"""
__version__ = '0.1.0'
