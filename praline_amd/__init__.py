"""praline_amd - MI355X-native pairwise profile-profile DP hot path of PRALINE 2.

The package holds only what the hot path needs: csrc/ (HIP kernels + the C ABI of
libpraline_dp.so), native.py (ctypes binding) and the host-side mirror of the reference's
operator interface (core / container / component / util).  See DESIGN.md.
"""
__version__ = "0.1.0"
