"""saena_amd -- MI355X-native V-cycle hot path of the Saena AMG solver.

capi : ctypes view of the C ABI of libsaena_amd.so (include/saena_gpu.h)
host : ctypes view of the host-side mirror of saena::matrix (include/saena_c.h)
"""
__all__ = ["capi", "host"]
