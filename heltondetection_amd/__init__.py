"""MI355X-native detection hot path. HIP kernels live in csrc/, behind include/hd_ops.h."""
__version__ = "0.1.0"
