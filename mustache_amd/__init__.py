"""mustache_amd -- MI355X-native scale-space chromatin-loop caller (drop-in for ay-lab/mustache's per-chromosome run).

The arithmetic of the hot path lives in hand-written HIP kernels (``mustache_amd/csrc``) behind a C ABI
(``include/mustache_hip.h``); this package is the Python host that mirrors the reference's interface.
"""
__version__ = "0.1.0"
