"""mustache_amd -- MI355X-native scale-space chromatin-loop caller (drop-in for ay-lab/mustache's per-chromosome run).

The arithmetic of the hot path lives in hand-written HIP kernels (``mustache_amd/csrc``) behind a C ABI
(``include/mustache_hip.h``); this package is the Python host that mirrors the reference's interface.
"""
__version__ = "0.1.0"

import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); this package uses the default
# stream + three side streams per device (engine.device_streams) and PyTorch has streams of its own: ask for 8 queues unless the
# user chose a value (effective only if the HIP runtime is not initialised yet -- it is not at `import torch`).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
