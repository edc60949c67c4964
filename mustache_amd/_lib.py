"""ctypes binding of libmustache_hip.so (include/mustache_hip.h).  There is no CPU fallback: if the HIP library is
missing or stale the import fails loudly."""
import ctypes
import os

MST_MAX_LEVELS = 64
MST_MAX_RADIUS = 32
MST_MAX_TESTED = 48
MST_ABI_VERSION = 3

MST_OK, MST_E_ARG, MST_E_HIP, MST_E_OVERFLOW, MST_E_NONFINITE = 0, -1, -2, -3, -4

_HERE = os.path.dirname(os.path.abspath(__file__))
# MUSTACHE_HIP_LIB: developer override to A/B alternative builds of the same ABI (still an in-tree HIP library)
LIB_PATH = os.environ.get("MUSTACHE_HIP_LIB") or os.path.join(_HERE, "libmustache_hip.so")


class MstLevels(ctypes.Structure):
    _fields_ = [
        ("n_octaves", ctypes.c_int32),
        ("levels_per_octave", ctypes.c_int32),
        ("radius", ctypes.c_int32 * MST_MAX_LEVELS),
        ("_pad", ctypes.c_int32),
        ("sigma", ctypes.c_double * MST_MAX_LEVELS),
        ("taps", (ctypes.c_double * (MST_MAX_RADIUS + 1)) * MST_MAX_LEVELS),
    ]


class MstError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmustache_hip error %d: %s" % (code, msg))
        self.code = code


class MstOverflow(MstError):
    pass


_p = ctypes.c_void_p
_i32 = ctypes.c_int32
_u32 = ctypes.c_uint32
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64

_SIGNATURES = {
    "mst_abi_version": (ctypes.c_int, []),
    "mst_last_error": (ctypes.c_char_p, []),
    "mst_scatter_blocks": (ctypes.c_int, [_p, _p, _p, _i64, ctypes.POINTER(_i64), _i32, _i32, _p, _p]),
    "mst_block_prologue": (ctypes.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "mst_gauss_blur": (ctypes.c_int, [_p, _p, _p, _i32, _i32, _i32, ctypes.POINTER(ctypes.c_double), _i32, _p]),
    "mst_scale_space": (ctypes.c_int, [_p, _p, _i32, _i32, ctypes.POINTER(MstLevels), _p, _u32, _p, _p, _i32, _p,
                                       _u64, _p]),
    "mst_scale_space_workspace_bytes": (_u64, [_i32, _i32, ctypes.POINTER(MstLevels)]),
    "mst_found_pvalues": (ctypes.c_int, [_p, _u32, _p, _p, _p, _i32, _i32, _p, _p, _p]),
    "mst_found_summary_bytes": (_u64, [_i32]),
    "mst_found_finish": (ctypes.c_int, [_p, _u32, _p, _p, _p, _i32, _i32, _p, _p, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "mst_bh_workspace_bytes": (_u64, [_i32, _u32]),
    "mst_bh_fdr": (ctypes.c_int, [_p, _p, _i32, _u32, _p, _p, _u64, _p]),
    "mst_bh_select": (ctypes.c_int, [_p, _p, _p, _i32, _u32, ctypes.c_double, _u32, _p, _p, _p, _p, _p, _u64, _p]),
    "mst_scale_space_band_tiles": (ctypes.c_int, [_i32, _i32, _p, _p]),
    "mst_scale_space_band_items": (ctypes.c_int, [ctypes.POINTER(_i64), _i32, _i32, _i32, _p, _i32, _p, _p]),
    "mst_bh_select_records": (ctypes.c_int, [_p, _p, _p, _i32, _u32, ctypes.c_double, _u32, _p, _p, _p, _p, _p, _p, _u64, _p]),
    "mst_bh_select_nowait": (ctypes.c_int, [_p, _p, _p, _i32, _u32, ctypes.c_double, _u32, _p, _p, _p, _p, _p, _u32, _p, _u64, _p]),
    "mst_found_summary_status": (ctypes.c_int, [_p, _u32]),
    "mst_select_below": (ctypes.c_int, [_p, _p, _p, _i32, _u32, ctypes.c_double, _u32, _p, _p, _p, _p, _p]),
    "mst_candidate_features": (ctypes.c_int, [_p, _p, _i32, _i32, _p, _p, _i32, _p, _p, _p, _p]),
    "mst_gather_diagonals": (ctypes.c_int, [_p, _i32, _i32, _p, _i32, _p, _p]),
    "mst_band_from_coo": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p]),
    "mst_band_from_packed": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p]),
    "mst_band_scatter_packed": (ctypes.c_int, [_p, _p, _i32, _p, _i64, _i64, _i32, _p, _p]),
    "mst_band_verify_packed": (ctypes.c_int, [_p, _p, _i32, _p, _i64, _i64, _i32, _p, _p, _p]),
    "mst_band_scatter_hic_rows": (ctypes.c_int, [_p, _p, _i32, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _i32, _p]),
    "mst_band_to_coo": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p]),
    "mst_normalize_band": (ctypes.c_int, [_p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    "mst_blocks_from_band": (ctypes.c_int, [_p, _i64, _i32, ctypes.POINTER(_i64), _i32, _i32, _p, _p, _p, _p]),
    "mst_scale_space_band": (ctypes.c_int, [_p, _i64, _i32, ctypes.POINTER(_i64), _i32, _i32, ctypes.POINTER(MstLevels), _p,
                                            _u32, _p, _p, _p, _i32, _p, _u64, _p]),
    "mst_scale_space_band_stage": (ctypes.c_int, [_p, _i64, _i32, ctypes.POINTER(_i64), _i32, _i32, ctypes.POINTER(MstLevels), _p,
                                                  _u32, _p, _p, _p, _i32, _p, _u64, ctypes.POINTER(_i32), _i32, _i32, _p]),
    "mst_scale_space_band_pair": (ctypes.c_int, [_p, _p, _i32, _i64, _i32, ctypes.POINTER(_i64), _i32, _i32, ctypes.POINTER(MstLevels),
                                                 _p, _u32, _p, _p, _p, _i32, _p, _u64, _p]),
    "mst_candidate_features_band": (ctypes.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _i32, _p, _p, _p, _p]),
    "mst_candidate_features_band_multi": (ctypes.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _i32, _p, _p, _p, _p]),
    "mst_gather_diagonals_band": (ctypes.c_int, [_p, _i64, _i32, _i64, _i32, _p, _i32, _p, _p]),
    "mst_diag_means": (ctypes.c_int, [_p, _i32, _i32, _p, _i32, _p, _p]),
    "mst_diag_means_band": (ctypes.c_int, [_p, _i64, _i32, _i64, _i32, _p, _i32, _p, _p]),
    "mst_diag_means_band_multi": (ctypes.c_int, [_p, _i64, _i32, _p, _i32, _p, _i32, _p, _p]),
    "mst_cluster_workspace_bytes": (_u64, [_u32]),
    "mst_cluster_representatives": (ctypes.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _u32, _p, _p, _p, _u64, _p]),
    "mst_diff_image": (ctypes.c_int, [_p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p]),
    "mst_masked_normfit": (ctypes.c_int, [_p, _p, _p, _p, _i32, _i64, _p, _p, _u64, _p]),
    "mst_diff_dog_workspace_bytes": (_u64, [_i32, _i32, ctypes.POINTER(MstLevels)]),
    "mst_diff_dog_band": (ctypes.c_int, [_p, _p, _i64, _i32, ctypes.POINTER(_i64), _i32, _i32, ctypes.POINTER(MstLevels),
                                         _p, _p, _p, _p, _u64, _p]),
    "mst_pair_pvalues_dog": (ctypes.c_int, [_p, _u32, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "mst_pair_gather": (ctypes.c_int, [_p, _u32, _p, _p, _i32, _p, _p, _p, _u32, _u32, _p, _p, _p, _p]),
    "mst_pair_pvalues": (ctypes.c_int, [_p, _u32, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
}

_lib = None


def load():
    """Load the shared library once; raise if it is absent (build it with `python -c 'import __graft_entry__ as g;
    g.build()'` or `make -C mustache_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mustache_amd: %s not found -- the HIP extension is required (no CPU fallback). "
            "Build it with `make -C mustache_amd/csrc` (hipcc, --offload-arch=gfx950)." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64.so.7; it must be in the process BEFORE our library is opened so that
    # the dynamic linker binds us to that same runtime instance (one HIP runtime per process, shared streams and
    # allocations).  Opening ours first would pull in /opt/rocm's copy and leave two runtimes fighting over the GPU.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError("mustache_amd: %s lacks symbols %s (stale build?)" % (LIB_PATH, missing))
    if lib.mst_abi_version() != MST_ABI_VERSION:
        raise ImportError("mustache_amd: ABI version mismatch (library %d, binding %d)"
                          % (lib.mst_abi_version(), MST_ABI_VERSION))
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc):
    if rc == MST_OK:
        return
    msg = load().mst_last_error().decode("utf-8", "replace")
    if rc == MST_E_OVERFLOW:
        raise MstOverflow(rc, msg)
    raise MstError(rc, msg)


# ---- stage markers (observability).  MUSTACHE_ROCTX=1: the host stages of a run (read / normalise / scale-space /
# tail) are pushed as roctx ranges through librocprofiler-sdk-roctx, so `rocprofv3 --marker-trace --kernel-trace` shows them above
# the kernels (the PROFILE library marks its entry points the same way).  Off by default: no library is loaded, no call is made.
_ROCTX = None


class stage:
    """with stage("read"): ...   -- a roctx range when MUSTACHE_ROCTX=1, nothing otherwise"""
    __slots__ = ("name", "on")

    def __init__(self, name):
        self.name, self.on = name, False

    def __enter__(self):
        global _ROCTX
        if os.environ.get("MUSTACHE_ROCTX") == "1":
            if _ROCTX is None:
                try:
                    _ROCTX = ctypes.CDLL("librocprofiler-sdk-roctx.so")
                    _ROCTX.roctxRangePushA.argtypes = [ctypes.c_char_p]
                except OSError:
                    _ROCTX = False
            if _ROCTX:
                _ROCTX.roctxRangePushA(("mustache: " + self.name).encode())
                self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            _ROCTX.roctxRangePop()
        return False
