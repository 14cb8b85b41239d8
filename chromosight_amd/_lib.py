"""ctypes binding of libchromosight_hip.so (the C ABI in include/chromosight_hip.h).

The library is the only compute backend of this package: if it cannot be loaded, or no
MI355X/HIP device is usable, every device entry point raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import pathlib
import threading
import weakref

import numpy as np

CS_F32, CS_F64 = 0, 1
LAYOUT_DENSE, LAYOUT_BAND, LAYOUT_BAND_LAZY, LAYOUT_BAND_PADDED, LAYOUT_BAND_COUNTS, LAYOUT_BAND_COUNTS_VIEW = 0, 1, 2, 3, 4, 5
COUNTS_HEADER_BYTES = 128      # CS_COUNTS_HEADER_BYTES: in front of a band of raw counts (CS_LAYOUT_BAND_COUNTS)
LAZY_BAND_BYTES = 128          # CS_LAZY_BAND_BYTES: descriptor of a lazily evaluated float64 band (cs_stage_block)
MASK_NONE, MASK_BINS, MASK_EXPLICIT = 0, 1, 2

# CHROMOSIGHT_HIP_LIBRARY: another build of the same library (diagnostics, e.g. the section-timing build)
_LIB_PATH = pathlib.Path(os.environ.get("CHROMOSIGHT_HIP_LIBRARY") or pathlib.Path(__file__).with_name("libchromosight_hip.so"))


class CsMatrix(C.Structure):
    _fields_ = [
        ("d_ptr", C.c_void_p),
        ("dtype", C.c_int32),
        ("layout", C.c_int32),
        ("ld", C.c_int64),
        ("band_lo", C.c_int32),
        ("band_w", C.c_int32),
        ("row0", C.c_int64),      # matrix row stored at d_ptr (row windows)
    ]


class CsKernel(C.Structure):
    _fields_ = [
        ("km", C.c_int32),
        ("kn", C.c_int32),
        ("h_kernel", C.POINTER(C.c_double)),
        ("h_kernel_conv", C.POINTER(C.c_double)),
        ("h_kernel_sq", C.POINTER(C.c_double)),
    ]


class CsNormxcorr2Params(C.Structure):
    _fields_ = [
        ("ms", C.c_int32),
        ("ns", C.c_int32),
        ("full", C.c_int32),
        ("sym_upper", C.c_int32),
        ("max_dist", C.c_int32),
        ("mask_mode", C.c_int32),
        ("d_miss_row", C.c_void_p),
        ("d_miss_col", C.c_void_p),
        ("d_mask", C.c_void_p),
        ("min_present", C.c_int32),
        ("compute_dtype", C.c_int32),
        ("xcorr_threshold", C.c_double),
        ("denom_eps", C.c_double),
        ("row_begin", C.c_int32),  # output row window, (0, 0) = all rows
        ("row_end", C.c_int32),
    ]


class CsCsr(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int32),
        ("n_cols", C.c_int32),
        ("nnz", C.c_int64),
        ("d_indptr", C.c_void_p),
        ("d_indices", C.c_void_p),
        ("d_data", C.c_void_p),
        ("dtype", C.c_int32),
        # view extensions (include/chromosight_hip.h): zero / NULL = plain CSR
        ("col0", C.c_int32),
        ("d_row_end", C.c_void_p),
        ("d_row_weight", C.c_void_p),
        ("d_col_weight", C.c_void_p),
    ]


class CsStageBlock(C.Structure):
    _fields_ = [
        ("row0", C.c_int64),
        ("n", C.c_int32),
        ("keep", C.c_int32),
        ("layout", C.c_int32),
        ("band_w", C.c_int32),
        ("ld", C.c_int64),
        ("d_band64", C.c_void_p),
        ("d_band32", C.c_void_p),
        ("d_law", C.c_void_p),
        ("ld64", C.c_int64),
        ("f64_diags", C.c_int32),
        ("band32_counts", C.c_int32),
        ("d_lazy", C.c_void_p),
    ]


class CsFociParams(C.Structure):
    _fields_ = [
        ("pearson", C.c_double),
        ("rescore_margin", C.c_double),
        ("min_size", C.c_int32),
        ("diag_only", C.c_int32),
        ("lo_diag", C.c_int32),
        ("hi_diag", C.c_int32),
        ("inter", C.c_int32),
        ("want_windows", C.c_int32),
        ("exclusive", C.c_int32),
        ("reserved", C.c_int32),
    ]


class CsFocus(C.Structure):
    _fields_ = [
        ("bin1", C.c_int32),
        ("bin2", C.c_int32),
        ("inside", C.c_int32),
        ("n_zero", C.c_int32),
        ("n_missing", C.c_int32),
        ("focus_size", C.c_int32),
        ("score", C.c_double),
        ("n_obs", C.c_double),
        ("pval", C.c_double),
    ]


class CsCall(C.Structure):
    """One entry of a cs_run_calls list (include/chromosight_hip.h)."""
    _fields_ = [
        ("fn", C.c_int32),
        ("lane", C.c_int32),
        ("after", C.c_int32),
        ("rc", C.c_int32),
        ("p", C.c_void_p * 12),
        ("i", C.c_int64 * 6),
        ("d", C.c_double * 2),
    ]


CALL_STAGE_BLOCKS, CALL_EVENT_RECORD, CALL_STREAM_WAIT_EVENT, CALL_DETECT_FOCI_BLOCKS, CALL_DETECT_FOCI_BATCH_TEMPLATES, \
    CALL_ACCEPT_RECORDS, CALL_DETECT_FOCI_BATCH_FINISH, CALL_STREAM_WAIT_TILES = 1, 2, 3, 4, 5, 6, 7, 9

# While a list is installed here, the entries named in _CAPTURED append (name, arguments) to it AFTER running as usual:
# chromosight_amd/plan.py turns the calls of one genome step into a cs_run_calls list.
CAPTURE = None
_CAPTURED = ("cs_stage_blocks", "cs_event_record", "cs_stream_wait_event", "cs_detect_foci_blocks", "cs_detect_foci_batch_templates")


# numpy view of an array of cs_focus records
FOCUS_DTYPE = np.dtype([("bin1", "<i4"), ("bin2", "<i4"), ("inside", "<i4"), ("n_zero", "<i4"), ("n_missing", "<i4"),
                        ("focus_size", "<i4"), ("score", "<f8"), ("n_obs", "<f8"), ("pval", "<f8")])


# name -> (restype, argtypes); every symbol include/chromosight_hip.h declares
_PROTOTYPES = {
    "cs_version": (C.c_char_p, []),
    "cs_last_kernel": (C.c_int, [C.c_void_p]),
    "cs_ctx_set_range_check": (C.c_int, [C.c_void_p, C.c_int32]),
    "cs_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "cs_ctx_destroy": (None, [C.c_void_p]),
    "cs_last_error": (C.c_char_p, [C.c_void_p]),
    "cs_device_cu_count": (C.c_int, [C.c_void_p]),
    "cs_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "cs_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cs_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "cs_stream_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cs_stream_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cs_stream_create_priority": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "cs_stream_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cs_event_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cs_event_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cs_event_record": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_stream_wait_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "cs_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "cs_normxcorr2": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsMatrix), C.POINTER(CsKernel),
                                C.POINTER(CsNormxcorr2Params), C.POINTER(CsMatrix), C.POINTER(CsMatrix)]),
    "cs_xcorr2": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsMatrix), C.c_int32, C.c_int32,
                            C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_double, C.c_int32,
                            C.POINTER(CsMatrix)]),
    "cs_rescore_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsMatrix), C.POINTER(CsKernel),
                                 C.POINTER(CsNormxcorr2Params), C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p]),
    "cs_compact_ge": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsMatrix), C.c_int32, C.c_int32,
                                C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_void_p]),
    "cs_distance_law_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsCsr), C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p]),
    "cs_detrend_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsCsr), C.c_void_p, C.c_int32,
                                 C.c_double, C.c_void_p]),
    "cs_csr_to_band": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsCsr), C.c_void_p, C.c_int32,
                                 C.c_double, C.POINTER(CsMatrix)]),
    "cs_csr_band_extent": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsCsr), C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p]),
    "cs_distance_law_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "cs_detect_foci_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(CsKernel),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cs_comm_available": (C.c_int, []),
    "cs_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cs_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cs_comm_destroy": (None, [C.c_void_p]),
    "cs_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "cs_comm_rank": (C.c_int, [C.c_void_p]),
    "cs_comm_world": (C.c_int, [C.c_void_p]),
    "cs_comm_allgather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "cs_comm_allgather_rows_once": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "cs_comm_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "cs_stage_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsCsr), C.POINTER(CsStageBlock), C.c_int32, C.c_double]),
    "cs_csr_median": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsCsr), C.POINTER(C.c_double)]),
    "cs_csr_median_many": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsCsr), C.c_int32, C.POINTER(C.c_double)]),
    "cs_detect_foci": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsMatrix), C.POINTER(CsKernel),
                                 C.POINTER(CsNormxcorr2Params), C.POINTER(CsFociParams), C.c_void_p, C.c_int64,
                                 C.POINTER(C.c_int64), C.c_void_p]),
    "cs_detect_foci_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(CsKernel), C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cs_detect_foci_batch_templates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cs_quantify_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsMatrix), C.POINTER(CsKernel),
                                     C.POINTER(CsNormxcorr2Params), C.POINTER(CsFociParams), C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_void_p]),
    "cs_detect_foci_batch_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_run_calls": (C.c_int, [C.POINTER(CsCall), C.c_int32]),
    "cs_quantify_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(CsMatrix), C.POINTER(CsKernel),
                                     C.POINTER(CsNormxcorr2Params), C.POINTER(CsFociParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_void_p]),
    "cs_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CsMatrix), C.POINTER(CsKernel),
                                C.POINTER(CsNormxcorr2Params), C.POINTER(CsFociParams), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "cs_label_foci": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                C.POINTER(C.c_int64)]),
    "cs_normxcorr2_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.POINTER(CsKernel),
                                     C.POINTER(CsNormxcorr2Params), C.c_void_p, C.c_int32, C.c_int64]),
    "cs_remove_neighbours": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "cs_accept_records": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "cs_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
}

ABI_SYMBOLS = tuple(sorted(_PROTOTYPES))

_lib = None
_lib_lock = threading.Lock()


class HipLibraryError(RuntimeError):
    """libchromosight_hip.so is missing, or a HIP call failed."""


def load_library():
    """dlopen the in-tree shared library and declare the prototypes (no GPU needed)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not _LIB_PATH.exists():
            raise HipLibraryError(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C chromosight_amd/csrc`. There is no CPU fallback."
            )
        lib = C.CDLL(str(_LIB_PATH))
        for name, (restype, argtypes) in _PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        for name in _CAPTURED:
            setattr(lib, name, _capturing(name, getattr(lib, name)))
        _lib = lib
        return lib


def _capturing(name, fn):
    def call(*args):
        rc = fn(*args)
        cap = CAPTURE
        if cap is not None:
            cap.append((name, args, threading.get_ident()))
        return rc
    call.__name__ = name
    return call


def raw_arg(arg):
    """The integer / float a ctypes call would pass for `arg` (None, int, float, byref(x), array, structure, c_void_p)."""
    if arg is None:
        return 0
    if isinstance(arg, (int, float)):
        return arg
    if hasattr(arg, "_obj"):                      # byref(x)
        return C.addressof(arg._obj)
    if isinstance(arg, (C.Array, C.Structure)):
        return C.addressof(arg)
    if isinstance(arg, C.c_void_p):
        return arg.value or 0
    return C.cast(arg, C.c_void_p).value or 0


def np_dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return CS_F32
    if dtype == np.float64:
        return CS_F64
    raise TypeError(f"unsupported dtype {dtype}")


class DeviceBuffer:
    """A typed HBM allocation owned by a Device (freed on garbage collection)."""

    def __init__(self, device, shape, dtype):
        self.device = device
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        ptr = C.c_void_p()
        device._check(device.lib.cs_malloc(device.ctx, max(self.nbytes, 1), C.byref(ptr)))
        self.ptr = ptr.value

    def __del__(self):
        try:
            if getattr(self, "ptr", None) and self.device.ctx:
                self.device.lib.cs_free(self.device.ctx, self.ptr)
        except Exception:
            pass
        self.ptr = None

    def upload(self, array, stream=None):
        array = np.ascontiguousarray(array, dtype=self.dtype)
        if array.nbytes != self.nbytes:
            raise ValueError("size mismatch in upload")
        self.device._check(self.device.lib.cs_memcpy_h2d(
            self.device.ctx, self.ptr, array.ctypes.data, array.nbytes, stream))
        # pageable host memory: the runtime has consumed `array` when the call returns
        return self

    def download(self, stream=None, pinned=False):
        out = self.device.pinned_result(self.shape, self.dtype) if pinned else np.empty(self.shape, dtype=self.dtype)
        self.device._check(self.device.lib.cs_memcpy_d2h(
            self.device.ctx, out.ctypes.data, self.ptr, self.nbytes, stream))
        return out

    def zero(self, stream=None):
        self.device._check(self.device.lib.cs_memset(self.device.ctx, self.ptr, 0, self.nbytes, stream))
        return self


class PinnedArray:
    """Page-locked host buffer exposed as a numpy array; freed with the array's last reference."""

    def __init__(self, device, shape, dtype):
        shape = tuple(int(x) for x in np.atleast_1d(shape))
        dtype = np.dtype(dtype)
        nbytes = max(int(np.prod(shape, dtype=np.int64)) * dtype.itemsize, 1)
        ptr = C.c_void_p()
        device._check(device.lib.cs_host_alloc(device.ctx, nbytes, C.byref(ptr)))
        self.device, self.ptr = device, ptr.value
        raw = (C.c_char * nbytes).from_address(self.ptr)
        raw._pinned_owner = self          # keeps this object (and the allocation) alive with the view
        self.array = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)

    def __del__(self):
        try:
            if getattr(self, "ptr", None) and self.device.ctx:
                self.device.lib.cs_host_free(self.device.ctx, self.ptr)
        except Exception:
            pass
        self.ptr = None


class _PinnedLease:
    """Returns a pinned block to its pool when the numpy array built on it dies."""

    def __init__(self, block, free_list):
        self.block, self.free_list = block, free_list

    def __del__(self):
        try:
            if len(self.free_list) < 4:
                self.free_list.append(self.block)
        except Exception:
            pass


# every live Device by its context handle: who holds a raw context (a recorded call list, plan.StepPlan) finds the object whose
# lock serialises its use
_DEVICES_BY_CTX = weakref.WeakValueDictionary()


def device_of(ctx):
    """The Device that owns the context handle `ctx`, or None."""
    return _DEVICES_BY_CTX.get(int(ctx)) if ctx else None


class Device:
    """One context on one GPU (one per process in multi-GPU runs)."""

    def __init__(self, index=0):
        self.lib = load_library()
        self.index = int(index)
        # a context serves one call in flight: the engine's entry points hold this while they use it (callers that want
        # concurrency on one GPU take one Device per host thread, as the genome drivers do)
        self.lock = threading.RLock()
        ctx = C.c_void_p()
        rc = self.lib.cs_ctx_create(self.index, C.byref(ctx))
        if rc != 0:
            self.ctx = None
            raise HipLibraryError(
                f"cs_ctx_create(device={index}) failed with status {rc}: no usable HIP device. "
                "This package has no CPU fallback."
            )
        self.ctx = ctx.value
        _DEVICES_BY_CTX[self.ctx] = self

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.cs_ctx_destroy(self.ctx)
        except Exception:
            pass
        self.ctx = None

    # -- helpers --------------------------------------------------------------------------
    def _check(self, rc):
        if rc == 0:
            return
        msg = self.lib.cs_last_error(self.ctx).decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        if rc == -3:
            raise NotImplementedError(msg)
        raise HipLibraryError(f"status {rc}: {msg}")

    @property
    def cu_count(self):
        return self.lib.cs_device_cu_count(self.ctx)

    def empty(self, shape, dtype):
        return DeviceBuffer(self, shape, dtype)

    def zeros(self, shape, dtype, stream=None):
        return DeviceBuffer(self, shape, dtype).zero(stream)

    def to_device(self, array, dtype=None, stream=None):
        array = np.ascontiguousarray(array, dtype=dtype)
        return DeviceBuffer(self, array.shape, array.dtype).upload(array, stream)

    def sync(self, stream=None):
        self._check(self.lib.cs_stream_sync(self.ctx, stream))

    def pinned_empty(self, shape, dtype):
        """numpy array over page-locked host memory (cs_host_alloc): D2H / H2D at link speed."""
        return PinnedArray(self, shape, dtype).array

    def pinned_result(self, shape, dtype):
        """Like pinned_empty, from a pool: page-locking costs more than the copy it speeds up, so the
        buffer of a result array goes back to the pool when the array is garbage collected and the
        next call of the same size reuses it."""
        dtype = np.dtype(dtype)
        shape = tuple(int(x) for x in np.atleast_1d(shape))
        nbytes = max(int(np.prod(shape, dtype=np.int64)) * dtype.itemsize, 1)
        pool = self.__dict__.setdefault("_pinned_pool", {})
        free = pool.setdefault(nbytes, [])
        block = free.pop() if free else PinnedArray(self, nbytes, np.uint8)
        lease = _PinnedLease(block, free)
        raw = (C.c_char * nbytes).from_address(block.ptr)
        raw._lease = lease
        return np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)

    def new_stream(self, high_priority=False):
        s = C.c_void_p()
        if high_priority:
            self._check(self.lib.cs_stream_create_priority(self.ctx, 1, C.byref(s)))
        else:
            self._check(self.lib.cs_stream_create(self.ctx, C.byref(s)))
        return s.value

    def stream_wait_tiles(self, stream, tiles_dev, epoch, timeout_us=1000):
        """cs_stream_wait_tiles: work enqueued on `stream` afterwards starts when the tile workgroups of tiles_dev's tile launch with
        this epoch are resident (or after the time-out)."""
        self._check(self.lib.cs_stream_wait_tiles(self.ctx, stream, tiles_dev.ctx, int(epoch), int(timeout_us)))

    def new_event(self):
        e = C.c_void_p()
        self._check(self.lib.cs_event_create(self.ctx, C.byref(e)))
        return e.value

    def record(self, event, stream=None):
        self._check(self.lib.cs_event_record(self.ctx, event, stream))

    def wait_event(self, event, stream=None):
        """Later work on `stream` waits for `event` (device-side, no host synchronisation)."""
        self._check(self.lib.cs_stream_wait_event(self.ctx, stream, event))

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        self._check(self.lib.cs_event_elapsed_ms(self.ctx, start, stop, C.byref(ms)))
        return float(ms.value)


_devices = {}


def get_device(index=None):
    """Process-wide Device for `index` (default: $CHROMOSIGHT_HIP_DEVICE, $LOCAL_RANK or 0)."""
    if index is None:
        index = int(os.environ.get("CHROMOSIGHT_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    dev = _devices.get(index)
    if dev is None:
        dev = Device(index)
        _devices[index] = dev
    return dev


def dense_matrix(buf, ld=None):
    rows, cols = buf.shape
    return CsMatrix(buf.ptr, np_dtype_code(buf.dtype), LAYOUT_DENSE, cols if ld is None else ld, 0, 0)


def band_matrix(buf, band_lo, band_w):
    rows, ld = buf.shape
    return CsMatrix(buf.ptr, np_dtype_code(buf.dtype), LAYOUT_BAND, ld, int(band_lo), int(band_w))
