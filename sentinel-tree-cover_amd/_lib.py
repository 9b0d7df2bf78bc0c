"""ctypes binding of libttc_hip.so (include/ttc.h) -- the stub a maintainer of the
reference would add next to src/download_and_predict_job.py.

PyTorch-ROCm is used only as the device allocator / stream provider: tensors are passed
to the library as raw device pointers.  There is NO CPU fallback: if the shared library
is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTC_LIB", os.path.join(_HERE, "libttc_hip.so"))      # TTC_LIB: experiment builds only

EXPORTS = [
    "ttc_version", "ttc_config_size", "ttc_create", "ttc_destroy", "ttc_last_error", "ttc_device_bytes",
    "ttc_load_weights", "ttc_load_dsen2_weights", "ttc_forward_windows", "ttc_process_subtiles",
    "ttc_tile_missing_counts", "ttc_tile_fix_missing", "ttc_mosaic", "ttc_dsen2_forward",
    "ttc_superresolve_tile", "ttc_upsample_20m", "ttc_debug_fetch", "ttc_debug_timing",
    "ttc_debug_kernel_ms", "ttc_feather", "ttc_aligned_mosaic", "ttc_remove_cloud_and_shadows",
    "ttc_u16_to_float", "ttc_float_to_u16", "ttc_s1_to_db", "ttc_forward_taps", "ttc_float_to_int16", "ttc_mosaic_features", "ttc_debug_keep", "ttc_identify_clouds_shadows",
    "ttc_sen2cor_clean", "ttc_median5", "ttc_snow_map", "ttc_merge_cloud_masks", "ttc_count_positive", "ttc_clip01", "ttc_divide",
    "ttc_border_subtiles", "ttc_seam_adjust", "ttc_reseg_mosaic", "ttc_smooth_strip", "ttc_superresolve_windows", "ttc_count_equal", "ttc_write_geotiff_u8",
    "ttc_predict_tile", "ttc_read_hkl", "ttc_read_hkl_error",
    "ttc_create_v2", "ttc_predict_tile_shaped", "ttc_adjust_shape", "ttc_debug_kernel_flops", "ttc_calibrate_precision",
]

# exported for tools/probes and the detector-stage tests, declared in csrc/ttc_internal.h -- not part of the drop-in surface (include/ttc.h)
INTERNAL_EXPORTS = ["ttc_debug_clouds_stage", "ttc_debug_knob", "ttc_debug_check_guards"]
GUARD_STATS = {"contexts_checked": 0, "buffers_checked": 0, "violations": []}      # TTC_GUARD runs (tools/run_guarded_gpu_tests.sh)

SAMPLER_FN = C.CFUNCTYPE(C.c_int64, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_void_p)


class TTCConfig(C.Structure):
    _fields_ = [("win_in", C.c_int32), ("length", C.c_int32), ("max_windows", C.c_int32),
                ("n_bands", C.c_int32), ("hidden", C.c_int32), ("base_filters", C.c_int32),
                ("zoneout", C.c_float), ("precision", C.c_int32), ("win_rows", C.c_int32),
                ("one_term_layers", C.c_uint32), ("fp32_conv_form", C.c_int32), ("dsen2_precision", C.c_int32), ("two_term_layers", C.c_uint32)]


# ttc_config.precision values.  The 16-bit engine multiplies three split products in every layer by default
# (one_term_layers = 0).  Measured with as-stored-scale kernels: running only the ConvGRU gates conv (bit 0) with plain fp16
# operands keeps max |dprob| at 2e-4 (L = 4) .. 6.5e-4 (L = 12) on white-noise windows, but reaches 3.0e-3 on a real
# (spatially smooth) 618^2 tile, outside the 1e-3 contract -- so no layer runs one product unless the caller asks for it.
PRECISIONS = {"fp32": 0, "fp16": 2, "bf16": 3}     # 1 / 4 were the retired bf16x3 / fp32-blocked engines (csrc/experiments/)


class TTCTileShapes(C.Structure):
    """ttc_tile_shapes: rows x cols of each raw array as stored (ttc_predict_tile_shaped)"""
    _fields_ = [("s2_10", C.c_int32 * 2), ("s2_20", C.c_int32 * 2), ("s1", C.c_int32 * 2), ("dem", C.c_int32 * 2), ("mask", C.c_int32 * 2)]


CAL_LAYERS = ("gates", "candidate", "conv_median", "conv_concat", "conv1", "conv2", "up2", "up2_out", "up3", "out")     # bits 0..9 of one_term_layers


class TTCPrecisionReport(C.Structure):
    """ttc_precision_report (ttc_calibrate_precision)"""
    _fields_ = [("one_term_layers", C.c_uint32), ("two_term_layers", C.c_uint32), ("max_dprob", C.c_float), ("dprob_all_three", C.c_float),
                ("layer_dprob_one", C.c_float * 10), ("layer_dprob_two", C.c_float * 10), ("budget", C.c_float), ("within_budget", C.c_int32),
                ("matrix_work_ratio", C.c_double), ("trials", C.c_int32), ("n_windows", C.c_int32)]


class TTCResegWindow(C.Structure):
    _fields_ = [("kind", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("pred_off", C.c_int64), ("weight_off", C.c_int64)]


class TTCTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


_lib = None


def load():
    """dlopen libttc_hip.so and declare prototypes; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sentinel-tree-cover_amd/csrc`).  There is no CPU fallback.")
    try:
        # PyTorch-ROCm wheels bundle their own HIP runtime: it has to be in the process BEFORE this library is dlopen'ed, or the library binds
        # to the system's libamdhip64 and the process ends up with two runtimes (ttc_create then fails with TTC_ERR_HIP: no device) --
        # measured with __graft_entry__.build() followed by smoke() in one process
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    P, I32, F32P, VP = C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.c_void_p
    lib.ttc_version.restype = C.c_char_p
    lib.ttc_config_size.restype = C.c_size_t
    if lib.ttc_config_size() != C.sizeof(TTCConfig):       # a stale .so (or binding): fail loudly instead of handing over a short struct
        raise RuntimeError(f"{LIB_PATH}: ttc_config is {lib.ttc_config_size()} bytes in the library, {C.sizeof(TTCConfig)} in this binding -- "
                           "rebuild the library (`make -C sentinel-tree-cover_amd/csrc`)")
    lib.ttc_create.argtypes = [C.POINTER(P), I32, C.POINTER(TTCConfig)]
    lib.ttc_create_v2.argtypes = [C.POINTER(P), I32, C.POINTER(TTCConfig), C.c_size_t]
    lib.ttc_destroy.argtypes = [P]
    lib.ttc_destroy.restype = None
    lib.ttc_last_error.argtypes = [P]
    lib.ttc_last_error.restype = C.c_char_p
    lib.ttc_device_bytes.argtypes = [P]
    lib.ttc_device_bytes.restype = C.c_size_t
    lib.ttc_load_weights.argtypes = [P, C.POINTER(TTCTensor), I32]
    lib.ttc_load_dsen2_weights.argtypes = [P, C.POINTER(TTCTensor), I32]
    lib.ttc_forward_windows.argtypes = [P, VP, I32, VP, VP]
    lib.ttc_process_subtiles.argtypes = [P, VP, I32, I32, I32, F32P, C.POINTER(C.c_int32), VP, VP, VP,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), I32, I32, VP, VP, VP]
    lib.ttc_tile_missing_counts.argtypes = [P, VP, I32, I32, I32, VP, VP]
    lib.ttc_tile_fix_missing.argtypes = [P, VP, I32, I32, I32, I32, I32, VP]
    lib.ttc_mosaic.argtypes = [P, VP, I32, C.POINTER(C.c_int32), I32, I32, I32, VP, VP, VP]
    lib.ttc_dsen2_forward.argtypes = [P, VP, VP, I32, I32, I32, VP, VP]
    lib.ttc_superresolve_tile.argtypes = [P, VP, I32, I32, I32, I32, VP]
    lib.ttc_upsample_20m.argtypes = [P, VP, VP, I32, I32, I32, VP, VP]
    lib.ttc_feather.argtypes = [P, VP, I32, I32, I32, I32, I32, VP, VP]
    lib.ttc_aligned_mosaic.argtypes = [P, VP, VP, I32, I32, I32, VP, VP]
    lib.ttc_remove_cloud_and_shadows.argtypes = [P, VP, VP, VP, I32, I32, I32, SAMPLER_FN, VP, VP, VP,
                                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), VP]
    lib.ttc_sen2cor_clean.argtypes = [P, VP, I32, I32, I32, VP, VP]
    lib.ttc_median5.argtypes = [P, VP, I32, I32, VP, VP]
    lib.ttc_snow_map.argtypes = [P, VP, I32, I32, I32, VP, VP, VP]
    lib.ttc_merge_cloud_masks.argtypes = [P, VP, VP, VP, C.c_int64, VP]
    lib.ttc_count_positive.argtypes = [P, VP, I32, I32, VP, VP]
    lib.ttc_clip01.argtypes = [P, VP, C.c_int64, VP]
    lib.ttc_divide.argtypes = [P, VP, C.c_int64, C.c_float, VP]
    lib.ttc_debug_keep.argtypes = [P, I32]
    lib.ttc_debug_clouds_stage.argtypes = [P, I32]
    lib.ttc_identify_clouds_shadows.argtypes = [P, VP, I32, I32, I32, VP, VP, VP, VP, VP, VP, VP]
    lib.ttc_mosaic_features.argtypes = [P, VP, I32, VP, I32, I32, I32, I32, VP, VP]
    lib.ttc_forward_taps.argtypes = [P, VP, I32, VP, VP, VP, VP]
    lib.ttc_float_to_int16.argtypes = [P, VP, C.c_int64, C.c_float, VP, VP]
    lib.ttc_u16_to_float.argtypes = [P, VP, C.c_int64, VP, VP]
    lib.ttc_float_to_u16.argtypes = [P, VP, C.c_int64, VP, VP]
    lib.ttc_s1_to_db.argtypes = [P, VP, I32, I32, I32, VP, VP]
    F64P, I32P = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.ttc_predict_tile.argtypes = [P, VP, VP, VP, VP, VP, VP, VP, I32, I32, I32, F64P, F64P, I32, I32, VP, VP, VP, VP, VP]
    lib.ttc_predict_tile_shaped.argtypes = [P, VP, VP, VP, VP, VP, VP, VP, I32, C.POINTER(TTCTileShapes), F64P, F64P, I32, I32, VP, VP, VP, VP, VP]
    lib.ttc_adjust_shape.argtypes = [P, VP, I32, I32, I32, I32, I32, I32, VP, VP]
    lib.ttc_calibrate_precision.argtypes = [P, P, VP, I32, C.c_float, C.POINTER(TTCPrecisionReport), VP]
    lib.ttc_border_subtiles.argtypes = [P, VP, VP, VP, I32, I32P, I32, F32P, F32P, I32, I32, VP, F32P, I32P, VP]
    lib.ttc_smooth_strip.argtypes = [P, VP, I32, I32, I32, F32P, VP, VP]
    lib.ttc_superresolve_windows.argtypes = [P, VP, I32, I32, I32, I32, I32, I32, VP]
    lib.ttc_seam_adjust.argtypes = [P, VP, I32, I32, I32, F32P, VP]
    lib.ttc_reseg_mosaic.argtypes = [P, VP, VP, I32, VP, VP, I32, I32, VP, VP, VP]
    lib.ttc_count_equal.argtypes = [P, VP, I32, I32, C.c_float, I32P, VP]
    lib.ttc_debug_fetch.argtypes = [P, C.c_char_p, F32P, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.ttc_debug_timing.argtypes = [P, I32]
    lib.ttc_debug_knob.argtypes = [I32, I32]
    lib.ttc_debug_check_guards.argtypes = [P, C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]
    lib.ttc_debug_kernel_ms.argtypes = [P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.ttc_debug_kernel_flops.argtypes = [P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    for name in EXPORTS:
        fn = getattr(lib, name)          # AttributeError here == missing export
        if name not in ("ttc_version", "ttc_destroy", "ttc_last_error", "ttc_device_bytes", "ttc_read_hkl_error"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def write_geotiff_u8(path, raster, west, south, east, north):
    """ttc_write_geotiff_u8: raster uint8 [rows, cols] (numpy) -> LZW GeoTIFF at `path` (host-side, no GPU needed)"""
    a = np.ascontiguousarray(raster, dtype=np.uint8)
    lib = load()
    lib.ttc_write_geotiff_u8.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]
    st = lib.ttc_write_geotiff_u8(str(path).encode(), a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], float(west), float(south),
                                  float(east), float(north))
    if st != 0:
        raise RuntimeError(f"ttc_write_geotiff_u8: status {st} ({path})")
    return str(path)


def read_hkl(path, name=None, alloc=None):
    """ttc_read_hkl: the numeric array of a hickle file (hkl.load for the raw folder's arrays, job.py:684-714) -> numpy array.
    Host-side, no GPU needed.  alloc(shape, dtype) -> C-contiguous numpy array to fill (e.g. a view of page-locked memory, so that
    the array can be uploaded without a staging copy: job.PinnedArena); default np.empty."""
    lib = load()
    lib.ttc_read_hkl.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ttc_read_hkl_error.restype = C.c_char_p
    shape = (C.c_int64 * 8)()
    nd, es, tc, sg = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    nm = name.encode() if name else None

    def call(buf, cap):
        st = lib.ttc_read_hkl(str(path).encode(), nm, buf, cap, shape, C.byref(nd), C.byref(es), C.byref(tc), C.byref(sg))
        if st != 0:
            raise RuntimeError(f"ttc_read_hkl: status {st}: {lib.ttc_read_hkl_error().decode()}")
    call(None, 0)
    kind = "f" if tc.value == 1 else ("i" if sg.value else "u")
    shp, dt = tuple(shape[i] for i in range(nd.value)), np.dtype(f"<{kind}{es.value}")
    out = np.empty(shp, dtype=dt) if alloc is None else alloc(shp, dt)
    if out.shape != shp or out.dtype != dt or not out.flags["C_CONTIGUOUS"]:
        raise ValueError("read_hkl: alloc() must return a C-contiguous array of the requested shape and dtype")
    call(out.ctypes.data_as(C.c_void_p), out.nbytes)
    return out


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible to PyTorch-ROCm (torch.cuda.is_available() is False); "
                           "the tree-cover hot path has no CPU fallback")
    return torch


def pack_tensors(weights: dict):
    """dict name -> float32 ndarray  ==>  (ctypes array of TTCTensor, keepalive list)."""
    arr = (TTCTensor * len(weights))()
    keep = []
    for i, (k, v) in enumerate(weights.items()):
        a = np.ascontiguousarray(v, dtype=np.float32)
        keep.append(a)
        arr[i].name = k.encode()
        arr[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].ndim = min(a.ndim, 4)
        for d in range(min(a.ndim, 4)):
            arr[i].shape[d] = a.shape[d]
    return arr, keep


class Context:
    """One libttc context: (device, window geometry, weights, workspace)."""

    def __init__(self, win_in=172, length=4, max_windows=36, device=0, zoneout=0.75, precision=0, win_rows=0,
                 one_term_layers=None, fp32_conv_form=0, dsen2_precision=None, two_term_layers=0):
        self.lib = load()
        self.torch = _torch()
        precision = PRECISIONS.get(precision, precision)
        if one_term_layers is None:
            one_term_layers = 0                  # every layer multiplies three split products (ttc.h)
        if isinstance(dsen2_precision, str):
            dsen2_precision = PRECISIONS[dsen2_precision]
        self.cfg = TTCConfig(win_in, length, max_windows, 17, 32, 64, zoneout, precision, win_rows, one_term_layers, int(fp32_conv_form),
                             int(dsen2_precision or 0), int(two_term_layers or 0))
        self.device = device
        self._h = C.c_void_p()
        st = self.lib.ttc_create_v2(C.byref(self._h), device, C.byref(self.cfg), C.sizeof(self.cfg))
        if st != 0:
            msg = self.lib.ttc_last_error(self._h).decode() if self._h else "ttc_create failed"
            if self._h:
                self.lib.ttc_destroy(self._h)
                self._h = C.c_void_p()
            raise RuntimeError(f"ttc_create: status {st}: {msg}")

    # -- helpers ------------------------------------------------------------------------
    def _check(self, st, what):
        if st != 0:
            raise RuntimeError(f"{what}: status {st}: {self.lib.ttc_last_error(self._h).decode()}")

    def _dev(self, x, dtype=None):
        """numpy / torch -> contiguous tensor on this context's device."""
        t = self.torch
        if not isinstance(x, t.Tensor):
            x = t.from_numpy(np.ascontiguousarray(x))
        if dtype is not None and x.dtype != dtype:
            x = x.to(dtype)
        return x.to(f"cuda:{self.device}", non_blocking=False).contiguous()

    def keep_intermediates(self, on=True):
        self._check(self.lib.ttc_debug_keep(self._h, 1 if on else 0), "ttc_debug_keep")

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def check_guards(self):
        """TTC_GUARD=<KiB> runs only: scan the guard zones around every device buffer the context owns -> (bytes overwritten, first offender)"""
        n, msg = C.c_int64(0), C.create_string_buffer(400)
        self._check(self.lib.ttc_debug_check_guards(self._h, C.byref(n), msg, 400), "ttc_debug_check_guards")
        return int(n.value), msg.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            if os.environ.get("TTC_GUARD", "0") not in ("", "0"):
                try:                                             # a device-memory check of everything this context ran (csrc/ttc_internal.h)
                    n, msg = self.check_guards()
                    GUARD_STATS["contexts_checked"] += 1
                    if n:
                        GUARD_STATS["violations"].append(msg)
                finally:
                    pass
            self.lib.ttc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self):
        return int(self.lib.ttc_device_bytes(self._h))

    # -- weights ------------------------------------------------------------------------
    def load_weights(self, weights: dict):
        arr, keep = pack_tensors(weights)
        self._check(self.lib.ttc_load_weights(self._h, arr, len(weights)), "ttc_load_weights")

    def load_dsen2_weights(self, weights: dict):
        arr, keep = pack_tensors(weights)
        self._check(self.lib.ttc_load_dsen2_weights(self._h, arr, len(weights)), "ttc_load_dsen2_weights")

    # -- model ----------------------------------------------------------------------------
    def forward_windows(self, x, out=None):
        """x [n, L+1, H, W, 17] float32 (numpy or cuda tensor) -> cuda tensor [n, H-14, W-14]; H = win_rows or W."""
        t = self.torch
        xd = self._dev(x, t.float32)
        n, W = xd.shape[0], self.cfg.win_in
        H = self.cfg.win_rows or W
        assert tuple(xd.shape[1:]) == (self.cfg.length + 1, H, W, 17), xd.shape
        if out is None:
            out = t.empty((n, H - 14, W - 14), dtype=t.float32, device=xd.device)
        self._check(self.lib.ttc_forward_windows(self._h, C.c_void_p(xd.data_ptr()), n,
                                                 C.c_void_p(out.data_ptr()), self._stream()),
                    "ttc_forward_windows")
        return out

    def calibrate_precision(self, ref_ctx, windows, budget=5e-4):
        """ttc_calibrate_precision: windows [n, L+1, H, W, 17] (numpy / cuda: REAL model inputs) through `ref_ctx` (fp32 Context, same
        geometry, same weights) and through candidate product maps on this 16-bit context; the cheapest map within `budget` of the fp32
        probabilities stays applied.  -> dict (one_term_layers, two_term_layers, max_dprob, per-layer table, matrix_work_ratio ...)"""
        t = self.torch
        xd = self._dev(windows, t.float32)
        W, H = self.cfg.win_in, self.cfg.win_rows or self.cfg.win_in
        if xd.dim() != 5 or tuple(xd.shape[1:]) != (self.cfg.length + 1, H, W, 17):
            raise ValueError(f"calibrate_precision: windows must be [n, {self.cfg.length + 1}, {H}, {W}, 17], got {tuple(xd.shape)}")
        rep = TTCPrecisionReport()
        self._check(self.lib.ttc_calibrate_precision(self._h, ref_ctx._h, C.c_void_p(xd.data_ptr()), int(xd.shape[0]), float(budget),
                                                     C.byref(rep), self._stream()), "ttc_calibrate_precision")
        self.cfg.one_term_layers, self.cfg.two_term_layers = rep.one_term_layers | (self.cfg.one_term_layers & ~0x3FF), rep.two_term_layers
        terms = {name: (1 if (rep.one_term_layers >> i) & 1 else (2 if (rep.two_term_layers >> i) & 1 else 3)) for i, name in enumerate(CAL_LAYERS)}
        return {"one_term_layers": int(rep.one_term_layers), "two_term_layers": int(rep.two_term_layers), "products_per_layer": terms,
                "max_dprob": float(rep.max_dprob), "dprob_all_three": float(rep.dprob_all_three), "budget": float(rep.budget),
                "within_budget": bool(rep.within_budget), "matrix_work_ratio": float(rep.matrix_work_ratio), "trials": int(rep.trials),
                "n_windows": int(rep.n_windows),
                "layer_alone_one_product": {name: float(rep.layer_dprob_one[i]) for i, name in enumerate(CAL_LAYERS)},
                "layer_alone_two_products": {name: float(rep.layer_dprob_two[i]) for i, name in enumerate(CAL_LAYERS)}}

    # -- per-tile core -------------------------------------------------------------------
    def tile_fix_missing(self, s2d, do_nan=True, do_zero_one=False):
        """in place on a cuda tensor [T, X, Y, 10]"""
        T, X, Y = (int(v) for v in s2d.shape[:3])
        self._check(self.lib.ttc_tile_fix_missing(self._h, C.c_void_p(s2d.data_ptr()), T, X, Y, int(do_nan),
                                                  int(do_zero_one), self._stream()), "ttc_tile_fix_missing")
        return s2d

    def tile_missing_counts(self, s2d):
        t = self.torch
        T, X, Y = (int(v) for v in s2d.shape[:3])
        counts = t.zeros(T, dtype=t.int32, device=s2d.device)
        self._check(self.lib.ttc_tile_missing_counts(self._h, C.c_void_p(s2d.data_ptr()), T, X, Y,
                                                     C.c_void_p(counts.data_ptr()), self._stream()),
                    "ttc_tile_missing_counts")
        return counts.cpu().numpy()

    def process_subtiles(self, s2, wmat, keep, interp, s1, dem, min_all, max_all, size, n_dates_ok, want_raw=False):
        """-> (windows [n, size, size] cuda float32, raw or None).  Window order = job.py iteration order."""
        t = self.torch
        s2d, itp, s1d, demd = (self._dev(a, t.float32) for a in (s2, interp, s1, dem))
        T, X, Y = (int(v) for v in s2d.shape[:3])
        assert tuple(itp.shape) == (T, X, Y) and tuple(s1d.shape) == (12, X, Y, 2) and tuple(demd.shape) == (X, Y)
        w = np.ascontiguousarray(wmat, dtype=np.float32)
        assert w.shape == (12, T)
        k = np.ascontiguousarray(keep, dtype=np.int32)
        mn = np.ascontiguousarray(min_all, dtype=np.float64)
        mx = np.ascontiguousarray(max_all, dtype=np.float64)
        nwin = 36
        out = t.empty((nwin, size, size), dtype=t.float32, device=s2d.device)
        raw = t.empty((nwin, size, size), dtype=t.float32, device=s2d.device) if want_raw else None
        self._check(self.lib.ttc_process_subtiles(
            self._h, C.c_void_p(s2d.data_ptr()), T, X, Y, w.ctypes.data_as(C.POINTER(C.c_float)),
            k.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(itp.data_ptr()), C.c_void_p(s1d.data_ptr()),
            C.c_void_p(demd.data_ptr()), mn.ctypes.data_as(C.POINTER(C.c_double)),
            mx.ctypes.data_as(C.POINTER(C.c_double)), size, n_dates_ok, C.c_void_p(out.data_ptr()),
            C.c_void_p(raw.data_ptr()) if want_raw else None, self._stream()), "ttc_process_subtiles")
        return out, raw

    TILE_DETECT, TILE_INPUTS_ONLY, TILE_NO_SUPERRES = 1, 2, 4

    @staticmethod
    def tile_needs_staged(status):
        """status: the 4 words of ttc_predict_tile (host array / sequence).  True -> one of process_tile's date-dropping rules fired
        on the device (words 0, 2, 3; word 1 is a count, not a flag): re-run the tile through the staged path."""
        return bool(int(status[0]) or int(status[2]) or int(status[3]))

    def predict_tile_raw(self, s2_10, s2_20, s1, dem, mask, dates, min_all, max_all, size, dem_m=None, flags=0,
                         want_float=False, want_inputs=False, out=None, status=None):
        """ttc_predict_tile_shaped: the whole per-tile chain in ONE enqueue, no host round trip.  s2_20 [T, h, w, 6] uint16 decides the
        tile's grid X, Y = 2h, 2w like process_tile (job.py:716-717); s2_10 [T, ~X, ~Y, 4] / s1 [12, ~X, ~Y, 2] uint16 as stored (cuda
        int16 / uint16 views or numpy) and dem [~X, ~Y] (median-filtered, /90) may be a pixel or an even number of pixels off that grid:
        they are reconciled by adjust_shape (:260-310) as an index map on the device; every array's shape is CHECKED here (dates per
        array, channel counts, the mask's [T, X, Y]) -- a raw device pointer carries none.  mask: float32 cloud + shadow mask (None with
        TILE_DETECT), dates [T] (cuda int32 tensor or sequence).
        -> (u8 cuda [Y, X] | None, f32 | None, model inputs | None, status cuda int32[4]); read `status` after a stream
        synchronisation and hand it to Context.tile_needs_staged: status[0] (a date the mosaic cannot align), status[2] (dates
        the gap-fill marks fully interpolated) or status[3] (bit 1 a date with > 10 % missing pixels, bit 2 more than 10 snowy
        dates, bit 4 a date > 90 % feathered) non-zero -> the rasters are NOT the reference's and the tile needs the staged
        calls (job.predict_tile_raw_checked does that; see ttc.h).  status[1] = dates kept by the model-side screening."""
        t = self.torch
        dev = f"cuda:{self.device}"

        def u16(a, what, last):
            if isinstance(a, t.Tensor):
                if a.dtype not in (t.int16, t.uint16):
                    raise ValueError(f"predict_tile_raw: {what} must hold the uint16 values as stored (int16 / uint16 tensor), got {a.dtype}")
                a = a.to(dev).contiguous()
            else:
                a = np.ascontiguousarray(a)
                if a.dtype not in (np.uint16, np.int16):
                    raise ValueError(f"predict_tile_raw: {what} must be the uint16 array as stored, got {a.dtype}")
                a = t.from_numpy(a.view(np.int16)).to(dev)
            if a.dim() == 3 and what != "s1":          # a single image: hkl stores [X, Y, C] (job.py:724-727)
                a = a[None]
            if a.dim() != 4 or int(a.shape[3]) != last:
                raise ValueError(f"predict_tile_raw: {what} must be [T, rows, cols, {last}], got {tuple(a.shape)}")
            return a
        d10, d20, ds1 = u16(s2_10, "s2_10", 4), u16(s2_20, "s2_20", 6), u16(s1, "s1", 2)
        T = int(d20.shape[0])
        # the 20 m stack decides the tile's grid (job.py:716-717); the other arrays are reconciled to it by adjust_shape on the device
        X, Y = 2 * int(d20.shape[1]), 2 * int(d20.shape[2])
        if int(d10.shape[0]) != T:
            raise ValueError(f"predict_tile_raw: s2_10 holds {int(d10.shape[0])} dates, s2_20 {T}")
        if int(ds1.shape[0]) != 12:
            raise ValueError(f"predict_tile_raw: s1 must hold the 12 monthly composites, got {tuple(ds1.shape)}")
        ddem = self._dev(dem, t.float32)
        dmask = self._dev(mask, t.float32) if mask is not None else None
        ddem_m = self._dev(dem_m, t.float32) if dem_m is not None else None
        if ddem.dim() != 2 or (ddem_m is not None and tuple(ddem_m.shape) != tuple(ddem.shape)):
            raise ValueError(f"predict_tile_raw: dem / dem_m must be [rows, cols] arrays of one shape, got {tuple(ddem.shape)}"
                             + (f" and {tuple(ddem_m.shape)}" if ddem_m is not None else ""))
        detect = bool(flags & self.TILE_DETECT)
        if dmask is not None and not detect and tuple(dmask.shape) != (T, X, Y):
            raise ValueError(f"predict_tile_raw: the cloud / shadow mask is {tuple(dmask.shape)}, the tile is {(T, X, Y)} "
                             "(T dates on the grid of 2 x the 20 m stack, job.py:716-717)")
        ddates = dates if isinstance(dates, t.Tensor) else t.tensor([int(d) for d in np.asarray(dates).ravel()], dtype=t.int32)
        ddates = ddates.to(dev, t.int32).contiguous()
        if int(ddates.numel()) != T:
            raise ValueError(f"predict_tile_raw: {int(ddates.numel())} dates for {T} images")
        shp = TTCTileShapes()
        for name, a in (("s2_10", d10), ("s2_20", d20), ("s1", ds1)):
            getattr(shp, name)[0], getattr(shp, name)[1] = int(a.shape[1]), int(a.shape[2])
        shp.dem[0], shp.dem[1] = int(ddem.shape[0]), int(ddem.shape[1])
        if dmask is not None:
            shp.mask[0], shp.mask[1] = int(dmask.shape[1]), int(dmask.shape[2])
        inputs_only = bool(flags & self.TILE_INPUTS_ONLY)
        # rasters are [Y, X]: transposed like load_mosaic_predictions' (job.py:1578)
        if out is not None and (tuple(out.shape) != (Y, X) or out.dtype != t.uint8 or not out.is_cuda or not out.is_contiguous()):
            raise ValueError(f"predict_tile_raw: out must be a contiguous cuda uint8 tensor [{Y}, {X}]")
        u8 = out if out is not None else (None if inputs_only else t.empty((Y, X), dtype=t.uint8, device=dev))
        f32 = t.empty((Y, X), dtype=t.float32, device=dev) if (want_float and not inputs_only) else None
        W = self.cfg.win_in
        frames = t.empty((36, self.cfg.length + 1, 17, W + 2, W + 2), dtype=t.float32, device=dev) if want_inputs else None
        status = status if status is not None else t.zeros(4, dtype=t.int32, device=dev)
        mn = (C.c_double * 17)(*[float(v) for v in min_all])
        mx = (C.c_double * 17)(*[float(v) for v in max_all])
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None      # noqa: E731
        self._check(self.lib.ttc_predict_tile_shaped(self._h, ptr(d10), ptr(d20), ptr(ds1), ptr(ddem), ptr(ddem_m), ptr(dmask), ptr(ddates),
                                                     T, C.byref(shp), mn, mx, int(size), int(flags), ptr(u8), ptr(f32), ptr(frames),
                                                     ptr(status), self._stream()), "ttc_predict_tile_shaped")
        self._keep = (d10, d20, ds1, ddem, dmask, ddem_m, ddates)      # inputs must outlive the enqueued work
        return u8, f32, frames, status

    def adjust_shape(self, a, width, height):
        """adjust_shape (job.py:260-310) on the device: float32 [T, n1, n2, C] / [T, n1, n2] / [n1, n2] -> the same rank at width x height"""
        t = self.torch
        x = self._dev(a, t.float32)
        nd = x.dim()
        x4 = x[None, :, :, None] if nd == 2 else (x[..., None] if nd == 3 else x)
        T, n1, n2, ch = (int(v) for v in x4.shape)
        if (n1, n2) == (int(width), int(height)):
            return x
        out = t.empty((T, int(width), int(height), ch), dtype=t.float32, device=x.device)
        self._check(self.lib.ttc_adjust_shape(self._h, C.c_void_p(x4.contiguous().data_ptr()), T, n1, n2, ch, int(width), int(height),
                                              C.c_void_p(out.data_ptr()), self._stream()), "ttc_adjust_shape")
        return out[0, :, :, 0] if nd == 2 else (out[..., 0] if nd == 3 else out)

    def mosaic(self, windows, xy, size, rows, cols, want_float=False):
        """windows [n, size, size] (numpy / cuda), xy [n, 2] int32 (folder_x, folder_y) -> (u8 [rows, cols], f32 | None)"""
        t = self.torch
        wd = self._dev(windows, t.float32)
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        u8 = t.empty((rows, cols), dtype=t.uint8, device=wd.device)
        f32 = t.empty((rows, cols), dtype=t.float32, device=wd.device) if want_float else None
        self._check(self.lib.ttc_mosaic(self._h, C.c_void_p(wd.data_ptr()), int(wd.shape[0]),
                                        xy.ctypes.data_as(C.POINTER(C.c_int32)), size, rows, cols,
                                        C.c_void_p(u8.data_ptr()), C.c_void_p(f32.data_ptr()) if want_float else None,
                                        self._stream()), "ttc_mosaic")
        return u8, f32

    def mosaic_features(self, feats, xy, size, depth, rows, cols):
        """feats [n, size, size, depth] int16 (numpy / cuda), xy [n, 2] int32 (folder_x, folder_y) -> cuda int16 [depth, rows, cols]"""
        t = self.torch
        fd = feats if isinstance(feats, t.Tensor) else t.from_numpy(np.ascontiguousarray(feats, dtype=np.int16))
        fd = fd.to(f"cuda:{self.device}").contiguous()
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        out = t.empty((depth, rows, cols), dtype=t.int16, device=fd.device)
        self._check(self.lib.ttc_mosaic_features(self._h, C.c_void_p(fd.data_ptr()), int(fd.shape[0]),
                                                 xy.ctypes.data_as(C.POINTER(C.c_int32)), size, depth, rows, cols,
                                                 C.c_void_p(out.data_ptr()), self._stream()), "ttc_mosaic_features")
        return out

    def forward_taps(self, x, early=True, late=True):
        """forward + feature taps: -> (probs [n, oh, ow], early [n, H, W, 64] | None, late [n, oh, ow, 64] | None), cuda float32"""
        t = self.torch
        a = self._dev(x, t.float32)
        n, W = int(a.shape[0]), self.cfg.win_in
        H = self.cfg.win_rows or W
        assert tuple(a.shape[1:]) == (self.cfg.length + 1, H, W, 17), a.shape
        oh, ow = H - 14, W - 14
        out = t.empty((n, oh, ow), dtype=t.float32, device=a.device)
        e = t.empty((n, H, W, 64), dtype=t.float32, device=a.device) if early else None
        l = t.empty((n, oh, ow, 64), dtype=t.float32, device=a.device) if late else None
        self._check(self.lib.ttc_forward_taps(self._h, C.c_void_p(a.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                              C.c_void_p(e.data_ptr()) if early else None,
                                              C.c_void_p(l.data_ptr()) if late else None, self._stream()), "ttc_forward_taps")
        return out, e, l

    def border_subtiles(self, s2, s1, dem, rows, min_all, max_all, hist_align, n_dates_ok):
        """resegment_tiles_wide.py:360-616 on the device (see ttc_border_subtiles).
        s2 [12, X, W, 14], s1 [12, X, W, 2], dem [X, W]; rows [n, 4] int32 (start, rows, pad before, pad after)
        -> (preds cuda [n, H-14, W-14], stats np [n, 4], applied np [n, 5])"""
        t = self.torch
        a, b, d = self._dev(s2, t.float32), self._dev(s1, t.float32), self._dev(dem, t.float32)
        X, W = int(a.shape[1]), int(a.shape[2])
        H = self.cfg.win_rows or self.cfg.win_in
        assert W == self.cfg.win_in and tuple(a.shape) == (12, X, W, 14) and tuple(b.shape) == (12, X, W, 2) and tuple(d.shape) == (X, W)
        rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
        n = rows.shape[0]
        mn, mx = (np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(17)) for v in (min_all, max_all))
        preds = t.empty((n, H - 14, W - 14), dtype=t.float32, device=a.device)
        stats = np.zeros((n, 4), np.float32)
        applied = np.zeros((n, 5), np.int32)
        FP = C.POINTER(C.c_float)
        self._check(self.lib.ttc_border_subtiles(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(d.data_ptr()),
                                                 X, rows.ctypes.data_as(C.POINTER(C.c_int32)), n, mn.ctypes.data_as(FP),
                                                 mx.ctypes.data_as(FP), int(bool(hist_align)), int(n_dates_ok),
                                                 C.c_void_p(preds.data_ptr()), stats.ctypes.data_as(FP),
                                                 applied.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()), "ttc_border_subtiles")
        return preds, stats, applied

    def reseg_mosaic(self, preds, table, weights, ramps, X, Y, want_sums=False):
        """ttc_reseg_mosaic: preds / weights flat float32 (numpy or cuda), table = [(kind, x, y, rows, cols, pred_off,
        weight_off)], ramps float64 [5, X, Y] -> cuda float32 [X, Y] (and the weight sums)"""
        t = self.torch
        p, w = self._dev(preds, t.float32), self._dev(weights, t.float32)
        r = self._dev(ramps, t.float64)
        assert tuple(r.shape) == (5, X, Y)
        arr = (TTCResegWindow * len(table))()
        for i, row in enumerate(table):
            arr[i] = TTCResegWindow(*[int(v) for v in row])
        out = t.empty((X, Y), dtype=t.float32, device=p.device)
        sums = t.empty((X, Y), dtype=t.float32, device=p.device) if want_sums else None
        self._check(self.lib.ttc_reseg_mosaic(self._h, C.c_void_p(p.data_ptr()), arr, len(table), C.c_void_p(w.data_ptr()),
                                              C.c_void_p(r.data_ptr()), int(X), int(Y), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(sums.data_ptr()) if want_sums else None, self._stream()), "ttc_reseg_mosaic")
        return (out, sums) if want_sums else out

    def smooth_strip(self, s2, wmat):
        """s2 [T, X, Y, 10] + operator [12, T] -> cuda [12, X, Y, 14] (bands | indices), see ttc_smooth_strip"""
        t = self.torch
        a = self._dev(s2, t.float32)
        T, X, Y = (int(v) for v in a.shape[:3])
        w = np.ascontiguousarray(wmat, dtype=np.float32)
        assert w.shape == (12, T) and a.shape[3] == 10
        out = t.empty((12, X, Y, 14), dtype=t.float32, device=a.device)
        self._check(self.lib.ttc_smooth_strip(self._h, C.c_void_p(a.data_ptr()), T, X, Y, w.ctypes.data_as(C.POINTER(C.c_float)),
                                              C.c_void_p(out.data_ptr()), self._stream()), "ttc_smooth_strip")
        return out

    def superresolve_windows(self, arr, wsize=125, quirks=1):
        """in place on a cuda tensor [T, X, Y, C >= 10]: DSen2 over wsize windows (resegment_tiles_wide.py:144-179)"""
        T, X, Y, Cc = (int(v) for v in arr.shape)
        assert arr.is_cuda and arr.is_contiguous() and arr.dtype == self.torch.float32
        self._check(self.lib.ttc_superresolve_windows(self._h, C.c_void_p(arr.data_ptr()), T, X, Y, Cc, int(wsize), int(quirks),
                                                      self._stream()), "ttc_superresolve_windows")
        return arr

    def seam_adjust(self, preds):
        """resegment_tiles_wide.py:518-531 on [n, rows, cols] (a copy is adjusted) -> (cuda tensor, stats np [n, 4])"""
        t = self.torch
        a = self._dev(preds, t.float32).clone()
        n, rows, cols = (int(v) for v in a.shape)
        stats = np.zeros((n, 4), np.float32)
        self._check(self.lib.ttc_seam_adjust(self._h, C.c_void_p(a.data_ptr()), n, rows, cols,
                                             stats.ctypes.data_as(C.POINTER(C.c_float)), self._stream()), "ttc_seam_adjust")
        return a, stats

    def float_to_int16(self, x, precision=1000):
        """job.py:174-180 on the device -> cuda int16"""
        t = self.torch
        a = self._dev(x, t.float32)
        out = t.empty(a.shape, dtype=t.int16, device=a.device)
        self._check(self.lib.ttc_float_to_int16(self._h, C.c_void_p(a.data_ptr()), a.numel(), float(precision),
                                                C.c_void_p(out.data_ptr()), self._stream()), "ttc_float_to_int16")
        return out

    # -- codecs ---------------------------------------------------------------------------
    def to_float32(self, u16):
        """uint16 numpy / cuda (viewed as int16 storage) -> cuda float32 (tof_downloading.py:64-72)"""
        t = self.torch
        a = u16 if isinstance(u16, t.Tensor) else t.from_numpy(np.ascontiguousarray(u16).view(np.int16))
        a = a.to(f"cuda:{self.device}").contiguous()
        out = t.empty(a.shape, dtype=t.float32, device=a.device)
        self._check(self.lib.ttc_u16_to_float(self._h, C.c_void_p(a.data_ptr()), a.numel(), C.c_void_p(out.data_ptr()),
                                              self._stream()), "ttc_u16_to_float")
        return out

    def to_int16(self, x):
        """float32 -> uint16 (returned as a cuda int16 tensor holding the uint16 bit patterns)"""
        t = self.torch
        a = self._dev(x, t.float32)
        out = t.empty(a.shape, dtype=t.int16, device=a.device)
        self._check(self.lib.ttc_float_to_u16(self._h, C.c_void_p(a.data_ptr()), a.numel(), C.c_void_p(out.data_ptr()),
                                              self._stream()), "ttc_float_to_u16")
        return out

    def s1_to_db(self, s1_u16):
        """uint16 [T, X, Y, 2] -> cuda float32 dB-scaled (job.py:699-708)"""
        t = self.torch
        a = s1_u16 if isinstance(s1_u16, t.Tensor) else t.from_numpy(np.ascontiguousarray(s1_u16).view(np.int16))
        a = a.to(f"cuda:{self.device}").contiguous()
        T, X, Y = (int(v) for v in a.shape[:3])
        out = t.empty(a.shape, dtype=t.float32, device=a.device)
        self._check(self.lib.ttc_s1_to_db(self._h, C.c_void_p(a.data_ptr()), T, X, Y, C.c_void_p(out.data_ptr()),
                                          self._stream()), "ttc_s1_to_db")
        return out

    # -- small raster steps of process_tile ----------------------------------------------
    def sen2cor_clean(self, clm20):
        t = self.torch
        a = self._dev(clm20, t.float32)
        T, w, h = (int(v) for v in a.shape)
        out = t.empty((T, 2 * w, 2 * h), dtype=t.float32, device=a.device)
        self._check(self.lib.ttc_sen2cor_clean(self._h, C.c_void_p(a.data_ptr()), T, w, h, C.c_void_p(out.data_ptr()), self._stream()),
                    "ttc_sen2cor_clean")
        return out

    def median5(self, dem):
        t = self.torch
        a = self._dev(dem, t.float32)
        out = t.empty_like(a)
        self._check(self.lib.ttc_median5(self._h, C.c_void_p(a.data_ptr()), int(a.shape[0]), int(a.shape[1]), C.c_void_p(out.data_ptr()),
                                         self._stream()), "ttc_median5")
        return out

    def snow_map(self, s2):
        """-> (snow cuda uint8 [X, Y], per-image snow fraction numpy float64 [T])"""
        t = self.torch
        T, X, Y = (int(v) for v in s2.shape[:3])
        snow = t.empty((X, Y), dtype=t.uint8, device=s2.device)
        cnt = (C.c_int32 * T)()
        self._check(self.lib.ttc_snow_map(self._h, C.c_void_p(s2.data_ptr()), T, X, Y, C.c_void_p(snow.data_ptr()), cnt, self._stream()),
                    "ttc_snow_map")
        return snow, np.array([cnt[i] for i in range(T)], dtype=np.float64) / (X * Y)

    def merge_cloud_masks(self, cloudshad, clm, fcps=None):
        self._check(self.lib.ttc_merge_cloud_masks(self._h, C.c_void_p(cloudshad.data_ptr()), C.c_void_p(clm.data_ptr()),
                                                   C.c_void_p(fcps.data_ptr()) if fcps is not None else None, cloudshad.numel(),
                                                   self._stream()), "ttc_merge_cloud_masks")

    def fraction_equal(self, a, value):
        """a cuda float32 [T, X, Y] -> numpy float64 [T]: np.mean(a == value, axis=(1, 2))"""
        T, npix = int(a.shape[0]), int(a.shape[1] * a.shape[2])
        cnt = (C.c_int32 * T)()
        self._check(self.lib.ttc_count_equal(self._h, C.c_void_p(a.data_ptr()), T, npix, C.c_float(value), cnt, self._stream()),
                    "ttc_count_equal")
        return np.array([cnt[i] for i in range(T)], dtype=np.float64) / npix

    def fraction_positive(self, a):
        """a cuda float32 [T, X, Y] -> numpy float64 [T]: np.mean(a > 0, axis=(1, 2))"""
        T, npix = int(a.shape[0]), int(a.shape[1] * a.shape[2])
        cnt = (C.c_int32 * T)()
        self._check(self.lib.ttc_count_positive(self._h, C.c_void_p(a.data_ptr()), T, npix, cnt, self._stream()), "ttc_count_positive")
        return np.array([cnt[i] for i in range(T)], dtype=np.float64) / npix

    def divide(self, a, divisor):
        self._check(self.lib.ttc_divide(self._h, C.c_void_p(a.data_ptr()), a.numel(), float(divisor), self._stream()), "ttc_divide")
        return a

    def clip01(self, a):
        self._check(self.lib.ttc_clip01(self._h, C.c_void_p(a.data_ptr()), a.numel(), self._stream()), "ttc_clip01")
        return a

    # -- cloud / shadow detection --------------------------------------------------------
    def identify_clouds_shadows(self, img, dem, forest=None, urban=None, debug_stage=0):
        """img [T,X,Y,10] float32, dem [X,Y]; forest [X,Y] (0/1) or None; urban = (core, near) masks or None
        -> (clouds cuda float32 [T,X,Y], fcps cuda uint8 [T,X,Y])"""
        t = self.torch
        a = self._dev(img, t.float32)
        T, X, Y = (int(v) for v in a.shape[:3])
        d = self._dev(dem, t.float32)
        u8 = lambda m: None if m is None else self._dev(np.ascontiguousarray(np.asarray(m) != 0).astype(np.uint8), t.uint8)
        f = u8(forest)
        core, near = (u8(urban[0]), u8(urban[1])) if urban is not None else (None, None)
        clouds = t.empty((T, X, Y), dtype=t.float32, device=a.device)
        fcps = t.empty((T, X, Y), dtype=t.uint8, device=a.device)
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        self._check(self.lib.ttc_debug_clouds_stage(self._h, int(debug_stage)), "ttc_debug_clouds_stage")
        try:
            self._check(self.lib.ttc_identify_clouds_shadows(self._h, ptr(a), T, X, Y, ptr(d), ptr(f), ptr(core), ptr(near), ptr(clouds),
                                                             ptr(fcps), self._stream()), "ttc_identify_clouds_shadows")
        finally:
            self.lib.ttc_debug_clouds_stage(self._h, 0)
        return clouds, fcps

    # -- cloud gap-fill ------------------------------------------------------------------
    def feather(self, mask, closing=20, clip=False):
        t = self.torch
        m = self._dev(mask, t.float32)
        T, X, Y = (int(v) for v in m.shape)
        w = t.empty_like(m)
        self._check(self.lib.ttc_feather(self._h, C.c_void_p(m.data_ptr()), T, X, Y, closing, int(clip),
                                         C.c_void_p(w.data_ptr()), self._stream()), "ttc_feather")
        return w

    def aligned_mosaic(self, tiles, w):
        """tiles [T,X,Y,10], w [T,X,Y] cuda float32 (w is updated in place) -> mosaic [X,Y,10]"""
        t = self.torch
        T, X, Y = (int(v) for v in tiles.shape[:3])
        mosaic = t.empty((X, Y, 10), dtype=t.float32, device=tiles.device)
        self._check(self.lib.ttc_aligned_mosaic(self._h, C.c_void_p(tiles.data_ptr()), C.c_void_p(w.data_ptr()), T, X, Y,
                                                C.c_void_p(mosaic.data_ptr()), self._stream()), "ttc_aligned_mosaic")
        return mosaic

    def remove_cloud_and_shadows(self, tiles, probs, pfcps=None, sampler=None, want_mosaic=False):
        """tiles [T,X,Y,10] cuda float32 (in place), probs [T,X,Y]; sampler: None (deterministic, on device) or a
        Python callable (evi float32[n]) -> int64 row indices.  -> (interp cuda [T,X,Y], to_remove list, mosaic | None)"""
        t = self.torch
        T, X, Y = (int(v) for v in tiles.shape[:3])
        pr = self._dev(probs, t.float32)
        pf = None
        if pfcps is not None:                       # [X, Y] or, like the reference (cloud_removal.py:709-711), [T, X, Y] -> date 0
            if isinstance(pfcps, t.Tensor):
                pf = pfcps[0] if pfcps.dim() == 3 else pfcps
                pf = (pf != 0).to(t.uint8).contiguous()
            else:
                a = np.asarray(pfcps)
                pf = self._dev(np.ascontiguousarray(a[0] if a.ndim == 3 else a).astype(np.uint8), t.uint8)
        interp = t.empty((T, X, Y), dtype=t.float32, device=tiles.device)
        mosaic = t.empty((X, Y, 10), dtype=t.float32, device=tiles.device) if want_mosaic else None
        rem = (C.c_int32 * T)()
        nrem = C.c_int32(0)
        err = []

        def _cb(evi_p, n, out_p, cap, user):
            try:
                evi = np.ctypeslib.as_array(evi_p, shape=(n,))
                idx = np.asarray(sampler(evi), dtype=np.int64)
                if idx.size > cap:
                    raise ValueError("sampler returned more rows than the capacity")
                np.ctypeslib.as_array(out_p, shape=(idx.size,))[:] = idx
                return int(idx.size)
            except Exception as e:      # never raise through the C frame
                err.append(e)
                return -1
        cb = SAMPLER_FN(_cb) if sampler is not None else SAMPLER_FN()
        st = self.lib.ttc_remove_cloud_and_shadows(
            self._h, C.c_void_p(tiles.data_ptr()), C.c_void_p(pr.data_ptr()), C.c_void_p(pf.data_ptr()) if pf is not None else None,
            T, X, Y, cb, None, C.c_void_p(interp.data_ptr()), C.c_void_p(mosaic.data_ptr()) if want_mosaic else None,
            rem, C.byref(nrem), self._stream())
        if err:
            raise err[0]
        self._check(st, "ttc_remove_cloud_and_shadows")
        return interp, [int(rem[i]) for i in range(nrem.value)], mosaic

    # -- 20 m -> 10 m ---------------------------------------------------------------------
    def dsen2_forward(self, x, bilinear):
        t = self.torch
        xd, bd = self._dev(x, t.float32), self._dev(bilinear, t.float32)
        n, H, W = (int(v) for v in xd.shape[:3])
        out = t.empty((n, H, W, 6), dtype=t.float32, device=xd.device)
        self._check(self.lib.ttc_dsen2_forward(self._h, C.c_void_p(xd.data_ptr()), C.c_void_p(bd.data_ptr()), n, H, W,
                                               C.c_void_p(out.data_ptr()), self._stream()), "ttc_dsen2_forward")
        return out

    def superresolve_tile(self, s2d, quirks=True):
        T, X, Y = (int(v) for v in s2d.shape[:3])
        self._check(self.lib.ttc_superresolve_tile(self._h, C.c_void_p(s2d.data_ptr()), T, X, Y, int(quirks),
                                                   self._stream()), "ttc_superresolve_tile")
        return s2d

    def upsample_20m(self, s2_10, s2_20):
        t = self.torch
        a, b = self._dev(s2_10, t.float32), self._dev(s2_20, t.float32)
        T, h, w = (int(v) for v in b.shape[:3])
        out = t.empty((T, 2 * h, 2 * w, 10), dtype=t.float32, device=a.device)
        self._check(self.lib.ttc_upsample_20m(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), T, h, w,
                                              C.c_void_p(out.data_ptr()), self._stream()), "ttc_upsample_20m")
        return out

    # -- debug ----------------------------------------------------------------------------
    def debug_fetch(self, name, shape=None):
        n = C.c_size_t(0)
        self._check(self.lib.ttc_debug_fetch(self._h, name.encode(), None, 0, C.byref(n)), "ttc_debug_fetch")
        buf = np.empty(n.value, dtype=np.float32)
        self._check(self.lib.ttc_debug_fetch(self._h, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)),
                                             n.value, C.byref(n)), "ttc_debug_fetch")
        if shape is not None:
            buf = buf[:int(np.prod(shape))].reshape(shape)
        return buf

    def timing(self, enable=True):
        self._check(self.lib.ttc_debug_timing(self._h, int(enable)), "ttc_debug_timing")

    def kernel_ms(self, name=None):
        if name is None:
            self._check(self.lib.ttc_debug_kernel_ms(self._h, None, None, None), "ttc_debug_kernel_ms")
            return None
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._check(self.lib.ttc_debug_kernel_ms(self._h, name.encode(), C.byref(ms), C.byref(cnt)),
                    "ttc_debug_kernel_ms")
        return ms.value, cnt.value

    def kernel_flops(self, name):
        """-> (matrix-instruction flops ISSUED per launch of the conv family `name`, launches noted) since the last kernel_ms(None)"""
        fl, cnt = C.c_double(0), C.c_int64(0)
        self._check(self.lib.ttc_debug_kernel_flops(self._h, name.encode(), C.byref(fl), C.byref(cnt)), "ttc_debug_kernel_flops")
        return fl.value, cnt.value
