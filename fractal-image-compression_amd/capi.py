"""ctypes binding of libfic_hip.so (include/fic.h).  Fails loudly when the library is missing:
there is no Python or CPU fallback for the hot path."""
import ctypes as C
import os
import sys
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libfic_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fic.h")

FIC_OK = 0
ERROR_NAMES = {
    -1: "FIC_E_GEOMETRY", -2: "FIC_E_WINDOW", -3: "FIC_E_ARGUMENT", -4: "FIC_E_NO_DEVICE",
    -5: "FIC_E_HIP", -6: "FIC_E_NOT_GREY", -7: "FIC_E_STATE", -8: "FIC_E_CAPACITY",
}


class FicError(Exception):
    """Raised for every negative return code of the C ABI (the reference throws unchecked
    exceptions in the same situations; CTL:184-186 catches them)."""

    def __init__(self, code, message):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


def declared_symbols():
    """Names of every FIC_API function declared in include/fic.h."""
    text = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"FIC_API\s+[\w\s\*]+?\b(fic_\w+)\s*\(", text)))


_lib = None


def _torch_first():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; libfic_hip.so uses /opt/rocm's.  Two HIP
    runtimes share a process as long as torch's initialises first -- the other order leaves torch with "No HIP GPUs are
    available".  So when torch is already imported, bring its runtime up before this library's first HIP call.  (A
    process that imports torch only later should call torch.cuda.init() before its first fic call; bench.py and the
    multi-GPU layer import torch at the top.)"""
    t = sys.modules.get("torch")
    if t is not None:
        try:
            if t.cuda.is_available():
                t.cuda.init()
        except Exception:       # a CPU-only box: the library itself reports FIC_E_NO_DEVICE
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("FIC_HIP_SO") or SO_PATH          # FIC_HIP_SO: another build of the same library (A/B runs of kernel variants)
    if not os.path.exists(so):
        raise FicError(-5, f"{so} is missing: build it with __graft_entry__.build() "
                           "(hipcc --offload-arch=gfx950); there is no fallback path")
    _torch_first()
    L = C.CDLL(so)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    i32p, f32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    sig = {
        "fic_version": (C.c_char_p, []),
        "fic_last_error": (C.c_char_p, []),
        "fic_last_error_code": (C.c_int, []),
        "fic_device_count": (C.c_int, []),
        "fic_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, ip, ip, ip, ip]),
        "fic_is_greyscale_argb": (C.c_int, [i32p, C.c_int, C.c_int]),
        "fic_encode_gray_argb": (C.c_int, [i32p] + [C.c_int] * 6 + [i32p, f32p, f32p, i32p, i32p]),
        "fic_encode_gray_u8": (C.c_int, [u8p] + [C.c_int] * 6 + [i32p, f32p, f32p, i32p, i32p]),
        "fic_encode_gray_argb_multi": (C.c_int, [i32p] + [C.c_int] * 6 + [i32p, f32p, f32p, i32p, i32p]),
        "fic_encode_gray_u8_multi": (C.c_int, [u8p] + [C.c_int] * 6 + [i32p, f32p, f32p, i32p, i32p]),
        "fic_write_run_gray": (C.c_int64, [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int64]),
        "fic_ctx_create": (vp, [C.c_int] * 7),
        "fic_ctx_destroy": (None, [vp]),
        "fic_ctx_set_gray_host": (C.c_int, [vp, u8p]),
        "fic_ctx_set_argb_host": (C.c_int, [vp, i32p]),
        "fic_ctx_set_gray_device": (C.c_int, [vp, vp]),
        "fic_ctx_encode": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "fic_ctx_sync": (C.c_int, [vp]),
        "fic_ctx_get_results_host": (C.c_int, [vp, i32p, f32p, f32p, i32p, i32p, i32p, f32p]),
        "fic_ctx_result_device_ptrs": (C.c_int, [vp] + [C.POINTER(vp)] * 7),
        "fic_ctx_records_device_ptr": (C.c_int, [vp, C.POINTER(vp)]),
        "fic_ctx_collage_host": (C.c_int, [vp, i32p]),
        "fic_encode_rgb_argb": (C.c_int, [i32p] + [C.c_int] * 5 + [i32p, f32p, f32p, f32p, f32p, i32p, i32p]),
        "fic_rgb_ctx_create": (vp, [C.c_int] * 6),
        "fic_rgb_ctx_destroy": (None, [vp]),
        "fic_rgb_ctx_set_argb_host": (C.c_int, [vp, i32p]),
        "fic_rgb_ctx_set_argb_device": (C.c_int, [vp, vp]),
        "fic_rgb_ctx_encode": (C.c_int, [vp, C.c_int, vp]),
        "fic_rgb_ctx_sync": (C.c_int, [vp]),
        "fic_rgb_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "fic_rgb_ctx_last_sweep": (C.c_int, [vp]),
        "fic_rgb_ctx_get_results_host": (C.c_int, [vp, i32p, f32p, f32p, f32p, f32p, i32p, i32p]),
        "fic_rgb_ctx_decode_host": (C.c_int, [vp, i32p, f32p, ip]),
        "fic_write_run_rgb": (C.c_int64, [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int64]),
        "fic_decode_gray_run": (C.c_int, [u8p, C.c_int64, C.c_int, u8p, C.c_int64, ip, ip, f32p, ip]),
        "fic_ctx_decode_host": (C.c_int, [vp, u8p, f32p, ip]),
        "fic_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "fic_ctx_sweep_time": (C.c_int, [vp, C.POINTER(C.c_double), ip, C.c_int]),
        "fic_ctx_info": (C.c_int, [vp, ip]),
        "fic_sweep_ranges_per_pool_read": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "fic_ctx_last_kernel": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "fic_ctx_sweep_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
        "fic_debug_rccl_selftest": (C.c_int, [C.c_int]),
        "fic_debug_gather_fallbacks": (C.c_int, []),
        "fic_debug_float_sum": (C.c_int, [C.c_int, C.c_float, C.POINTER(C.c_uint32), C.c_int, f32p]),
        "fic_debug_float_sum_fallbacks": (C.c_int, []),
        "fic_debug_decode_gray_run": (C.c_int, [u8p, C.c_int64, C.c_int, u8p, C.c_int64, f32p, ip, ip]),
        "fic_debug_sqrt_f64": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
        "fic_ctx_debug_pool_host": (C.c_int, [vp, u8p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u8p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def has_xcheck():
    """True when the library was built with round 1's exact-covariance matrix-core sweeps ("sweep" = 3 / 4; build.py,
    FIC_BUILD_XCHECK, on by default) -- the test-suite's independent cross-checks of the default sweep."""
    return b"+xcheck" in lib().fic_version()


def last_error():
    return lib().fic_last_error().decode("utf-8", "replace")


def check(rc):
    if rc < 0:
        raise FicError(rc, last_error())
    return rc


def ptr(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def geometry(w, h, B):
    v = [C.c_int() for _ in range(4)]
    check(lib().fic_geometry(w, h, B, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def decode_gray_run(run, device=0, avg_error_in=0.0):
    """FractalCompression.decode on a grey .run stream (FC:547-553, 356-421), on the GPU.
    Returns (gray uint8 [H,W], avgError float32 after the call, iterations)."""
    buf = np.frombuffer(bytes(run), np.uint8)
    if buf.size < 20:
        raise FicError(-3, "run stream shorter than its header")
    w = int.from_bytes(bytes(run[4:8]), "big", signed=True)
    h = int.from_bytes(bytes(run[8:12]), "big", signed=True)
    cap = max(w, 0) * max(h, 0)
    out = np.zeros(max(cap, 1), np.uint8)
    avg = C.c_float(avg_error_in)
    it, wo, ho = C.c_int(), C.c_int(), C.c_int()
    check(lib().fic_decode_gray_run(ptr(buf, C.c_uint8), buf.size, device, ptr(out, C.c_uint8), cap, C.byref(wo),
                                    C.byref(ho), C.byref(avg), C.byref(it)))
    return out[:cap].reshape(h, w), np.float32(avg.value), it.value


def encode_gray_oneshot(gray, B, wK, n_iso=1, device=0):
    """The one-shot C entry point fic_encode_gray_u8 (host buffers in and out; what the JNI shim binds)."""
    g = np.ascontiguousarray(gray, np.uint8)
    h, w = g.shape
    Rw, Rh, Dw, Dh = geometry(w, h, B)
    nr = Rw * Rh
    r = {"idx_local": np.zeros(nr, np.int32), "a": np.zeros(nr, np.float32), "b": np.zeros(nr, np.float32),
         "iso": np.zeros(nr, np.int32), "qrows": np.zeros((nr, 3), np.int32)}
    check(lib().fic_encode_gray_u8(ptr(g, C.c_uint8), w, h, B, Dw if wK is None else wK, n_iso, device,
                                   ptr(r["idx_local"], C.c_int32), ptr(r["a"], C.c_float), ptr(r["b"], C.c_float),
                                   ptr(r["iso"], C.c_int32), ptr(r["qrows"], C.c_int32)))
    return r


def encode_gray_multi(gray, B, wK, n_iso=1, n_gpus=1, width=None, height=None):
    """fic_encode_gray_u8_multi / fic_encode_gray_argb_multi: one synchronous call, range blocks sharded over the first
    n_gpus devices, RCCL gather in the library (what the JNI host calls on a multi-GPU node).  `gray`: uint8 [H, W], or --
    with width and height given -- the int32 ARGB pixels of RasterImage.argb."""
    if width is not None:
        g = np.ascontiguousarray(gray, np.int32).reshape(-1)
        w, h = int(width), int(height)
        if g.size != w * h:
            raise FicError(-3, "argb has the wrong number of pixels")
        fn, p = lib().fic_encode_gray_argb_multi, ptr(g, C.c_int32)
    else:
        g = np.ascontiguousarray(gray, np.uint8)
        h, w = g.shape
        fn, p = lib().fic_encode_gray_u8_multi, ptr(g, C.c_uint8)
    Rw, Rh, Dw, Dh = geometry(w, h, B)
    nr = Rw * Rh
    r = {"idx_local": np.zeros(nr, np.int32), "a": np.zeros(nr, np.float32), "b": np.zeros(nr, np.float32),
         "iso": np.zeros(nr, np.int32), "qrows": np.zeros((nr, 3), np.int32)}
    check(fn(p, w, h, B, Dw if wK is None else wK, n_iso, n_gpus, ptr(r["idx_local"], C.c_int32), ptr(r["a"], C.c_float),
             ptr(r["b"], C.c_float), ptr(r["iso"], C.c_int32), ptr(r["qrows"], C.c_int32)))
    return r


def release_cache():
    lib().fic_release_cache.restype = None
    lib().fic_release_cache()


def decode_rgb_run(run, device=0, avg_error_in=0.0):
    """decodeRGB (FC:430-508) on the GPU.  Returns (argb int32 [H*W], avgError float32, iterations, w, h)."""
    L = lib()
    L.fic_decode_rgb_run.restype = C.c_int
    L.fic_decode_rgb_run.argtypes = [C.POINTER(C.c_uint8), C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int64,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    buf = np.frombuffer(bytes(run), np.uint8)
    if buf.size < 20:
        raise FicError(-3, "run stream shorter than its header")
    w = int.from_bytes(bytes(run[4:8]), "big", signed=True)
    h = int.from_bytes(bytes(run[8:12]), "big", signed=True)
    cap = max(w, 0) * max(h, 0)
    out = np.zeros(max(cap, 1), np.int32)
    avg = C.c_float(avg_error_in)
    it, wo, ho = C.c_int(), C.c_int(), C.c_int()
    check(L.fic_decode_rgb_run(ptr(buf, C.c_uint8), buf.size, device, ptr(out, C.c_int32), cap, C.byref(wo), C.byref(ho),
                               C.byref(avg), C.byref(it)))
    return out[:cap], np.float32(avg.value), it.value, w, h


def encode_rgb(argb, w, h, B, wK, device=0, want_collage=False):
    """encodeRGB (FC:171-219) on the GPU.  argb: int32 [h*w].  Returns a dict of [N_r] arrays
    (idx_local, a, bR, bG, bB, qrows [N_r,5]) and, when asked, the collage (int32 [h*w])."""
    Rw, Rh, Dw, Dh = geometry(w, h, B)
    nr = Rw * Rh
    argb = np.ascontiguousarray(argb, np.int32).reshape(-1)
    if argb.size != w * h:
        raise FicError(-3, "argb has the wrong number of pixels")
    r = {"idx_local": np.zeros(nr, np.int32), "a": np.zeros(nr, np.float32), "bR": np.zeros(nr, np.float32),
         "bG": np.zeros(nr, np.float32), "bB": np.zeros(nr, np.float32), "qrows": np.zeros((nr, 5), np.int32)}
    col = np.zeros(w * h, np.int32) if want_collage else None
    check(lib().fic_encode_rgb_argb(ptr(argb, C.c_int32), w, h, B, wK, device, ptr(r["idx_local"], C.c_int32),
                                    ptr(r["a"], C.c_float), ptr(r["bR"], C.c_float), ptr(r["bG"], C.c_float),
                                    ptr(r["bB"], C.c_float), ptr(r["qrows"], C.c_int32), ptr(col, C.c_int32)))
    if want_collage:
        r["collage"] = col
    return r


class RgbEncoder:
    """Handle API of the joint-RGB path (fic_rgb_ctx_*): `planes` colour images of one geometry, device resident."""

    def __init__(self, width, height, B, wK, planes=1, device=0):
        L = lib()
        self.width, self.height, self.B, self.wK, self.planes, self.device = width, height, B, wK, planes, device
        Rw, Rh, Dw, Dh = geometry(width, height, B)
        self.n_ranges = Rw * Rh
        self._h = L.fic_rgb_ctx_create(device, width, height, B, wK, planes)
        if not self._h:
            raise FicError(L.fic_last_error_code() or -3, last_error())

    def close(self):
        if getattr(self, "_h", None):
            lib().fic_rgb_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_argb(self, argb):
        """int32 [planes, H*W] numpy array (copied) or a torch CUDA int32 tensor of that many elements (used in place)."""
        n = self.planes * self.width * self.height
        if hasattr(argb, "data_ptr"):
            if not argb.is_cuda or argb.element_size() != 4 or not argb.is_contiguous() or argb.numel() != n:
                raise FicError(-3, "device input must be a contiguous int32 CUDA tensor [planes,H,W]")
            self._keep = argb
            check(lib().fic_rgb_ctx_set_argb_device(self._h, C.c_void_p(argb.data_ptr())))
        else:
            a = np.ascontiguousarray(argb, np.int32)
            if a.size != n:
                raise FicError(-3, "argb has the wrong number of pixels")
            check(lib().fic_rgb_ctx_set_argb_host(self._h, ptr(a, C.c_int32)))

    def encode(self, with_collage=False, stream=None):
        s = 0 if stream is None else int(getattr(stream, "cuda_stream", stream))
        check(lib().fic_rgb_ctx_encode(self._h, 1 if with_collage else 0, C.c_void_p(s)))
        self._collage = bool(with_collage)

    def sync(self):
        check(lib().fic_rgb_ctx_sync(self._h))

    def set_option(self, name, value):
        """"sweep": 0 auto, 1 the VALU sweeps, 2 the matrix-core full search."""
        check(lib().fic_rgb_ctx_set_option(self._h, name.encode(), int(value)))

    def last_sweep(self):
        return check(lib().fic_rgb_ctx_last_sweep(self._h))

    def results(self):
        P, N = self.planes, self.n_ranges
        r = {"idx_local": np.zeros((P, N), np.int32), "a": np.zeros((P, N), np.float32), "bR": np.zeros((P, N), np.float32),
             "bG": np.zeros((P, N), np.float32), "bB": np.zeros((P, N), np.float32), "qrows": np.zeros((P, N, 5), np.int32)}
        col = np.zeros((P, self.height * self.width), np.int32) if getattr(self, "_collage", False) else None
        check(lib().fic_rgb_ctx_get_results_host(self._h, ptr(r["idx_local"], C.c_int32), ptr(r["a"], C.c_float),
                                                 ptr(r["bR"], C.c_float), ptr(r["bG"], C.c_float), ptr(r["bB"], C.c_float),
                                                 ptr(r["qrows"], C.c_int32), ptr(col, C.c_int32)))
        if col is not None:
            r["collage"] = col
        return r

    def decode(self):
        """decodeRGB from the context's quantised rows: (argb int32 [planes, H*W], avgError float32 [planes], iterations)."""
        P = self.planes
        out = np.zeros((P, self.height * self.width), np.int32)
        avg = np.zeros(P, np.float32)
        it = np.zeros(P, np.int32)
        check(lib().fic_rgb_ctx_decode_host(self._h, ptr(out, C.c_int32), ptr(avg, C.c_float), ptr(it, C.c_int)))
        return out, avg, it


def write_run_rgb(qrows5, w, h, B, wK):
    """writeData RGB branch (FC:230-238, 248-257): header + 5 ints per row, big-endian."""
    q = np.ascontiguousarray(qrows5, np.int32).reshape(-1, 5)
    out = np.zeros(20 + 20 * q.shape[0], np.uint8)
    n = lib().fic_write_run_rgb(ptr(q, C.c_int32), q.shape[0], w, h, B, wK, ptr(out, C.c_uint8), out.size)
    check(int(n))
    return out.tobytes()


def write_run_gray(qrows, w, h, B, wK):
    """writeData grey branch (FC:230-246): header + rows, big-endian."""
    q = np.ascontiguousarray(qrows, np.int32).reshape(-1, 3)
    out = np.zeros(20 + 12 * q.shape[0], np.uint8)
    n = lib().fic_write_run_gray(ptr(q, C.c_int32), q.shape[0], w, h, B, wK, ptr(out, C.c_uint8), out.size)
    check(int(n))
    return out.tobytes()
