"""Host-side mirror of the reference's interface for the grey encode path.

Same names, argument meaning and error behaviour as bvk_ss19 (src/bvk_ss19/):
  RasterImage           RasterImage.java:18-24   public fields argb, width, height
  FractalCompression    FractalCompression.java  statics blockgroesse (:14), widthKernel (:15),
                        encode (:54), encodeGrayScale (:109), isGreyScale (:32), writeData (:230),
                        getBestGeneratedCollage (:269)
All block search runs in libfic_hip.so on the GPU (capi.py); nothing here computes a match.
"""
import ctypes as C
import io

import numpy as np

from . import capi
from .capi import FicError


class RasterImage:
    """RasterImage.java:18-24 -- int[] argb (ARGB, scanline order), width, height."""

    def __init__(self, width, height, argb=None):
        self.width = int(width)
        self.height = int(height)
        if argb is None:
            argb = np.zeros(self.width * self.height, np.int32)          # new int[w*h]
        self.argb = np.ascontiguousarray(argb, np.int32).reshape(-1)
        if self.argb.size != self.width * self.height:
            raise ValueError("argb has the wrong number of pixels")

    @classmethod
    def from_gray(cls, gray):
        g = np.ascontiguousarray(gray, np.uint8)
        u = g.astype(np.uint32)
        argb = (0xFF000000 | (u << 16) | (u << 8) | u).astype(np.uint32).view(np.int32)
        return cls(g.shape[1], g.shape[0], argb.reshape(-1))

    def red(self):
        return ((self.argb.view(np.uint32) >> 16) & 0xFF).astype(np.uint8).reshape(self.height, self.width)


class _DevArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class Encoder:
    """Handle API (fic_ctx_*): the working set for `planes` grey images of one geometry on one
    device.  Inputs may be numpy arrays (copied to the device) or torch CUDA tensors (used in
    place: this is what bench.py times)."""

    def __init__(self, width, height, B, wK=None, n_iso=1, planes=1, device=0):
        L = capi.lib()
        self.width, self.height, self.B, self.n_iso, self.planes, self.device = width, height, B, n_iso, planes, device
        Rw, Rh, Dw, Dh = capi.geometry(width, height, B)
        if wK is None:
            if Dw != Dh:
                raise FicError(-2, "full search (wK=None) needs a square block grid, as in the reference (FC:89-96)")
            wK = Dw
        self.wK = wK
        self.Rw, self.Rh, self.Dw, self.Dh = Rw, Rh, Dw, Dh
        self.n_ranges, self.n_domains = Rw * Rh, Dw * Dh
        self._h = L.fic_ctx_create(device, width, height, B, wK, n_iso, planes)
        if not self._h:
            raise FicError(L.fic_last_error_code() or -3, capi.last_error())
        self._keep = None
        info = (C.c_int * 10)()
        capi.check(L.fic_ctx_info(self._h, info))
        self.ranges_per_tile = 64 * info[6]

    def close(self):
        if getattr(self, "_h", None):
            capi.lib().fic_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- input -----------------------------------------------------------------------------
    def set_gray(self, gray):
        """gray: uint8 [planes,H,W] / [H,W] numpy array, or a torch CUDA uint8 tensor of that shape."""
        L = capi.lib()
        n = self.planes * self.height * self.width
        if hasattr(gray, "data_ptr"):                                  # torch tensor, device resident
            if not gray.is_cuda or gray.dtype.itemsize != 1 or not gray.is_contiguous() or gray.numel() != n:
                raise FicError(-3, "device input must be a contiguous uint8 CUDA tensor [planes,H,W]")
            self._keep = gray
            capi.check(L.fic_ctx_set_gray_device(self._h, C.c_void_p(gray.data_ptr())))
        else:
            g = np.ascontiguousarray(gray, np.uint8)
            if g.size != n:
                raise FicError(-3, f"input has {g.size} pixels, expected {n}")
            capi.check(L.fic_ctx_set_gray_host(self._h, capi.ptr(g, C.c_uint8)))

    def set_argb(self, argb):
        a = np.ascontiguousarray(argb, np.int32)
        if a.size != self.planes * self.height * self.width:
            raise FicError(-3, "argb has the wrong number of pixels")
        capi.check(capi.lib().fic_ctx_set_argb_host(self._h, capi.ptr(a, C.c_int32)))

    # ---- compute ---------------------------------------------------------------------------
    def set_option(self, name, value):
        capi.check(capi.lib().fic_ctx_set_option(self._h, name.encode(), int(value)))

    def encode(self, range_begin=0, range_count=-1, stream=None):
        """Asynchronous on `stream` (a raw hipStream_t handle / torch stream, None = default)."""
        s = 0 if stream is None else int(getattr(stream, "cuda_stream", stream))
        capi.check(capi.lib().fic_ctx_encode(self._h, range_begin, range_count, C.c_void_p(s)))

    def sync(self):
        capi.check(capi.lib().fic_ctx_sync(self._h))

    def sweep_time(self, reset=True):
        ms, n = C.c_double(), C.c_int()
        capi.check(capi.lib().fic_ctx_sweep_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def sweep_stats(self, reset=True):
        """k_sweep_q counters (after set_option("sweep_stats", 1)): tile epilogues, flagged tiles, exact pairs, waves,
        and of a sample of the waves the shader-clock cycles / 100 MHz ticks they lived (`clock_ghz` = their ratio)."""
        v = (C.c_uint64 * 8)()
        capi.check(capi.lib().fic_ctx_sweep_stats(self._h, v, 1 if reset else 0))
        d = dict(zip(["tiles", "flagged_tiles", "exact_pairs", "waves", "wave_cycles", "wave_ticks", "waves_sampled"], [int(x) for x in v]))
        d["clock_ghz"] = d["wave_cycles"] / d["wave_ticks"] / 10.0 if d["wave_ticks"] else None
        return d

    def last_kernel(self):
        """Name of the sweep kernel the last encode launched, as rocprofv3 prints it (without the argument list)."""
        buf = C.create_string_buffer(96)
        capi.check(capi.lib().fic_ctx_last_kernel(self._h, buf, 96))
        return buf.value.decode()

    def info(self):
        v = (C.c_int * 10)()
        capi.check(capi.lib().fic_ctx_info(self._h, v))
        keys = ["Rw", "Rh", "Nr", "Dw", "Dh", "Nd", "NR", "tiles", "chunks", "sweep_kind"]
        return dict(zip(keys, list(v)))

    # ---- results ---------------------------------------------------------------------------
    def results(self):
        P, N = self.planes, self.n_ranges
        r = {
            "idx_local": np.zeros((P, N), np.int32), "a": np.zeros((P, N), np.float32),
            "b": np.zeros((P, N), np.float32), "iso": np.zeros((P, N), np.int32),
            "qrows": np.zeros((P, N, 3), np.int32), "idx_global": np.zeros((P, N), np.int32),
            "err": np.zeros((P, N), np.float32),
        }
        capi.check(capi.lib().fic_ctx_get_results_host(
            self._h, capi.ptr(r["idx_local"], C.c_int32), capi.ptr(r["a"], C.c_float), capi.ptr(r["b"], C.c_float),
            capi.ptr(r["iso"], C.c_int32), capi.ptr(r["qrows"], C.c_int32), capi.ptr(r["idx_global"], C.c_int32),
            capi.ptr(r["err"], C.c_float)))
        return r

    def results_device(self):
        """torch tensors aliasing the context's device result arrays (no copy)."""
        import torch
        p = [C.c_void_p() for _ in range(7)]
        capi.check(capi.lib().fic_ctx_result_device_ptrs(self._h, *[C.byref(x) for x in p]))
        P, N = self.planes, self.n_ranges
        dev = torch.device("cuda", self.device)

        def t(ptr, shape, ts):
            return torch.as_tensor(_DevArray(ptr.value, shape, ts), device=dev)

        return {"idx_local": t(p[0], (P, N), "<i4"), "a": t(p[1], (P, N), "<f4"), "b": t(p[2], (P, N), "<f4"),
                "iso": t(p[3], (P, N), "<i4"), "qrows": t(p[4], (P, N, 3), "<i4"),
                "idx_global": t(p[5], (P, N), "<i4"), "err": t(p[6], (P, N), "<f4")}

    def records_device(self):
        """torch int32 tensor [planes, N_r, 6] aliasing the context's packed codebook records (the unit of the gather)."""
        import torch
        p = C.c_void_p()
        capi.check(capi.lib().fic_ctx_records_device_ptr(self._h, C.byref(p)))
        return torch.as_tensor(_DevArray(p.value, (self.planes, self.n_ranges, 6), "<i4"), device=torch.device("cuda", self.device))

    def collage(self):
        out = np.zeros((self.planes, self.height * self.width), np.int32)
        capi.check(capi.lib().fic_ctx_collage_host(self._h, capi.ptr(out, C.c_int32)))
        return out

    def decode(self):
        """Reference decoder loop (FC:356-421) on the device from the last encode's quantised rows
        (+ isometries).  Returns (gray uint8 [planes,H,W], avgError float32 [planes], iterations int32 [planes])."""
        P = self.planes
        out = np.zeros((P, self.height, self.width), np.uint8)
        avg = np.zeros(P, np.float32)
        it = np.zeros(P, np.int32)
        capi.check(capi.lib().fic_ctx_decode_host(self._h, capi.ptr(out, C.c_uint8), capi.ptr(avg, C.c_float),
                                                  capi.ptr(it, C.c_int)))
        return out, avg, it

    def debug_pool(self):
        P, Nd, n = self.planes, self.n_domains, self.B * self.B
        pix = np.zeros((P, Nd, n), np.uint8)
        s = np.zeros((P, Nd), np.uint32)
        v = np.zeros((P, Nd), np.uint32)
        sc = np.zeros((P, self.height // 2, self.width // 2), np.uint8)
        capi.check(capi.lib().fic_ctx_debug_pool_host(self._h, capi.ptr(pix, C.c_uint8), capi.ptr(s, C.c_uint32),
                                                      capi.ptr(v, C.c_uint32), capi.ptr(sc, C.c_uint8)))
        return {"pix": pix, "sum": s, "var": v, "scaled": sc}


def encode_rgb_per_channel(rgb, B, wK=None, n_iso=1, device=0, sweep=0):
    """BASELINE.json config 5: a batch of colour images encoded as 3 independent grey planes each (NOT the
    reference's joint-RGB fit, which is `capi.encode_rgb`).  rgb: uint8 [N,H,W,3] (or [H,W,3]).
    Returns the result dict with arrays shaped [N, 3, N_r] (qrows [N, 3, N_r, 3]); channel order R, G, B."""
    a = np.ascontiguousarray(rgb, np.uint8)
    if a.ndim == 3:
        a = a[None]
    n, h, w, _ = a.shape
    planes = np.ascontiguousarray(a.transpose(0, 3, 1, 2)).reshape(n * 3, h, w)
    with Encoder(w, h, B, wK, n_iso, n * 3, device) as enc:
        if sweep:
            enc.set_option("sweep", sweep)
        enc.set_gray(planes)
        enc.encode()
        r = enc.results()
    return {k: v.reshape((n, 3) + v.shape[1:]) for k, v in r.items()}


def encode_gray(gray, B, wK=None, n_iso=1, device=0, sweep=0, chunks=0, q_shape=0):
    """One grey image (uint8 [H,W]) -> result dict with [N_r] arrays (plane axis dropped)."""
    g = np.ascontiguousarray(gray, np.uint8)
    with Encoder(g.shape[1], g.shape[0], B, wK, n_iso, 1, device) as enc:
        if sweep:
            enc.set_option("sweep", sweep)
        if chunks:
            enc.set_option("chunks", chunks)
        if q_shape:
            enc.set_option("q_shape", q_shape)
        enc.set_gray(g)
        enc.encode()
        r = enc.results()
        out = {k: v[0] for k, v in r.items()}
        out["wK"] = enc.wK
        return out


class FractalCompression:
    """Drop-in mirror of bvk_ss19.FractalCompression for the grey encode path.

    Static state like the reference (FC:14-20): set `blockgroesse` / `widthKernel`, call
    `encode(image, out)`.  `n_iso` and `device` are this build's additions (n_iso=1 = reference).
    """

    blockgroesse = 8          # FC:14
    widthKernel = 2           # FC:15
    n_iso = 1                 # extension; 1 == the reference algorithm
    device = 0
    imageInfo = None          # float32 [N_r][3] = {i_local, a, b}   (FC:17,124)
    imageIso = None           # int32 [N_r] winning isometry (extension)
    _last = None

    @staticmethod
    def isGreyScale(image):   # FC:32-45
        rc = capi.lib().fic_is_greyscale_argb(capi.ptr(image.argb, C.c_int32), image.width, image.height)
        return bool(capi.check(rc))

    @classmethod
    def encode(cls, image, out):   # FC:54-59
        if cls.isGreyScale(image):
            return cls.encodeGrayScale(image, out)
        return cls.encodeRGB(image, out)

    avgError = np.float32(0.0)   # FC:20 -- static, never reset between decode calls

    @classmethod
    def getAvgError(cls):   # FC:22-24
        return cls.avgError

    @classmethod
    def decode(cls, inputStream):   # FC:547-553: readInt() isRGB, then the matching decoder on the rest of the stream
        data = inputStream.read() if hasattr(inputStream, "read") else bytes(inputStream)
        if len(data) >= 4 and int.from_bytes(data[:4], "big", signed=True) != 0:      # FC:549-552 -> decodeRGB
            return cls._decode_rgb(data)
        return cls._decode_gray(data)

    @classmethod
    def _decode_gray(cls, run):
        gray, avg, _ = capi.decode_gray_run(run, cls.device, float(cls.avgError))
        cls.avgError = avg
        return RasterImage.from_gray(gray)

    @classmethod
    def _decode_rgb(cls, run):
        argb, avg, _, w, h = capi.decode_rgb_run(run, cls.device, float(cls.avgError))
        cls.avgError = avg
        return RasterImage(w, h, argb)

    @classmethod
    def decodeGreyScale(cls, inputStream):   # FC:356-421: the stream positioned AFTER the isRGB int (as FC.decode leaves it)
        rest = inputStream.read() if hasattr(inputStream, "read") else bytes(inputStream)
        return cls._decode_gray((0).to_bytes(4, "big") + rest)

    @classmethod
    def decodeRGB(cls, inputStream):   # FC:430-508: same convention
        rest = inputStream.read() if hasattr(inputStream, "read") else bytes(inputStream)
        return cls._decode_rgb((1).to_bytes(4, "big") + rest)

    @classmethod
    def generateKernel(cls, domainbloeckePerWidth, domainbloeckePerHeight, index):   # FC:84-100 (host integer logic)
        wK = cls.widthKernel
        dy = index // domainbloeckePerWidth - wK // 2
        dx = index % domainbloeckePerWidth - wK // 2
        dx, dy = max(dx, 0), max(dy, 0)
        if dx + wK >= domainbloeckePerWidth:
            dx = domainbloeckePerWidth - wK
        if dy + wK >= domainbloeckePerHeight:
            dy = domainbloeckePerHeight - wK
        return [dy, dx]

    @staticmethod
    def generateGrayImage(width, height):   # FC:1142-1148
        return RasterImage(width, height, np.full(width * height, np.int32(-8355712)))   # 0xff808080

    _collage = None
    _collageRGB = None

    @classmethod
    def getBestGeneratedCollage(cls, originalImage):   # FC:269-300: the collage of the last encodeGrayScale
        if cls._collage is None or cls._collage.size != originalImage.width * originalImage.height:
            raise FicError(-7, "getBestGeneratedCollage: no grey encode of this image size yet")
        return RasterImage(originalImage.width, originalImage.height, cls._collage.copy())

    @classmethod
    def getBestGeneratedCollageRGB(cls, originalImage):   # FC:308-347
        if cls._collageRGB is None or cls._collageRGB.size != originalImage.width * originalImage.height:
            raise FicError(-7, "getBestGeneratedCollageRGB: no RGB encode of this image size yet")
        return RasterImage(originalImage.width, originalImage.height, cls._collageRGB.copy())

    imageInfoRGB = None       # float32 [N_r][5] = {i_local, a, bR, bG, bB}   (FC:18,185)
    _lastRGB = None

    @classmethod
    def encodeRGB(cls, image, out):   # FC:171-219
        B, wK = cls.blockgroesse, cls.widthKernel
        r = capi.encode_rgb(image.argb, image.width, image.height, B, wK, cls.device, want_collage=True)
        cls.imageInfoRGB = np.stack([r["idx_local"].astype(np.float32), r["a"], r["bR"], r["bG"], r["bB"]], axis=1)
        cls._lastRGB = {"qrows": r["qrows"]}
        cls._collageRGB = np.asarray(r["collage"], np.int32).reshape(-1).copy()
        cls.writeData(out, 1, image.width, image.height)              # FC:217
        return cls.getBestGeneratedCollageRGB(image)                   # FC:218

    @classmethod
    def encodeGrayScale(cls, image, out):   # FC:109-162
        B, wK = cls.blockgroesse, cls.widthKernel
        with Encoder(image.width, image.height, B, wK, cls.n_iso, 1, cls.device) as enc:
            enc.set_argb(image.argb)
            enc.encode()
            r = enc.results()
            cls.imageInfo = np.stack([r["idx_local"][0].astype(np.float32), r["a"][0], r["b"][0]], axis=1)
            cls.imageIso = r["iso"][0].copy()
            cls._last = {"qrows": r["qrows"][0].copy(), "w": image.width, "h": image.height, "B": B, "wK": wK}
            cls._collage = np.asarray(enc.collage()[0], np.int32).reshape(-1).copy()
            cls.writeData(out, 0, image.width, image.height)          # FC:160
        return cls.getBestGeneratedCollage(image)                      # FC:161

    @classmethod
    def writeData(cls, out, isRGB, width, height):   # FC:230-261
        if isRGB != 0:                                               # FC:248-257
            if cls._lastRGB is None:
                raise FicError(-7, "writeData before encode")
            out.write(capi.write_run_rgb(cls._lastRGB["qrows"], width, height, cls.blockgroesse, cls.widthKernel))
            if hasattr(out, "close") and not isinstance(out, io.BytesIO):
                out.close()
            return
        if cls._last is None:
            raise FicError(-7, "writeData before encode")
        run = capi.write_run_gray(cls._last["qrows"], width, height, cls.blockgroesse, cls.widthKernel)
        out.write(run)
        if hasattr(out, "close") and not isinstance(out, io.BytesIO):
            out.close()                                               # FC:259
