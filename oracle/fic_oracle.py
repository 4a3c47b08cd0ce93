"""ctypes binding of oracle/libfic_oracle.so (the CPU restatement of the Java codec).

TEST INFRASTRUCTURE ONLY -- see the header of fic_oracle.c.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfic_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "fic_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        i32p = C.POINTER(C.c_int32)
        f32p = C.POINTER(C.c_float)
        u8p = C.POINTER(C.c_uint8)
        ip = C.POINTER(C.c_int)
        L.fo_java_f2i.argtypes = [C.c_float]
        L.fo_java_f2i.restype = C.c_int32
        L.fo_geometry.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, ip, ip]
        L.fo_scale_image.argtypes = [i32p, C.c_int, C.c_int, i32p]
        L.fo_scale_image.restype = None
        L.fo_pool.argtypes = [i32p, C.c_int, C.c_int, C.c_int, i32p, i32p, f32p]
        L.fo_generate_kernel.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip, ip]
        L.fo_generate_kernel.restype = None
        L.fo_domain_block_index.argtypes = [C.c_int] * 6
        L.fo_iso_source.argtypes = [C.c_int] * 4
        L.fo_encode_gray.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     f32p, i32p, f32p]
        L.fo_write_run_gray.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.fo_write_run_gray.restype = C.c_int64
        L.fo_calculate_indices.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.fo_calculate_indices.restype = None
        L.fo_decode_gray.argtypes = [u8p, C.c_int64, i32p, C.c_int, f32p, ip, ip, ip]
        L.fo_collage_gray.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, i32p]
        L.fo_is_greyscale.argtypes = [i32p, C.c_int, C.c_int]
        L.fo_encode_rgb.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        L.fo_write_run_rgb.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.fo_write_run_rgb.restype = C.c_int64
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def gray_to_argb(gray):
    """uint8 [H,W] -> int32 ARGB [H*W] with r=g=b, alpha 255 (RasterImage.argb, RI:22-24)."""
    g = np.ascontiguousarray(gray, dtype=np.uint8).astype(np.uint32)
    return (0xFF000000 | (g << 16) | (g << 8) | g).astype(np.uint32).view(np.int32).reshape(-1).copy()


def rgb_to_argb(rgb):
    c = np.ascontiguousarray(rgb, dtype=np.uint8).astype(np.uint32)
    return (0xFF000000 | (c[..., 0] << 16) | (c[..., 1] << 8) | c[..., 2]).astype(np.uint32).view(np.int32).reshape(-1).copy()


def geometry(w, h, B):
    v = [C.c_int() for _ in range(4)]
    rc = lib().fo_geometry(w, h, B, *[C.byref(x) for x in v])
    if rc:
        raise ValueError(f"oracle geometry rc={rc} for w={w} h={h} B={B}")
    return tuple(x.value for x in v)  # Rw, Rh, Dw, Dh


def pool(argb, w, h, B):
    Rw, Rh, Dw, Dh = geometry(w, h, B)
    nd, n = Dw * Dh, B * B
    pix = np.zeros((nd, n), np.int32)
    mean = np.zeros(nd, np.int32)
    var = np.zeros(nd, np.float32)
    rc = lib().fo_pool(_p(argb, C.c_int32), w, h, B, _p(pix, C.c_int32), _p(mean, C.c_int32), _p(var, C.c_float))
    if rc:
        raise ValueError(f"fo_pool rc={rc}")
    return pix, mean, var


def encode_gray(argb, w, h, B, wK, n_iso=1, r0=0, r1=None):
    """Returns dict(info float32[N_r,3]={i_local,a,b}, iso int32[N_r], err float32[N_r]).
    Only rows r0:r1 are filled."""
    Rw, Rh, Dw, Dh = geometry(w, h, B)
    nr = Rw * Rh
    if r1 is None:
        r1 = nr
    info = np.zeros((nr, 3), np.float32)
    iso = np.zeros(nr, np.int32)
    err = np.zeros(nr, np.float32)
    argb = np.ascontiguousarray(argb, np.int32)
    rc = lib().fo_encode_gray(_p(argb, C.c_int32), w, h, B, wK, n_iso, r0, r1, _p(info, C.c_float),
                              _p(iso, C.c_int32), _p(err, C.c_float))
    if rc:
        raise ValueError(f"fo_encode_gray rc={rc}")
    return {"info": info, "iso": iso, "err": err}


def write_run_gray(info, w, h, B, wK):
    info = np.ascontiguousarray(info, np.float32)
    nr = info.shape[0]
    out = np.zeros(20 + 12 * nr, np.uint8)
    n = lib().fo_write_run_gray(_p(info, C.c_float), nr, w, h, B, wK, _p(out, C.c_uint8))
    assert n == out.size
    return out.tobytes()


def quantise_gray(info):
    """int32 [N_r,3] rows exactly as writeData emits them (FC:242-244)."""
    run = write_run_gray(info, 0, 0, 0, 0)
    return np.frombuffer(run[20:], dtype=">i4").astype(np.int32).reshape(-1, 3)


def decode_gray(run, avg_error_in=0.0):
    """Returns (gray uint8 [H,W], avgError float32 after the call, iterations)."""
    buf = np.frombuffer(run, np.uint8).copy()
    w = int.from_bytes(run[4:8], "big", signed=True)
    h = int.from_bytes(run[8:12], "big", signed=True)
    out = np.zeros(w * h, np.int32)
    avg = C.c_float(avg_error_in)
    iters, wo, ho = C.c_int(), C.c_int(), C.c_int()
    rc = lib().fo_decode_gray(_p(buf, C.c_uint8), buf.size, _p(out, C.c_int32), out.size, C.byref(avg),
                              C.byref(iters), C.byref(wo), C.byref(ho))
    if rc:
        raise ValueError(f"fo_decode_gray rc={rc}")
    gray = ((out.view(np.uint32) >> 16) & 0xFF).astype(np.uint8).reshape(h, w)
    return gray, np.float32(avg.value), iters.value


def decode_rows(qrows, iso, w, h, B, wK, avg_error_in=0.0):
    """Decoder loop from quantised rows + isometry ids (n_iso = 8 extension twin of decode_gray)."""
    L = lib()
    L.fo_decode_rows.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    q = np.ascontiguousarray(qrows, np.int32)
    k = None if iso is None else np.ascontiguousarray(iso, np.int32)
    out = np.zeros(w * h, np.int32)
    avg = C.c_float(avg_error_in)
    iters = C.c_int()
    rc = L.fo_decode_rows(_p(q, C.c_int32), None if k is None else _p(k, C.c_int32), w, h, B, wK, _p(out, C.c_int32),
                          C.byref(avg), C.byref(iters))
    if rc:
        raise ValueError(f"fo_decode_rows rc={rc}")
    gray = ((out.view(np.uint32) >> 16) & 0xFF).astype(np.uint8).reshape(h, w)
    return gray, np.float32(avg.value), iters.value


def collage_gray(argb, w, h, B, wK, info):
    info = np.ascontiguousarray(info, np.float32).copy()
    out = np.zeros(w * h, np.int32)
    argb = np.ascontiguousarray(argb, np.int32)
    rc = lib().fo_collage_gray(_p(argb, C.c_int32), w, h, B, wK, _p(info, C.c_float), _p(out, C.c_int32))
    if rc:
        raise ValueError(f"fo_collage_gray rc={rc}")
    return out


def is_greyscale(argb, w, h):
    argb = np.ascontiguousarray(argb, np.int32)
    return bool(lib().fo_is_greyscale(_p(argb, C.c_int32), w, h))


def encode_rgb(argb, w, h, B, wK):
    Rw, Rh, Dw, Dh = geometry(w, h, B)
    info = np.zeros((Rw * Rh, 5), np.float32)
    argb = np.ascontiguousarray(argb, np.int32)
    rc = lib().fo_encode_rgb(_p(argb, C.c_int32), w, h, B, wK, _p(info, C.c_float))
    if rc:
        raise ValueError(f"fo_encode_rgb rc={rc}")
    return info


def decode_rgb(run, avg_error_in=0.0):
    """decodeRGB (FC:430-508).  Returns (rgb uint8 [H,W,3], avgError float32, iterations)."""
    L = lib()
    L.fo_decode_rgb.argtypes = [C.POINTER(C.c_uint8), C.c_int64, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float),
                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    buf = np.frombuffer(run, np.uint8).copy()
    w = int.from_bytes(run[4:8], "big", signed=True)
    h = int.from_bytes(run[8:12], "big", signed=True)
    out = np.zeros(w * h, np.int32)
    avg = C.c_float(avg_error_in)
    iters, wo, ho = C.c_int(), C.c_int(), C.c_int()
    rc = L.fo_decode_rgb(_p(buf, C.c_uint8), buf.size, _p(out, C.c_int32), out.size, C.byref(avg), C.byref(iters),
                         C.byref(wo), C.byref(ho))
    if rc:
        raise ValueError(f"fo_decode_rgb rc={rc}")
    u = out.view(np.uint32)
    rgb = np.stack([(u >> 16) & 0xFF, (u >> 8) & 0xFF, u & 0xFF], axis=-1).astype(np.uint8).reshape(h, w, 3)
    return rgb, np.float32(avg.value), iters.value


def collage_rgb(argb, w, h, B, wK, info):
    L = lib()
    L.fo_collage_rgb.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                 C.POINTER(C.c_int32)]
    info = np.ascontiguousarray(info, np.float32).copy()
    out = np.zeros(w * h, np.int32)
    argb = np.ascontiguousarray(argb, np.int32)
    rc = L.fo_collage_rgb(_p(argb, C.c_int32), w, h, B, wK, _p(info, C.c_float), _p(out, C.c_int32))
    if rc:
        raise ValueError(f"fo_collage_rgb rc={rc}")
    return out


def quantise_rgb(info):
    run = write_run_rgb(info, 0, 0, 0, 0)
    return np.frombuffer(run[20:], dtype=">i4").astype(np.int32).reshape(-1, 5)


def write_run_rgb(info, w, h, B, wK):
    info = np.ascontiguousarray(info, np.float32)
    nr = info.shape[0]
    out = np.zeros(20 + 20 * nr, np.uint8)
    n = lib().fo_write_run_rgb(_p(info, C.c_float), nr, w, h, B, wK, _p(out, C.c_uint8))
    assert n == out.size
    return out.tobytes()


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
