"""ctypes front-end of the CPU oracle (oracle/jpeg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (nvimagecodec_amd) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_jpeg.so")
_lib = None

FMT_RGB, FMT_BGR, FMT_GRAY = 0, 1, 2
CS_GRAY, CS_YCC, CS_RGB, CS_CMYK, CS_YCCK = 0, 1, 2, 3, 4


class OracleError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"oracle {what} failed with code {code}")
        self.code = code


class _Info(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("width", "height", "ncomp", "sof", "colorspace", "restart_interval")] + [
        ("h", ctypes.c_int32 * 4), ("v", ctypes.c_int32 * 4), ("bw", ctypes.c_int32 * 4), ("bh", ctypes.c_int32 * 4),
        ("dw", ctypes.c_int32 * 4), ("dh", ctypes.c_int32 * 4), ("hmax", ctypes.c_int32), ("vmax", ctypes.c_int32)]


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "jpeg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_jpeg.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.c_void_p
        L.oj_read_info.argtypes = [u8p, ctypes.c_size_t, ctypes.POINTER(_Info)]
        L.oj_decode_coefficients.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, u8p]
        L.oj_decode_component_plane.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_int]
        L.oj_decode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int]
        L.oj_decode_cmyk.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_int]
        L.oj_quality_tables.argtypes = [ctypes.c_int, u8p, u8p]
        L.oj_quality_tables.restype = None
        L.oj_enc_geometry.argtypes = [ctypes.c_int] * 5 + [u8p, u8p]
        L.oj_enc_geometry.restype = None
        L.oj_forward.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p, u8p,
                                 u8p, u8p, u8p]
        L.oj_forward_planes.argtypes = [u8p, ctypes.c_int, u8p, ctypes.c_int, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        u8p, u8p, u8p, u8p, u8p]
        L.oj_encode.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, u8p, ctypes.c_size_t]
        L.oj_encode.restype = ctypes.c_long
        L.oj_set_idct_variant.argtypes = [ctypes.c_int]
        L.oj_set_idct_variant.restype = None
        _lib = L
    return _lib


IDCT_SIMD, IDCT_C = 0, 1


def set_idct_variant(variant):
    """Which jpeg_idct_islow the decode functions restate: IDCT_SIMD (default; libjpeg-turbo's x86-64 SIMD routine, what the reference's
    CPU path runs) or IDCT_C (jidctint.c, what JSIMD_FORCENONE=1 selects).  They agree on every stream an encoder can write."""
    lib().oj_set_idct_variant(int(variant))


def _buf(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return a, a.ctypes.data, a.size


def read_info(data):
    a, p, n = _buf(data)
    info = _Info()
    rc = lib().oj_read_info(p, n, ctypes.byref(info))
    if rc:
        raise OracleError(rc, "read_info")
    nc = info.ncomp
    return dict(width=info.width, height=info.height, ncomp=nc, sof=info.sof, colorspace=info.colorspace,
                restart_interval=info.restart_interval, hmax=info.hmax, vmax=info.vmax,
                h=list(info.h)[:nc], v=list(info.v)[:nc], bw=list(info.bw)[:nc], bh=list(info.bh)[:nc],
                dw=list(info.dw)[:nc], dh=list(info.dh)[:nc])


def decode_coefficients(data):
    """-> (list of int16 arrays [bh, bw, 64] natural order, list of uint16[64] quant tables)"""
    info = read_info(data)
    a, p, n = _buf(data)
    coefs, qts = [], []
    for c in range(info["ncomp"]):
        co = np.zeros((info["bh"][c], info["bw"][c], 64), dtype=np.int16)
        qt = np.zeros(64, dtype=np.uint16)
        rc = lib().oj_decode_coefficients(p, n, c, co.ctypes.data, qt.ctypes.data)
        if rc:
            raise OracleError(rc, "decode_coefficients")
        coefs.append(co)
        qts.append(qt)
    return coefs, qts


def decode_planes(data):
    """Raw IDCT output per component at its own resolution (dh x dw): what P_YUV / P_UNCHANGED return."""
    info = read_info(data)
    a, p, n = _buf(data)
    out = []
    for c in range(info["ncomp"]):
        pl = np.zeros((info["dh"][c], info["dw"][c]), dtype=np.uint8)
        rc = lib().oj_decode_component_plane(p, n, c, pl.ctypes.data, pl.strides[0])
        if rc:
            raise OracleError(rc, "decode_component_plane")
        out.append(pl)
    return out


def decode(data, fmt=FMT_RGB, fancy=True):
    """Full decode -> HxWx3 (RGB/BGR) or HxW (gray) uint8, the libjpeg_turbo_ext result for I_RGB / I_BGR / P_Y."""
    info = read_info(data)
    a, p, n = _buf(data)
    h, w = info["height"], info["width"]
    out = np.zeros((h, w) if fmt == FMT_GRAY else (h, w, 3), dtype=np.uint8)
    rc = lib().oj_decode(p, n, fmt, 1 if fancy else 0, out.ctypes.data, out.strides[0])
    if rc:
        raise OracleError(rc, "decode")
    return out


def decode_cmyk(data, fancy=True):
    """Four-component frames: libjpeg-turbo's JCS_CMYK output, HxWx4 (YCCK frames converted the way jdcolor.c does)."""
    info = read_info(data)
    a, p, n = _buf(data)
    out = np.zeros((info["height"], info["width"], 4), dtype=np.uint8)
    rc = lib().oj_decode_cmyk(p, n, 1 if fancy else 0, out.ctypes.data, out.strides[0])
    if rc:
        raise OracleError(rc, "decode_cmyk")
    return out


def quality_tables(quality):
    ql = np.zeros(64, dtype=np.uint16)
    qc = np.zeros(64, dtype=np.uint16)
    lib().oj_quality_tables(int(quality), ql.ctypes.data, qc.ctypes.data)
    return ql, qc


_SUBS = {"444": (1, 1), "422": (2, 1), "420": (2, 2), "440": (1, 2), "411": (4, 1), "410": (4, 2), "gray": (1, 1)}


def enc_geometry(w, h, subsampling):
    ncomp = 1 if subsampling == "gray" else 3
    hs, vs = _SUBS[subsampling]
    bw = (ctypes.c_int32 * 3)()
    bh = (ctypes.c_int32 * 3)()
    lib().oj_enc_geometry(w, h, ncomp, hs, vs, ctypes.addressof(bw), ctypes.addressof(bh))
    return ncomp, hs, vs, list(bw)[:ncomp], list(bh)[:ncomp]


def forward(rgb, subsampling="420", quality=90):
    """RGB HxWx3 -> quantized coefficient arrays per component ([bh, bw, 64] int16 natural order) + quant tables."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    ncomp, hs, vs, bw, bh = enc_geometry(w, h, subsampling)
    ql, qc = quality_tables(quality)
    coefs = [np.zeros((bh[c], bw[c], 64), dtype=np.int16) for c in range(ncomp)]
    ptrs = [coefs[c].ctypes.data if c < ncomp else None for c in range(3)]
    rc = lib().oj_forward(rgb.ctypes.data, rgb.strides[0], w, h, ncomp, hs, vs, ql.ctypes.data, qc.ctypes.data, *ptrs)
    if rc:
        raise OracleError(rc, "forward")
    return coefs, (ql, qc)


def forward_planes(planes, width, height, subsampling="420", quality=90):
    """Y, Cb, Cr planes that are the stream's components already (P_YUV input) -> quantized coefficient arrays per component
    ([bh, bw, 64] int16 natural order) + quant tables.  Edge rule: replicate the plane's last column / row (parity unpinned)."""
    ncomp, hs, vs, bw, bh = enc_geometry(width, height, subsampling)
    assert ncomp == 3 and len(planes) == 3
    pl = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
    ql, qc = quality_tables(quality)
    coefs = [np.zeros((bh[c], bw[c], 64), dtype=np.int16) for c in range(3)]
    rc = lib().oj_forward_planes(pl[0].ctypes.data, pl[0].strides[0], pl[1].ctypes.data, pl[1].strides[0], pl[2].ctypes.data, pl[2].strides[0],
                                 width, height, hs, vs, ql.ctypes.data, qc.ctypes.data, *[c.ctypes.data for c in coefs])
    if rc:
        raise OracleError(rc, "forward_planes")
    return coefs, (ql, qc)


def encode(rgb, subsampling="420", quality=90, restart_interval=0):
    """Baseline JPEG bytes (Annex-K Huffman tables, libjpeg marker order)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    ncomp = 1 if subsampling == "gray" else 3
    hs, vs = _SUBS[subsampling]
    cap = w * h * 6 + (1 << 16)  # noise at quality 100 codes to more than the raw picture
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().oj_encode(rgb.ctypes.data, rgb.strides[0], w, h, ncomp, hs, vs, int(quality), int(restart_interval), out.ctypes.data, cap)
    if n < 0 or n > cap:
        raise OracleError(n, "encode")
    return out[:n].tobytes()


def scan_bytes(jpeg):
    """Entropy-coded segment(s) of a JPEG: everything after the first SOS header up to EOI."""
    b = bytes(jpeg)
    i = 2
    while i < len(b):
        assert b[i] == 0xFF
        m = b[i + 1]
        L = (b[i + 2] << 8) | b[i + 3]
        if m == 0xDA:
            start = i + 2 + L
            end = b.rfind(b"\xff\xd9")
            return b[start:end]
        i += 2 + L
    raise ValueError("no SOS")
