"""Deployment-side codecs: ctypes binding of libfpv_link.so (include/fpv_link.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfpv_link.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fpv_link.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-Wall", "-shared", "-o", LIB_PATH, src])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.fpv_crc8_dvb_s2.restype = C.c_uint8
        _lib.fpv_crc8_dvb_s2.argtypes = [C.c_char_p, C.c_size_t]
        _lib.fpv_ppm_frame.restype = C.c_int
    return _lib


def crc8(data: bytes) -> int:
    return int(lib().fpv_crc8_dvb_s2(bytes(data), len(data)))


def pack_channels(ch) -> bytes:
    a = (C.c_uint16 * 16)(*[int(c) for c in ch])
    out = (C.c_uint8 * 22)()
    lib().fpv_crsf_pack_channels(a, out)
    return bytes(out)


def unpack_channels(b: bytes):
    a = (C.c_uint8 * 22)(*b)
    out = (C.c_uint16 * 16)()
    lib().fpv_crsf_unpack_channels(a, out)
    return list(out)


def channels_frame(ch) -> bytes:
    a = (C.c_uint16 * 16)(*[int(c) for c in ch])
    out = (C.c_uint8 * 26)()
    lib().fpv_crsf_channels_frame(a, out)
    return bytes(out)


def validate(frame: bytes) -> bool:
    return bool(lib().fpv_crsf_validate(bytes(frame), len(frame)))


class _Parser(C.Structure):
    _fields_ = [("buf", C.c_uint8 * 128), ("fill", C.c_size_t)]


class CrsfParser:
    """Resumable form of the reference's read loop (controller/receive_telemetry.py:150-164)."""

    def __init__(self):
        self._p = _Parser()
        lib().fpv_crsf_parser_init(C.byref(self._p))

    def feed(self, data: bytes):
        """Returns a list of (ok, frame_bytes)."""
        out, data = [], bytes(data)
        frame = (C.c_uint8 * 64)()
        while True:
            used, flen = C.c_size_t(), C.c_size_t()
            rc = lib().fpv_crsf_parser_feed(C.byref(self._p), data, len(data), C.byref(used), frame, C.byref(flen))
            data = data[used.value:]
            if rc == 0:
                return out
            out.append((rc == 1, bytes(frame[: flen.value])))


class _Att(C.Structure):
    _fields_ = [("pitch", C.c_double), ("roll", C.c_double), ("yaw", C.c_double)]


class _Bat(C.Structure):
    _fields_ = [("vbat", C.c_double), ("curr", C.c_double), ("mah", C.c_uint32), ("pct", C.c_uint8)]


def decode_attitude(frame: bytes):
    o = _Att()
    if lib().fpv_crsf_decode_attitude(bytes(frame), len(frame), C.byref(o)) != 0:
        raise ValueError("not an ATTITUDE frame")
    return o.pitch, o.roll, o.yaw


def decode_battery(frame: bytes):
    o = _Bat()
    if lib().fpv_crsf_decode_battery(bytes(frame), len(frame), C.byref(o)) != 0:
        raise ValueError("not a BATTERY_SENSOR frame")
    return o.vbat, o.curr, int(o.mah), int(o.pct)


def ppm_frame(channel_us, invert: bool = False) -> np.ndarray:
    a = (C.c_int * 8)(*[int(c) for c in channel_us])
    out = np.zeros(4096, np.float32)
    n = lib().fpv_ppm_frame(a, int(invert), out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
    if n < 0:
        raise RuntimeError("PPM frame does not fit")
    return out[:n]
