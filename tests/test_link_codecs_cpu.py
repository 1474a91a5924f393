"""CRSF / PPM codecs against restatements of controller/receive_telemetry.py and
controller/send_controll_commands_through_jack.py written here in the test (the reference files import pyserial /
sounddevice at module top and cannot be imported)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from fpv_drone_rl_agent_amd import deploy as D


def ref_crc8(data):                      # receive_telemetry.py:24-37
    crc = 0
    for a in data:
        crc ^= a
        for _ in range(8):
            crc = ((crc << 1) ^ 0xD5) if crc & 0x80 else (crc << 1)
        crc &= 0xFF
    return crc


def ref_pack(channels):                  # receive_telemetry.py:45-71
    result, dest_shift, new_val = bytearray(), 0, 0
    for ch in channels:
        new_val |= (ch << dest_shift) & 0xFF
        result.append(new_val)
        src_bits_left = 11 - 8 + dest_shift
        new_val = ch >> (11 - src_bits_left)
        if src_bits_left >= 8:
            result.append(new_val & 0xFF)
            new_val >>= 8
            src_bits_left -= 8
        dest_shift = src_bits_left
    return bytes(result)


def test_crc8_dvb_s2_check_value_and_reference():
    assert D.crc8(b"123456789") == 0xBC                       # catalogue check value of CRC-8/DVB-S2
    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 24, 64):
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert D.crc8(b) == ref_crc8(b)


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 0x7FF), min_size=16, max_size=16))
def test_channel_packing_matches_reference_and_roundtrips(ch):
    packed = D.pack_channels(ch)
    assert len(packed) == 22 and packed == ref_pack(ch)
    assert D.unpack_channels(packed) == ch


def test_channel_bit_layout():
    # "bits BA987654321 -> 87654321, 00000BA9" (receive_telemetry.py:47)
    p = D.pack_channels([0x7FF] + [0] * 15)
    assert p[0] == 0xFF and p[1] == 0x07 and p[2:] == bytes(20)
    p = D.pack_channels([0, 0x7FF] + [0] * 14)
    assert p[0] == 0 and p[1] == 0xF8 and p[2] == 0x3F


def test_channels_frame_and_validation():
    f = D.channels_frame([992] * 16)                          # 1500 us in CRSF units
    assert len(f) == 26 and f[0] == 0xC8 and f[1] == 24 and f[2] == 0x16     # :74
    assert f[-1] == ref_crc8(f[2:-1]) and D.validate(f)                      # :39-40, :76
    bad = bytearray(f); bad[10] ^= 1
    assert not D.validate(bytes(bad))


def _frame(ptype, payload):
    body = bytes([ptype]) + bytes(payload)
    return bytes([0xC8, len(body) + 1]) + body + bytes([ref_crc8(body)])


def test_stream_parser_resyncs_like_the_reference():
    att = _frame(0x1E, (1000).to_bytes(2, "big", signed=True) + (-2500).to_bytes(2, "big", signed=True) + (31416).to_bytes(2, "big", signed=True))
    bat = _frame(0x08, (168).to_bytes(2, "big") + (53).to_bytes(2, "big") + bytes([0x01, 0x02, 0x03, 87]))
    corrupt = bytearray(att); corrupt[4] ^= 0xFF
    stream = att + bat + bytes(corrupt) + bytes([0x00, 0xFF, 0x01]) + att   # 0xFF length flushes the buffer (:156-157)
    p = D.CrsfParser()
    got = []
    for i in range(0, len(stream), 7):                                      # arbitrary chunking
        got += p.feed(stream[i:i + 7])
    assert [ok for ok, _ in got[:3]] == [True, True, False]
    assert got[0][1] == att and got[1][1] == bat
    pitch, roll, yaw = D.decode_attitude(got[0][1])
    assert np.isclose(pitch, 1000 / 10000 * 57.2958) and np.isclose(roll, -2500 / 10000 * 57.2958)   # :97-99 (degrees)
    vbat, curr, mah, pct = D.decode_battery(got[1][1])
    assert (vbat, curr, pct) == (16.8, 5.3, 87) and mah == (0x01 << 16 | 0x02 << 7 | 0x03)           # :105-108, `<< 7` kept
    with pytest.raises(ValueError):
        D.decode_attitude(bat)


def ref_ppm(values, invert=False):       # send_controll_commands_through_jack.py:19-66
    us = lambda u: int(u * 48000 / 1_000_000)
    hi, lo = (-0.8, 0.8) if invert else (0.8, -0.8)
    parts, total = [], 0
    for v in values:
        v = int(np.clip(v, 700, 2300))
        parts += [np.full(us(300), hi, np.float32), np.full(us(v - 300), lo, np.float32)]
        total += v
    parts.append(np.full(us(300), hi, np.float32))
    total += 300
    gap = 22.5 * 1000 - total
    if gap < 3000:
        gap = 4000
    parts.append(np.full(us(gap), lo, np.float32))
    return np.concatenate(parts)


@pytest.mark.parametrize("values,invert", [([1500] * 8, False), ([1000, 2000, 1500, 700, 2300, 1234, 1777, 600], True),
                                           ([2300] * 8, False)])
def test_ppm_frame_matches_reference(values, invert):
    w = D.ppm_frame(values, invert)
    r = ref_ppm(values, invert)
    assert w.shape == r.shape and np.array_equal(w, r)
    if values == [1500] * 8:
        assert abs(len(w) / 48000 - 0.0225) < 2e-4            # a 22.5 ms frame
