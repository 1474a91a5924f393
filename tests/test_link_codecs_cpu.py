"""CRSF / PPM codecs (controller/receive_telemetry.py, controller/send_controll_commands_through_jack.py)
against committed input / output vectors (tests/golden/link_codecs_v1.json, made by tests/golden/make_link_codecs.py from
the formats' definitions: CRC-8/DVB-S2 by polynomial division, channel packing as one 176-bit little-endian integer, the
PPM frame as run lengths).  The reference files import pyserial / sounddevice at module top and cannot be imported."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from fpv_drone_rl_agent_amd import deploy as D


GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "link_codecs_v1.json")))


def test_crc8_dvb_s2_check_value_and_vectors():
    assert D.crc8(b"123456789") == 0xBC                       # catalogue check value of CRC-8/DVB-S2
    for v in GOLD["crc8_dvb_s2"]:
        assert D.crc8(bytes.fromhex(v["hex"])) == v["crc"]


def test_channel_packing_matches_vectors():
    for v in GOLD["rc_channels_packed"]:
        packed = D.pack_channels(v["channels"])
        assert len(packed) == 22 and packed.hex() == v["hex"]
        assert D.unpack_channels(packed) == v["channels"]


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 0x7FF), min_size=16, max_size=16))
def test_channel_packing_is_the_176_bit_little_endian_integer_and_roundtrips(ch):
    packed = D.pack_channels(ch)
    assert packed == sum(c << (11 * i) for i, c in enumerate(ch)).to_bytes(22, "little")
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
    assert f[-1] == D.crc8(f[2:-1]) and D.validate(f)                        # :39-40, :76 (crc8 itself: vectors above)
    bad = bytearray(f); bad[10] ^= 1
    assert not D.validate(bytes(bad))


def _frame(ptype, payload):
    body = bytes([ptype]) + bytes(payload)
    return bytes([0xC8, len(body) + 1]) + body + bytes([D.crc8(body)])


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


@pytest.mark.parametrize("case", range(len(GOLD["ppm_48k"])))
def test_ppm_frame_matches_vectors(case):
    v = GOLD["ppm_48k"][case]
    w = D.ppm_frame(v["values"], v["invert"])
    expect = np.concatenate([np.full(n, level, np.float32) for level, n in v["runs"]])
    assert w.shape == expect.shape and np.array_equal(w, expect)
    if v["values"] == [1500] * 8:
        assert abs(len(w) / 48000 - 0.0225) < 2e-4            # a 22.5 ms frame
