"""Golden vectors for the CRSF / PPM codecs (controller/receive_telemetry.py:24-77,
controller/send_controll_commands_through_jack.py:19-66) -> tests/golden/link_codecs_v1.json.

The expected outputs are worked out here from the formats' definitions, in forms that share nothing with an
implementation's loops:
  * CRC-8/DVB-S2 as the remainder of message(x) * x^8 modulo x^8 + x^7 + x^6 + x^4 + x^2 + 1 over GF(2), by long division
    on one big integer (catalogue check value 0xBC for "123456789");
  * CRSF RC channels: sixteen 11-bit values, least significant bit first, = the 176-bit little-endian integer
    sum(ch[i] << 11 i);
  * the PPM frame as run lengths in samples at 48 kHz: per channel a 300 us mark and a (value - 300) us space with the value
    clamped to 700..2300, a closing 300 us mark, then a space filling the frame to 22.5 ms (4 000 us if less than 3 000 us would
    remain); mark / space levels +0.8 / -0.8, swapped when inverted; every duration truncated to whole samples.
Run: python tests/golden/make_link_codecs.py
"""
import json
import os
import random


def crc8_dvb_s2(data: bytes) -> int:
    poly = 0x1D5                                       # x^8 + x^7 + x^6 + x^4 + x^2 + 1
    n = int.from_bytes(data, "big") << 8
    for bit in range(n.bit_length() - 1, 7, -1):
        if n >> bit & 1:
            n ^= poly << (bit - 8)
    return n & 0xFF


def pack16(ch):
    return sum(c << (11 * i) for i, c in enumerate(ch)).to_bytes(22, "little")


def ppm_runs(values, invert):
    samples = lambda us: int(us * 48000 / 1_000_000)
    mark, space = (-0.8, 0.8) if invert else (0.8, -0.8)
    runs, used = [], 0
    for v in values:
        v = min(max(int(v), 700), 2300)
        runs += [[mark, samples(300)], [space, samples(v - 300)]]
        used += v
    runs.append([mark, samples(300)])
    rest = 22500 - (used + 300)
    runs.append([space, samples(rest if rest >= 3000 else 4000)])
    return runs


def main():
    rnd = random.Random(20251004)
    assert crc8_dvb_s2(b"123456789") == 0xBC
    crc = [{"hex": b.hex(), "crc": crc8_dvb_s2(b)}
           for b in [b"", b"\x00", b"\xff", b"123456789"] + [bytes(rnd.randrange(256) for _ in range(n)) for n in (1, 5, 24, 24, 64)]]
    chans = [[0x7FF] + [0] * 15, [0, 0x7FF] + [0] * 14, [992] * 16, list(range(16)), [0x7FF] * 16] + \
            [[rnd.randrange(0x800) for _ in range(16)] for _ in range(24)]
    pack = [{"channels": c, "hex": pack16(c).hex()} for c in chans]
    ppm = [{"values": v, "invert": inv, "runs": ppm_runs(v, inv)}
           for v, inv in [([1500] * 8, False), ([1000, 2000, 1500, 700, 2300, 1234, 1777, 600], True), ([2300] * 8, False),
                          ([700] * 8, True), ([rnd.randrange(500, 2500) for _ in range(8)], False)]]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "link_codecs_v1.json")
    json.dump({"crc8_dvb_s2": crc, "rc_channels_packed": pack, "ppm_48k": ppm}, open(out, "w"), indent=0)
    print("wrote", out)


if __name__ == "__main__":
    main()
