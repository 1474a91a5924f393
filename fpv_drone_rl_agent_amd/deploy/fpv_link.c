/* fpv_link.c — CRSF / PPM codecs (include/fpv_link.h).  Written from the behaviour of controller/receive_telemetry.py
 * and controller/send_controll_commands_through_jack.py; see the header for the line references. */
#include "../../include/fpv_link.h"

#include <string.h>

uint8_t fpv_crc8_dvb_s2(const uint8_t* data, size_t n) {
    uint8_t crc = 0;
    for (size_t i = 0; i < n; ++i) {
        crc ^= data[i];
        for (int b = 0; b < 8; ++b) crc = (crc & 0x80) ? (uint8_t)((crc << 1) ^ 0xD5) : (uint8_t)(crc << 1);
    }
    return crc;
}

void fpv_crsf_pack_channels(const uint16_t ch[16], uint8_t out[22]) {
    uint32_t acc = 0;      /* bit accumulator, least significant bits leave first */
    int bits = 0, o = 0;
    for (int c = 0; c < 16; ++c) {
        acc |= (uint32_t)(ch[c] & 0x7FF) << bits;
        bits += 11;
        while (bits >= 8) {
            out[o++] = (uint8_t)(acc & 0xFF);
            acc >>= 8;
            bits -= 8;
        }
    }
}

void fpv_crsf_unpack_channels(const uint8_t in[22], uint16_t ch[16]) {
    uint32_t acc = 0;
    int bits = 0, i = 0;
    for (int c = 0; c < 16; ++c) {
        while (bits < 11) {
            acc |= (uint32_t)in[i++] << bits;
            bits += 8;
        }
        ch[c] = (uint16_t)(acc & 0x7FF);
        acc >>= 11;
        bits -= 11;
    }
}

void fpv_crsf_channels_frame(const uint16_t ch[16], uint8_t out[26]) {
    out[0] = FPV_CRSF_SYNC;
    out[1] = 24;
    out[2] = FPV_CRSF_RC_CHANNELS_PACKED;
    fpv_crsf_pack_channels(ch, out + 3);
    out[25] = fpv_crc8_dvb_s2(out + 2, 23);
}

int fpv_crsf_validate(const uint8_t* frame, size_t n) {
    if (n < 4) return 0;
    return fpv_crc8_dvb_s2(frame + 2, n - 3) == frame[n - 1];
}

void fpv_crsf_parser_init(fpv_crsf_parser* p) { p->fill = 0; }

int fpv_crsf_parser_feed(fpv_crsf_parser* p, const uint8_t* data, size_t n, size_t* consumed, uint8_t* frame,
                         size_t* frame_len) {
    size_t used = 0;
    for (;;) {
        /* top the buffer up */
        while (p->fill < sizeof p->buf && used < n) p->buf[p->fill++] = data[used++];
        if (p->fill <= 2) break;
        size_t expected = (size_t)p->buf[1] + 2;
        if (expected > 64 || expected < 4) {       /* reference: input = bytearray() */
            p->fill = 0;
            if (used == n) break;
            continue;
        }
        if (p->fill < expected) {
            if (used == n) break;
            continue;
        }
        memcpy(frame, p->buf, expected);
        *frame_len = expected;
        memmove(p->buf, p->buf + expected, p->fill - expected);
        p->fill -= expected;
        *consumed = used;
        return fpv_crsf_validate(frame, expected) ? 1 : -1;
    }
    *consumed = used;
    return 0;
}

static int be16s(const uint8_t* b) { return (int16_t)((b[0] << 8) | b[1]); }
static int32_t be32s(const uint8_t* b) {
    return (int32_t)(((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]);
}
static int sbyte(uint8_t b) { return b >= 128 ? (int)b - 256 : (int)b; }

int fpv_crsf_decode_attitude(const uint8_t* f, size_t n, fpv_crsf_attitude* o) {
    if (n < 10 || f[2] != FPV_CRSF_ATTITUDE) return -1;
    o->pitch = be16s(f + 3) / 10000.0 * 57.2958;
    o->roll = be16s(f + 5) / 10000.0 * 57.2958;
    o->yaw = be16s(f + 7) / 10000.0 * 57.2958;
    return 0;
}

int fpv_crsf_decode_battery(const uint8_t* f, size_t n, fpv_crsf_battery* o) {
    if (n < 12 || f[2] != FPV_CRSF_BATTERY_SENSOR) return -1;
    o->vbat = be16s(f + 3) / 10.0;
    o->curr = be16s(f + 5) / 10.0;
    o->mah = ((uint32_t)f[7] << 16) | ((uint32_t)f[8] << 7) | f[9];   /* `<< 7` as in the reference */
    o->pct = f[10];
    return 0;
}

int fpv_crsf_decode_link(const uint8_t* f, size_t n, fpv_crsf_link* o) {
    if (n < 14 || f[2] != FPV_CRSF_LINK_STATISTICS) return -1;
    o->rssi1 = sbyte(f[3]); o->rssi2 = sbyte(f[4]); o->lq = f[5]; o->snr = sbyte(f[6]);
    o->antenna = f[7]; o->mode = f[8]; o->power = f[9];
    o->downlink_rssi = sbyte(f[10]); o->downlink_lq = f[11]; o->downlink_snr = sbyte(f[12]);
    return 0;
}

int fpv_crsf_decode_gps(const uint8_t* f, size_t n, fpv_crsf_gps* o) {
    if (n < 19 || f[2] != FPV_CRSF_GPS) return -1;
    o->lat = be32s(f + 3) / 1e7;
    o->lon = be32s(f + 7) / 1e7;
    o->gspd = be16s(f + 11) / 36.0;
    o->hdg = be16s(f + 13) / 100.0;
    o->alt = be16s(f + 15) - 1000;
    o->sats = f[17];
    return 0;
}

static int us_to_samples(double us) { return (int)(us * 48000.0 / 1000000.0); }

int fpv_ppm_frame(const int channel_us[8], int invert, float* out, int cap) {
    const float pulse = invert ? -0.8f : 0.8f, wait = invert ? 0.8f : -0.8f;
    const int sep = us_to_samples(300.0);
    int n = 0;
    double total_us = 0.0;
    for (int c = 0; c < 8; ++c) {
        int v = channel_us[c] < 700 ? 700 : (channel_us[c] > 2300 ? 2300 : channel_us[c]);
        int w = us_to_samples((double)(v - 300));
        if (n + sep + w > cap) return -1;
        for (int i = 0; i < sep; ++i) out[n++] = pulse;
        for (int i = 0; i < w; ++i) out[n++] = wait;
        total_us += v;
    }
    if (n + sep > cap) return -1;
    for (int i = 0; i < sep; ++i) out[n++] = pulse;
    total_us += 300.0;
    double gap = 22.5 * 1000.0 - total_us;
    if (gap < 3000.0) gap = 4000.0;
    int g = us_to_samples(gap);
    if (n + g > cap) return -1;
    for (int i = 0; i < g; ++i) out[n++] = wait;
    return n;
}
