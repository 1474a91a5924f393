/*
 * fpv_link.h — deployment-side codecs: CRSF serial frames and PPM trainer frames.
 * Host C, no GPU.  Each entry point names the reference routine it replaces; quirks of the reference are kept
 * and called out.
 */
#ifndef FPV_LINK_H
#define FPV_LINK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPV_CRSF_SYNC 0xC8                 /* controller/receive_telemetry.py:6 */
#define FPV_CRSF_RC_CHANNELS_PACKED 0x16   /* :16 */
#define FPV_CRSF_ATTITUDE 0x1E             /* :17 */
#define FPV_CRSF_BATTERY_SENSOR 0x08       /* :10 */
#define FPV_CRSF_LINK_STATISTICS 0x14      /* :15 */
#define FPV_CRSF_GPS 0x02                  /* :9 */

/* crc8_data(): CRC-8/DVB-S2 (poly 0xD5, init 0) — receive_telemetry.py:24-37 */
uint8_t fpv_crc8_dvb_s2(const uint8_t* data, size_t n);

/* packCrsfToBytes(): 16 channels x 11 bits -> 22 bytes, little-endian bit order — :45-71.  Values are masked to
 * 11 bits (the reference lets a larger value spill into the next channel). */
void fpv_crsf_pack_channels(const uint16_t ch[16], uint8_t out[22]);
void fpv_crsf_unpack_channels(const uint8_t in[22], uint16_t ch[16]);

/* channelsCrsfToChannelsPacket(): [0xC8, 24, 0x16, 22 payload bytes, crc] = 26 bytes — :73-77 */
void fpv_crsf_channels_frame(const uint16_t ch[16], uint8_t out[26]);

/* crsf_validate_frame(): crc over frame[2 : n-1] equals frame[n-1] — :39-40 */
int fpv_crsf_validate(const uint8_t* frame, size_t n);

/* The read loop of :150-164 as a resumable parser.  Feed bytes; whenever a complete frame is found it is copied to
 * `frame` (<= 64 bytes), *frame_len is set, and 1 is returned with *consumed = bytes taken from `data` so far; call
 * again with the rest.  Returns 0 when more bytes are needed, -1 for a frame whose CRC failed (it is still copied
 * out, as the reference prints it).  Like the reference it does not look for the sync byte: a length byte outside
 * 2..62 flushes the buffer. */
typedef struct {
    uint8_t buf[128];
    size_t fill;
} fpv_crsf_parser;
void fpv_crsf_parser_init(fpv_crsf_parser* p);
int fpv_crsf_parser_feed(fpv_crsf_parser* p, const uint8_t* data, size_t n, size_t* consumed, uint8_t* frame,
                         size_t* frame_len);

/* handleCrsfPacket() decoders — :96-100 (attitude), :104-108 (battery), :79-91 (link statistics), :116-122 (GPS).
 * Attitude: big-endian int16 / 10000 * 57.2958, i.e. DEGREES although the reference's print says "(rad)".
 * Battery capacity keeps the reference's `data[8] << 7` (an 8 was probably meant). */
typedef struct { double pitch, roll, yaw; } fpv_crsf_attitude;
typedef struct { double vbat, curr; uint32_t mah; uint8_t pct; } fpv_crsf_battery;
typedef struct { int rssi1, rssi2, lq, snr, antenna, mode, power, downlink_rssi, downlink_lq, downlink_snr; } fpv_crsf_link;
typedef struct { double lat, lon, gspd, hdg; int alt, sats; } fpv_crsf_gps;
int fpv_crsf_decode_attitude(const uint8_t* frame, size_t n, fpv_crsf_attitude* out);
int fpv_crsf_decode_battery(const uint8_t* frame, size_t n, fpv_crsf_battery* out);
int fpv_crsf_decode_link(const uint8_t* frame, size_t n, fpv_crsf_link* out);
int fpv_crsf_decode_gps(const uint8_t* frame, size_t n, fpv_crsf_gps* out);

/* generate_ppm_frame(): 8 channels, 300 us separator pulses, 22.5 ms frame (sync gap >= 3 ms else 4 ms), 48 kHz,
 * amplitude 0.8 — controller/send_controll_commands_through_jack.py:6-66.  Writes the samples of one frame to `out`
 * (capacity `cap`) and returns their number, or -1 if `cap` is too small. */
int fpv_ppm_frame(const int channel_us[8], int invert_polarity, float* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
