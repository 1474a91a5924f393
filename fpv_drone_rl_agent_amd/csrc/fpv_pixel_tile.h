// fpv_pixel_tile.h — the pixel detector (hover.py:157-222, SPEC §6b) of a POOL of 64-drone tiles: one blob, one group of lanes.
//
// What the reference does with the thresholded camera image — the border test, cv2.findContours, the largest contour,
// cv2.arcLength, cv2.approxPolyDP, "four vertices or invisible", cv2.contourArea, cv2.boundingRect (hover.py:180-213) — is
// integer work on a blob of 1 .. ~60 image rows that a fifth of the drones have at all.  One lane per drone would keep its wave
// busy for the slowest blob of the tile while most lanes idle, and a wave issues one vector instruction per ~5 cycles whatever
// its ILP.  A launch is as slow as its slowest workgroup, so what counts is the worst wave of ~1 000, not the average.  So:
//
//  1. The lanes that own the drones project the panel's corners and drop the blobs that cannot be seen whole (behind the camera,
//     beside the image, touching its border: pixel_touches_border is exact); corners, slopes and row range go to LDS.
//  2. The waves of the pool — the tile's one wave in the stand-alone step kernels, all eight waves of the workgroup for its four
//     tiles in fpv_rollstep.hip — each derive the same schedule from the row ranges: the tallest blobs first, 16 lanes (one DPP
//     row) each — as many of them as one round of the pool's lanes has room for —, then the others, 8 lanes each; "octet" o of
//     the schedule is worked by lanes 8 (o mod 8) .. of wave (o / 8) mod waves in round o / (8 waves).  A group evaluates its blob's rows into its private span table, then
//     runs fpv_polydp.h's poly_blob on it: statistics and cv2's CHAIN_APPROX_SIMPLE contour in one pass over the table, the
//     contour into registers, arcLength, Douglas-Peucker with its farthest-point sweeps spread over the group's lanes (DPP
//     reductions), the clean-up pass.  The result (visible, vertex sums, 2 x contourArea, boundingRect) goes to the drone's
//     result slot.
//  3. The owner lanes turn their drones' results into obs[7:9, 11, 13, 14].
//
// Integers and a few doubles only: fp32 and fp64 handles take the same decisions from the same spans, and the CPU oracle's bit
// for bit.
//
// LDS of a pool, `pool` = PixPool<T, TILES, WAVES>::kWords 4-byte words (32-byte aligned):
//   per tile: corner block [4][64] Vec4<T> (u, v, edge slopes, {jlo, jhi, -, -} of every drone; jlo > jhi: nothing to do) and
//   results [64] int4 ({visible | w << 8 | h << 16, sum x | sum y << 16, 2 x contourArea, -});
//   cv2.arcLength's float32 length of a run of k = 0 .. 127 diagonal steps in units of 2^-23 (128 words);
//   the census: per tile the numbers of its blobs of more than 0 / 16 / 20 / 24 / 28 / 36 rows (two words), then the octet count;
//   the schedule: one u16 per octet (drone | second half << 8 | 16 lanes << 9);
//   per octet of lanes: span table (a | b << 8 per row, 0x00FF = no red pixel, framed by two empty rows) and contour list; a
//   16-lane group uses the space of its two octets.
// The caller orders the phases (setup -> contours -> features) with whatever makes LDS writes of one phase visible to the
// next: a workgroup barrier between waves, a wavefront fence within one.
#pragma once
#include "fpv_dynamics.h"
#include "fpv_polydp.h"

namespace fpv {

typedef int pi32x4 __attribute__((ext_vector_type(4)));

constexpr int kPolyPoints = 12;                                 // contour points per lane: 95 for an 8-lane group, 191 for a 16-lane one

template <typename T, int TILES, int WAVES>
struct PixPool {
    static constexpr int kTileWords = 4 * kTile * 4 * (int)(sizeof(T) / 4) + kTile * 4;       // corner block + results
    static constexpr int kRes = 4 * kTile * 4 * (int)(sizeof(T) / 4);                         // (inside a tile's block)
    static constexpr int kRun = TILES * kTileWords;
    static constexpr int kCensus = kRun + 128;                                                // per tile 2 words of counts, then {octets, -}
    static constexpr int kSched = kCensus + 2 * TILES + 2;
    static constexpr int kSchedWords = TILES * kTile;                                         // 2 octets per drone at most, u16 each
    static constexpr int kOct = kSched + kSchedWords + 1;                                     // (+ a spare entry for lanes with nothing to enter)
    static constexpr int kTabWords = 34, kListWords = 4 * kPolyPoints;                        // per octet: 68 u16 of table, 8 kPolyPoints of list
    static constexpr int kOctWords = kTabWords + kListWords;
    static constexpr int kWords = (kOct + 8 * WAVES * kOctWords + 7) / 8 * 8;
    static_assert((kTileWords * 4) % 32 == 0 && 2 * kTabWords * 2 >= 132, "aligned tiles; a 16-lane group's table holds 128 rows + frame");
};

// Inclusive sum over the 64 lanes on the VALU (DPP row shifts, then the row broadcasts gfx9 keeps): lane 63 holds the total.
__device__ __forceinline__ int wave_scan_inclusive(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);    // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);    // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ void pixel_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A group of 8 lanes (half a DPP row) or 16 (a row).  All-reduce by a butterfly of lane permutations that stay inside the half
// row — neighbours, pairs of neighbours (quad_perm), the two quads (row_half_mirror) — plus, for 16 lanes, the two halves
// (row_mirror).  The inclusive scan shifts by 1, 2, 4 (and 8) inside the ROW (row_shr) and masks what would come from outside the
// group.  Every lane of the group must be active (the callers' control flow is uniform over the group).
struct DppGroup {
    int l;           // lane in the group
    bool wide;       // 16 lanes
    __device__ __forceinline__ int lane() const { return l; }
    __device__ __forceinline__ int size() const { return wide ? 16 : 8; }
    __device__ __forceinline__ uint32_t max_u32(uint32_t x) const {
        uint32_t y;
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false); x = x > y ? x : y;    // quad_perm:[1,0,3,2]
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false); x = x > y ? x : y;    // quad_perm:[2,3,0,1]
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false); x = x > y ? x : y;   // row_half_mirror
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false); x = (wide && y > x) ? y : x;     // row_mirror
        return x;
    }
    __device__ __forceinline__ int max_i32(int x) const {
        int y;
        y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); x = x > y ? x : y;
        y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false); x = x > y ? x : y;
        y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false); x = x > y ? x : y;
        y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false); x = (wide && y > x) ? y : x;
        return x;
    }
    __device__ __forceinline__ int add_i32(int x) const {
        x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);
        const int y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);
        return x + (wide ? y : 0);
    }
    __device__ __forceinline__ int scan_add(int x) const {
        int y;
        y = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); x += l >= 1 ? y : 0;                // row_shr:1
        y = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); x += l >= 2 ? y : 0;                // row_shr:2
        y = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); x += l >= 4 ? y : 0;                // row_shr:4
        y = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); x += l >= 8 ? y : 0;                // row_shr:8 (l >= 8: a wide group)
        return x;
    }
    __device__ __forceinline__ void fence() const { pixel_wave_fence(); }
};

// Owner lanes (one per drone): corners, slopes and row range of this lane's drone into its tile's corner block; a blob that
// touches the border, lies behind the camera, beside the image or between two rows of pixel centres gets the empty range (1, 0).
// The owners of tile 0 also write the pool's run-length table.  Returns the number of rows of the lane's blob (<= 0: none) and
// publishes the tile's census (pixel_tile_schedule below).
template <typename T, int TILES, int WAVES>
__device__ __forceinline__ int pixel_tile_setup(const Drone<T>& d, float* pool, int tile, int lane) {
    using P = PixPool<T, TILES, WAVES>;
    T u[4], v[4], sl[4], jlo, jhi;
    pixel_setup(d, u, v, sl, jlo, jhi);
    if (pixel_touches_border(u, v, sl, jlo, jhi)) { jlo = T(1); jhi = T(0); }                    // invisible: no rows to look at
    Vec4<T>* cb = reinterpret_cast<Vec4<T>*>(pool + tile * P::kTileWords);
    cb[lane] = Vec4<T>{u[0], u[1], u[2], u[3]};
    cb[kTile + lane] = Vec4<T>{v[0], v[1], v[2], v[3]};
    cb[2 * kTile + lane] = Vec4<T>{sl[0], sl[1], sl[2], sl[3]};
    cb[3 * kTile + lane] = Vec4<T>{jlo, jhi, T(0), T(0)};
    if (tile == 0) {
        uint32_t* run = reinterpret_cast<uint32_t*>(pool + P::kRun);
        run[lane] = poly_run_length_q23(lane);
        run[kTile + lane] = poly_run_length_q23(kTile + lane);
    }
    const int rows = (int)jhi - (int)jlo + 1;
    if (WAVES > 1) {
        const uint32_t c0 = (uint32_t)__popcll(__ballot(rows > 0)) | (uint32_t)__popcll(__ballot(rows > 16)) << 8 |
                            (uint32_t)__popcll(__ballot(rows > 20)) << 16 | (uint32_t)__popcll(__ballot(rows > 24)) << 24;
        const uint32_t c1 = (uint32_t)__popcll(__ballot(rows > 28)) | (uint32_t)__popcll(__ballot(rows > 36)) << 8;
        if (lane == 0) {
            uint32_t* cen = reinterpret_cast<uint32_t*>(pool + P::kCensus);
            cen[2 * tile] = c0;
            cen[2 * tile + 1] = c1;
        }
    }
    return rows;
}

// Owner lanes of a pool of several waves, once every tile's census is in LDS (a workgroup barrier behind pixel_tile_setup): the
// schedule entries of THIS tile's blobs.  Blobs of more than `big` rows get 16 lanes; `big` is the smallest of 16, 20, 24, 28 with
// which one round serves every blob (8 WAVES octets), 36 when none does.  Octets: the tall blobs first, two each, tile by tile in
// lane order; then the others.  Every owner wave derives the same bases from the same census; a lane with nothing to enter
// writes the spare entry (no branch around the stores).  Round 4 had every wave of the pool derive the whole schedule for itself
// behind barrier P — a microsecond of every wave's contour time for what 1/8 of it, on the owners, before the barrier, does.
template <typename T, int TILES, int WAVES>
__device__ __forceinline__ void pixel_tile_schedule(float* pool, int tile, int lane, int rows) {
    using P = PixPool<T, TILES, WAVES>;
    const uint32_t* cen = reinterpret_cast<const uint32_t*>(pool + P::kCensus);
    int ncand = 0, n16 = 0, n20 = 0, n24 = 0, n28 = 0;
    uint32_t c0[TILES], c1[TILES];
#pragma unroll
    for (int k = 0; k < TILES; ++k) {
        c0[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)cen[2 * k]);
        c1[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)cen[2 * k + 1]);
        ncand += (int)(c0[k] & 255u); n16 += (int)((c0[k] >> 8) & 255u); n20 += (int)((c0[k] >> 16) & 255u);
        n24 += (int)(c0[k] >> 24); n28 += (int)(c1[k] & 255u);
    }
    const int room = 8 * WAVES - ncand;                          // octets left for second halves
    const int big = n16 <= room ? 16 : (n20 <= room ? 20 : (n24 <= room ? 24 : (n28 <= room ? 28 : 36)));
    int nbig = 0, bb = 0, sb = 0;                                // all tall blobs; those / the others of the tiles before this one
#pragma unroll
    for (int k = 0; k < TILES; ++k) {
        const int cand = (int)(c0[k] & 255u);
        const int tall = (int)(big == 16 ? (c0[k] >> 8) & 255u : (big == 20 ? (c0[k] >> 16) & 255u : (big == 24 ? c0[k] >> 24 :
                                                                 (big == 28 ? c1[k] & 255u : (c1[k] >> 8) & 255u))));
        nbig += tall;
        bb += k < tile ? tall : 0;
        sb += k < tile ? cand - tall : 0;
    }
    const bool isb = rows > big, any = rows > 0;
    const unsigned long long mb = __ballot(isb), ms = __ballot(any && !isb), below = (1ull << lane) - 1ull;
    uint16_t* sched = reinterpret_cast<uint16_t*>(pool + P::kSched);
    constexpr int kSpare = 2 * P::kSchedWords;                    // (one u16 behind the 2 TILES kTile entries a pool can need)
    const uint32_t id = (uint32_t)(tile * kTile + lane);
    const int o = isb ? 2 * (bb + __popcll(mb & below)) : 2 * nbig + sb + __popcll(ms & below);
    sched[any ? o : kSpare] = (uint16_t)(id | (isb ? 1u << 9 : 0u));
    sched[isb ? o + 1 : kSpare] = (uint16_t)(id | 1u << 8 | 1u << 9);
    if (tile == 0 && lane == 0) reinterpret_cast<uint32_t*>(pool + P::kCensus)[2 * TILES] = (uint32_t)(ncand + nbig);   // octets
}

// Every lane of the pool's waves (wv = 0 .. WAVES - 1), once the corner blocks are complete: the blobs.
template <typename T, int TILES, int WAVES>
__device__ __forceinline__ void pixel_pool_contours(float* pool, int wv, int lane, uint64_t* probe = nullptr) {
    using P = PixPool<T, TILES, WAVES>;
    int octets;
    uint16_t* sched = reinterpret_cast<uint16_t*>(pool + P::kSched);
    if (WAVES > 1) {
        // the owners wrote the schedule before the barrier in front of this call (pixel_tile_schedule)
        octets = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(pool + P::kCensus)[2 * TILES]);
    } else {
        // ---- one wave, its own tile: the schedule derived on the spot (the wave's ballots ARE the census)
        static_assert(WAVES > 1 || TILES == 1, "a pool of one wave holds one tile");
        const Vec4<T> qr = reinterpret_cast<const Vec4<T>*>(pool)[3 * kTile + lane];
        const int rows = (int)qr.y - (int)qr.x + 1;
        const int ncand = __popcll(__ballot(rows > 0)), n16 = __popcll(__ballot(rows > 16)), n20 = __popcll(__ballot(rows > 20));
        const int n24 = __popcll(__ballot(rows > 24)), n28 = __popcll(__ballot(rows > 28));
        const int room = 8 * WAVES - ncand;                      // octets left for second halves
        const int big = n16 <= room ? 16 : (n20 <= room ? 20 : (n24 <= room ? 24 : (n28 <= room ? 28 : 36)));
        const bool isb = rows > big, any = rows > 0;
        const unsigned long long mb = __ballot(isb), ms = __ballot(any && !isb), below = (1ull << lane) - 1ull;
        constexpr int kSpare = 2 * P::kSchedWords;
        const int nbig = __popcll(mb);
        const int o = isb ? 2 * __popcll(mb & below) : 2 * nbig + __popcll(ms & below);
        sched[any ? o : kSpare] = (uint16_t)((uint32_t)lane | (isb ? 1u << 9 : 0u));
        sched[isb ? o + 1 : kSpare] = (uint16_t)((uint32_t)lane | 1u << 8 | 1u << 9);
        octets = ncand + nbig;
    }
    pixel_wave_fence();
    if (probe && lane == 0) *probe = __builtin_amdgcn_s_memrealtime();      // (diagnostic builds only: the schedule is done)
    const uint32_t* run = reinterpret_cast<const uint32_t*>(pool + P::kRun);
    for (int base = wv * 8; base < octets; base += 8 * WAVES) {                                   // (uniform over the wave)
        const int o = base + (lane >> 3);
        if (o < octets) {
            const uint32_t en = sched[o];
            const int id = (int)(en & 255u), half = (int)((en >> 8) & 1u);
            const DppGroup g{(lane & 7) + 8 * half, ((en >> 9) & 1u) != 0u};
            float* blk = pool + (id >> 6) * P::kTileWords;
            const int d = id & (kTile - 1);
            const Vec4<T>* cb = reinterpret_cast<const Vec4<T>*>(blk);
            // the group's scratch: that of its first octet (a 16-lane group's second octet lies right behind): table, then list
            float* oct = pool + P::kOct + (wv * 8 + (lane >> 3) - half) * P::kOctWords;
            uint16_t* tab = reinterpret_cast<uint16_t*>(oct) + 2;                                 // (tab[-1] and tab[rows] frame the rows)
            uint16_t* lst = reinterpret_cast<uint16_t*>(oct + (g.wide ? 2 : 1) * P::kTabWords);
            const Vec4<T> qu = cb[d], qv = cb[kTile + d], qs = cb[2 * kTile + d], qr = cb[3 * kTile + d];
            const T u[4] = {qu.x, qu.y, qu.z, qu.w}, v[4] = {qv.x, qv.y, qv.z, qv.w}, sl[4] = {qs.x, qs.y, qs.z, qs.w};
            const int jlo = (int)qr.x, nrows = (int)qr.y - jlo + 1;
            for (int k = g.lane(); k < nrows; k += g.size()) {
                T fa, fb;
                const bool have = row_span(u, v, sl, (T)(jlo + k) + T(0.5), fa, fb);
                tab[k] = have ? (uint16_t)(int)(fb * T(256) + fa) : (uint16_t)0x00FFu;
            }
            if (g.lane() == 0) { tab[-1] = (uint16_t)0x00FFu; tab[nrows] = (uint16_t)0x00FFu; }
            pixel_wave_fence();
            const PolyResult pr = poly_blob<kPolyPoints>(g, tab, lst, run, jlo, nrows);
            if (g.lane() == 0)
                reinterpret_cast<pi32x4*>(blk + P::kRes)[d] = pi32x4{pr.vis | pr.w << 8 | pr.h << 16, pr.sx | pr.sy << 16, pr.area2, 0};
        }
        pixel_wave_fence();                                    // (the next round reuses tables and lists)
    }
}

// Owner lanes, after the contours: obs[7:9, 11, 13, 14] of this lane's drone (hover.py:197-213, :245-248).  Returns visible.
template <typename T, int TILES, int WAVES>
__device__ __forceinline__ bool pixel_tile_features(const float* pool, int tile, int lane, T& cx, T& cy, T& area, T& ratio) {
    using P = PixPool<T, TILES, WAVES>;
    const float* blk = pool + tile * P::kTileWords;
    const Vec4<T> mine = reinterpret_cast<const Vec4<T>*>(blk)[3 * kTile + lane];
    const pi32x4 r = reinterpret_cast<const pi32x4*>(blk + P::kRes)[lane];
    const bool vis = mine.x <= mine.y && (r.x & 1) != 0;
    cx = cy = area = ratio = T(0);
    if (vis) {
        cx = (T)(r.y & 0xFFFF) * T(1.0 / 256.0) - T(1);        // hover.py:197-203: mean of four integers / 64 - 1, exact (also in the
        cy = (T)(r.y >> 16) * T(1.0 / 256.0) - T(1);           // float32 the reference stores it in)
        area = (T)r.z * T(1.0 / 32768.0);                      // :206
        ratio = div_((T)((r.x >> 8) & 255), (T)((r.x >> 16) & 255));                  // :209-213
    }
    return vis;
}

// One wave owns its tile and works its blobs itself (the stand-alone env-step kernels): LDS operations of a wave execute in
// program order, so between the phases only the compiler has to be kept from reordering them.
template <typename T>
using OneWavePixT = PixPool<T, 1, 1>;
template <typename T>
__device__ __forceinline__ bool camera_features_pixel_tile(const Drone<T>& d, float* pix, int lane, T& cx, T& cy, T& area, T& ratio) {
    (void)pixel_tile_setup<T, 1, 1>(d, pix, 0, lane);
    pixel_wave_fence();
    pixel_pool_contours<T, 1, 1>(pix, 0, lane);
    pixel_wave_fence();
    return pixel_tile_features<T, 1, 1>(pix, 0, lane, cx, cy, area, ratio);
}

}  // namespace fpv
