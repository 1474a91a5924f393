// fpv_pixel_tile.h — the pixel detector (hover.py:157-222, SPEC §6b) of one 64-drone tile, worked by the tile's lanes together.
//
// A drone's blob is 0 .. ~45 image rows and a third of the drones see the panel at all: one lane walking its own drone's rows
// keeps its wave busy for the tallest blob of the tile while most lanes idle (measured on bench.py's states: 35 rows per tile
// against 7.8 per drone), and a wave issues one vector instruction per ~5 cycles whatever its ILP — the detector is bound by
// instructions per lane.  So the work is cut twice:
//
//  ROWS -> ITEMS.  The lanes that own the drones project the corners, drop the blobs that touch the image border
//  (pixel_touches_border: exact), cut the others' rows into chunks and post (drone, chunk) items through LDS; WORKERS lanes (the
//  tile's wave, or its two waves in fpv_rollstep.hip) take one item each, add its PixelRun to the drone's integer accumulators
//  with LDS atomics and leave every row's span (or "empty") in a table.  Every pair of consecutive rows is counted by exactly
//  one item (an item that is not its drone's first evaluates the row above itself), so the accumulators need sums, minima and
//  maxima only: exact integers, order-free.
//
//  CONTOURS -> LANE GROUPS (round 4).  What the reference then does with the mask — cv2.findContours, the largest contour,
//  cv2.arcLength, cv2.approxPolyDP, "four vertices or invisible" (hover.py:187-203) — is worked per surviving blob by a group of
//  16 lanes (one DPP row): fpv_polydp.h runs approx.cpp's Douglas-Peucker over six slot points per row taken from the span table,
//  its farthest-point searches spread over the group's lanes and reduced with four DPP steps.  Each wave lists the tile's
//  candidates (blobs with rows that do not touch the border) itself from the accumulators, group G of the tile takes candidates
//  G, G + groups, ...; the result (visible, vertex sums, 2 x contourArea, boundingRect) goes to the drone's result slot, which
//  its owner lane turns into obs[7:9, 11, 13, 14].  Integers and a few doubles only: the same decisions for fp32 and fp64
//  handles, and the CPU oracle's bit for bit given the same spans.
//
// Per tile, `pix` points at PixTile<T, WORKERS, CAP>::kWords 4-byte words of LDS (32-byte aligned):
//   corner block [4][64] Vec4<T>: u, v, edge slopes, {first span-table entry, jlo, jhi, rows per item} of every drone;
//   accumulators [3][64] int4: {n, W, S, H}, {viol, amin, bmax, j0}, {jn};
//   results [64] int4: {visible | w << 8 | h << 16, sum x | sum y << 16, 2 x contourArea, -};
//   item list: WORKERS packed items (drone | chunk << 8), then the item count;
//   candidate lists: 64 bytes per wave of the tile;
//   span table: CAP u16 entries a | b << 8 per row (0x00FF: no red pixel), the tile's drones one after the other (rows beyond
//   CAP are evaluated again where they are needed: CAP = 64 x 128 never overflows).
// The caller orders the phases (post -> work -> contours -> features) with whatever makes LDS writes of one phase visible to the
// next: a workgroup barrier between two waves, a wavefront fence within one.
#pragma once
#include "fpv_dynamics.h"
#include "fpv_polydp.h"

namespace fpv {

typedef int pi32x4 __attribute__((ext_vector_type(4)));

template <typename T, int WORKERS, int CAP>
struct PixTile {
    static constexpr int kAcc = 4 * kTile * 4 * (int)(sizeof(T) / 4);      // words from the tile's base
    static constexpr int kRes = kAcc + 3 * kTile * 4;
    static constexpr int kSlots = kRes + kTile * 4;
    static constexpr int kCand = kSlots + WORKERS + 8;      // (items, the item count, padding to 32 bytes)
    static constexpr int kTab = kCand + (WORKERS / kTile) * (kTile / 4);
    static constexpr int kWords = kTab + CAP / 2;
    static_assert((kAcc * 4) % 32 == 0 && (kTab * 4) % 16 == 0 && (kWords * 4) % 32 == 0 && CAP % 16 == 0 && WORKERS % kTile == 0,
                  "aligned areas");
};
constexpr int kPixMinChunk = 2;                                 // rows per item at least
constexpr uint16_t kPixNoRow = 0x00FFu;                         // span-table entry of a row without red pixels (a = 255 > b = 0)

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

// A group of 16 lanes = one DPP row.  All-reduce by a butterfly of lane permutations inside the row: neighbours, pairs of
// neighbours (quad_perm), the two quads of a half row (row_half_mirror), the two half rows (row_mirror).  Every lane of the
// row must be active (the callers' control flow is uniform over the group).
struct DppGroup16 {
    int l;
    __device__ __forceinline__ int lane() const { return l; }
    __device__ __forceinline__ static constexpr int size() { return 16; }
    __device__ __forceinline__ uint32_t max_u32(uint32_t x) const {
        uint32_t y;
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false); x = x > y ? x : y;    // quad_perm:[1,0,3,2]
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false); x = x > y ? x : y;    // quad_perm:[2,3,0,1]
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false); x = x > y ? x : y;   // row_half_mirror
        y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false); x = x > y ? x : y;   // row_mirror
        return x;
    }
    __device__ __forceinline__ int add_i32(int x) const {
        x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);
        return x;
    }
};

// Owner lanes (one per drone): corners, slopes and row range of this lane's drone; blobs that touch the border get no rows.
// Writes the corner block's first three vectors and clears the accumulators.  Returns ok (all corners in front of the camera,
// some row of pixel centres inside, border untouched so far).
template <typename T, int WORKERS, int CAP>
__device__ __forceinline__ bool pixel_tile_setup(const Drone<T>& d, float* pix, int lane, T& jlo, T& jhi) {
    T u[4], v[4], sl[4];
    bool ok = pixel_setup(d, u, v, sl, jlo, jhi);
    if (pixel_touches_border(u, v, sl, jlo, jhi)) { ok = false; jlo = T(1); jhi = T(0); }        // invisible: no rows to walk
    Vec4<T>* cb = reinterpret_cast<Vec4<T>*>(pix);
    cb[lane] = Vec4<T>{u[0], u[1], u[2], u[3]};
    cb[kTile + lane] = Vec4<T>{v[0], v[1], v[2], v[3]};
    cb[2 * kTile + lane] = Vec4<T>{sl[0], sl[1], sl[2], sl[3]};
    pi32x4* acc = reinterpret_cast<pi32x4*>(pix + PixTile<T, WORKERS, CAP>::kAcc);
    acc[lane] = pi32x4{0, 0, 0, 0};                        // n, W, S, H
    acc[kTile + lane] = pi32x4{-1000, 128, -1, 128};       // viol, amin, bmax, j0
    acc[2 * kTile + lane] = pi32x4{-1, 0, 0, 0};           // jn
    return ok;
}

// Owner lanes: cut the drone's rows [jlo, jhi] into chunks of ch rows — the fewest rows per item that let the WORKERS lanes take
// all items in one round (items <= rows / ch + one per drone with rows) — and append the items (drone | chunk << 8) at this
// lane's place in the list.  Leaves {first entry of the drone in the span table, jlo, jhi, ch} in the corner block.  A lane
// without rows (jlo > jhi) appends nothing.
template <typename T, int WORKERS, int CAP>
__device__ __forceinline__ void pixel_post_items(float* pix, int lane, int ilo, int ihi) {
    using P = PixTile<T, WORKERS, CAP>;
    const int rows = max(ihi - ilo + 1, 0);
    const int rows_incl = wave_scan_inclusive(rows);
    const int tile_rows = __builtin_amdgcn_readlane(rows_incl, 63);
    const int with_rows = __popcll(__ballot(rows > 0));
    const int room = max(WORKERS - with_rows, 1);              // (WORKERS = 64 and every drone with rows: one item per drone)
    const int ch = max(with_rows >= WORKERS ? 128 : kPixMinChunk, (tile_rows + room - 1) / room);      // uniform
    // ceil(rows / ch) for 0 <= rows <= 128 without an integer division: floor((rows + ch - 1/2) / ch), the half keeps the
    // quotient away from the integers by 1 / (2 ch), far more than the rounding of the reciprocal
    const int items = (int)(((float)(rows + ch) - 0.5f) * (1.0f / (float)ch));
    const int incl = wave_scan_inclusive(items);
    uint32_t* slots = reinterpret_cast<uint32_t*>(pix + P::kSlots);
    for (int k = 0, pos = incl - items; k < items; ++k, ++pos) slots[pos] = (uint32_t)lane | (uint32_t)k << 8;
    if (lane == kTile - 1) slots[WORKERS] = (uint32_t)incl;
    reinterpret_cast<Vec4<T>*>(pix)[3 * kTile + lane] = Vec4<T>{(T)(rows_incl - rows), (T)ilo, (T)ihi, (T)ch};
}

// Worker lanes (worker = 0 .. WORKERS - 1): walk the rows of this lane's item and add its run to the drone's accumulators.
template <typename T, int WORKERS, int CAP>
__device__ __forceinline__ void pixel_work_items(float* pix, int worker) {
    using P = PixTile<T, WORKERS, CAP>;
    const uint32_t* slots = reinterpret_cast<const uint32_t*>(pix + P::kSlots);
    const Vec4<T>* cb = reinterpret_cast<const Vec4<T>*>(pix);
    int* acc = reinterpret_cast<int*>(pix + P::kAcc);
    uint16_t* tab = reinterpret_cast<uint16_t*>(pix + P::kTab);
    const int total = (int)slots[WORKERS];
    for (int item = worker; item < total; item += WORKERS) {   // (one round: the chunk size is chosen for it)
        const uint32_t rec = slots[item];
        const int d = (int)(rec & 63u), k = (int)(rec >> 8);
        const Vec4<T> qu = cb[d], qv = cb[kTile + d], qs = cb[2 * kTile + d], qr = cb[3 * kTile + d];
        const T u[4] = {qu.x, qu.y, qu.z, qu.w}, v[4] = {qv.x, qv.y, qv.z, qv.w}, sl[4] = {qs.x, qs.y, qs.z, qs.w};
        const T jf = qr.y + (T)k * qr.w, jl = min_(jf + qr.w - T(1), qr.z);
        const int idx = (int)qr.x + (int)(jf - qr.y);            // this item's first entry in the span table
        T pa = T(0), pb = T(0);
        const bool linked = k > 0 && row_span(u, v, sl, jf - T(0.5), pa, pb);        // the row above, if the blob has it
        PixelRun<T> r;
        pixel_walk(u, v, sl, jf, jl, r, tab, idx, CAP, linked, pa, pb);
        if (r.n > T(0)) {
            int* a = acc + 4 * d;
            atomicAdd(a, (int)r.n); atomicAdd(a + 1, (int)r.W); atomicAdd(a + 2, (int)r.S); atomicAdd(a + 3, (int)r.H);
            a += 4 * kTile;
            atomicMax(a, (int)r.viol); atomicMin(a + 1, (int)r.amin); atomicMax(a + 2, (int)r.bmax); atomicMin(a + 3, (int)r.j0);
            atomicMax(a + 4 * kTile, (int)r.jn);
        }
    }
}

// a drone's rows in the tile's span table (fpv_polydp.h's Span policy)
template <typename T, int CAP>
struct TileSpan {
    const uint16_t* tab;       // the drone's first entry minus its first row
    int cap_row;               // rows >= cap_row lie beyond the table
    T u[4], v[4], sl[4];       // (CAP < 64 x 128 only) the drone's corners, for the rows beyond the table
    __device__ __forceinline__ bool operator()(int row, int& a, int& b) const {
        if (CAP >= kTile * 128 || row < cap_row) {
            const uint32_t ab = tab[row];
            a = (int)(ab & 255u); b = (int)(ab >> 8);
            return a <= b;
        }
        T fa, fb;
        const bool have = row_span(u, v, sl, (T)row + T(0.5), fa, fb);
        a = have ? (int)fa : 255; b = have ? (int)fb : 0;
        return have;
    }
};

// "this drone's blob has rows and does not touch the image border" (hover.py:180-185) from its accumulators
__device__ __forceinline__ bool pixel_candidate(const pi32x4& a0, const pi32x4& a1, const pi32x4& a2) {
    return a0.x > 0 && !(a1.y == 0 || a1.z == 127 || a1.w == 0 || a2.x == 127);
}

// Every worker lane, after every item of the tile is done: the contours (see the head of this file).
template <typename T, int WORKERS, int CAP>
__device__ __forceinline__ void pixel_tile_contours(float* pix, int worker) {
    using P = PixTile<T, WORKERS, CAP>;
    constexpr int kGroups = WORKERS / 16;
    const int lane = worker & (kTile - 1), wv = worker / kTile;
    const pi32x4* acc = reinterpret_cast<const pi32x4*>(pix + P::kAcc);
    const Vec4<T>* cb = reinterpret_cast<const Vec4<T>*>(pix);
    // this wave's list of the tile's candidates, in drone order (both waves of a tile derive the same list)
    const bool cand = pixel_candidate(acc[lane], acc[kTile + lane], acc[2 * kTile + lane]);
    const unsigned long long m = __ballot(cand);
    const int ncand = __popcll(m);
    uint8_t* list = reinterpret_cast<uint8_t*>(pix + P::kCand) + kTile * wv;
    if (cand) list[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const DppGroup16 g{lane & 15};
    pi32x4* res = reinterpret_cast<pi32x4*>(pix + P::kRes);
    for (int r = worker >> 4; r < ncand; r += kGroups) {
        const int d = (int)list[r];
        const pi32x4 a0 = acc[d], a1 = acc[kTile + d], a2 = acc[2 * kTile + d];
        const Vec4<T> qr = cb[3 * kTile + d];
        TileSpan<T, CAP> span;
        const int rowbase = (int)qr.x, ilo = (int)qr.y;
        span.tab = reinterpret_cast<const uint16_t*>(pix + P::kTab) + (rowbase - ilo);
        span.cap_row = CAP - rowbase + ilo;
        if (CAP < kTile * 128) {
            const Vec4<T> qu = cb[d], qv = cb[kTile + d], qs = cb[2 * kTile + d];
            span.u[0] = qu.x; span.u[1] = qu.y; span.u[2] = qu.z; span.u[3] = qu.w;
            span.v[0] = qv.x; span.v[1] = qv.y; span.v[2] = qv.z; span.v[3] = qv.w;
            span.sl[0] = qs.x; span.sl[1] = qs.y; span.sl[2] = qs.z; span.sl[3] = qs.w;
        }
        const int j0 = a1.w, jn = a2.x;
        PolyBlob bl = {j0, jn, a0.y, a0.z, a0.w, a1.y, a1.z};
        if (jn - j0 + 1 != a0.x || a1.x > 1) bl = poly_largest_component(span, j0, jn);      // several pieces: cv2 takes the largest
        const PolyResult pr = poly_detect(g, span, bl);
        if (g.lane() == 0) res[d] = pi32x4{pr.vis | pr.w << 8 | pr.h << 16, pr.sx | pr.sy << 16, pr.area2, 0};
    }
}

// Owner lanes, after the contours: obs[7:9, 11, 13, 14] of this lane's drone (hover.py:197-213, :245-248).  Returns visible.
template <typename T, int WORKERS, int CAP>
__device__ __forceinline__ bool pixel_tile_features(const float* pix, int lane, T& cx, T& cy, T& area, T& ratio) {
    using P = PixTile<T, WORKERS, CAP>;
    const pi32x4* acc = reinterpret_cast<const pi32x4*>(pix + P::kAcc);
    const bool cand = pixel_candidate(acc[lane], acc[kTile + lane], acc[2 * kTile + lane]);
    const pi32x4 r = reinterpret_cast<const pi32x4*>(pix + P::kRes)[lane];
    const bool vis = cand && (r.x & 1) != 0;
    cx = cy = area = ratio = T(0);
    if (vis) {
        cx = (T)(r.y & 0xFFFF) * T(1.0 / 256.0) - T(1);        // hover.py:197-203: mean of four integers / 64 - 1, exact (also in the
        cy = (T)(r.y >> 16) * T(1.0 / 256.0) - T(1);           // float32 the reference stores it in)
        area = (T)r.z * T(1.0 / 32768.0);                      // :206
        ratio = div_((T)((r.x >> 8) & 255), (T)((r.x >> 16) & 255));                  // :209-213
    }
    return vis;
}

// One wave owns the tile and works its items itself (the stand-alone env-step kernels): LDS operations of a wave execute in
// program order, so between the phases only the compiler has to be kept from reordering them.
__device__ __forceinline__ void pixel_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr int kPixCapOneWave = 2048;                           // 4 KB of span table per tile: four tiles still share a workgroup
template <typename T>
using OneWavePixT = PixTile<T, kTile, kPixCapOneWave>;
template <typename T>
__device__ __forceinline__ bool camera_features_pixel_tile(const Drone<T>& d, float* pix, int lane, T& cx, T& cy, T& area, T& ratio) {
    T jlo, jhi;
    pixel_tile_setup<T, kTile, kPixCapOneWave>(d, pix, lane, jlo, jhi);
    pixel_post_items<T, kTile, kPixCapOneWave>(pix, lane, (int)jlo, (int)jhi);
    pixel_wave_fence();
    pixel_work_items<T, kTile, kPixCapOneWave>(pix, lane);
    pixel_wave_fence();
    pixel_tile_contours<T, kTile, kPixCapOneWave>(pix, lane);
    pixel_wave_fence();
    return pixel_tile_features<T, kTile, kPixCapOneWave>(pix, lane, cx, cy, area, ratio);
}

}  // namespace fpv
