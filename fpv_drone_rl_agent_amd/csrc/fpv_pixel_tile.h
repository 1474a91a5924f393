// fpv_pixel_tile.h — the pixel detector's rows (hover.py:157-222, SPEC §6b) as work items of one 64-drone tile.
//
// A drone's blob is 0 .. ~45 image rows and a third of the drones see the panel at all: one lane walking its own drone's rows
// keeps its wave busy for the tallest blob of the tile while most lanes idle (measured on bench.py's states: 35 rows per tile
// against 7.8 per drone), and a wave issues one vector instruction per ~5 cycles whatever its ILP — the walk is bound by
// instructions per lane.  So the lanes that own the drones project the corners, drop the blobs that touch the image border
// (pixel_touches_border: exact), cut the others' rows into chunks and post (drone, chunk) items through LDS; WORKERS lanes
// (the tile's wave, or its two waves in fpv_rollstep.hip) take one item each, add its PixelRun to the drone's integer
// accumulators with LDS atomics and leave the rows' spans in a table; the owner lanes read accumulators and table and apply
// the gates (pixel_finish).  Every pair of consecutive rows is counted by exactly one item (an item that is not its drone's
// first evaluates the row above itself), so the accumulators need sums, minima and maxima only: exact integers, order-free —
// the same features, bit for bit, as one lane walking all rows (camera_features_pixel).
//
// Per tile, `pix` points at PixTile<WORKERS, CAP>::kFloats floats of LDS (16-byte aligned):
//   corner block [4][64] f32x4: u, v, edge slopes, {first span-table entry, jlo, jhi, rows per item} of every drone;
//   accumulators [3][64] int4: {n, W, S, H}, {viol, amin, bmax, j0}, {jn};
//   item list: WORKERS packed items (drone | chunk << 8), then the item count;
//   span table: CAP u16 entries a | b << 8 per row, the tile's drones one after the other (rows beyond CAP are evaluated again
//   by the vertex search: CAP = 64 x 128 never overflows).
// The caller orders the phases (post -> work -> features) with whatever makes LDS writes of one phase visible to the next:
// a workgroup barrier between two waves, a wavefront fence within one.
#pragma once
#include "fpv_dynamics.h"

namespace fpv {

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef int pi32x4 __attribute__((ext_vector_type(4)));

template <int WORKERS, int CAP>
struct PixTile {
    static constexpr int kAcc = 4 * kTile * 4;                  // floats from the tile's base
    static constexpr int kSlots = kAcc + 3 * kTile * 4;
    static constexpr int kTab = kSlots + WORKERS + 4;
    static constexpr int kFloats = kTab + CAP / 2;
    static_assert((kAcc * 4) % 16 == 0 && (kFloats * 4) % 16 == 0 && CAP % 8 == 0, "16-byte areas must be 16-byte aligned");
};
constexpr int kPixMinChunk = 2;                                 // rows per item at least

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

// Owner lanes (one per drone): corners, slopes and row range of this lane's drone; blobs that touch the border get no rows.
// Writes the corner block's first three vectors and clears the accumulators.  Returns ok (all corners in front of the camera,
// some row of pixel centres inside, border untouched so far).
template <int WORKERS, int CAP>
__device__ __forceinline__ bool pixel_tile_setup(const Drone<float>& d, float* pix, int lane, float (&u)[4], float (&v)[4], float (&sl)[4],
                                                 float& jlo, float& jhi) {
    bool ok = pixel_setup(d, u, v, sl, jlo, jhi);
    if (pixel_touches_border(u, v, sl, jlo, jhi)) { ok = false; jlo = 1.f; jhi = 0.f; }        // invisible: no rows to walk
    pf32x4* cb = reinterpret_cast<pf32x4*>(pix);
    cb[lane] = pf32x4{u[0], u[1], u[2], u[3]};
    cb[kTile + lane] = pf32x4{v[0], v[1], v[2], v[3]};
    cb[2 * kTile + lane] = pf32x4{sl[0], sl[1], sl[2], sl[3]};
    pi32x4* acc = reinterpret_cast<pi32x4*>(pix + PixTile<WORKERS, CAP>::kAcc);
    acc[lane] = pi32x4{0, 0, 0, 0};                        // n, W, S, H
    acc[kTile + lane] = pi32x4{-1000, 128, -1, 128};       // viol, amin, bmax, j0
    acc[2 * kTile + lane] = pi32x4{-1, 0, 0, 0};           // jn
    return ok;
}

// Owner lanes: cut the drone's rows [jlo, jhi] into chunks of ch rows — the fewest rows per item that let the WORKERS lanes take
// all items in one round (items <= rows / ch + one per drone with rows) — and append the items (drone | chunk << 8) at this
// lane's place in the list.  Leaves {first entry of the drone in the span table, jlo, jhi, ch} in the corner block.  A lane
// without rows (jlo > jhi) appends nothing.
template <int WORKERS, int CAP>
__device__ __forceinline__ void pixel_post_items(float* pix, int lane, int ilo, int ihi) {
    using P = PixTile<WORKERS, CAP>;
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
    reinterpret_cast<pf32x4*>(pix)[3 * kTile + lane] = pf32x4{(float)(rows_incl - rows), (float)ilo, (float)ihi, (float)ch};
}

// Worker lanes (worker = 0 .. WORKERS - 1): walk the rows of this lane's item and add its run to the drone's accumulators.
template <int WORKERS, int CAP>
__device__ __forceinline__ void pixel_work_items(float* pix, int worker) {
    using P = PixTile<WORKERS, CAP>;
    const uint32_t* slots = reinterpret_cast<const uint32_t*>(pix + P::kSlots);
    const pf32x4* cb = reinterpret_cast<const pf32x4*>(pix);
    int* acc = reinterpret_cast<int*>(pix + P::kAcc);
    uint16_t* tab = reinterpret_cast<uint16_t*>(pix + P::kTab);
    const int total = (int)slots[WORKERS];
    for (int item = worker; item < total; item += WORKERS) {   // (one round: the chunk size is chosen for it)
        const uint32_t rec = slots[item];
        const int d = (int)(rec & 63u), k = (int)(rec >> 8);
        const pf32x4 qu = cb[d], qv = cb[kTile + d], qs = cb[2 * kTile + d], qr = cb[3 * kTile + d];
        const float u[4] = {qu.x, qu.y, qu.z, qu.w}, v[4] = {qv.x, qv.y, qv.z, qv.w}, sl[4] = {qs.x, qs.y, qs.z, qs.w};
        const float jf = qr.y + (float)k * qr.w, jl = fminf(jf + qr.w - 1.0f, qr.z);
        const int idx = (int)qr.x + (int)(jf - qr.y);            // this item's first entry in the span table
        float pa = 0.f, pb = 0.f;
        const bool linked = k > 0 && row_span(u, v, sl, jf - 0.5f, pa, pb);        // the row above, if the blob has it
        PixelRun<float> r;
        pixel_walk(u, v, sl, jf, jl, r, tab, idx, CAP, linked, pa, pb);
        if (r.n > 0.f) {
            int* a = acc + 4 * d;
            atomicAdd(a, (int)r.n); atomicAdd(a + 1, (int)r.W); atomicAdd(a + 2, (int)r.S); atomicAdd(a + 3, (int)r.H);
            a += 4 * kTile;
            atomicMax(a, (int)r.viol); atomicMin(a + 1, (int)r.amin); atomicMax(a + 2, (int)r.bmax); atomicMin(a + 3, (int)r.j0);
            atomicMax(a + 4 * kTile, (int)r.jn);
        }
    }
}

// Owner lanes, after every item of the tile is done: the blob's run from the accumulators and the span table, then the gates and
// features of SPEC §6b items 3-7.  Returns visible.
template <int WORKERS, int CAP>
__device__ __forceinline__ bool pixel_tile_features(const float* pix, int lane, const float (&u)[4], const float (&v)[4], const float (&sl)[4],
                                                    float jlo, bool ok, float& cx, float& cy, float& area, float& ratio) {
    using P = PixTile<WORKERS, CAP>;
    const pi32x4* acc = reinterpret_cast<const pi32x4*>(pix + P::kAcc);
    const pi32x4 a0 = acc[lane], a1 = acc[kTile + lane], a2 = acc[2 * kTile + lane];
    const int rowbase = (int)reinterpret_cast<const pf32x4*>(pix)[3 * kTile + lane].x, ilo = (int)jlo;
    const uint16_t* tab = reinterpret_cast<const uint16_t*>(pix + P::kTab);
    auto entry = [&](int row, float& a, float& b) {        // (rows outside the blob may be asked for — result unused)
        const int idx = rowbase + row - ilo;
        if (CAP >= kTile * 128 || idx < CAP) {
            const uint32_t ab = tab[min(max(idx, 0), CAP - 1)];
            a = (float)(ab & 255u); b = (float)(ab >> 8);
        } else {                                           // a tile with more blob rows than the table holds: evaluated again
            row_span(u, v, sl, (float)row + 0.5f, a, b);
        }
    };
    PixelRun<float> r;
    r.n = (float)a0.x; r.W = (float)a0.y; r.S = (float)a0.z; r.H = (float)a0.w;
    r.viol = (float)a1.x; r.amin = (float)a1.y; r.bmax = (float)a1.z; r.j0 = (float)a1.w; r.jn = (float)a2.x;
    entry(a1.w, r.a0, r.b0);                               // first and last row of the blob (unused when the blob is empty)
    entry(a2.x, r.pa, r.pb);
    const bool seen = pixel_finish(r, u, v, [&](float jc, float (&a)[3], float (&b)[3]) {
        const int j = (int)jc;
        entry(j - 1, a[0], b[0]); entry(j, a[1], b[1]); entry(j + 1, a[2], b[2]);
    }, cx, cy, area, ratio);
    return ok && seen;
}

// One wave owns the tile and works its items itself (the stand-alone env-step kernels): LDS operations of a wave execute in
// program order, so between the phases only the compiler has to be kept from reordering them.
__device__ __forceinline__ void pixel_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr int kPixCapOneWave = 2048;                           // 4 KB of span table per tile: four tiles still share a workgroup
__device__ __forceinline__ bool camera_features_pixel_tile(const Drone<float>& d, float* pix, int lane, float& cx, float& cy, float& area,
                                                           float& ratio) {
    float u[4], v[4], sl[4], jlo, jhi;
    const bool ok = pixel_tile_setup<kTile, kPixCapOneWave>(d, pix, lane, u, v, sl, jlo, jhi);
    pixel_post_items<kTile, kPixCapOneWave>(pix, lane, (int)jlo, (int)jhi);
    pixel_wave_fence();
    pixel_work_items<kTile, kPixCapOneWave>(pix, lane);
    pixel_wave_fence();
    return pixel_tile_features<kTile, kPixCapOneWave>(pix, lane, u, v, sl, jlo, ok, cx, cy, area, ratio);
}

}  // namespace fpv
