// fpv_ppo.hip — gradient of the PPO minibatch loss (SB3 PPO.train, the learner north_star puts in place of the SAC of
// simulation/train_hover.py:47-60) for the MlpPolicy of fpv_policy.hip, as hand-written gfx950 kernels:
//
//   U0 ppo_pack_kernel     live nn.Linear tensors -> MFMA-ordered training image (forward operands + W2^T, Wa^T)
//                          and, for the one-kernel path, W2 cut into bf16 pieces as MFMA A fragments (both directions)
//   U1 ppo_fwd_bwd_kernel  per row: pi and V forward, clipped-surrogate / value-loss derivatives, backward through the
//                          heads and layer 2 — all in registers, transposed like the rollout kernel (drones on the MFMA
//                          column, an accumulator tile is the next product's B operand as it stands, in both directions).
//                          Emits H1, H2, D1 = dL/dZ1, D2 = dL/dZ2 per net (row-major [rows][128]) and d mu, d v.
//   U2 ppo_wgrad_kernel    dW = D^T H as MFMA products whose K index is the row: both operands are read from the
//                          row-major matrices as they lie (32 consecutive columns per half-wave, no transposes);
//                          every workgroup reduces a slab of rows, slabs are summed in fixed order by
//   U3 ppo_reduce_kernel   -> the 13 gradient tensors + {pg loss, value loss, approx KL, clip fraction}.
// fp32 throughout (v_mfma_f32_32x32x2_f32); no atomics: the result is bitwise reproducible for a given launch shape.
#include "../../include/fpv_env.h"   // FPV_PPO_* path constants
#include "fpv_mlp.h"

#ifndef FPV_PPO_THREADS
#define FPV_PPO_THREADS 256
#endif

namespace fpv {

namespace {

constexpr int kHid = 128;
// per-net training image (floats): W1p | b1 | W2p | b2 | W2Tp
constexpr int kTOffW1 = 0, kTOffB1 = 3072, kTOffW2 = 3200, kTOffB2 = 19584, kTOffW2T = 19712, kTNet = 36096;
// head block: Wa[h][kt][r][4] | Wv[h][kt][r] | ba[4] | bv (pad 4) | log_std[4] | inv_std[4] | logp const (pad 4) | WaT[m][s][lane]
constexpr int kTHead = 2 * kTNet;
constexpr int kHWa = 0, kHWv = 512, kHBa = 640, kHBv = 644, kHLogStd = 648, kHInvStd = 652, kHLpc = 656, kHWaT = 660;
constexpr int kHeadFloats = 660 + 512;                  // 1172
constexpr int kTImage = kTHead + kHeadFloats;
// U2 partial layout per (net, slab)
constexpr int kPW1 = 0, kPB1 = 3072, kPW2 = 3200, kPB2 = 19584, kPWh = 19712, kPBh = 20224, kPartStride = 20232;
constexpr int kFwdBwdThreads = FPV_PPO_THREADS;          // U1 workgroup: 256 = one wave per SIMD (512 measured slower: 175 vs 133 us)
constexpr int kWaveStats = 8;                           // pg, vl, kl, clip fraction, d log_std[4]
// W2 cut into bf16 pieces once per update for the one-kernel path (U0 writes it behind the training image's users): per net the
// A operands of layer 2 (dir 0) and of its transpose on the way back (dir 1), [dir][k-step 8][tile 4][piece 3][lane 64] x 16 bytes
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPieceStep = 4 * 3 * 64;                  // 16-byte words per k-step: four tiles x three pieces x 64 lanes (12 KB)
constexpr int kPieceNet = 2 * 8 * kPieceStep;           // per net (192 KB)
constexpr int kPieceFloats = 2 * kPieceNet * 4;         // both nets, in floats of workspace

// Storage type of the activation / delta matrices that U1 hands to U2 through memory: float (default, exact) or bf16
// (`fpv_ppo_io.storage = 1`).  The matrices are 4 KB per row in fp32 — U1 writes and U2 reads 268 MB per 65 536-row
// minibatch, which costs both kernels about as much time as their matrix products — and half that in bf16; the products
// stay fp32 MFMA on the rounded values (relative rounding 2^-9 per stored value, zero-mean over the rows of a sum).
typedef unsigned short bf16_bits;
__device__ __forceinline__ float from_store(float v) { return v; }
__device__ __forceinline__ float from_store(bf16_bits v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ bf16_bits to_bf16(float v) {     // round to nearest even
    const uint32_t u = __float_as_uint(v);
    return (bf16_bits)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// lane (col, h) holds rows tile_row(r, h) of each 32-row tile: four consecutive rows per register quad -> one 16-byte
// (fp32) or 8-byte (bf16) store
__device__ __forceinline__ void store_tile(float* __restrict__ mat, int64_t n, const f32x16& t, int m, int h) {
    f32x4* row = reinterpret_cast<f32x4*>(mat + n * kHid);
#pragma unroll
    for (int q = 0; q < 4; ++q) row[m * 8 + 2 * q + h] = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
__device__ __forceinline__ void store_tile(bf16_bits* __restrict__ mat, int64_t n, const f32x16& t, int m, int h) {
    uint2* row = reinterpret_cast<uint2*>(mat + n * kHid);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        row[m * 8 + 2 * q + h] = uint2{(uint32_t)to_bf16(t[4 * q]) | ((uint32_t)to_bf16(t[4 * q + 1]) << 16),
                                       (uint32_t)to_bf16(t[4 * q + 2]) | ((uint32_t)to_bf16(t[4 * q + 3]) << 16)};
}
template <typename ST>
__device__ __forceinline__ void store_rows(ST* __restrict__ mat, int64_t n, const Hidden& t, int h) {
#pragma unroll
    for (int m = 0; m < 4; ++m) store_tile(mat, n, t.t[m], m, h);
}

__device__ __forceinline__ void load_tile(const float* __restrict__ mat, int64_t n, f32x16& t, int m, int h) {
    const f32x4* row = reinterpret_cast<const f32x4*>(mat + n * kHid);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = row[m * 8 + 2 * q + h];
        t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ void load_tile(const bf16_bits* __restrict__ mat, int64_t n, f32x16& t, int m, int h) {
    const uint2* row = reinterpret_cast<const uint2*>(mat + n * kHid);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint2 v = row[m * 8 + 2 * q + h];
        t[4 * q] = __uint_as_float(v.x << 16); t[4 * q + 1] = __uint_as_float(v.x & 0xFFFF0000u);
        t[4 * q + 2] = __uint_as_float(v.y << 16); t[4 * q + 3] = __uint_as_float(v.y & 0xFFFF0000u);
    }
}

// ---- fp32 operands as three bf16 pieces (the 6-term split of fpv_policy_split.h, cut where the operand is used)
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
struct Pieces { pbf16x8 p1, p2, p3; };
__device__ __forceinline__ Pieces pieces_of(const float (&v)[8]) {
    Pieces o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float r1 = v[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        o.p1[j] = a; o.p2[j] = b; o.p3[j] = (__bf16)(r1 - (float)b);
    }
    return o;
}
// TERMS = 6: every piece product down to 2^-24 of the product; TERMS = 3 (FPV_PPO_TWO_KERNELS_BF16X3, kept to have its error on
// record): w2 x1 + w1 x2 + w1 x1 only — operands good to 16 bits, products to ~2^-16
template <int TERMS>
__device__ __forceinline__ f32x16 mfma6_pieces(const Pieces& w, const Pieces& x, f32x16 acc) {
    if (TERMS == 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p3, x.p1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x.p3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p2, x.p2, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p2, x.p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x.p2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x.p1, acc, 0, 0, 0);
    return acc;
}

// the same six products for four accumulator tiles that share x, piece by piece ACROSS the tiles: two products into one tile are
// four apart in the instruction stream (the order per tile is mfma6_pieces<6>'s: the same bits)
__device__ __forceinline__ void mfma6x4_pieces(const Pieces (&w)[4], const Pieces& x, f32x16 (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m].p3, x.p1, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m].p1, x.p3, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m].p2, x.p2, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m].p2, x.p1, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m].p1, x.p2, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m].p1, x.p1, acc[m], 0, 0, 0);
}

// ... and for four tiles that share w
__device__ __forceinline__ void mfma6x4_pieces_b(const Pieces& w, const Pieces (&x)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p3, x[m].p1, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x[m].p3, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p2, x[m].p2, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p2, x[m].p1, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x[m].p2, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x[m].p1, acc[m], 0, 0, 0);
}

}  // namespace

// ---------------------------------------------------------------- U0: training image
struct PpoParams {
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2, *wa, *ba, *log_std;
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2, *wv, *bv;
    int in_dim, act_dim;
};

constexpr int kPackBlocks = (kTImage + 255) / 256;      // the training image; 2 * kPieceNet / (3 * 256) more workgroups cut W2 into pieces

__global__ __launch_bounds__(256) void ppo_pack_kernel(PpoParams p, float* __restrict__ img, u32x4* __restrict__ pieces) {
    if ((int)blockIdx.x >= kPackBlocks) {       // one thread per (net, dir, k-step, tile, lane): its eight fp32 weights -> 3 x 16 bytes
        const int q = ((int)blockIdx.x - kPackBlocks) * 256 + (int)threadIdx.x;
        const int lane = q & 63, m = (q >> 6) & 3, ks = (q >> 8) & 7, dir = (q >> 11) & 1, net = q >> 12;
        const int col = lane & 31, h = lane >> 5;
        const float* w2 = net ? p.vf_w2 : p.pi_w2;
        float av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);          // the k-step's summation index held by this lane half
            av[j] = dir == 0 ? w2[(32 * m + col) * kHid + k]                  // layer 2: A = row 32 m + col of W2
                             : w2[k * kHid + 32 * m + col];                   // on the way back: A = column 32 m + col
        }
        const Pieces o = pieces_of(av);
        u32x4* dst = pieces + (size_t)net * kPieceNet + ((dir * 8 + ks) * 4 + m) * 192 + lane;
        dst[0] = (u32x4)o.p1; dst[64] = (u32x4)o.p2; dst[128] = (u32x4)o.p3;
        return;
    }
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kTImage) return;
    float v = 0.0f;
    if (e < kTHead) {
        const int net = e / kTNet, o = e % kTNet;
        const float* w1 = net ? p.vf_w1 : p.pi_w1;
        const float* b1 = net ? p.vf_b1 : p.pi_b1;
        const float* w2 = net ? p.vf_w2 : p.pi_w2;
        const float* b2 = net ? p.vf_b2 : p.pi_b2;
        if (o < kTOffB1) {                      // W1p[m][s][lane] = W1[m*32 + i][s + 12 h]
            const int lane = o % 64, s = (o / 64) % 12, m = o / (64 * 12);
            const int k = s + 12 * (lane >> 5);
            v = k < p.in_dim ? w1[(m * 32 + (lane & 31)) * p.in_dim + k] : 0.0f;
        } else if (o < kTOffW2) {               // b1p[m][h][r]
            const int q = o - kTOffB1, r = q % 16, h = (q / 16) % 2, m = q / 32;
            v = b1[m * 32 + tile_row(r, h)];
        } else if (o < kTOffB2) {               // W2p[m][kt][r4][lane][rr] = W2[m*32 + i][kt*32 + row(4 r4 + rr, h)]
            const int q = o - kTOffW2, rr = q % 4, lane = (q / 4) % 64, r4 = (q / 256) % 4, kt = (q / 1024) % 4, m = q / 4096;
            v = w2[(m * 32 + (lane & 31)) * kHid + kt * 32 + tile_row(4 * r4 + rr, lane >> 5)];
        } else if (o < kTOffW2T) {
            const int q = o - kTOffB2, r = q % 16, h = (q / 16) % 2, m = q / 32;
            v = b2[m * 32 + tile_row(r, h)];
        } else {                                // W2Tp[m][kt][r4][lane][rr] = W2[kt*32 + row(4 r4 + rr, h)][m*32 + i]
            const int q = o - kTOffW2T, rr = q % 4, lane = (q / 4) % 64, r4 = (q / 256) % 4, kt = (q / 1024) % 4, m = q / 4096;
            v = w2[(kt * 32 + tile_row(4 * r4 + rr, lane >> 5)) * kHid + m * 32 + (lane & 31)];
        }
    } else {
        const int q = e - kTHead;
        if (q < 512) {                          // Wa[h][kt][r][o]
            const int o = q % 4, r = (q / 4) % 16, kt = (q / 64) % 4, h = q / 256;
            v = o < p.act_dim ? p.wa[o * kHid + kt * 32 + tile_row(r, h)] : 0.0f;
        } else if (q < 640) {                   // Wv[h][kt][r]
            const int z = q - 512, r = z % 16, kt = (z / 16) % 4, h = z / 64;
            v = p.wv[kt * 32 + tile_row(r, h)];
        } else if (q < 644) v = q - 640 < p.act_dim ? p.ba[q - 640] : 0.0f;
        else if (q == 644) v = p.bv[0];
        else if (q >= 648 && q < 652) v = q - 648 < p.act_dim ? p.log_std[q - 648] : 0.0f;
        else if (q >= 652 && q < 656) v = q - 652 < p.act_dim ? expf(-p.log_std[q - 652]) : 0.0f;
        else if (q == 656) {
            for (int j = 0; j < p.act_dim; ++j) v -= p.log_std[j] + 0.5f * 1.8378770664093453f;
        } else if (q >= kHWaT) {                // WaT[m][s][lane] = Wa[h + 2 s][m*32 + i]
            const int z = q - kHWaT, lane = z % 64, s = (z / 64) % 2, m = z / 128;
            const int j = (lane >> 5) + 2 * s;
            v = j < p.act_dim ? p.wa[j * kHid + m * 32 + (lane & 31)] : 0.0f;
        }
    }
    img[e] = v;
}

// ---------------------------------------------------------------- U1: forward + backward through the activations
struct PpoArgs {
    const float* image;
    const int64_t* index;     // [rows] gather indices into the rollout arrays, or null
    const float* obs;         // [*][in_dim] normalised observations
    const float* actions;     // [*][act_dim]
    const float* old_logp;    // [*]
    const float* adv;         // [*]
    const float* ret;         // [*]
    void* mats;               // [2][4][rows][128] of ST: H1, H2, D1, D2 per net
    float* xg;                // [rows][in_dim] gathered observations (U2's operand for dW1)
    float* dmu;               // [rows][4]
    float* dv;                // [rows]
    float* wave_stats;        // [blocks * waves][8]
    int64_t rows;
    float clip_range, vf_coef, inv_rows;
    int32_t in_dim, act_dim;
};

template <int THREADS, typename ST>
__global__ __launch_bounds__(THREADS) void ppo_fwd_bwd_kernel(PpoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hb = lds + kTNet;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    constexpr int kChunk = THREADS / 2;
    constexpr bool kReloadH1 = THREADS > 256;
    float st[kWaveStats] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int net = 0; net < 2; ++net) {
        if (net) __syncthreads();               // every wave is done reading the pi image
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(p.image + (size_t)net * kTNet);
            f32x4* dst = reinterpret_cast<f32x4*>(lds);
            constexpr int kGran = kTNet / 4, kPer = (kGran + THREADS - 1) / THREADS, kBatch = 12;
            for (int q0 = 0; q0 < kPer; q0 += kBatch) {
                f32x4 stage[kBatch];
#pragma unroll
                for (int q = 0; q < kBatch; ++q) {
                    const int g = (q0 + q) * THREADS + threadIdx.x;
                    stage[q] = src[g < kGran ? g : 0];
                }
#pragma unroll
                for (int q = 0; q < kBatch; ++q) {
                    const int g = (q0 + q) * THREADS + threadIdx.x;
                    if (g < kGran) dst[g] = stage[q];
                }
            }
            if (net == 0)
                for (int g = threadIdx.x; g < kHeadFloats / 4; g += THREADS)
                    reinterpret_cast<f32x4*>(hb)[g] = reinterpret_cast<const f32x4*>(p.image + kTHead)[g];
        }
        __syncthreads();
        ST* const mats = static_cast<ST*>(p.mats);
        ST* H1 = mats + (size_t)(net * 4 + 0) * p.rows * kHid;
        ST* H2 = mats + (size_t)(net * 4 + 1) * p.rows * kHid;
        ST* D1 = mats + (size_t)(net * 4 + 2) * p.rows * kHid;
        ST* D2 = mats + (size_t)(net * 4 + 3) * p.rows * kHid;

#pragma unroll 1
        for (int64_t base = (int64_t)blockIdx.x * kChunk; base < p.rows; base += (int64_t)gridDim.x * kChunk) {
            const int64_t n = base + wave * 32 + col;
            const bool live = n < p.rows;
            const int64_t src = live ? (p.index ? p.index[n] : n) : 0;
            float x[12];
            {
                const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + src * p.in_dim);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bool have = live && 4 * (3 * h + q) < p.in_dim;
                    const f32x4 v = have ? row[3 * h + q] : f32x4{0.f, 0.f, 0.f, 0.f};
                    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                    if (net == 0 && have) reinterpret_cast<f32x4*>(p.xg + n * p.in_dim)[3 * h + q] = v;
                }
            }
            // ---- forward
            Hidden a;
            bias_init(a, lds + kTOffB1, h);
#pragma unroll
            for (int s = 0; s < 12; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    a.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[kTOffW1 + (m * 12 + s) * 64 + lane], x[s], a.t[m], 0, 0, 0);
            // tanh of layer-1 tile kt + 1 and its store ride inside the matrix products that consume tile kt
#pragma unroll
            for (int r = 0; r < 16; ++r) a.t[0][r] = fast_tanh(a.t[0][r]);
            if (live) store_tile(H1, n, a.t[0], 0, h);
            Hidden c;
            bias_init(c, lds + kTOffB2, h);
            {
                f32x4 wc[4], wn[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wc[m] = *reinterpret_cast<const f32x4*>(lds + kTOffW2 + (((m * 4 * 4) * 64 + lane) * 4));
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int kt = t >> 2, r4 = t & 3;
                    if (t < 15) {
                        const int kt1 = (t + 1) >> 2, r41 = (t + 1) & 3;
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            wn[m] = *reinterpret_cast<const f32x4*>(lds + kTOffW2 + ((((m * 4 + kt1) * 4 + r41) * 64 + lane) * 4));
                    }
                    if (kt < 3) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) a.t[kt + 1][4 * r4 + rr] = fast_tanh(a.t[kt + 1][4 * r4 + rr]);
                        if (r4 == 3 && live) store_tile(H1, n, a.t[kt + 1], kt + 1, h);
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            c.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[m][rr], a.t[kt][4 * r4 + rr], c.t[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);   // keep later k-steps' operands and VALU work out of this one
#pragma unroll
                    for (int m = 0; m < 4; ++m) wc[m] = wn[m];
                }
            }
            tanh_all(c);
            if (live) store_rows(H2, n, c, h);

            // ---- heads, loss derivatives, g = dL/dH2
            Hidden g;
            if (net == 0) {
                float mu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(hb + kHWa + (((h * 4 + kt) * 16 + r) * 4));
                        const float v = c.t[kt][r];
                        mu[0] += w.x * v; mu[1] += w.y * v; mu[2] += w.z * v; mu[3] += w.w * v;
                        if (r == 15) __builtin_amdgcn_sched_barrier(0);   // bound the look-ahead of the weight reads
                    }
                float act[4] = {0.f, 0.f, 0.f, 0.f};
                if (live) {
                    if (p.act_dim == 4) {
                        const f32x4 v = reinterpret_cast<const f32x4*>(p.actions)[src];
                        act[0] = v.x; act[1] = v.y; act[2] = v.z; act[3] = v.w;
                    } else {
                        for (int j = 0; j < p.act_dim; ++j) act[j] = p.actions[src * p.act_dim + j];
                    }
                }
                const float old_lp = live ? p.old_logp[src] : 0.f, adv = live ? p.adv[src] : 0.f;
                float z[4], lp = hb[kHLpc];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mu[j] = half_sum(mu[j]);
                    mu[j] += hb[kHBa + j];
                    z[j] = (act[j] - mu[j]) * hb[kHInvStd + j];      // inv_std is 0 beyond act_dim
                    lp -= 0.5f * z[j] * z[j];
                }
                const float lr = lp - old_lp, ratio = __expf(lr);
                const float lo = 1.0f - p.clip_range, hi = 1.0f + p.clip_range;
                const float pg1 = adv * ratio, pg2 = adv * fminf(fmaxf(ratio, lo), hi);
                const float glp = (live && pg1 <= pg2) ? -adv * ratio * p.inv_rows : 0.f;
                float dmu[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dmu[j] = glp * z[j] * hb[kHInvStd + j];
                if (live && h == 0) {
                    reinterpret_cast<f32x4*>(p.dmu)[n] = f32x4{dmu[0], dmu[1], dmu[2], dmu[3]};
                    st[0] -= fminf(pg1, pg2);
                    st[2] += (ratio - 1.0f) - lr;
                    st[3] += fabsf(ratio - 1.0f) > p.clip_range ? 1.0f : 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) st[4 + j] += j < p.act_dim ? glp * (z[j] * z[j] - 1.0f) : 0.f;
                }
                // g^T = Wa^T . dmu^T : K = 4 action components = two k-steps; this half supplies component h + 2 s
                const float b0 = h ? dmu[1] : dmu[0], b1 = h ? dmu[3] : dmu[2];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hb[kHWaT + (m * 2 + 0) * 64 + lane], b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hb[kHWaT + (m * 2 + 1) * 64 + lane], b1, acc, 0, 0, 0);
                    g.t[m] = acc;
                }
            } else {
                float v = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        v += hb[kHWv + (h * 4 + kt) * 16 + r] * c.t[kt][r];
                        if (r == 15) __builtin_amdgcn_sched_barrier(0);
                    }
                v = half_sum(v);
                v += hb[kHBv];
                const float ret = live ? p.ret[src] : 0.f;
                const float err = live ? v - ret : 0.f;
                const float dv = p.vf_coef * 2.0f * err * p.inv_rows;
                if (live && h == 1) {
                    p.dv[n] = dv;
                    st[1] += err * err;
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) g.t[m][r] = hb[kHWv + (h * 4 + m) * 16 + r] * dv;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- D2 = g * (1 - H2^2), then dH1^T = W2^T . D2^T : the D2 tiles are the B operands as they stand; tile
            // kt + 1 is finished (and stored) inside the products that consume tile kt
#pragma unroll
            for (int r = 0; r < 16; ++r) g.t[0][r] *= 1.0f - c.t[0][r] * c.t[0][r];
            if (live) store_tile(D2, n, g.t[0], 0, h);
            Hidden e;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) e.t[m][r] = 0.f;
            {
                f32x4 wc[4], wn[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wc[m] = *reinterpret_cast<const f32x4*>(lds + kTOffW2T + (((m * 4 * 4) * 64 + lane) * 4));
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int kt = t >> 2, r4 = t & 3;
                    if (t < 15) {
                        const int kt1 = (t + 1) >> 2, r41 = (t + 1) & 3;
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            wn[m] = *reinterpret_cast<const f32x4*>(lds + kTOffW2T + ((((m * 4 + kt1) * 4 + r41) * 64 + lane) * 4));
                    }
                    if (kt < 3) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const float hv = c.t[kt + 1][4 * r4 + rr];
                            g.t[kt + 1][4 * r4 + rr] *= 1.0f - hv * hv;
                        }
                        if (r4 == 3 && live) store_tile(D2, n, g.t[kt + 1], kt + 1, h);
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            e.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[m][rr], g.t[kt][4 * r4 + rr], e.t[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) wc[m] = wn[m];
                }
            }
            // D1 = dH1 * (1 - H1^2), a tile at a time.  H1 went to memory after layer 1 (U2 needs it there anyway); with
            // two waves per SIMD there is no room to keep it in registers through the backward products, so it comes
            // back from L2 here (unconditionally: a predicated load would keep the old tiles alive)
            if (kReloadH1) asm volatile("" ::: "memory");
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x16 hm = a.t[m];
                if (kReloadH1) load_tile(H1, live ? n : 0, hm, m, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) e.t[m][r] *= 1.0f - hm[r] * hm[r];
                if (live) store_tile(D1, n, e.t[m], m, h);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- per-wave statistics (fixed lane order)
#pragma unroll
    for (int k = 0; k < kWaveStats; ++k) {
        float v = st[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) p.wave_stats[((size_t)blockIdx.x * (THREADS / 64) + wave) * kWaveStats + k] = v;
    }
}

// ---------------------------------------------------------------- U2: weight gradients, K = rows
struct WgradArgs {
    const void* mats;         // [2][4][rows][128] of ST
    const float* xg;          // [rows][in_dim]
    const float* dmu;         // [rows][4]
    const float* dv;          // [rows]
    float* part;              // [2][slabs][kPartStride]
    int64_t rows;
    int32_t rows_per_slab, slabs, in_dim, act_dim;
};

namespace {
template <typename ST>
struct Ops {                  // one k-step's operands as they come from memory (converted only when they are multiplied)
    ST a2, a1, hh, b[4];
    float ah, x;
};
}  // namespace

// 512 threads = two sets of four waves.  Wave w of a set owns output rows 32 w .. 32 w + 31 of dW2 (four 32-column
// tiles) and dW1, and columns 32 w .. 32 w + 31 of the head gradient; the two sets take alternate 8-row groups of the
// slab (two waves per SIMD hide each other's load latency) and are added through LDS at the end, set 0 + set 1.
// blockIdx.y = net, blockIdx.x = slab of rows.
constexpr int kWgradXchFloats = 4 * 99 * 64;

template <typename ST>
__global__ __launch_bounds__(512) void ppo_wgrad_kernel(WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float xch[];     // [4 waves][99][64 lanes]
    const int net = blockIdx.y, slab = blockIdx.x;
    const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3, set = threadIdx.x >> 8;
    const int i = lane & 31, kk = lane >> 5;
    const int nh = net == 0 ? p.act_dim : 1;
    const int64_t n0 = (int64_t)slab * p.rows_per_slab;
    const int64_t n1 = n0 + p.rows_per_slab < p.rows ? n0 + p.rows_per_slab : p.rows;
    const int len = (int)(n1 - n0);
    // lane-fixed column offsets; rows are 32-bit offsets from the slab's first row
    const ST* const mats = static_cast<const ST*>(p.mats);
    const ST* H1 = mats + ((size_t)(net * 4 + 0) * p.rows + n0) * kHid + i;
    const ST* H2 = mats + ((size_t)(net * 4 + 1) * p.rows + n0) * kHid + 32 * w + i;
    const ST* D1 = mats + ((size_t)(net * 4 + 2) * p.rows + n0) * kHid + 32 * w + i;
    const ST* D2 = mats + ((size_t)(net * 4 + 3) * p.rows + n0) * kHid + 32 * w + i;
    const float* XG = p.xg + n0 * p.in_dim + (i < p.in_dim ? i : p.in_dim - 1);
    const float* AH = net == 0 ? p.dmu + n0 * 4 + (i < 4 ? i : 3) : p.dv + n0;
    const int ah_stride = net == 0 ? 4 : 1;
    const bool x_on = i < p.in_dim, ah_on = i < nh;

    f32x16 w2[4], w1, hd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        w2[0][r] = 0.f; w2[1][r] = 0.f; w2[2][r] = 0.f; w2[3][r] = 0.f; w1[r] = 0.f; hd[r] = 0.f;
    }
    float sb2 = 0.f, sb1 = 0.f, sbh = 0.f;

    // one k-step = rows row0 + kk (slab-relative).  load() only requests; masking waits until fma(), so that a whole
    // group of requests is in flight before anything depends on one of them
    auto load = [&](Ops<ST>& o, int row0) {
        const int row = row0 + kk;
        const int r = row < len ? row : 0;
        o.a2 = D2[r * kHid]; o.a1 = D1[r * kHid]; o.ah = AH[r * ah_stride];
        o.hh = H2[r * kHid];
        o.x = XG[r * p.in_dim];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) o.b[jt] = H1[r * kHid + 32 * jt];
    };
    auto fma = [&](const Ops<ST>& o, int row0) {
        const bool ok = row0 + kk < len;
        const float a2 = ok ? from_store(o.a2) : 0.f, a1 = ok ? from_store(o.a1) : 0.f, ah = (ok && ah_on) ? o.ah : 0.f,
                    x = x_on ? o.x : 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) w2[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, from_store(o.b[jt]), w2[jt], 0, 0, 0);
        w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x, w1, 0, 0, 0);
        hd = __builtin_amdgcn_mfma_f32_32x32x2f32(ah, from_store(o.hh), hd, 0, 0, 0);
        sb2 += a2; sb1 += a1; sbh += ah;
    };

    constexpr int U = 4;                        // k-steps (pairs of rows) per register group = 8 rows
    Ops<ST> A[U], B[U];
    int g0 = 8 * set;                           // this set's groups: rows g0, g0 + 16, ...
#pragma unroll
    for (int u = 0; u < U; ++u) load(A[u], g0 + 2 * u);
#pragma unroll 1
    for (; g0 < len; g0 += 32) {
        // the fences pin the order "request the next group, then multiply the current one": left alone, the scheduler
        // sinks every load next to its use and waits for each (measured 187 us instead of 100)
#pragma unroll
        for (int u = 0; u < U; ++u) load(B[u], g0 + 16 + 2 * u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) fma(A[u], g0 + 2 * u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) load(A[u], g0 + 32 + 2 * u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) fma(B[u], g0 + 16 + 2 * u);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- set 1 hands its sums to set 0
    float* mine = xch + (size_t)w * 99 * 64 + lane;
    if (set == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) mine[(jt * 16 + r) * 64] = w2[jt][r];
            mine[(64 + r) * 64] = w1[r];
            mine[(80 + r) * 64] = hd[r];
        }
        mine[96 * 64] = sb2; mine[97 * 64] = sb1; mine[98 * 64] = sbh;
    }
    __syncthreads();
    if (set == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) w2[jt][r] += mine[(jt * 16 + r) * 64];
        w1[r] += mine[(64 + r) * 64];
        hd[r] += mine[(80 + r) * 64];
    }
    sb2 += mine[96 * 64]; sb1 += mine[97 * 64]; sbh += mine[98 * 64];

    float* out = p.part + ((size_t)net * p.slabs + slab) * kPartStride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 32 * w + tile_row(r, kk);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) out[kPW2 + o * kHid + 32 * jt + i] = w2[jt][r];
        if (x_on) out[kPW1 + o * p.in_dim + i] = w1[r];
    }
    if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nh) out[kPWh + r * kHid + 32 * w + i] = hd[r];
    }
    sb2 = half_sum(sb2); sb1 = half_sum(sb1); sbh = half_sum(sbh);
    if (kk == 0) {
        out[kPB2 + 32 * w + i] = sb2;
        out[kPB1 + 32 * w + i] = sb1;
        if (w == 0 && ah_on) out[kPBh + i] = sbh;
    }
}

// ---------------------------------------------------------------- U2 on the bf16 matrix pipe (fpv_ppo_io.path = FPV_PPO_TWO_KERNELS_BF16X6)
// The same weight gradients with every fp32 operand cut into three bf16 pieces and the six piece products that matter per
// product (fpv_policy_split.h's mfma6: w3 x1, w1 x3, w2 x2, w2 x1, w1 x2, w1 x1, small terms first; fp32 accumulation): the
// 32x32x16 bf16 product takes 16 rows of the slab per instruction where the fp32 one takes 2, at 16 x the rate — 6 / 16 of the
// matrix-pipe time.  A lane holds eight ROWS of one column per operand (k = 8 (lane >> 5) + j on both operands, so any
// consistent numbering of the 16 rows will do): eight dword loads per fragment, 128 contiguous bytes per half-wave each.  Same
// work split as above (wave w of a set: output rows 32 w .. of dW2 / dW1, columns 32 w .. of the head gradient), the two sets
// take alternate 16-row groups, set 0 + set 1 through LDS, the same partial layout: U3 does not know which kernel ran.
// fp32 storage only (a bf16-stored operand has one piece: that combination keeps the fp32-MFMA kernel).
template <int TERMS>
__global__ __launch_bounds__(512) void ppo_wgrad_split_kernel(WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float xch[];     // [4 waves][99][64 lanes]
    const int net = blockIdx.y, slab = blockIdx.x;
    const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3, set = threadIdx.x >> 8;
    const int i = lane & 31, kk = lane >> 5;
    const int nh = net == 0 ? p.act_dim : 1;
    const int64_t n0 = (int64_t)slab * p.rows_per_slab;
    const int64_t n1 = n0 + p.rows_per_slab < p.rows ? n0 + p.rows_per_slab : p.rows;
    const int len = (int)(n1 - n0);
    const float* const mats = static_cast<const float*>(p.mats);
    const float* H1 = mats + ((size_t)(net * 4 + 0) * p.rows + n0) * kHid + i;
    const float* H2 = mats + ((size_t)(net * 4 + 1) * p.rows + n0) * kHid + 32 * w + i;
    const float* D1 = mats + ((size_t)(net * 4 + 2) * p.rows + n0) * kHid + 32 * w + i;
    const float* D2 = mats + ((size_t)(net * 4 + 3) * p.rows + n0) * kHid + 32 * w + i;
    const float* XG = p.xg + n0 * p.in_dim + (i < p.in_dim ? i : p.in_dim - 1);
    const float* AH = net == 0 ? p.dmu + n0 * 4 + (i < 4 ? i : 3) : p.dv + n0;
    const int ah_stride = net == 0 ? 4 : 1;
    const bool x_on = i < p.in_dim, ah_on = i < nh;

    f32x16 w2[4], w1, hd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        w2[0][r] = 0.f; w2[1][r] = 0.f; w2[2][r] = 0.f; w2[3][r] = 0.f; w1[r] = 0.f; hd[r] = 0.f;
    }
    float sb2 = 0.f, sb1 = 0.f, sbh = 0.f;
#pragma unroll 1
    for (int g0 = 16 * set; g0 < len; g0 += 32) {
        // this lane's eight rows of the group (clamped for the loads, masked after them)
        int rr[8];
        bool ok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = g0 + 8 * kk + j;
            ok[j] = row < len;
            rr[j] = ok[j] ? row : 0;
        }
        float a2[8], a1[8], ah[8], hh[8], x[8], b[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                      // every request of the group before the first use
            a2[j] = D2[rr[j] * kHid]; a1[j] = D1[rr[j] * kHid]; ah[j] = AH[rr[j] * ah_stride];
            hh[j] = H2[rr[j] * kHid]; x[j] = XG[rr[j] * p.in_dim];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) b[jt][j] = H1[rr[j] * kHid + 32 * jt];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a2[j] = ok[j] ? a2[j] : 0.f; a1[j] = ok[j] ? a1[j] : 0.f; ah[j] = (ok[j] && ah_on) ? ah[j] : 0.f;
            x[j] = x_on ? x[j] : 0.f;
            sb2 += a2[j]; sb1 += a1[j]; sbh += ah[j];
        }
        const Pieces pa2 = pieces_of(a2);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) w2[jt] = mfma6_pieces<TERMS>(pa2, pieces_of(b[jt]), w2[jt]);
        w1 = mfma6_pieces<TERMS>(pieces_of(a1), pieces_of(x), w1);
        hd = mfma6_pieces<TERMS>(pieces_of(ah), pieces_of(hh), hd);
    }

    // ---- set 1 hands its sums to set 0 (as in ppo_wgrad_kernel)
    float* mine = xch + (size_t)w * 99 * 64 + lane;
    if (set == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) mine[(jt * 16 + r) * 64] = w2[jt][r];
            mine[(64 + r) * 64] = w1[r];
            mine[(80 + r) * 64] = hd[r];
        }
        mine[96 * 64] = sb2; mine[97 * 64] = sb1; mine[98 * 64] = sbh;
    }
    __syncthreads();
    if (set == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) w2[jt][r] += mine[(jt * 16 + r) * 64];
        w1[r] += mine[(64 + r) * 64];
        hd[r] += mine[(80 + r) * 64];
    }
    sb2 += mine[96 * 64]; sb1 += mine[97 * 64]; sbh += mine[98 * 64];

    float* out = p.part + ((size_t)net * p.slabs + slab) * kPartStride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 32 * w + tile_row(r, kk);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) out[kPW2 + o * kHid + 32 * jt + i] = w2[jt][r];
        if (x_on) out[kPW1 + o * p.in_dim + i] = w1[r];
    }
    if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nh) out[kPWh + r * kHid + 32 * w + i] = hd[r];
    }
    sb2 = half_sum(sb2); sb1 = half_sum(sb1); sbh = half_sum(sbh);
    if (kk == 0) {
        out[kPB2 + 32 * w + i] = sb2;
        out[kPB1 + 32 * w + i] = sb1;
        if (w == 0 && ah_on) out[kPBh + i] = sbh;
    }
}

// ---------------------------------------------------------------- U12: U1 and U2 as one kernel (fpv_ppo_io.path = FPV_PPO_ONE_KERNEL, the default)
// The matrices U1 hands to U2 are 4 KB per row and net: written and read back they are 537 MB of HBM traffic per 65 536-row
// minibatch, and neither trip hides behind the products (U1 135 us with the stores, 66 without; U2 91 us, 60 with its operands in
// registers: profiles/r04_ppo_update_profile.txt).  Here they never leave the CU.  A workgroup (four waves, one per SIMD) takes one
// net and one slab of rows, 128 rows at a time; per chunk
//   phase A  every wave: forward, loss derivatives and backward of its own 32 rows as in U1 (rows on the MFMA column, the matrices
//            in accumulator tiles); H2 and d out are laid down in LDS as soon as D2 exists, so the backward runs with three tile
//            sets live, not four;
//   phase B  the weight gradients of the chunk's 128 rows as in U2 — K = the row index, wave w owns output rows 32 w .. 32 w + 31 of
//            dW2 and dW1 and columns 32 w .. of the head gradient — with the operands read from ONE row-major LDS matrix
//            [128 rows][132] that all four waves fill together, three times: H2 (head gradient, with d out), D2 (each wave takes
//            its 64 A operands into registers), D1 (with x: dW1 += D1^T x); H1, dW2's B operand (dW2 += D2^T H1), is laid down
//            in the same room as bf16 PIECES and read back transposed (ds_read_b64_tr_b16).  The transposition U2 got from the
//            row-major matrices in memory is the LDS write (or that read) here.
// W2 is not held in LDS: U0 cuts it once per update into a piece image (both directions' MFMA A fragments), and the kernel streams
// that image through a two-slot ring, one k-step per slot (round 5; until then W2 lay in LDS as fp32 and every wave cut its
// fragments again in every chunk).  W1's 48 A operands per lane come from U0's image in memory (the same for every chunk; asked
// for during the previous chunk's last products).  The partial sums of a slab stay in registers over its rows and are written once,
// in U2's layout: U3 does not know which kernel ran.  The barriers order LDS traffic only.  The three 128 x 128 products per row —
// layer 2, its transpose on the way back, dW2 — run as 6-term bf16 splits of fp32 operands (fp32-level: the golden-vector error is
// the fp32 kernels'): the weights' pieces from U0, H1's cut once per chunk by the wave that holds the row, a wave's own activations
// and deltas cut where they are used; everything else on the fp32-input MFMA; no atomics: bitwise reproducible for a given launch shape.
constexpr int kFRow = 132;                      // floats per row of the staged matrix
constexpr int kFOffB1 = 0, kFOffB2 = 128, kFOffRing = 256, kFOffHead = kFOffRing + 2 * kPieceStep * 4;
constexpr int kFOffX = (kFOffHead + kHeadFloats + 3) / 4 * 4;
constexpr int kFOffDo = kFOffX + 128 * 24, kFOffM = kFOffDo + 128 * 4;
constexpr int kH1PieceBytes = 128 * 256;         // one piece of the chunk's H1 as bf16, 256-byte rows
constexpr int kFusedFloats = kFOffM + 3 * kH1PieceBytes / 4;    // the staged matrix [128][132] fp32 lies in the same place at other times
static_assert(128 * kFRow <= 3 * kH1PieceBytes / 4, "the fp32 matrix fits into the piece image's room");
static_assert(kFusedFloats * 4 <= 160 * 1024, "the fused PPO kernel's LDS image");
static_assert(kFOffX % 4 == 0 && kFOffM % 4 == 0 && kFOffHead % 4 == 0 && kFOffRing % 4 == 0, "16-byte rows");

struct FusedArgs {
    const float* image;       // U0's training image: W1p, b1, b2 and the head block are taken from it
    const u32x4* pieces;      // U0's piece image of W2 (both nets, both directions)
    const int64_t* index;
    const float *obs, *actions, *old_logp, *adv, *ret;
    float* part;              // [2][slabs][kPartStride]
    float* wave_stats;        // [2 * slabs * 4][8]
    int64_t rows;
    float clip_range, vf_coef, inv_rows;
    int32_t rows_per_slab, slabs, in_dim, act_dim;
};

// workgroup barrier that orders LDS traffic only: global reads stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// a wave's 32 rows of one matrix, accumulator tiles -> row-major LDS rows (four consecutive columns per register quad)
__device__ __forceinline__ void stage_tile_rows(float* __restrict__ mat, int row, const Hidden& t, int h) {
    f32x4* dst = reinterpret_cast<f32x4*>(mat + row * kFRow);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[m * 8 + 2 * q + h] = f32x4{t.t[m][4 * q], t.t[m][4 * q + 1], t.t[m][4 * q + 2], t.t[m][4 * q + 3]};
}

// Phase stamps (s_memrealtime, 100 MHz) of the diagnostic build -DFPV_PPO_STAMPS=1 (tools/ppo_stamps.py): two workgroups write
// them over the start of the matrices' workspace, which this path does not use.  In the product the markers compile to nothing.
#ifndef FPV_PPO_STAMPS
#define FPV_PPO_STAMPS 0
#endif
#if FPV_PPO_STAMPS
#define PPO_STAMP_SETUP()                                                                                                        \
    unsigned long long* stamps_ = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.image) + kTImage) +               \
                                  (size_t)((blockIdx.x == 0 ? 0 : 1) * 4 + (threadIdx.x >> 6)) * 128;                           \
    const bool stamp_on_ = ((blockIdx.x == 0 && blockIdx.y == 0) || (blockIdx.x == 77 && blockIdx.y == 1)) && (threadIdx.x & 63) == 0
#define PPO_STAMP(k) do { if (stamp_on_ && (k) < 120) { stamps_[k] = __builtin_amdgcn_s_memrealtime(); if ((k) < 4) stamps_[120 + (k)] = __builtin_readcyclecounter(); } } while (0)
#else
#define PPO_STAMP_SETUP() ((void)0)
#define PPO_STAMP(k) ((void)0)
#endif

__global__ __launch_bounds__(256) void ppo_fused_kernel(FusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const hb = lds + kFOffHead;
    const int net = blockIdx.y, slab = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int nh = net == 0 ? p.act_dim : 1;
    const bool x_on = col < p.in_dim, ah_on = col < nh;
    const int64_t n0 = (int64_t)slab * p.rows_per_slab;
    const int64_t n1 = n0 + p.rows_per_slab < p.rows ? n0 + p.rows_per_slab : p.rows;
    float st[kWaveStats] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    PPO_STAMP_SETUP();
    PPO_STAMP(0);
    const float* const img = p.image + (size_t)net * kTNet;

    // ---- the net's image: b1 | b2 and the head block from U0's image
    {
        for (int g = threadIdx.x; g < kHid / 4; g += 256) {
            reinterpret_cast<f32x4*>(lds + kFOffB1)[g] = reinterpret_cast<const f32x4*>(img + kTOffB1)[g];
            reinterpret_cast<f32x4*>(lds + kFOffB2)[g] = reinterpret_cast<const f32x4*>(img + kTOffB2)[g];
        }
        for (int g = threadIdx.x; g < kHeadFloats / 4; g += 256)
            reinterpret_cast<f32x4*>(hb)[g] = reinterpret_cast<const f32x4*>(p.image + kTHead)[g];
    }
    // ---- W2's pieces stream through a two-slot LDS ring, one k-step (12 KB: four tiles x three pieces) per slot: step t of a chunk
    // (t = 0..7 layer 2, 8..15 the way back) is read from slot t & 1 by all four waves; wave w fetched its tile's quarter of step
    // t + 1 from U0's piece image two steps ahead (registers G) and lays it down in the other slot while step t is multiplied; the
    // barrier that ends a step publishes the slot and frees the one just read.  The stream is cyclic (step 16 = the next chunk's 0).
    u32x4* const ring = reinterpret_cast<u32x4*>(lds + kFOffRing);
    const u32x4* wstream = p.pieces + (size_t)net * kPieceNet + wave * 192 + lane;
    u32x4 G[3];
    auto stream_load = [&](int t) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) G[pc] = wstream[(t & 15) * kPieceStep + pc * 64];
    };
    auto ring_put = [&](int slot) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) ring[slot * kPieceStep + (wave * 3 + pc) * 64 + lane] = G[pc];
    };
    auto ring_get = [&](int slot, Pieces (&w)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            w[m].p1 = (pbf16x8)ring[slot * kPieceStep + (m * 3 + 0) * 64 + lane];
            w[m].p2 = (pbf16x8)ring[slot * kPieceStep + (m * 3 + 1) * 64 + lane];
            w[m].p3 = (pbf16x8)ring[slot * kPieceStep + (m * 3 + 2) * 64 + lane];
        }
    };
    stream_load(0);
    ring_put(0);
    stream_load(1);
    // W1's A operands: the same for every chunk, but 48 registers are not to be had across a chunk: read again per chunk (the
    // opaque zero keeps the reads inside the loop), during the previous chunk's last products
    float w1r[48];
    auto load_w1 = [&]() {
        int zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        const float* w1p = img + kTOffW1 + lane + zero;
#pragma unroll
        for (int q = 0; q < 48; ++q) w1r[q] = w1p[q * 64];
    };
    load_w1();

    // weight-gradient sums of this wave's output rows over the whole slab
    f32x16 w2s[4], w1s;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        w2s[0][r] = 0.f; w2s[1][r] = 0.f; w2s[2][r] = 0.f; w2s[3][r] = 0.f; w1s[r] = 0.f;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 hd01 = {0.f, 0.f}, hd23 = {0.f, 0.f};                  // head gradient of hidden unit 32 wave + i (outputs 0,1 and 2,3)
    float sb4[4] = {0.f, 0.f, 0.f, 0.f};                         // head bias gradient: this lane's rows' d out (lane half 0 counts)
    float sb2 = 0.f, sb1 = 0.f;
    float* const M = lds + kFOffM;
    float* const XS = lds + kFOffX;
    float* const DO = lds + kFOffDo;
    const int kk = h, i = col;
    const int srow = wave * 32 + col;                            // this lane's row inside the chunk
    // the first chunk's row index and x (every later chunk's are asked for during the chunk before)
    int64_t src_next = n0 + srow < n1 ? (p.index ? p.index[n0 + srow] : n0 + srow) : 0;
    int64_t src_after = n0 + 128 + srow < n1 ? (p.index ? p.index[n0 + 128 + srow] : n0 + 128 + srow) : 0;
    f32x4 xn[3];
    {
        const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + src_next * p.in_dim);
#pragma unroll
        for (int q = 0; q < 3; ++q) xn[q] = row[4 * (3 * h + q) < p.in_dim ? 3 * h + q : 0];
    }
    __syncthreads();                                             // the image is in place
    PPO_STAMP(1);

#pragma unroll 1
    for (int64_t base = n0; base < n1; base += 128) {
        // ================================================================ phase A: this wave's 32 rows (as ppo_fwd_bwd_kernel)
        const int64_t n = base + srow;
        [[maybe_unused]] const int ck = 4 + 16 * (int)((base - n0) >> 7);
        {                                                        // (the stream's addresses repeat chunk after chunk: an opaque zero keeps
            int zero;                                            // its reads inside the loop, as for W1 below)
            asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
            wstream = p.pieces + (size_t)net * kPieceNet + wave * 192 + lane + zero;
        }
        const bool live = n < n1;
        const int64_t src = src_next;
        // the row's other inputs: asked for now, used after the forward (a gathered row is an HBM miss)
        f32x4 act4 = {0.f, 0.f, 0.f, 0.f};
        float in0 = 0.f, in1 = 0.f;                              // pi: old log-probability, advantage; V: return
        if (net == 0) {
            if (p.act_dim == 4) act4 = reinterpret_cast<const f32x4*>(p.actions)[src];
            else
                for (int j = 0; j < p.act_dim; ++j) act4[j] = p.actions[src * p.act_dim + j];
            in0 = p.old_logp[src]; in1 = p.adv[src];
        } else {
            in0 = p.ret[src];
        }
        Hidden a;
        {
            float x[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bool have = live && 4 * (3 * h + q) < p.in_dim;
                const f32x4 v = have ? xn[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
            bias_init(a, lds + kFOffB1, h);
#pragma unroll
            for (int s = 0; s < 12; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    a.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1r[m * 12 + s], x[s], a.t[m], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a.t[0][r] = fast_tanh(a.t[0][r]);
        Hidden c;
        bias_init(c, lds + kFOffB2, h);
        PPO_STAMP(ck + 15);                                      // (layer 1 is behind us)
        {
            // layer 2 on the bf16 pipe: k-step (kt, sp) sums over hidden units 32 kt + 16 sp + 8 (j >> 2) + 4 h + (j & 3), j = 0..7 — what
            // registers 8 sp .. 8 sp + 7 of accumulator tile kt hold in lane half h.  B = those eight values of H1, cut into three bf16
            // pieces here; A = the same eight columns of W2's row 32 m + col, already in pieces (U0's piece image, through the ring):
            // six piece products per tile, issued piece by piece across the four tiles — fp32-level results at 6/16 of the matrix
            // cycles.  (Until round 5 W2 lay in LDS as fp32 and every wave cut its A operands again per chunk: 64 cuts of eight values
            // per wave and chunk, more vector work than everything else in the chunk together.)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int kt = ks >> 1, sp = ks & 1;
                Pieces wa[4];
                ring_get(ks & 1, wa);
                ring_put((ks + 1) & 1);
                stream_load(ks + 2);
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = a.t[kt][8 * sp + j];
                const Pieces pb = pieces_of(bv);
                if (kt < 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) a.t[kt + 1][8 * sp + j] = fast_tanh(a.t[kt + 1][8 * sp + j]);
                }
                mfma6x4_pieces(wa, pb, c.t);
                lds_barrier();
            }
        }
        PPO_STAMP(ck + 0);
        tanh_all(c);

        // ---- heads, loss derivatives, g = dL/dH2
        Hidden g;
        float dout[4] = {0.f, 0.f, 0.f, 0.f};
        if (net == 0) {
            float mu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(hb + kHWa + (((h * 4 + kt) * 16 + r) * 4));
                    const float v = c.t[kt][r];
                    mu[0] += w.x * v; mu[1] += w.y * v; mu[2] += w.z * v; mu[3] += w.w * v;
                    if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            const float act[4] = {live ? act4.x : 0.f, live ? act4.y : 0.f, live ? act4.z : 0.f, live ? act4.w : 0.f};
            const float old_lp = live ? in0 : 0.f, adv = live ? in1 : 0.f;
            float z[4], lp = hb[kHLpc];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mu[j] = half_sum(mu[j]);
                mu[j] += hb[kHBa + j];
                z[j] = (act[j] - mu[j]) * hb[kHInvStd + j];
                lp -= 0.5f * z[j] * z[j];
            }
            const float lr = lp - old_lp, ratio = __expf(lr);
            const float lo = 1.0f - p.clip_range, hi = 1.0f + p.clip_range;
            const float pg1 = adv * ratio, pg2 = adv * fminf(fmaxf(ratio, lo), hi);
            const float glp = (live && pg1 <= pg2) ? -adv * ratio * p.inv_rows : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) dout[j] = glp * z[j] * hb[kHInvStd + j];
            if (live && h == 0) {
                st[0] -= fminf(pg1, pg2);
                st[2] += (ratio - 1.0f) - lr;
                st[3] += fabsf(ratio - 1.0f) > p.clip_range ? 1.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) st[4 + j] += j < p.act_dim ? glp * (z[j] * z[j] - 1.0f) : 0.f;
            }
            const float b0 = h ? dout[1] : dout[0], b1 = h ? dout[3] : dout[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hb[kHWaT + (m * 2 + 0) * 64 + lane], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hb[kHWaT + (m * 2 + 1) * 64 + lane], b1, acc, 0, 0, 0);
                g.t[m] = acc;
            }
        } else {
            float v = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v += hb[kHWv + (h * 4 + kt) * 16 + r] * c.t[kt][r];
                    if (r == 15) __builtin_amdgcn_sched_barrier(0);
                }
            v = half_sum(v);
            v += hb[kHBv];
            const float ret = live ? in0 : 0.f;
            const float err = live ? v - ret : 0.f;
            dout[0] = p.vf_coef * 2.0f * err * p.inv_rows;
            if (live && h == 1) st[1] += err * err;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) g.t[m][r] = hb[kHWv + (h * 4 + m) * 16 + r] * dout[0];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- D2 = g (1 - H2^2); H2 and d out go to LDS now (the previous chunk's last readers are past this barrier)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) g.t[m][r] *= 1.0f - c.t[m][r] * c.t[m][r];
        PPO_STAMP(ck + 1);
        lds_barrier();
        PPO_STAMP(ck + 2);
        stage_tile_rows(M, srow, c, h);
        if (h == 0) {
            reinterpret_cast<f32x4*>(DO)[srow] = f32x4{dout[0], dout[1], dout[2], dout[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) sb4[j] += dout[j];
        }
        // ---- dH1^T = W2^T . D2^T (the W2 column reads), D1 = dH1 (1 - H1^2)
        Hidden e;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) e.t[m][r] = 0.f;
        {
            // the same on the way back: A = column 32 m + col of W2 at the k-step's eight rows (steps 8..15 of the stream), B = D2
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int kt = ks >> 1, sp = ks & 1;
                Pieces wa[4];
                ring_get(ks & 1, wa);
                ring_put((ks + 1) & 1);
                stream_load(8 + ks + 2);
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = g.t[kt][8 * sp + j];
                const Pieces pb = pieces_of(bv);
                mfma6x4_pieces(wa, pb, e.t);
                lds_barrier();
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) e.t[m][r] *= 1.0f - a.t[m][r] * a.t[m][r];

        PPO_STAMP(ck + 3);
        // ================================================================ phase B: weight gradients of the chunk's 128 rows
        // x again, from L2: asked for here, laid down with D1 three fillings later
        f32x4 xr[3];
        {
            const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + src * p.in_dim);
#pragma unroll
            for (int q = 0; q < 3; ++q) xr[q] = row[4 * (3 * h + q) < p.in_dim ? 3 * h + q : 0];
        }
        // and the next chunk's x, through the row index asked for a chunk ago; the index of the chunk after that is asked for now
        // (a gathered x behind a gathered index would be two misses in a row)
        {
            src_next = src_after;
            const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + src_next * p.in_dim);
#pragma unroll
            for (int q = 0; q < 3; ++q) xn[q] = row[4 * (3 * h + q) < p.in_dim ? 3 * h + q : 0];
            const int64_t na = n + 256;
            src_after = na < n1 ? (p.index ? p.index[na] : na) : 0;
        }
        // K = the chunk's 128 rows, two per product (row 2 s + kk); operands of the next group of k-steps are asked for before the
        // current group is multiplied (pinned with scheduling fences, as in ppo_wgrad_kernel).
        lds_barrier();                                           // H2 and d out of all four waves are in place
        PPO_STAMP(ck + 4);
        {
            // head gradient += d out^T H2 on the vector pipe: four (or one) outputs per hidden unit would fill an eighth of an MFMA
            // tile, and a lone wave's two LDS reads per product cost more than the product.  Lane (i, kk) owns hidden unit
            // 32 wave + i and the rows 2 s + kk: 64 reads of H2, 64 broadcast reads of d out, 256 fmas.
            const float* HH = M + kk * kFRow + 32 * wave + i;
            const f32x4* DV = reinterpret_cast<const f32x4*>(DO) + kk;
            // two groups of eight k-steps in flight: the next group's reads are asked for before the current one is used (left alone,
            // the compiler waits for every pair of reads)
            float hP[8], hQ[8];
            f32x4 dP[8], dQ[8];
            auto load = [&](float (&hh)[8], f32x4 (&d)[8], int s0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { hh[u] = HH[2 * (s0 + u) * kFRow]; d[u] = DV[2 * (s0 + u)]; }
            };
            auto use = [&](const float (&hh)[8], const f32x4 (&d)[8]) {      // two packed fmas per row (the head BIAS gradient is summed
#pragma unroll                                                                        // row by row where d out is formed, not here)
                for (int u = 0; u < 8; ++u) {
                    const f32x2 h2 = {hh[u], hh[u]};
                    hd01 = __builtin_elementwise_fma(f32x2{d[u].x, d[u].y}, h2, hd01);
                    hd23 = __builtin_elementwise_fma(f32x2{d[u].z, d[u].w}, h2, hd23);
                }
            };
            load(hP, dP, 0);
#pragma unroll 1
            for (int s0 = 0; s0 < 64; s0 += 16) {
                load(hQ, dQ, s0 + 8);
                __builtin_amdgcn_sched_barrier(0);
                use(hP, dP);
                __builtin_amdgcn_sched_barrier(0);
                load(hP, dP, s0 + 16 < 64 ? s0 + 16 : 0);
                __builtin_amdgcn_sched_barrier(0);
                use(hQ, dQ);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PPO_STAMP(ck + 5);
        lds_barrier();
        PPO_STAMP(ck + 6);
        stage_tile_rows(M, srow, g, h);                          // D2 of the chunk
        lds_barrier();
        PPO_STAMP(ck + 7);
        // this wave's A operands of dW2, as the bf16 product wants them: a k-step is 16 rows and lane half h supplies rows
        // 16 S + 8 h + j, j = 0..7, of column 32 wave + i (cut into pieces k-step by k-step)
        float a2r[64];
        {
            const float* A2 = M + 8 * kk * kFRow + 32 * wave + i;
#pragma unroll
            for (int s = 0; s < 64; ++s) { a2r[s] = A2[(16 * (s >> 3) + (s & 7)) * kFRow]; sb2 += a2r[s]; }
        }
        PPO_STAMP(ck + 8);
        lds_barrier();
        {
            // H1 of the chunk as bf16 PIECES, row-major [piece][chunk row][128 units] with 256-byte rows: the cut is made once, by the
            // wave that holds the row (eight cuts; each wave used to cut all of H1 again for its own products, 32 cuts), and dW2's B
            // operand — eight ROWS of one unit per lane — comes back through ds_read_b64_tr_b16, which hands a lane four rows of one
            // column out of four 8-byte row segments.  16-byte chunk ch of row r lies at 256 r + 16 (ch ^ ((r & 3) << 2 | (r >> 2) & 3)):
            // the transposed reads (one 32-lane half = 4 rows x 64 bytes) fall on 64 banks; the 8-byte writes of 16 rows are 2-way.
            unsigned char* const HP = reinterpret_cast<unsigned char*>(M);
            int fr = ((srow & 3) << 2) | ((srow >> 2) & 3);
            asm volatile("" : "+v"(fr));                         // (formed here, not ahead of the chunk loop: 16 addresses would be spilled)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int kt = ks >> 1, sp = ks & 1;
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bv[j] = a.t[kt][8 * sp + j];
                    asm volatile("" : "+v"(bv[j]));              // cut again: kept from the forward pass, the 96 piece registers went to scratch
                }
                const Pieces pv = pieces_of(bv);
                const u32x4 q1 = (u32x4)pv.p1, q2 = (u32x4)pv.p2, q3 = (u32x4)pv.p3;
                // j = 0..3: units 16 ks + 4 h + 0..3 (chunk 2 ks, bytes 8 h ..); j = 4..7: units 16 ks + 8 + 4 h + 0..3 (chunk 2 ks + 1)
                unsigned char* w0 = HP + 256 * srow + 16 * ((2 * ks) ^ fr) + 8 * h;
                unsigned char* w1 = HP + 256 * srow + 16 * ((2 * ks + 1) ^ fr) + 8 * h;
                *reinterpret_cast<uint2*>(w0) = uint2{q1.x, q1.y}; *reinterpret_cast<uint2*>(w1) = uint2{q1.z, q1.w};
                *reinterpret_cast<uint2*>(w0 + kH1PieceBytes) = uint2{q2.x, q2.y}; *reinterpret_cast<uint2*>(w1 + kH1PieceBytes) = uint2{q2.z, q2.w};
                *reinterpret_cast<uint2*>(w0 + 2 * kH1PieceBytes) = uint2{q3.x, q3.y}; *reinterpret_cast<uint2*>(w1 + 2 * kH1PieceBytes) = uint2{q3.z, q3.w};
            }
        }
        PPO_STAMP(ck + 9);
        lds_barrier();
        PPO_STAMP(ck + 10);
        {                                                        // dW2 += D2^T H1, six piece products per product (as layer 2 above)
            // lane l of a 16-lane group: q = (l & 15) >> 2 names the block's row, pp = l & 3 its 8-byte segment; the block = rows
            // 16 S + 8 kk + 4 half + 0..3, units 32 jt + 16 gb + 0..15 (gb = the group's parity): element q of the result is row q of
            // unit 32 jt + (l & 31).  With the row's swizzle f = 4 q + 2 kk + half the chunk index is ((jt ^ q) << 2) | (2 gb + (pp >> 1)) ^ (2 kk + half).
            typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
            typedef pbf16x4 __attribute__((address_space(3))) * lds_b4_ptr;
            const int pp = lane & 3, gb = (lane >> 4) & 1;
            int qq = (lane & 15) >> 2, lo0 = (2 * gb + (pp >> 1)) ^ (2 * kk);
            asm volatile("" : "+v"(qq), "+v"(lo0));
            int opaque0;                                         // (keeps the 192 read addresses from being formed ahead of the chunk loop and spilled)
            asm volatile("s_mov_b32 %0, 0" : "=s"(opaque0));
            const unsigned hp0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)(M)) + 256 * (8 * kk + qq) + 8 * (pp & 1) + opaque0;
            struct Grp { Pieces b[4]; };
            auto load = [&](Grp& o, int S) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    pbf16x4 v[3][2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const unsigned ad = hp0 + 4096 * S + 1024 * half + 16 * ((((jt ^ qq) << 2)) | (lo0 ^ half));
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            v[pc][half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)(size_t)(ad + pc * kH1PieceBytes));
                    }
                    o.b[jt].p1 = __builtin_shufflevector(v[0][0], v[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    o.b[jt].p2 = __builtin_shufflevector(v[1][0], v[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    o.b[jt].p3 = __builtin_shufflevector(v[2][0], v[2][1], 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            auto fma = [&](const Grp& o, int S) {
                float t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = a2r[8 * S + j];
                const Pieces pa = pieces_of(t8);
                mfma6x4_pieces_b(pa, o.b, w2s);
            };
            Grp P, Q;                                            // two register sets: step S + 1's reads are under way while step S is multiplied
            load(P, 0);
#pragma unroll
            for (int S = 0; S < 8; S += 2) {
                load(Q, S + 1);
                __builtin_amdgcn_sched_barrier(0);
                fma(P, S);
                __builtin_amdgcn_sched_barrier(0);
                if (S + 2 < 8) load(P, S + 2);
                __builtin_amdgcn_sched_barrier(0);
                fma(Q, S + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PPO_STAMP(ck + 11);
        lds_barrier();
        PPO_STAMP(ck + 12);
        stage_tile_rows(M, srow, e, h);                          // D1 of the chunk, and x
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const bool have = live && 4 * (3 * h + q) < p.in_dim;
            reinterpret_cast<f32x4*>(XS + srow * 24)[3 * h + q] = have ? xr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        PPO_STAMP(ck + 13);
        lds_barrier();
        PPO_STAMP(ck + 14);
        load_w1();                                               // for the next chunk
        {                                                        // dW1 += D1^T x
            constexpr int G = 8;
            struct Grp { float a1[G], xv[G]; };
            const float* A1 = M + kk * kFRow + 32 * wave + i;
            const float* XV = XS + kk * 24 + (i < 24 ? i : 23);
            auto load = [&](Grp& o, int s0) {
#pragma unroll
                for (int u = 0; u < G; ++u) { o.a1[u] = A1[2 * (s0 + u) * kFRow]; o.xv[u] = XV[2 * (s0 + u) * 24]; }
            };
            f32x16 odd;                                          // (the second chain, as in the head gradient)
#pragma unroll
            for (int r = 0; r < 16; ++r) odd[r] = 0.f;
            auto fma = [&](const Grp& o) {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const float xv = x_on ? o.xv[u] : 0.f;
                    if (u & 1) odd = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a1[u], xv, odd, 0, 0, 0);
                    else w1s = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a1[u], xv, w1s, 0, 0, 0);
                    sb1 += o.a1[u];
                }
            };
            Grp P, Q;
            load(P, 0);
#pragma unroll 1
            for (int s0 = 0; s0 < 64; s0 += 2 * G) {
                load(Q, s0 + G);
                __builtin_amdgcn_sched_barrier(0);
                fma(P);
                __builtin_amdgcn_sched_barrier(0);
                load(P, s0 + 2 * G < 64 ? s0 + 2 * G : 0);
                __builtin_amdgcn_sched_barrier(0);
                fma(Q);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) w1s[r] += odd[r];
        }
    }

    PPO_STAMP(2);
    // ---- the slab's partial sums, in ppo_wgrad_kernel's layout
    {
        const int w = wave;
        float* out = p.part + ((size_t)net * p.slabs + slab) * kPartStride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = 32 * w + tile_row(r, kk);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) out[kPW2 + o * kHid + 32 * jt + i] = w2s[jt][r];
            if (x_on) out[kPW1 + o * p.in_dim + i] = w1s[r];
        }
        float hd4[4] = {hd01.x, hd01.y, hd23.x, hd23.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hd4[j] = half_sum(hd4[j]);
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) sb4[j] += __shfl_xor(sb4[j], off);      // over the wave's 32 rows (lane half 0)
        }
        // the four waves' head-bias sums, added in wave order by wave 0 (the staged-matrix room is free: every wave is past its last read)
        __syncthreads();
        if (lane == 0) reinterpret_cast<f32x4*>(M)[wave] = f32x4{sb4[0], sb4[1], sb4[2], sb4[3]};
        __syncthreads();
        if (wave == 0) {
            const f32x4 s0 = reinterpret_cast<f32x4*>(M)[0], s1 = reinterpret_cast<f32x4*>(M)[1], s2 = reinterpret_cast<f32x4*>(M)[2],
                        s3 = reinterpret_cast<f32x4*>(M)[3];
            sb4[0] = ((s0.x + s1.x) + s2.x) + s3.x; sb4[1] = ((s0.y + s1.y) + s2.y) + s3.y;
            sb4[2] = ((s0.z + s1.z) + s2.z) + s3.z; sb4[3] = ((s0.w + s1.w) + s2.w) + s3.w;
        }
        sb2 = half_sum(sb2); sb1 = half_sum(sb1);
        if (kk == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nh) out[kPWh + j * kHid + 32 * w + i] = hd4[j];
            out[kPB2 + 32 * w + i] = sb2;
            out[kPB1 + 32 * w + i] = sb1;
            if (w == 0 && ah_on) out[kPBh + i] = i == 0 ? sb4[0] : (i == 1 ? sb4[1] : (i == 2 ? sb4[2] : sb4[3]));
        }
    }
#pragma unroll
    for (int k = 0; k < kWaveStats; ++k) {
        float v = st[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) p.wave_stats[(((size_t)net * p.slabs + slab) * 4 + wave) * kWaveStats + k] = v;
    }
    PPO_STAMP(3);
}

// ---------------------------------------------------------------- U3: fixed-order sums -> gradient tensors + statistics
struct ReduceArgs {
    const float* part;
    const float* wave_stats;
    float* g[13];             // PpoParams order
    float* stats;             // [4]
    int32_t slabs, waves, in_dim, act_dim;
    float inv_rows, ent_coef;
};

// 256 threads = 8 slab groups x 32 elements: group s sums slabs s, s + 8, ... (independent loads, all in flight), the
// eight group sums are added in fixed order through LDS.  The last workgroup sums the per-wave statistics instead.
__global__ __launch_bounds__(256) void ppo_reduce_kernel(ReduceArgs p) {
    __shared__ float sm[8][32];
    if (blockIdx.x == gridDim.x - 1) {          // statistics + d log_std: thread = (wave subset j, statistic k)
        const int k = threadIdx.x & 7, j = threadIdx.x >> 3;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        constexpr int kPer = 32;                 // 1 024 waves' statistics (256 workgroups of the one-kernel path): every read in flight at
        if (p.waves <= 32 * kPer) {              // once — this workgroup's eight dependent rounds of reads were the kernel's duration
            float v[kPer];
#pragma unroll
            for (int m = 0; m < kPer; ++m) {
                const int wv = j + 32 * m;
                v[m] = wv < p.waves ? p.wave_stats[(size_t)wv * kWaveStats + k] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < kPer; ++m) {     // the loop's order of additions: whole groups of four into s0..s3, the rest into s0
                const bool whole = j + 128 * (m >> 2) + 96 < p.waves;
                if ((m & 3) == 0) s0 += v[m];
                else if ((m & 3) == 1) { s1 += whole ? v[m] : 0.f; s0 += whole ? 0.f : v[m]; }
                else if ((m & 3) == 2) { s2 += whole ? v[m] : 0.f; s0 += whole ? 0.f : v[m]; }
                else { s3 += whole ? v[m] : 0.f; s0 += whole ? 0.f : v[m]; }
            }
        } else {
            int wv = j;
            for (; wv + 96 < p.waves; wv += 128) {
                s0 += p.wave_stats[(size_t)wv * kWaveStats + k]; s1 += p.wave_stats[(size_t)(wv + 32) * kWaveStats + k];
                s2 += p.wave_stats[(size_t)(wv + 64) * kWaveStats + k]; s3 += p.wave_stats[(size_t)(wv + 96) * kWaveStats + k];
            }
            for (; wv < p.waves; wv += 32) s0 += p.wave_stats[(size_t)wv * kWaveStats + k];
        }
        sm[k][j] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (threadIdx.x < kWaveStats) {
            float t = 0.f;
            for (int q = 0; q < 32; ++q) t += sm[threadIdx.x][q];
            if (threadIdx.x < 4) p.stats[threadIdx.x] = t * p.inv_rows;
            else if ((int)threadIdx.x - 4 < p.act_dim) p.g[6][threadIdx.x - 4] = t - p.ent_coef;   // -ent_coef * d entropy
        }
        return;
    }
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float* dst = nullptr;
    float sum = 0.f;
    if (e < 2 * kPartStride) {
        const int net = e / kPartStride, o = e % kPartStride;
        const int nh = net == 0 ? p.act_dim : 1;
        if (o < kPB1) { if (o < kHid * p.in_dim) dst = p.g[net ? 7 : 0] + o; }
        else if (o < kPW2) dst = p.g[net ? 8 : 1] + (o - kPB1);
        else if (o < kPB2) dst = p.g[net ? 9 : 2] + (o - kPW2);
        else if (o < kPWh) dst = p.g[net ? 10 : 3] + (o - kPB2);
        else if (o < kPBh) { if (o - kPWh < nh * kHid) dst = p.g[net ? 11 : 4] + (o - kPWh); }
        else if (o - kPBh < nh) dst = p.g[net ? 12 : 5] + (o - kPBh);
        if (dst) {
            const float* src = p.part + (size_t)net * p.slabs * kPartStride + o;
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int b = grp + 8 * q;
                v[q] = b < p.slabs ? src[(size_t)b * kPartStride] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += v[q];
            for (int b = grp + 128; b < p.slabs; b += 8) sum += src[(size_t)b * kPartStride];
        }
    }
    sm[grp][el] = sum;
    __syncthreads();
    if (grp == 0 && dst) {
        float t = sm[0][el];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += sm[q][el];
        *dst = t;
    }
}

// ---------------------------------------------------------------- U4: gradient clip + Adam on the flat parameter buffer
// torch.nn.utils.clip_grad_norm_(max_norm) followed by torch.optim.Adam.step() (no weight decay, no amsgrad), for ONE
// flat tensor of a few 10^4 elements, one launch.
struct AdamArgs {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t* step;            // device {step count, launch-internal arrival counter}
    float* grad_norm;         // out (may be null): the norm before clipping
    int64_t n;
    float lr, beta1, beta2, eps, max_norm;
    const int32_t* gate;      // device word or null: non-zero = this launch changes nothing (target_kl's early stop, decided on the device)
};

// Every 1024-thread workgroup recomputes the norm of the whole gradient (a few 10^4 elements out of L2, the same
// summation order in every workgroup) and updates its own 1024 elements; the last workgroup to arrive advances the step.
__global__ __launch_bounds__(1024) void adam_clip_kernel(AdamArgs p) {
    __shared__ float red[16];
    __shared__ float bc[3];
    if (p.gate && *p.gate != 0) return;          // (uniform over the grid: no workgroup takes a ticket)
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int kPer = 40;                    // the policy's 39 049 parameters: every read of the gradient in flight at once (the
    if (p.n <= (int64_t)kPer * 1024) {          // loop below waits for memory ten times in a row: 7 of the kernel's 8.8 us)
        float g[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int64_t e = (int64_t)threadIdx.x + 1024 * q;
            g[q] = e < p.n ? p.grad[e] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q) {        // the loop's order of additions: whole groups of four into ss[0..3], the rest into ss[0]
            const bool whole = (int64_t)threadIdx.x + 4096 * (q >> 2) + 3072 < p.n;
            const float sq = g[q] * g[q];
            if ((q & 3) == 0) ss[0] += sq;      // (a zero for an element past the end)
            else { ss[q & 3] += whole ? sq : 0.f; ss[0] += whole ? 0.f : sq; }
        }
    } else {
        int64_t i = threadIdx.x;
        for (; i + 3 * 1024 < p.n; i += 4 * 1024) {
            const float g0 = p.grad[i], g1 = p.grad[i + 1024], g2 = p.grad[i + 2048], g3 = p.grad[i + 3072];
            ss[0] += g0 * g0; ss[1] += g1 * g1; ss[2] += g2 * g2; ss[3] += g3 * g3;
        }
        for (; i < p.n; i += 1024) {
            const float g = p.grad[i];
            ss[0] += g * g;
        }
    }
    float s = (ss[0] + ss[1]) + (ss[2] + ss[3]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    const int64_t step = p.step[0] + 1;
    // the two bias corrections do not wait for the norm: one lane of two other waves each takes a powf (a microsecond of one lane's
    // dependent instructions apiece) while wave 0 sums
    if (threadIdx.x == 64) bc[1] = p.lr / (1.0f - powf(p.beta1, (float)step));
    if (threadIdx.x == 128) bc[2] = 1.0f / sqrtf(1.0f - powf(p.beta2, (float)step));
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        const float norm = sqrtf(t);
        if (p.grad_norm && blockIdx.x == 0) *p.grad_norm = norm;
        bc[0] = p.max_norm > 0.f ? fminf(p.max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    }
    // this thread's element: asked for before the barrier that publishes the clip factor
    const int64_t k = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    float g_k = 0.f, m_k = 0.f, v_k = 0.f, p_k = 0.f;
    if (k < p.n) { g_k = p.grad[k]; m_k = p.exp_avg[k]; v_k = p.exp_avg_sq[k]; p_k = p.param[k]; }
    __syncthreads();
    const float coef = bc[0], step_size = bc[1], inv_bc2 = bc[2];
    if (k < p.n) {
        const float g = g_k * coef;
        const float m = p.beta1 * m_k + (1.0f - p.beta1) * g;
        const float v = p.beta2 * v_k + (1.0f - p.beta2) * g * g;
        p.exp_avg[k] = m;
        p.exp_avg_sq[k] = v;
        p.param[k] = p_k - step_size * m / (sqrtf(v) * inv_bc2 + p.eps);
    }
    // every workgroup has read step[0] above; the one that arrives last publishes the new count
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        unsigned long long* arrive = reinterpret_cast<unsigned long long*>(p.step + 1);
        if (atomicAdd(arrive, 1ull) == (unsigned long long)gridDim.x - 1) {
            *arrive = 0ull;
            p.step[0] = step;
        }
    }
}

hipError_t launch_adam_clip(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                            float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_norm,
                            const int32_t* gate, hipStream_t s) {
    AdamArgs a{param, grad, exp_avg, exp_avg_sq, step, grad_norm, n, lr, beta1, beta2, eps, max_norm, gate};
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// SB3 PPO.train's target_kl rule for an epoch, on the device: gate = {stopped, epochs that counted}.  Behind an epoch's minibatches:
// an epoch that started unstopped counts, and stops the epochs behind it when the mean of its minibatches' approx_kl is above the limit.
__global__ void ppo_kl_gate_kernel(const float* __restrict__ stats, int n_minibatches, float limit, int32_t* gate) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || gate[0] != 0) return;
    float kl = 0.f;
    for (int k = 0; k < n_minibatches; ++k) kl += stats[4 * k + 2];
    kl /= (float)n_minibatches;
    gate[1] += 1;
    if (kl > limit) gate[0] = 1;
}

hipError_t launch_ppo_kl_gate(const float* stats, int n_minibatches, float limit, int32_t* gate, hipStream_t s) {
    hipLaunchKernelGGL(ppo_kl_gate_kernel, dim3(1), dim3(64), 0, s, stats, n_minibatches, limit, gate);
    return hipGetLastError();
}

// ---------------------------------------------------------------- U5: the epoch's row order
// SB3 draws np.random.permutation(rows) per epoch and cuts it into minibatches.  On the device a permutation by sorting random keys
// (torch.randperm) is a radix sort of 2 x 10^6 keys — 170 us per epoch, 3.5 % of a training iteration at 65 536 drones.  This is a
// KEYED BIJECTION instead: element i of the order is E(i), E an 8-round generalised Feistel network on the ceil(log2 n) bits of the
// index (round function: a 32-bit avalanche mix of the other half and the round key), with cycle walking for n that is not a power
// of two (E is applied until the value falls below n: a bijection of [0, 2^b) restricted this way is a bijection of [0, n)).  One
// thread per element, no memory read, every (seed, counter) another permutation.  It is a pseudo-random permutation, not a uniform
// draw from all n! of them (neither is a 64-bit-seeded shuffle); what minibatch SGD needs from it — every row once per epoch, no
// structure shared between epochs or with the buffer's (time, drone) layout — tests/test_ppo_grad_gpu.py checks.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {          // murmur3's finaliser
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void shuffle_indices_kernel(int64_t* __restrict__ out, int64_t n, int bits, uint32_t k0, uint32_t k1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int lb = bits >> 1, rb = bits - lb;                    // the halves' widths (rb = lb or lb + 1)
    const uint32_t lm = (1u << lb) - 1u, rm = (1u << rb) - 1u;
    uint64_t x = (uint64_t)i;
    do {
        uint32_t l = (uint32_t)(x >> rb) & lm, r = (uint32_t)x & rm;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            l ^= mix32(r ^ (k0 + 0x9E3779B9u * (uint32_t)q)) & lm;
            r ^= mix32(l ^ (k1 + 0x9E3779B9u * (uint32_t)(q + 1))) & rm;
        }
        x = ((uint64_t)l << rb) | r;
    } while ((int64_t)x >= n);
    out[i] = (int64_t)x;
}

hipError_t launch_shuffle_indices(int64_t* out, int64_t n, uint64_t seed, uint64_t counter, hipStream_t s) {
    int bits = 2;
    while (((int64_t)1 << bits) < n) ++bits;
    const uint64_t k = (seed ^ (counter * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull + counter;
    hipLaunchKernelGGL(shuffle_indices_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, bits, (uint32_t)k,
                       (uint32_t)(k >> 32));
    return hipGetLastError();
}

// ---------------------------------------------------------------- launchers
static int ppo_slabs(int64_t rows, int* rows_per_slab) {
    int64_t rps = (rows + 127) / 128;
    rps = (rps + 31) / 32 * 32;
    if (rps < 32) rps = 32;
    *rows_per_slab = (int)rps;
    return (int)((rows + rps - 1) / rps);
}

static int ppo_blocks(int64_t rows, int cus) {
    constexpr int chunk = kFwdBwdThreads / 2;
    const int64_t b = (rows + chunk - 1) / chunk;
    return (int)(b < cus ? b : cus);
}

// the one-kernel path: one workgroup per net and slab, slabs of whole 128-row chunks, cus / 2 slabs per net at full size
static int ppo_fused_slabs(int64_t rows, int cus, int* rows_per_slab) {
    const int64_t want = cus / 2 > 0 ? cus / 2 : 1;
    int64_t rps = (rows + want - 1) / want;
    rps = (rps + 127) / 128 * 128;
    if (rps < 128) rps = 128;
    *rows_per_slab = (int)rps;
    return (int)((rows + rps - 1) / rps);
}

static int64_t ppo_stat_waves(int64_t rows, int cus) {
    int rps;
    const int64_t two = (int64_t)ppo_blocks(rows, cus) * (kFwdBwdThreads / 64), one = 2 * (int64_t)ppo_fused_slabs(rows, cus, &rps) * 4;
    return two > one ? two : one;
}

int64_t ppo_workspace_floats(int64_t rows, int cus) {
    int rps, frps;
    const int two = ppo_slabs(rows, &rps), one = ppo_fused_slabs(rows, cus, &frps);
    const int slabs = two > one ? two : one;                     // the partial blocks of whichever path writes more (cus > 256: the one-kernel path)
    return (int64_t)kTImage + 8 * rows * kHid + rows * 24 + rows * 4 + rows + ppo_stat_waves(rows, cus) * kWaveStats +
           kPieceFloats + 2 * (int64_t)slabs * kPartStride + 64;
}

template <typename ST>
static hipError_t launch_ppo_grad_t(const float* const* w, float* const* grads, const int64_t* index, const float* obs,
                                    const float* actions, const float* old_logp, const float* adv, const float* ret,
                                    int64_t rows, float clip_range, float vf_coef, float ent_coef, float* stats, float* ws,
                                    int in_dim, int act_dim, int cus, int path, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    constexpr size_t lds_bytes = (size_t)(kTNet + kHeadFloats) * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)ppo_fwd_bwd_kernel<kFwdBwdThreads, ST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ppo_wgrad_kernel<ST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kWgradXchFloats * sizeof(float)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ppo_wgrad_split_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kWgradXchFloats * sizeof(float)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ppo_wgrad_split_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kWgradXchFloats * sizeof(float)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ppo_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kFusedFloats * sizeof(float)));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int rps;
    const int slabs = ppo_slabs(rows, &rps);
    const int blocks = ppo_blocks(rows, cus);
    // workspace carve-up (floats; every piece 16-byte aligned); the matrices take 8 rows x 128 floats, or half of that
    float* image = ws;
    float* mats = image + kTImage;
    float* xg = mats + 8 * rows * kHid;
    float* dmu = xg + rows * 24;
    float* dv = dmu + rows * 4;
    float* wave_stats = dv + (rows + 3) / 4 * 4;
    u32x4* pieces = reinterpret_cast<u32x4*>(wave_stats + (size_t)ppo_stat_waves(rows, cus) * kWaveStats);
    float* part = reinterpret_cast<float*>(pieces) + kPieceFloats;

    PpoParams pp{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], in_dim, act_dim};
    hipLaunchKernelGGL(ppo_pack_kernel, dim3(kPackBlocks + 2 * kPieceNet / (3 * 256)), dim3(256), 0, s, pp, image, pieces);
    if (path == FPV_PPO_ONE_KERNEL && sizeof(ST) == 4) {
        int frps;
        const int fslabs = ppo_fused_slabs(rows, cus, &frps);
        FusedArgs f{image, pieces, index, obs, actions, old_logp, adv, ret, part, wave_stats,
                    rows, clip_range, vf_coef, 1.0f / (float)rows, frps, fslabs, in_dim, act_dim};
        hipLaunchKernelGGL(ppo_fused_kernel, dim3(fslabs, 2), dim3(256), kFusedFloats * sizeof(float), s, f);
        ReduceArgs r{part, wave_stats, {}, stats, fslabs, 2 * fslabs * 4, in_dim, act_dim, 1.0f / (float)rows, ent_coef};
        for (int k = 0; k < 13; ++k) r.g[k] = grads[k];
        hipLaunchKernelGGL(ppo_reduce_kernel, dim3((2 * kPartStride + 31) / 32 + 1), dim3(256), 0, s, r);
        return hipGetLastError();
    }
    PpoArgs a{image, index, obs, actions, old_logp, adv, ret, mats, xg, dmu, dv, wave_stats,
              rows, clip_range, vf_coef, 1.0f / (float)rows, in_dim, act_dim};
    hipLaunchKernelGGL((ppo_fwd_bwd_kernel<kFwdBwdThreads, ST>), dim3(blocks), dim3(kFwdBwdThreads), lds_bytes, s, a);
    WgradArgs g{mats, xg, dmu, dv, part, rows, rps, slabs, in_dim, act_dim};
    if (path == FPV_PPO_TWO_KERNELS_BF16X6 && sizeof(ST) == 4)
        hipLaunchKernelGGL(ppo_wgrad_split_kernel<6>, dim3(slabs, 2), dim3(512), kWgradXchFloats * sizeof(float), s, g);
    else if (path == FPV_PPO_TWO_KERNELS_BF16X3 && sizeof(ST) == 4)
        hipLaunchKernelGGL(ppo_wgrad_split_kernel<3>, dim3(slabs, 2), dim3(512), kWgradXchFloats * sizeof(float), s, g);
    else
        hipLaunchKernelGGL(ppo_wgrad_kernel<ST>, dim3(slabs, 2), dim3(512), kWgradXchFloats * sizeof(float), s, g);
    ReduceArgs r{part, wave_stats, {}, stats, slabs, blocks * (kFwdBwdThreads / 64), in_dim, act_dim, 1.0f / (float)rows, ent_coef};
    for (int k = 0; k < 13; ++k) r.g[k] = grads[k];
    hipLaunchKernelGGL(ppo_reduce_kernel, dim3((2 * kPartStride + 31) / 32 + 1), dim3(256), 0, s, r);
    return hipGetLastError();
}

hipError_t launch_ppo_grad(const float* const* w, float* const* grads, const int64_t* index, const float* obs,
                           const float* actions, const float* old_logp, const float* adv, const float* ret, int64_t rows,
                           float clip_range, float vf_coef, float ent_coef, float* stats, float* ws, int in_dim, int act_dim,
                           int cus, int storage, int path, hipStream_t s) {
    if (storage == 1)
        return launch_ppo_grad_t<bf16_bits>(w, grads, index, obs, actions, old_logp, adv, ret, rows, clip_range, vf_coef,
                                            ent_coef, stats, ws, in_dim, act_dim, cus, FPV_PPO_TWO_KERNELS, s);
    return launch_ppo_grad_t<float>(w, grads, index, obs, actions, old_logp, adv, ret, rows, clip_range, vf_coef, ent_coef,
                                    stats, ws, in_dim, act_dim, cus, path, s);
}

}  // namespace fpv
