// fpv_policy_w16.hip — the rollout's actor-critic forward on 16-wave workgroups (gfx950, exact-fp32 MFMA).
//
// Same operator as fpv_policy.hip (SB3 MlpPolicy.forward + VecNormalize.normalize_obs + Box clip per rollout step,
// simulation/train_hover.py:43-58, net_arch=[128,128]), re-cut after the phase stamps of round 2
// (profiles/r02_policy_phase_stamps.md):
//   * the SIMD's matrix pipe is arbitrated oldest-wave-first.  With two 32-drone waves per SIMD (fpv_policy.hip) the
//     older wave runs its layers at full rate, the younger one only fills its gaps and then finishes the launch alone —
//     and every tanh / head / sampling phase of a lone wave leaves the pipe idle (23 % of a 65 536-drone launch).
//   * ~10 k cycles of the launch pass before the first product: 160 KB of weight image through registers into LDS,
//     then the observation load.
// Here a wave takes 16 drones on v_mfma_f32_16x16x4_f32 (same 64 flop / clk / SIMD): a layer's 128 outputs are eight
// 4-register tiles, a wave needs ~100 registers, so FOUR waves share a SIMD (1 024-thread workgroups, still 256 drones
// and one LDS weight image per workgroup) and the hardware's own wave arbitration keeps the matrix pipe fed while one
// wave is in its vector phases.  As in fpv_policy.hip everything is computed transposed (H^T = W . X^T): register r of
// an output tile holds hidden unit 16 t + 4 (lane >> 4) + r, so a tile is the next layer's B operand as it stands; the
// k order this induces is baked into the weight image (policy16_pack_kernel).
// The image comes in by LDS-DMA in consumption order {pi layer 1 | rest of pi + heads | V layer 1 | rest of V}; a wave
// waits (counted vmcnt + raw s_barrier) only for what its next phase reads, so the first product issues once 11 KB
// have landed.
#include "fpv_mlp.h"

namespace fpv {

namespace w16 {
constexpr int kW1 = 0, kB1 = 2560, kW2 = 2688, kB2 = 19072, kNet = 19200;   // floats, per net
constexpr int kHead = 2 * kNet;                                             // 38 400: Wa_p[512], Wv_p[128], scalars
constexpr int kImage = kHead + 660;                                         // 39 060 floats = 156 240 B
constexpr int kLdsFloats = kImage + 40;                                     // + normaliser mean[20], inv[20]
}  // namespace w16

struct Policy16Params {
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2, *wa, *ba, *log_std;
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2, *wv, *bv;
    int in_dim, act_dim;   // in_dim % 4 == 0, <= 20; act_dim <= 4
};

// nn.Linear layouts: w1 [128][in], w2 [128][128], wa [act][128], wv [1][128] (row = output).
__global__ void policy16_pack_kernel(Policy16Params p, float* __restrict__ img) {
    using namespace w16;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kImage) return;
    float v = 0.0f;
    if (e < kHead) {
        const int net = e / kNet, o = e % kNet;
        const float* w1 = net ? p.vf_w1 : p.pi_w1;
        const float* b1 = net ? p.vf_b1 : p.pi_b1;
        const float* w2 = net ? p.vf_w2 : p.pi_w2;
        const float* b2 = net ? p.vf_b2 : p.pi_b2;
        if (o < kB1) {                          // W1p[s][t][lane] = W1[16 t + (lane & 15)][4 s + (lane >> 4)]
            const int lane = o % 64, t = (o / 64) % 8, s = o / 512;
            const int k = 4 * s + (lane >> 4);
            v = k < p.in_dim ? w1[(16 * t + (lane & 15)) * p.in_dim + k] : 0.0f;
        } else if (o < kW2) {                   // b1p[t][g][r] = b1[16 t + 4 g + r]
            const int q = o - kB1;
            v = b1[16 * (q / 16) + (q % 16)];
        } else if (o < kB2) {                   // W2p[t][u][lane][r] = W2[16 t + (lane & 15)][16 u + 4 (lane >> 4) + r]
            const int q = o - kW2, r = q % 4, lane = (q / 4) % 64, u = (q / 256) % 8, t = q / 2048;
            v = w2[(16 * t + (lane & 15)) * 128 + 16 * u + 4 * (lane >> 4) + r];
        } else {
            const int q = o - kB2;
            v = b2[16 * (q / 16) + (q % 16)];
        }
    } else {
        const int q = e - kHead;
        if (q < 512) {                          // Wa_p[g][t][r][o] = Wa[o][16 t + 4 g + r]
            const int o = q % 4, r = (q / 4) % 4, t = (q / 16) % 8, g = q / 128;
            v = o < p.act_dim ? p.wa[o * 128 + 16 * t + 4 * g + r] : 0.0f;
        } else if (q < 640) {                   // Wv_p[g][t][r]
            const int z = q - 512, r = z % 4, t = (z / 4) % 8, g = z / 32;
            v = p.wv[16 * t + 4 * g + r];
        } else if (q < 644) v = q - 640 < p.act_dim ? p.ba[q - 640] : 0.0f;
        else if (q == 644) v = p.bv[0];
        else if (q >= 648 && q < 652) v = q - 648 < p.act_dim ? p.log_std[q - 648] : 0.0f;
        else if (q >= 652 && q < 656) v = q - 652 < p.act_dim ? expf(p.log_std[q - 652]) : 0.0f;
        else if (q == 656) {                    // -sum log_std - act_dim * 0.5 ln 2 pi
            for (int j = 0; j < p.act_dim; ++j) v -= p.log_std[j] + 0.5f * 1.8378770664093453f;
        }
    }
    img[e] = v;
}

struct Policy16Args {
    const float* image;      // packed weights (w16::kImage floats)
    const float* obs;        // [n][in_dim] raw observations
    const double* rms;       // mean[in], var[in], count — or null: observations are used as they are
    const uint64_t* tick;    // device counter keying the in-kernel RNG (the env's per-tile tick), or null
    float* obs_n;            // [n][in_dim] normalised observation (rollout buffer), may be null
    float* action;           // [n][act] sampled action (unclipped), may be null
    float* clipped;          // [n][act] clip(action, -1, 1) — what the env receives, may be null
    float* value;            // [n]
    float* logp;             // [n], may be null
    float* eps_out;          // [n][act], may be null
    int64_t n;
    uint64_t seed, id_offset;
    float clip_obs, eps_norm;
    int32_t in_dim, act_dim;
};

typedef __attribute__((address_space(3))) const float lfloat;
typedef __attribute__((address_space(3))) const f32x4 lf32x4;

// One 1-KiB piece of the image by LDS-DMA.  Inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write
// and drains vmcnt(0) in front of the next ds_read, which would serialise the staging with the first layer; an asm
// statement is outside its bookkeeping (cdna_hip_programming.md §5.7), the counted waits below are ours.  M0 (the
// wave-uniform LDS destination) is written and restored inside the statement.
__device__ __forceinline__ void stage_piece16(const float* __restrict__ image, unsigned lds_base, int piece, int lane) {
    const int byte = piece * 1024 + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)piece * 1024u);
    const float* src = image + piece * 256 + lane * 4;
    unsigned keep;
    if (byte < w16::kImage * 4)                 // the last piece is partial: exec-masked lanes write nothing
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

#define FPV_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// Fence for hipcc, no instruction: LDS reads do not move across it, neither in the IR passes (memory clobber) nor in the
// machine scheduler.  Without it the unrolled layers' fragment reads are pulled to the front of their phase — 64 x 16 B
// per lane for a layer 2 — and the 128-register budget of a four-waves-per-SIMD kernel spills.
#define FPV_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// sum over the four 16-lane groups of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48): the same value in all of them
__device__ __forceinline__ float group_sum(float v) {
    v += __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(v), 0x401F));   // bit mode: lane ^ 16
    return half_sum(v);                                                               // v_permlane32_swap: lane ^ 32
}

// ACT4: act_dim == 4 (hover) — the action rows leave as 16-byte stores.
template <bool ACT4>
__global__ __launch_bounds__(1024) void policy_forward_w16_kernel(Policy16Args p) {
    using namespace w16;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_f;
    lfloat* ldsc = (lfloat*)(size_t)lds0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;          // drone of the wave's 16, k-group
    const float clip = p.rms ? p.clip_obs : 3.0e38f;
    constexpr int kChunk = 256;
    FPV_STAMP(0);

    const int64_t rows = p.n;
    const int64_t base0 = (int64_t)blockIdx.x * kChunk;
    if (base0 >= rows) return;                       // uniform: nothing to stage for
    const uint64_t tick = p.tick ? p.tick[0] : 0ull;

    // ---- the wave's first vector-memory operations: normaliser statistics (wave 0), the first chunk's observation
    if ((int)threadIdx.x < p.in_dim) {
        lds_f[kImage + threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
        lds_f[kImage + 20 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[p.in_dim + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
    }
    // asm loads, so that hipcc's own wait for them does not become vmcnt(0) behind the LDS-DMA it cannot see
    f32x4 xq[5];
    {
        const int64_t d = base0 + wave * 16 + j;
        const float* row = p.obs + (d < rows ? d : 0) * p.in_dim;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const float* a = row + (4 * s < p.in_dim ? 4 * s : 0);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq[s]) : "v"(a) : "memory");
        }
    }
    // ---- weight image by LDS-DMA, 11 one-KiB pieces per wave, in consumption order (a slot past its group's end
    // re-loads the group's last piece: same bytes to the same place)
    stage_piece16(p.image, lds0, wave < 11 ? wave : 10, lane);                                  // G1: pi layer 1 (pieces 0..10)
#pragma unroll
    for (int it = 0; it < 5; ++it) {                                                            // G2: rest of pi (11..74), heads (150..152)
        const int q = it * 16 + wave;
        stage_piece16(p.image, lds0, q < 64 ? 11 + q : (q < 67 ? 150 + (q - 64) : 152), lane);
    }
    stage_piece16(p.image, lds0, 75 + (wave < 11 ? wave : 10), lane);                           // G3: V layer 1 (75..85)
#pragma unroll
    for (int it = 0; it < 4; ++it) stage_piece16(p.image, lds0, 86 + it * 16 + wave, lane);     // G4: rest of V (86..149)
    FPV_STAMP(1);
    // observation + G1 landed (G2, G3, G4: 10 pieces still in flight); lgkmcnt(0): the normaliser's LDS writes
    asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();

    bool first = true;
    for (int64_t base = base0; base < rows; base += (int64_t)gridDim.x * kChunk) {
        const int64_t d = base + wave * 16 + j;     // this lane's drone (the four 16-lane groups share it)
        const bool live = d < rows;
        if (!first) {                               // (the first chunk's rows came in ahead of the weight image)
            const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + (live ? d : 0) * p.in_dim);
#pragma unroll
            for (int s = 0; s < 5; ++s) xq[s] = row[4 * s < p.in_dim ? s : 0];
            // the wait for these loads belongs inside this branch: at the join hipcc would place a vmcnt(0) that the
            // first chunk pays with its LDS-DMA still in flight
            asm volatile("" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]));
        }
        FPV_STAMP(2);
        // ---- normalise the row (every lane its drone's whole row: the chunk s goes out from lane group s & 3), then
        // keep component g of every chunk: this lane's B operands, k = 4 s + g
        float x[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const bool have = live && 4 * s < p.in_dim;
            const f32x4 mv = *(lf32x4*)(ldsc + kImage + 4 * s), iv = *(lf32x4*)(ldsc + kImage + 20 + 4 * s);
            f32x4 v;
            v.x = have ? fminf(fmaxf((xq[s].x - mv.x) * iv.x, -clip), clip) : 0.0f;
            v.y = have ? fminf(fmaxf((xq[s].y - mv.y) * iv.y, -clip), clip) : 0.0f;
            v.z = have ? fminf(fmaxf((xq[s].z - mv.z) * iv.z, -clip), clip) : 0.0f;
            v.w = have ? fminf(fmaxf((xq[s].w - mv.w) * iv.w, -clip), clip) : 0.0f;
            if (p.obs_n && have && g == (s & 3)) reinterpret_cast<f32x4*>(p.obs_n + d * p.in_dim)[s] = v;
            x[s] = g == 0 ? v.x : (g == 1 ? v.y : (g == 2 ? v.z : v.w));
            xq[s] = f32x4{0.f, 0.f, 0.f, 0.f};      // dead from here on (else 20 registers stay live around the loop)
            asm volatile("" : "+v"(x[s]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        FPV_STAMP(3);

        float head[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // 4 action means + value (partial sums over this lane's 32 hidden units)
        f32x4 pend0 = {0.f, 0.f, 0.f, 0.f}, pend1 = {0.f, 0.f, 0.f, 0.f};   // the last layer-2 tile pair of the previous net
        lfloat* wa = ldsc + kHead + g * 128;        // Wa_p[g][t][r][o]
        lfloat* wv = ldsc + kHead + 512 + g * 32;   // Wv_p[g][t][r]
        // head terms of element r of finished layer-2 tile t: the same instructions for both nets (no branch between
        // the products) — s_pi / s_v are 1 / 0 for pi and 0 / 1 for V, so the other net's term adds an exact zero.
        // Results pass through an asm so that hipcc keeps the arithmetic where it is written, under the products.
        auto head_apply = [&](float v, f32x4 wh, float wvr, float s_pi, float s_v) {
            const float th = fast_tanh(v), tp = th * s_pi, tv = th * s_v;
            head[0] += wh.x * tp; head[1] += wh.y * tp; head[2] += wh.z * tp; head[3] += wh.w * tp;
            head[4] += wvr * tv;
        };
        // (the two nets as a real loop: one base register + immediate offsets address a whole net's image)
#pragma unroll 1
        for (int net = 0; net < 2; ++net) {
            lfloat* nb = ldsc + net * kNet + lane;  // + lane: W1p is read one float per lane
            lfloat* nb4 = ldsc + net * kNet + lane * 4;
            lfloat* nbg = ldsc + net * kNet + g * 4;
            const float s_pi = net == 0 ? 1.0f : 0.0f, s_v = 1.0f - s_pi;
            // ---- layer 1: in (<= 20) -> 128, one pair of 16-row output tiles at a time: five k-steps alternate between
            // the pair's accumulators (a product never waits for the one just issued); the vector work of the pair
            // before — tanh, or for V's first pair the tanh + action-head terms of pi's last layer-2 pair — sits between
            // the k-steps; the next pair's fragments are requested one pair ahead
            f32x4 a[8];
            float wf0[5], wf1[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) { wf0[s] = nb[kW1 + (s * 8 + 0) * 64]; wf1[s] = nb[kW1 + (s * 8 + 1) * 64]; }
#pragma unroll
            for (int P = 0; P < 4; ++P) {
                a[2 * P] = *(lf32x4*)(nbg + kB1 + (2 * P) * 16);
                a[2 * P + 1] = *(lf32x4*)(nbg + kB1 + (2 * P + 1) * 16);
                float nf0[5], nf1[5];
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    nf0[s] = P < 3 ? nb[kW1 + (s * 8 + 2 * P + 2) * 64] : 0.0f;
                    nf1[s] = P < 3 ? nb[kW1 + (s * 8 + 2 * P + 3) * 64] : 0.0f;
                }
                // (P == 0) head weights of the pending pair's elements, requested one slice ahead: slices take elements
                // {0,1} {2,3} {4,5} {6} {7} of the pair (tiles 6, 7)
                f32x4 hw0 = {0.f, 0.f, 0.f, 0.f}, hw1 = {0.f, 0.f, 0.f, 0.f};
                if (P == 0) { hw0 = *(lf32x4*)(wa + (6 * 4 + 0) * 4); hw1 = *(lf32x4*)(wa + (6 * 4 + 1) * 4); }
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const int e0 = s < 3 ? 2 * s : 3 + s, e1 = s < 3 ? 2 * s + 2 : 4 + s;        // this slice's elements
                    const int f0 = s + 1 < 3 ? 2 * (s + 1) : 3 + (s + 1);                        // the next slice's first
                    f32x4 nh0 = hw0, nh1 = hw1;
                    if (P == 0 && s < 4) {
                        nh0 = *(lf32x4*)(wa + ((6 + (f0 >> 2)) * 4 + (f0 & 3)) * 4);
                        if (s + 1 < 3) nh1 = *(lf32x4*)(wa + ((6 + ((f0 + 1) >> 2)) * 4 + ((f0 + 1) & 3)) * 4);
                    }
                    a[2 * P] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[s], x[s], a[2 * P], 0, 0, 0);
                    a[2 * P + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[s], x[s], a[2 * P + 1], 0, 0, 0);
                    // vector slice s of 5: elements e0..e1 of the 8 pending ones
#pragma unroll
                    for (int e = e0; e < e1; ++e) {
                        if (P == 0) head_apply(e < 4 ? pend0[e & 3] : pend1[e & 3], e == e0 ? hw0 : hw1, 0.0f, s_v, 0.0f);
                        else a[2 * P - 2 + (e >> 2)][e & 3] = fast_tanh(a[2 * P - 2 + (e >> 2)][e & 3]);
                    }
                    if (P == 0) asm volatile("" : "+v"(head[0]), "+v"(head[1]), "+v"(head[2]), "+v"(head[3]) :: "memory");
                    else asm volatile("" : "+v"(a[2 * P - 2]), "+v"(a[2 * P - 1]) :: "memory");
                    hw0 = nh0; hw1 = nh1;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int s = 0; s < 5; ++s) { wf0[s] = nf0[s]; wf1[s] = nf1[s]; }
            }
            if (net == 0) FPV_STAMP(4); else FPV_STAMP(7);
            if (first) {
                if (net == 0) FPV_WAIT_VM(5);       // G2 landed
                else FPV_WAIT_VM(0);                // G4 landed
                __builtin_amdgcn_s_barrier();
            }
            // ---- layer 2: 128 -> 128, one pair of output tiles at a time; the layer-1 tiles are the B operands as they
            // stand.  Per k-tile u: the next fragments are requested, eight products alternate between the pair's
            // accumulators, then one element of pending vector work: during the first pair the tanh of layer 1's last
            // pair (read from u = 6 on), afterwards tanh + head terms of the pair before
            f32x4 w0 = *(lf32x4*)(nb4 + kW2 + (0 * 8 + 0) * 256), w1 = *(lf32x4*)(nb4 + kW2 + (1 * 8 + 0) * 256);
            f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};           // the pair before
            f32x4 hw = {0.f, 0.f, 0.f, 0.f};                                       // head weights of the pending element,
            float hv = 0.0f;                                                       // requested one k-tile ahead
#pragma unroll
            for (int Q = 0; Q < 4; ++Q) {
                f32x4 c0 = *(lf32x4*)(nbg + kB2 + (2 * Q) * 16), c1 = *(lf32x4*)(nbg + kB2 + (2 * Q + 1) * 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int nq = u < 7 ? Q : Q + 1, nu = u < 7 ? u + 1 : 0;      // next (pair, k-tile)
                    f32x4 n0 = w0, n1 = w1;
                    if (nq < 4) {
                        n0 = *(lf32x4*)(nb4 + kW2 + ((2 * nq) * 8 + nu) * 256);
                        n1 = *(lf32x4*)(nb4 + kW2 + ((2 * nq + 1) * 8 + nu) * 256);
                    }
                    f32x4 nhw = hw;
                    float nhv = hv;
                    if (nq >= 1 && nq < 4) {        // the element the next k-tile finishes: nu of pair nq - 1
                        nhw = *(lf32x4*)(wa + ((2 * nq - 2 + (nu >> 2)) * 4 + (nu & 3)) * 4);
                        nhv = wv[(2 * nq - 2 + (nu >> 2)) * 4 + (nu & 3)];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], a[u][r], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], a[u][r], c1, 0, 0, 0);
                    }
                    if (Q == 0) {
                        if (u < 4) {
                            a[6 + (u >> 1)][2 * (u & 1)] = fast_tanh(a[6 + (u >> 1)][2 * (u & 1)]);
                            a[6 + (u >> 1)][2 * (u & 1) + 1] = fast_tanh(a[6 + (u >> 1)][2 * (u & 1) + 1]);
                            asm volatile("" : "+v"(a[6]), "+v"(a[7]) :: "memory");
                        }
                    } else {
                        head_apply(u < 4 ? q0[u & 3] : q1[u & 3], hw, hv, s_pi, s_v);
                        asm volatile("" : "+v"(head[0]), "+v"(head[1]), "+v"(head[2]), "+v"(head[3]), "+v"(head[4]) :: "memory");
                    }
                    w0 = n0; w1 = n1; hw = nhw; hv = nhv;
                    __builtin_amdgcn_sched_barrier(0);
                }
                q0 = c0; q1 = c1;
            }
            if (net == 0) {
                pend0 = q0; pend1 = q1;             // finished under V's first layer-1 pair
                FPV_STAMP(5);
                if (first) {
                    FPV_WAIT_VM(4);                 // G3 landed
                    __builtin_amdgcn_s_barrier();
                }
                FPV_STAMP(6);
            } else {
                FPV_STAMP(8);
                // ---- the exposed tail: tanh + value-head terms of V's last pair
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    head_apply(e < 4 ? q0[e & 3] : q1[e & 3], f32x4{0.f, 0.f, 0.f, 0.f}, wv[((e < 4 ? 6 : 7)) * 4 + (e & 3)], 0.0f, 1.0f);
            }
        }
        first = false;
        // ---- combine the four lane groups, add biases, sample (Philox4x32-10 keyed like the env noise, stream tag
        // 'POL1', Box-Muller), log-prob, clip, store — one output array per lane group
        lfloat* hs = ldsc + kHead + 640;            // ba[4], bv, -, -, -, log_std[4], std[4], logp const
        const float mu[4] = {group_sum(head[0]) + hs[0], group_sum(head[1]) + hs[1], group_sum(head[2]) + hs[2],
                             group_sum(head[3]) + hs[3]};
        const float val = group_sum(head[4]) + hs[4];
        uint32_t xr[4];
        philox4x32_10((uint32_t)(p.id_offset + (uint64_t)d), (uint32_t)tick, (uint32_t)(tick >> 32), 0x504F4C31u,
                      (uint32_t)p.seed, (uint32_t)(p.seed >> 32), xr);
        constexpr float k24 = 5.9604644775390625e-8f, m2ln2 = -1.3862943611198906f;
        const float u0 = ((float)(xr[0] >> 8) + 0.5f) * k24, u1 = ((float)(xr[1] >> 8) + 0.5f) * k24;
        const float u2 = ((float)(xr[2] >> 8) + 0.5f) * k24, u3 = ((float)(xr[3] >> 8) + 0.5f) * k24;
        const float ra = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u0)), rb = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u2));
        const float eps[4] = {ra * __builtin_amdgcn_cosf(u1), ra * __builtin_amdgcn_sinf(u1), rb * __builtin_amdgcn_cosf(u3),
                              rb * __builtin_amdgcn_sinf(u3)};
        float act[4], lp = hs[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            act[q] = mu[q] + hs[12 + q] * eps[q];
            if (q < p.act_dim) lp -= 0.5f * eps[q] * eps[q];
        }
        if (live) {
            if (g == 0) {
                if (p.action) {
                    if constexpr (ACT4) reinterpret_cast<f32x4*>(p.action)[d] = f32x4{act[0], act[1], act[2], act[3]};
                    else for (int q = 0; q < p.act_dim; ++q) p.action[d * p.act_dim + q] = act[q];
                }
            } else if (g == 1) {
                if (p.clipped) {
                    if constexpr (ACT4)
                        reinterpret_cast<f32x4*>(p.clipped)[d] = f32x4{fminf(fmaxf(act[0], -1.f), 1.f), fminf(fmaxf(act[1], -1.f), 1.f),
                                                                       fminf(fmaxf(act[2], -1.f), 1.f), fminf(fmaxf(act[3], -1.f), 1.f)};
                    else for (int q = 0; q < p.act_dim; ++q) p.clipped[d * p.act_dim + q] = fminf(fmaxf(act[q], -1.f), 1.f);
                }
            } else if (g == 2) {
                p.value[d] = val;
                if (p.logp) p.logp[d] = lp;
            } else if (p.eps_out) {
                if constexpr (ACT4) reinterpret_cast<f32x4*>(p.eps_out)[d] = f32x4{eps[0], eps[1], eps[2], eps[3]};
                else for (int q = 0; q < p.act_dim; ++q) p.eps_out[d * p.act_dim + q] = eps[q];
            }
        }
        FPV_STAMP(9);
    }
}

// ---------------------------------------------------------------- launchers
int policy16_image_floats() { return w16::kImage; }

hipError_t launch_policy16_pack(const float* const* w /* 13 device pointers, fpv_policy_load's order */, float* image,
                                int in_dim, int act_dim, hipStream_t s) {
    Policy16Params p{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], in_dim, act_dim};
    hipLaunchKernelGGL(policy16_pack_kernel, dim3((w16::kImage + 255) / 256), dim3(256), 0, s, p, image);
    return hipGetLastError();
}

// both nets, in-kernel sampling, in_dim <= 20: the rollout's launch
hipError_t launch_policy16_forward(const float* image, const float* obs, const double* rms, const uint64_t* tick, float* obs_n,
                                   float* action, float* clipped, float* value, float* logp, float* eps_out, int64_t n,
                                   uint64_t seed, uint64_t id_offset, float clip_obs, float eps_norm, int max_blocks, int in_dim,
                                   int act_dim, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    constexpr int kLds = w16::kLdsFloats * (int)sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)policy_forward_w16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)policy_forward_w16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    Policy16Args p{image, obs, rms, tick, obs_n, action, clipped, value, logp, eps_out, n, seed, id_offset, clip_obs, eps_norm,
                   in_dim, act_dim};
    int64_t blocks = (n + 255) / 256;
    if (blocks > max_blocks) blocks = max_blocks;
    if (act_dim == 4) hipLaunchKernelGGL(policy_forward_w16_kernel<true>, dim3((unsigned)blocks), dim3(1024), kLds, s, p);
    else hipLaunchKernelGGL(policy_forward_w16_kernel<false>, dim3((unsigned)blocks), dim3(1024), kLds, s, p);
    return hipGetLastError();
}

}  // namespace fpv
