// CPU build of the kernels' contour code (fpv_drone_rl_agent_amd/csrc/fpv_polydp.h) for tests/test_polydp_cpu.py: the header is
// plain C++ over a lane-group policy, compiled here by g++.  Three groups play it: a group of ONE lane, and groups of EIGHT and of
// SIXTEEN lanes played by as many threads in lockstep (every cross-lane operation of the policy is a rendezvous), which exercises the scan, the
// lane-strided positions and the all-reduces exactly as the 8-lane DPP groups of the kernels use them.  Test infrastructure; the
// product never loads this.
#define __host__
#define __device__
#define __forceinline__ inline
#include "../../fpv_drone_rl_agent_amd/csrc/fpv_polydp.h"

#include <pthread.h>
#include <string.h>

namespace {
struct OneLane {
    int lane() const { return 0; }
    int size() const { return 1; }
    uint32_t max_u32(uint32_t x) const { return x; }
    int max_i32(int x) const { return x; }
    int add_i32(int x) const { return x; }
    int scan_add(int x) const { return x; }
    void fence() const {}
};

// n threads, one per lane; slots[] + a barrier make every group operation an exchange
struct Shared {
    pthread_barrier_t bar;
    int64_t slot[16];
    int n;
};
struct EightLanes {
    int l;
    Shared* sh;
    int lane() const { return l; }
    int size() const { return sh->n; }
    template <class F>
    int64_t exchange(int64_t x, F f) const {
        sh->slot[l] = x;
        pthread_barrier_wait(&sh->bar);
        const int64_t r = f(sh->slot);
        pthread_barrier_wait(&sh->bar);
        return r;
    }
    uint32_t max_u32(uint32_t x) const {
        const int n = sh->n;
        return (uint32_t)exchange((int64_t)x, [n](const int64_t* s) { int64_t m = s[0]; for (int i = 1; i < n; ++i) m = s[i] > m ? s[i] : m; return m; });
    }
    int max_i32(int x) const {
        const int n = sh->n;
        return (int)exchange((int64_t)x, [n](const int64_t* s) { int64_t m = s[0]; for (int i = 1; i < n; ++i) m = s[i] > m ? s[i] : m; return m; });
    }
    int add_i32(int x) const {
        const int n = sh->n;
        return (int)exchange((int64_t)x, [n](const int64_t* s) { int64_t m = 0; for (int i = 0; i < n; ++i) m += s[i]; return m; });
    }
    int scan_add(int x) const {
        const int me = l;
        return (int)exchange((int64_t)x, [me](const int64_t* s) { int64_t m = 0; for (int i = 0; i <= me; ++i) m += s[i]; return m; });
    }
    void fence() const { pthread_barrier_wait(&sh->bar); }
};

constexpr int kS = 128;
void fill_table(const int16_t* spans, int& jlo, int& R, uint16_t* tab) {
    jlo = 128; int jhi = -1;
    for (int j = 0; j < 128; ++j) if (spans[2 * j] >= 0) { if (j < jlo) jlo = j; jhi = j; }
    R = jhi >= jlo ? jhi - jlo + 1 : 0;
    // like the kernels, the table may start above / end below the blob: one empty row on either side when there is room
    if (R > 0 && jlo > 0) { --jlo; ++R; }
    if (R > 0 && jlo + R < 128) ++R;
    tab[-1] = tab[R] = (uint16_t)0x00FFu;                           // the table's frame
    for (int r = 0; r < R; ++r) {
        const int a = spans[2 * (jlo + r)], b = spans[2 * (jlo + r) + 1];
        tab[r] = a >= 0 ? (uint16_t)(a | b << 8) : (uint16_t)0x00FFu;
    }
}
void put(const fpv::PolyResult& r, int out[6]) { out[0] = r.vis; out[1] = r.sx; out[2] = r.sy; out[3] = r.area2; out[4] = r.w; out[5] = r.h; }

uint32_t g_gtab[128];
struct Job { const uint16_t* tab; uint16_t* list; int jlo, R, lane; Shared* sh; fpv::PolyResult res; };
void* run_lane(void* p) {
    Job* j = (Job*)p;
    j->res = fpv::poly_blob<6>(EightLanes{j->lane, j->sh}, j->tab, j->list, g_gtab, j->jlo, j->R);      // 8 or 16 lanes x 6 points
    return nullptr;
}
}  // namespace

// spans: [128][2] (a, b; a < 0: no red pixel in the row).  out: {visible, sum x, sum y, 2 * contourArea, w, h}
// mode 0: poly_blob with one lane (contour in `registers`, falls back to the slot routine by itself); mode 1 / 3: poly_blob with
// eight / sixteen lanes in lockstep, six contour points per lane (larger contours go to the slot routine: exercised often); mode 2: the slot
// routine alone (poly_largest_component + poly_detect).
extern "C" int polydp_detect_spans(const int16_t* spans, int mode, int out[6]) {
    uint16_t tab_mem[132], list[kS + 64];
    uint16_t* tab = tab_mem + 1;
    int jlo, R;
    for (int i = 0; i < 6; ++i) out[i] = 0;
    fill_table(spans, jlo, R, tab);
    if (R == 0) return 0;
    for (int k = 0; k < 128; ++k) g_gtab[k] = fpv::poly_run_length_q23(k);
    if (mode == 0) {
        put(fpv::poly_blob<kS>(OneLane{}, tab, list, g_gtab, jlo, R), out);
    } else if (mode == 1 || mode == 3) {
        const int n = mode == 1 ? 8 : 16;
        Shared sh;
        sh.n = n;
        pthread_barrier_init(&sh.bar, nullptr, n);
        Job jobs[16];
        pthread_t th[16];
        for (int l = 0; l < n; ++l) { jobs[l] = Job{tab, list, jlo, R, l, &sh, {}}; pthread_create(&th[l], nullptr, run_lane, &jobs[l]); }
        for (int l = 0; l < n; ++l) pthread_join(th[l], nullptr);
        pthread_barrier_destroy(&sh.bar);
        for (int l = 1; l < n; ++l)
            if (memcmp(&jobs[l].res, &jobs[0].res, sizeof(fpv::PolyResult)) != 0) return -1;     // every lane must hold the same result
        put(jobs[0].res, out);
    } else {
        const fpv::TabSpan span{tab, jlo};
        int n = 0, j0 = 128, jn = -1, amin = 128, bmax = -1;
        for (int r = 0; r < R; ++r) {
            int a, b;
            if (!span(jlo + r, a, b)) continue;
            ++n; if (jlo + r < j0) j0 = jlo + r; jn = jlo + r;
            if (a < amin) amin = a;
            if (b > bmax) bmax = b;
        }
        if (n == 0 || amin == 0 || bmax == 127 || j0 == 0 || jn == 127) return 0;
        put(fpv::poly_detect(OneLane{}, span, fpv::poly_largest_component(span, j0, jn)), out);
    }
    return out[0];
}
