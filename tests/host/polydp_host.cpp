// CPU build of the kernels' approxPolyDP / arcLength code (fpv_drone_rl_agent_amd/csrc/fpv_polydp.h) for tests/test_polydp_cpu.py:
// the header is plain C++ over a lane-group policy, compiled here by g++ with a group of one lane.  Test infrastructure; the
// product never loads this.
#define __host__
#define __device__
#define __forceinline__ inline
#include "../../fpv_drone_rl_agent_amd/csrc/fpv_polydp.h"

namespace {
struct OneLane {
    int lane() const { return 0; }
    int size() const { return 1; }
    uint32_t max_u32(uint32_t x) const { return x; }
    int add_i32(int x) const { return x; }
};
struct TableSpan {
    const int16_t* tab;   // [128][2]: a, b; a < 0 for a row without red pixels
    bool operator()(int row, int& a, int& b) const {
        a = tab[2 * row]; b = tab[2 * row + 1];
        return a >= 0;
    }
};
}  // namespace

// spans: [128][2].  out: {visible, sum x, sum y, 2 * contourArea, w, h}.  What fpv_pixel_tile.h does around poly_detect is
// restated here: the accumulators over all rows, the border rule (hover.py:180-185), one 8-connected run of rows or the largest
// component.
extern "C" int polydp_detect_spans(const int16_t* spans, int out[6]) {
    TableSpan span{spans};
    int n = 0, j0 = 128, jn = -1, W = 0, S = 0, H = 0, viol = -1000, amin = 128, bmax = -1, pa = 0, pb = 0;
    for (int j = 0; j < 128; ++j) {
        int a, b;
        if (!span(j, a, b)) continue;
        if (n > 0 && jn == j - 1) {
            const int da = a > pa ? a - pa : pa - a, db = b > pb ? b - pb : pb - b;
            S += da + db; H += (da > 0) + (db > 0);
            if (a - pb > viol) viol = a - pb;
            if (pa - b > viol) viol = pa - b;
        }
        W += b - a; ++n;
        if (j < j0) j0 = j;
        jn = j; pa = a; pb = b;
        if (a < amin) amin = a;
        if (b > bmax) bmax = b;
    }
    for (int i = 0; i < 6; ++i) out[i] = 0;
    if (n == 0 || amin == 0 || bmax == 127 || j0 == 0 || jn == 127) return 0;
    const bool connected = jn - j0 + 1 == n && viol <= 1;
    fpv::PolyBlob bl = connected ? fpv::PolyBlob{j0, jn, W, S, H, amin, bmax} : fpv::poly_largest_component(span, j0, jn);
    const fpv::PolyResult r = fpv::poly_detect(OneLane{}, span, bl);
    out[0] = r.vis; out[1] = r.sx; out[2] = r.sy; out[3] = r.area2; out[4] = r.w; out[5] = r.h;
    return r.vis;
}
