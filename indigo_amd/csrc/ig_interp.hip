// Host-side construction of the 3-D gridding (interpolation) matrix in CSR form.
//
// Reference: indigo/interp.py:8-15 lin_interp, :18-80 _interp3_mat -- a numba-compiled loop nest (native code in
// the reference as well), called by Backend.Interp (indigo/backends/backend.py:392-401).  Same arithmetic, in the
// same order and in double precision, so the weights are bit-identical to the reference's:
//
//   pos_d  = N_d * coord[d, i] + N_d / 2                       (integer division)
//   taps   = ceil(pos_d - width) .. floor(pos_d + width) - 1   (end exclusive)
//   weight = (wz * wy) * wx,   w = lerp(table, |tap - pos| / width)   (0 at and beyond the table's end)
//   column = (x mod N0) + N0 * ((y mod N1) + N1 * (z mod N2))         (wrap-around)
//
// Two calls: ig_interp3_count fills the row pointers, ig_interp3_fill the column indices (sorted within a row, in
// the grid order asked for) and the float32 weights.  Rows are independent: a few host threads share them.
// No device work here (setup only): a 5e7-nonzero matrix takes about a second instead of the half minute of a
// vectorised numpy formulation.
#include "ig_common.h"
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace {

inline double lin_interp(const double* table, int64_t n, double x) {
    if (!(x < 1.0)) return 0.0;
    const double xs = x * (double)(n - 1);
    const int64_t idx = (int64_t)xs;
    const double frac = xs - (double)idx;
    const int64_t hi = idx + 1 < n - 1 ? idx + 1 : n - 1;
    // two roundings per product and one per sum, as numpy evaluates it (no fused multiply-add)
    const double a = (1.0 - frac) * table[idx];
    const double b = frac * table[hi];
    return a + b;
}

inline int64_t wrap(int64_t k, int64_t n) {
    int64_t r = k % n;
    return r < 0 ? r + n : r;
}

struct Taps { int64_t start, count; };

inline Taps taps_of(double pos, double width) {
    const int64_t s = (int64_t)std::ceil(pos - width), e = (int64_t)std::floor(pos + width);
    return Taps{s, e > s ? e - s : 0};
}

template <class F>
void parallel_rows(int64_t m, F&& body) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? hw : 4);
    if (nt > 16) nt = 16;
    if (m < 4096) nt = 1;
    std::vector<std::thread> th;
    const int64_t per = (m + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = (int64_t)t * per, hi = std::min<int64_t>(m, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=, &body]() { body(lo, hi); });
    }
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

int ig_interp3_count(int64_t m, const int64_t* N, double width, const double* coord, int32_t* rowptr) {
    if (m < 0 || !N || !coord || !rowptr || !(width > 0) || N[0] < 1 || N[1] < 1 || N[2] < 1)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_count: bad arguments");
    const double* cx = coord; const double* cy = coord + m; const double* cz = coord + 2 * m;
    std::vector<int64_t> cnt((size_t)m);
    parallel_rows(m, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const Taps tx = taps_of((double)N[0] * cx[i] + (double)(N[0] / 2), width);
            const Taps ty = taps_of((double)N[1] * cy[i] + (double)(N[1] / 2), width);
            const Taps tz = taps_of((double)N[2] * cz[i] + (double)(N[2] / 2), width);
            cnt[i] = tx.count * ty.count * tz.count;
        }
    });
    int64_t acc = 0;
    rowptr[0] = 0;
    for (int64_t i = 0; i < m; ++i) {
        acc += cnt[i];
        if (acc > 0x7fffffffLL) return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_count: more than 2^31 - 1 nonzeros");
        rowptr[i + 1] = (int32_t)acc;
    }
    return IG_OK;
}

int ig_interp3_fill(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable,
                    const double* coord, const int32_t* rowptr, int32_t* colind, float* weights, int grid_order) {
    if (m < 0 || !N || !coord || !rowptr || !table || ntable < 2 || !(width > 0) || (rowptr[m] > 0 && (!colind || !weights)))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: bad arguments");
    if (grid_order != 0 && grid_order != 1)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: grid_order must be 0 (x, y, z) or 1 (x, z, y)");
    const int64_t n0 = N[0], n1 = N[1], n2 = N[2];
    if (n0 * n1 * n2 > 0x7fffffffLL) return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: grid exceeds int32 column indices");
    const double* cx = coord; const double* cy = coord + m; const double* cz = coord + 2 * m;
    int bad = 0;
    parallel_rows(m, [&](int64_t lo, int64_t hi) {
        std::vector<std::pair<int32_t, float>> row;
        std::vector<double> wxs;
        std::vector<int64_t> jxs;
        for (int64_t i = lo; i < hi; ++i) {
            const double px = (double)n0 * cx[i] + (double)(n0 / 2);
            const double py = (double)n1 * cy[i] + (double)(n1 / 2);
            const double pz = (double)n2 * cz[i] + (double)(n2 / 2);
            const Taps tx = taps_of(px, width), ty = taps_of(py, width), tz = taps_of(pz, width);
            const int64_t cnt = tx.count * ty.count * tz.count;
            if (cnt != (int64_t)rowptr[i + 1] - rowptr[i]) { bad = 1; continue; }
            if (cnt == 0) continue;
            wxs.resize((size_t)tx.count); jxs.resize((size_t)tx.count);
            for (int64_t a = 0; a < tx.count; ++a) {
                const int64_t x = tx.start + a;
                wxs[a] = lin_interp(table, ntable, std::fabs((double)x - px) / width);
                jxs[a] = wrap(x, n0);
            }
            row.clear();
            for (int64_t c = 0; c < tz.count; ++c) {
                const int64_t z = tz.start + c;
                const double wz = lin_interp(table, ntable, std::fabs((double)z - pz) / width);
                const int64_t jz = wrap(z, n2);
                for (int64_t b = 0; b < ty.count; ++b) {
                    const int64_t y = ty.start + b;
                    const double wy = lin_interp(table, ntable, std::fabs((double)y - py) / width);
                    const int64_t jy = wrap(y, n1);
                    const double wzy = wz * wy;
                    const int64_t base = grid_order == 0 ? n0 * (jy + n1 * jz) : n0 * (jz + n2 * jy);
                    for (int64_t a = 0; a < tx.count; ++a)
                        row.emplace_back((int32_t)(base + jxs[a]), (float)(wzy * wxs[a]));
                }
            }
            bool sorted = true;
            for (size_t q = 1; q < row.size(); ++q) if (row[q].first <= row[q - 1].first) { sorted = false; break; }
            if (!sorted) {
                std::stable_sort(row.begin(), row.end(), [](const std::pair<int32_t, float>& u, const std::pair<int32_t, float>& v) { return u.first < v.first; });
                for (size_t q = 1; q < row.size(); ++q) if (row[q].first == row[q - 1].first) bad = 2;   // a row wraps onto one column twice
            }
            int32_t* ci = colind + rowptr[i];
            float* wv = weights + rowptr[i];
            for (size_t q = 0; q < row.size(); ++q) { ci[q] = row[q].first; wv[q] = row[q].second; }
        }
    });
    if (bad == 1) return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: rowptr does not come from ig_interp3_count on the same inputs");
    if (bad == 2) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_interp3_fill: a row wraps onto the same column twice (grid smaller than the kernel)");
    return IG_OK;
}

}  // extern "C"
