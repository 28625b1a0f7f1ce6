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
#include <atomic>
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

}  // extern "C"

// weights: float32 interpolation weights -- or, with per-axis phase tables (phase_x/y/z: N0 / N1 / N2 doubles), complex64
// values  (float)w * (complex64)exp(2 pi i (phase_x[kx] + phase_y[ky] + phase_z[kz])) * (float)scale  in `cvalues`: the gridding
// matrix times the centred transform's modulation and normalisation (the G' factor of the reference's -O3 SENSE tree,
// examples/pics.py:104-177) without three more passes over 5e7 nonzeros in numpy
static int interp3_fill(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable,
                        const double* coord, const int32_t* rowptr, int32_t* colind, float* weights, int grid_order,
                        const double* phase_x, const double* phase_y, const double* phase_z, double scale, float2* cvalues) {
    if (m < 0 || !N || !coord || !rowptr || !table || ntable < 2 || !(width > 0) || (rowptr[m] > 0 && (!colind || (!weights && !cvalues))))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: bad arguments");
    if (grid_order != 0 && grid_order != 1)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: grid_order must be 0 (x, y, z) or 1 (x, z, y)");
    const int64_t n0 = N[0], n1 = N[1], n2 = N[2];
    if (n0 * n1 * n2 > 0x7fffffffLL) return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: grid exceeds int32 column indices");
    const double* cx = coord; const double* cy = coord + m; const double* cz = coord + 2 * m;
    std::atomic<int> bad{0};        // set from several host threads
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
            for (size_t q = 0; q < row.size(); ++q) ci[q] = row[q].first;
            if (weights) {
                float* wv = weights + rowptr[i];
                for (size_t q = 0; q < row.size(); ++q) wv[q] = row[q].second;
            }
            if (cvalues) {
                float2* cv = cvalues + rowptr[i];
                const float sc = (float)scale;
                for (size_t q = 0; q < row.size(); ++q) {
                    const int64_t col = row[q].first;
                    const int64_t kx = col % n0, k1 = (col / n0) % (grid_order == 0 ? n1 : n2), k2 = col / (n0 * (grid_order == 0 ? n1 : n2));
                    const int64_t ky = grid_order == 0 ? k1 : k2, kz = grid_order == 0 ? k2 : k1;
                    // the phase summed x, then y, then z in double, as Backend.fftc_mod adds its per-axis terms
                    double ph = 0.0 + phase_x[kx];
                    ph += phase_y[ky];
                    ph += phase_z[kz];
                    const double a = 6.283185307179586 * ph;
                    const float mr = (float)std::cos(a), mi = (float)std::sin(a);
                    const float w = row[q].second;
                    cv[q] = make_float2((w * mr) * sc, (w * mi) * sc);
                }
            }
        }
    });
    if (bad == 1) return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill: rowptr does not come from ig_interp3_count on the same inputs");
    if (bad == 2) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_interp3_fill: a row wraps onto the same column twice (grid smaller than the kernel)");
    return IG_OK;
}

// ---- the separable form of a gridding matrix (round 6) --------------------------------------------------------------------
// The reference's weights are products of three per-axis factors by construction (w = wz * wy * wx, indigo/interp.py:42-52), and
// what `pics.py -O3` folds into the gridding matrix on an even grid -- the centred transform's modulation exp(i pi k) and its
// 1 / sqrt(P) (examples/pics.py:104-177) -- is a sign per axis and cell and a constant.  So a sample is fully described by its first
// tap and tap count per axis and 3 x tw per-axis weights: ONE record of 64 bytes (tw = 4) instead of 27 stored taps of 8 .. 12
// bytes each (216 .. 324 bytes; at the reference's default width 3: 125 taps, 1000 .. 1500 bytes against 128).  The gridding kernels
// of ig_gridsep.hip compute the taps from the records.
//   record (ig_interp3_sep_words(tw) 32-bit words; axes in MEMORY order of the grid: 0 = x, 1 = middle, 2 = slow -- (x, y, z) for
//   grid_order 0, (x, z, y) for grid_order 1):
//     words [0, tw)       float weights of the x taps       (w * sign_x[cell])
//     words [tw, 2 tw)    float weights of the middle axis  (w * sign[cell])
//     words [2 tw, 3 tw)  float weights of the slow axis    (w * sign[cell] * scale)       unused weights are 0
//     word 3 tw           first tap (wrapped into the grid) on axis 0 | on axis 1 << 16
//     word 3 tw + 1       first tap on axis 2 | taps on axis 0 << 16 | on axis 1 << 20 | on axis 2 << 24
// Each factor is rounded to float32 once; a tap's weight is the float32 product (w1 * w2) * w0: at most five roundings where the
// stored matrix has two -- 3e-7 relative in the worst case, against the 1e-5 the parity tests allow.
extern "C" int ig_interp3_sep_words(int tw) { return tw == 4 ? 16 : (tw == 6 || tw == 8) ? 32 : -1; }

extern "C" int ig_interp3_sep(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable, const double* coord,
                              int grid_order, const double* sign_x, const double* sign_y, const double* sign_z, double scale,
                              int tw, uint32_t* records) {
    const int rw = ig_interp3_sep_words(tw);
    if (m < 0 || !N || !coord || !table || ntable < 2 || !(width > 0) || rw < 0 || (m > 0 && !records) || (grid_order != 0 && grid_order != 1))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_sep: bad arguments (tw is 4, 6 or 8; grid_order 0 or 1)");
    for (int d = 0; d < 3; ++d)
        if (N[d] < tw || N[d] > 65535) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_interp3_sep: grid axis of %lld points (tw .. 65535)", (long long)N[d]);
    const double* c[3] = {coord, coord + m, coord + 2 * m};
    const double* sg[3] = {sign_x, sign_y, sign_z};
    const int axis_of[3] = {0, grid_order == 0 ? 1 : 2, grid_order == 0 ? 2 : 1};       // memory axis -> reference axis
    std::atomic<int> bad{0};
    parallel_rows(m, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            uint32_t* r = records + (size_t)i * rw;
            for (int q = 0; q < rw; ++q) r[q] = 0u;
            uint32_t first[3], count[3];
            for (int a = 0; a < 3; ++a) {
                const int d = axis_of[a];
                const int64_t n = N[d];
                const double pos = (double)n * c[d][i] + (double)(n / 2);
                const Taps t = taps_of(pos, width);
                if (t.count > tw) bad = 1;
                const int64_t cnt = t.count > tw ? tw : t.count;
                first[a] = (uint32_t)wrap(t.start, n);
                count[a] = (uint32_t)cnt;
                for (int64_t q = 0; q < cnt; ++q) {
                    const int64_t x = t.start + q;
                    double w = lin_interp(table, ntable, std::fabs((double)x - pos) / width);
                    if (sg[d]) w *= sg[d][wrap(x, n)];
                    if (a == 2) w *= scale;
                    const float wf = (float)w;
                    __builtin_memcpy(r + a * tw + q, &wf, 4);
                }
            }
            r[3 * tw] = first[0] | (first[1] << 16);
            r[3 * tw + 1] = first[2] | (count[0] << 16) | (count[1] << 20) | (count[2] << 24);
        }
    });
    if (bad) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_interp3_sep: a sample has more than tw = %d taps on an axis (kernel half-width %g)", tw, width);
    return IG_OK;
}

extern "C" {

int ig_interp3_fill(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable,
                    const double* coord, const int32_t* rowptr, int32_t* colind, float* weights, int grid_order) {
    return interp3_fill(m, N, width, table, ntable, coord, rowptr, colind, weights, grid_order, nullptr, nullptr, nullptr, 1.0, nullptr);
}

int ig_interp3_fill_modulated(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable,
                              const double* coord, const int32_t* rowptr, int32_t* colind, void* values, int grid_order,
                              const double* phase_x, const double* phase_y, const double* phase_z, double scale) {
    if (!phase_x || !phase_y || !phase_z || !values) return ig_fail(nullptr, IG_ERR_ARG, "ig_interp3_fill_modulated: bad arguments");
    return interp3_fill(m, N, width, table, ntable, coord, rowptr, colind, nullptr, grid_order, phase_x, phase_y, phase_z, scale, (float2*)values);
}

// k-space support table of a gridding matrix whose columns number an n0 x n2 x n1 grid as kx + n0*(kz + n2*ky) (grid layouts
// 1 and 2 of the fused transform): one pass over the column indices sets the segment bits, the hulls follow from the bits.
// Layout of `table` (int16): [z_lo, z_hi) per (ky, kx tile), [y_lo, y_hi) per kx tile, then the segment bitmaps: zw_in uint32
// words per (ky, kx tile) -- bit kz / zw_in of word kz % zw_in is set iff segment (kx tile, ky, kz) holds a nonzero -- and, if
// zw_out differs, the same bits once more with zw_out words per entry.  The two forms are what the z pass of the transform
// wants on its input side (cropped transform: thread b of zw_in holds rows b + zw_in a) and on its output side (padded
// transform: thread k1 of zw_out stores rows k1 + zw_out k2): ig_fft_support_words gives both for a grid axis; 16 / 16 for the
// 256- and 512-point axes, where the table is round 1's.  (indigo_amd/fused.py:grid_support.)
int ig_grid_support(int64_t nnz, const int32_t* colind, int64_t n0, int64_t n1, int64_t n2, int tile, int zw_in, int zw_out, int16_t* table) {
    if (nnz < 0 || (nnz > 0 && !colind) || !table || n0 < 1 || n1 < 1 || n2 < 1 || tile < 1 || n0 % tile || zw_in < 1 || zw_out < 1 ||
        zw_in > 64 || zw_out > 64 || n2 > 32 * (int64_t)zw_in || n2 > 32 * (int64_t)zw_out || n2 > 32767 || n1 > 32767)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_support: bad arguments (tile divides n0; n2 <= 32 * words per entry)");
    const int64_t nt = n0 / tile, ne = n1 * nt;
    int16_t* zr = table;
    int16_t* yr = table + 2 * ne;
    uint32_t* bits = reinterpret_cast<uint32_t*>(table + 2 * (ne + nt));
    uint32_t* bits_out = zw_out != zw_in ? bits + ne * zw_in : nullptr;
    std::fill(table, table + 2 * (ne + nt), (int16_t)0);
    std::fill(bits, bits + ne * zw_in + (bits_out ? ne * zw_out : 0), 0u);
    std::atomic<int> bad{0};        // set from several host threads
    parallel_rows(nnz, [&](int64_t lo, int64_t hi) {
        int32_t last = -1;
        for (int64_t p = lo; p < hi; ++p) {
            const int32_t col = colind[p];
            if (col == last) continue;
            last = col;
            const int64_t kx = col % n0, kz = (col / n0) % n2, ky = col / (n0 * n2);
            if (col < 0 || ky >= n1) { bad = 1; continue; }
            uint32_t* w = bits + (ky * nt + kx / tile) * zw_in + (kz % zw_in);
            const uint32_t m = 1u << (kz / zw_in);
            if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & m)) {
                __atomic_fetch_or(w, m, __ATOMIC_RELAXED);
                if (bits_out) __atomic_fetch_or(bits_out + (ky * nt + kx / tile) * zw_out + (kz % zw_out), 1u << (kz / zw_out), __ATOMIC_RELAXED);
            }
        }
    });
    if (bad) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_support: column index outside the grid");
    parallel_rows(ne, [&](int64_t lo, int64_t hi) {
        for (int64_t e = lo; e < hi; ++e) {
            int zlo = 1 << 30, zhi = -1;
            for (int t = 0; t < zw_in; ++t) {
                const uint32_t w = bits[e * zw_in + t];
                if (!w) continue;
                const int first = __builtin_ctz(w), lastb = 31 - __builtin_clz(w);
                zlo = std::min(zlo, t + zw_in * first);
                zhi = std::max(zhi, t + zw_in * lastb);
            }
            if (zhi >= 0) { zr[2 * e] = (int16_t)zlo; zr[2 * e + 1] = (int16_t)(zhi + 1); }
        }
    });
    for (int64_t t = 0; t < nt; ++t) {
        int ylo = -1, yhi = -1;
        for (int64_t ky = 0; ky < n1; ++ky)
            if (zr[2 * (ky * nt + t) + 1] > zr[2 * (ky * nt + t)]) { if (ylo < 0) ylo = (int)ky; yhi = (int)ky; }
        if (ylo >= 0) { yr[2 * t] = (int16_t)ylo; yr[2 * t + 1] = (int16_t)(yhi + 1); }
    }
    return IG_OK;
}

}  // extern "C"
