// Host-side (float64) construction of the steerable-pyramid filter bank constants.
//
// Restates, in plain C++, the mask arithmetic of the reference
//   api/steerable/math_utils.py:52-73   (prepare_grid, rcosFn, pointOp = np.interp)
//   api/steerable/SCFpyr_PyTorch.py:61-63, 94-107, 139-199  (LUTs, lo0/hi0, himask, anglemask,
//                                                            crop bounds, lomask)
// and folds them into the complex per-band tables the gfx950 kernels consume.  Masks are
// evaluated in float64 exactly as numpy does and rounded to fp32 once at the end
// (SCFpyr_PyTorch.py:106-107 rounds each factor; the difference is below 1 fp32 ulp of the product).
#include <cmath>
#include <algorithm>
#include "mm_common.h"

namespace mm {

namespace {

const double kPi = 3.14159265358979323846;

// np.interp(x, xp, fp) for increasing xp: clamped at the ends, linear in between.
double interp(double x, const std::vector<double>& xp, const std::vector<double>& fp) {
    const size_t n = xp.size();
    if (x <= xp[0]) return fp[0];
    if (x >= xp[n - 1]) return fp[n - 1];
    size_t j = std::upper_bound(xp.begin(), xp.end(), x) - xp.begin() - 1;  // xp[j] <= x < xp[j+1]
    if (j >= n - 1) j = n - 2;
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    return slope * (x - xp[j]) + fp[j];
}

struct Grid {
    int n;
    std::vector<double> log_rad, angle;  // [n][n]
};

// math_utils.py:52-60 for even square m=n: x_k = -1 + 2k/n
Grid prepare_grid(int n) {
    Grid g;
    g.n = n;
    g.log_rad.resize((size_t)n * n);
    g.angle.resize((size_t)n * n);
    std::vector<double> x(n);
    const double start = -(double)(n / 2) / (n / 2.0);
    const double stop = (double)(n / 2) / (n / 2.0) - (1 - n % 2) * 2.0 / n;
    const double step = (stop - start) / (n - 1);
    for (int k = 0; k < n; ++k) x[k] = start + k * step;  // np.linspace: start + k*step
    x[n - 1] = stop;
    std::vector<double> rad((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double xv = x[j], yv = x[i];
            g.angle[(size_t)i * n + j] = std::atan2(yv, xv);
            rad[(size_t)i * n + j] = std::sqrt(xv * xv + yv * yv);
        }
    rad[(size_t)(n / 2) * n + n / 2] = rad[(size_t)(n / 2) * n + n / 2 - 1];
    for (size_t k = 0; k < rad.size(); ++k) g.log_rad[k] = std::log2(rad[k]);
    return g;
}

// math_utils.py:62-69 with width=1, position=-0.5, then Yrcos = sqrt(Y) (SCFpyr_PyTorch.py:97-98)
void rcos(std::vector<double>& X, std::vector<double>& Y) {
    const int N = 256;
    X.resize(N + 3);
    Y.resize(N + 3);
    for (int k = 0; k < N + 3; ++k) {
        const double x = kPi * (double)(k - N - 1) / 2 / N;
        const double c = std::cos(x);
        Y[k] = c * c;
        X[k] = -0.5 + 2.0 * 1.0 / kPi * (x + kPi / 4);
    }
    Y[0] = Y[1];
    Y[N + 2] = Y[N + 1];
    for (auto& y : Y) y = std::sqrt(y);
}

// SCFpyr_PyTorch.py:182-183
void crop_bounds(int d, int& s, int& e) {
    s = (int)(std::ceil((d + 0.5) / 2) - std::ceil((std::ceil((d - 0.5) / 2) + 0.5) / 2));
    e = (int)(s + std::ceil((d - 0.5) / 2));
}

double factorial(int n) { return n <= 1 ? 1.0 : n * factorial(n - 1); }

}  // namespace

int host_level_mask(const PyramidConfig& c, int level, int band, std::vector<double>& out, int& side, int crop[2]) {
    if (level < 1 || level > c.height - 2 || band < 0 || band >= c.nbands) return MM_ERR_INVALID_ARG;
    const int n0 = 2 * c.size;
    Grid g = prepare_grid(n0);
    std::vector<double> Xr, Yr;
    rcos(Xr, Yr);
    std::vector<double> YIr(Yr.size());
    for (size_t k = 0; k < Yr.size(); ++k) YIr[k] = std::sqrt(1 - Yr[k] * Yr[k]);

    // angular LUT (SCFpyr_PyTorch.py:61-63,148-150)
    const int lut = 1024;
    const int nl = 3 * lut + 3;
    std::vector<double> Xc(nl), Yc(nl);
    const int order = c.nbands - 1;
    const double cst = std::pow(2.0, 2 * order) * factorial(order) * factorial(order) / (c.nbands * factorial(2 * order));
    for (int k = 0; k < nl; ++k) {
        Xc[k] = kPi * (double)(k - (2 * lut + 1)) / lut;
        double alpha = std::fmod(Xc[k] + kPi, 2 * kPi);
        if (alpha < 0) alpha += 2 * kPi;  // python % semantics
        alpha -= kPi;
        Yc[k] = 2 * std::sqrt(cst) * std::pow(std::cos(Xc[k]), order) * (std::fabs(alpha) < kPi / 2 ? 1.0 : 0.0);
    }

    // running product of low-pass masks, cropped alongside the grids
    int n = n0;
    std::vector<double> lo((size_t)n * n);
    for (size_t k = 0; k < lo.size(); ++k) lo[k] = interp(g.log_rad[k], Xr, YIr);
    crop[0] = 0;
    crop[1] = n0;
    for (int l = 1;; ++l) {
        for (auto& x : Xr) x -= std::log2((double)c.scale_factor);
        if (l == level) {
            std::vector<double> Xs(nl);
            for (int k = 0; k < nl; ++k) Xs[k] = Xc[k] + kPi * band / c.nbands;
            out.resize((size_t)n * n);
            for (size_t k = 0; k < out.size(); ++k)
                out[k] = lo[k] * interp(g.angle[k], Xs, Yc) * interp(g.log_rad[k], Xr, Yr);
            side = n;
            return MM_OK;
        }
        int s, e;
        crop_bounds(n, s, e);
        const int m = e - s;
        Grid g2;
        g2.n = m;
        g2.log_rad.resize((size_t)m * m);
        g2.angle.resize((size_t)m * m);
        std::vector<double> lo2((size_t)m * m);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                const size_t src = (size_t)(i + s) * n + (j + s), dst = (size_t)i * m + j;
                g2.log_rad[dst] = g.log_rad[src];
                g2.angle[dst] = g.angle[src];
                // lomask uses |sqrt(1 - Yrcos^2)| on the already-shifted Xrcos (SCFpyr_PyTorch.py:193-194)
                lo2[dst] = lo[src] * interp(g.log_rad[src], Xr, YIr);
            }
        g = g2;
        lo.swap(lo2);
        n = m;
        crop[0] = s;
        crop[1] = e;
    }
}

// Walks SCFpyr_PyTorch.build / _build_levels (SCFpyr_PyTorch.py:70-208) once for an n0 x n0 grid and emits the
// multiplier every returned tensor applies to the (shifted) image spectrum.
int build_scf_full_tables(int n0, int height, int nbands, int scale_factor, std::vector<ScfOutput>& outs) {
    outs.clear();
    if (n0 <= 0 || height < 2 || nbands < 2 || scale_factor < 1) return MM_ERR_INVALID_ARG;
    Grid g = prepare_grid(n0);
    std::vector<double> Xr, Yr;
    rcos(Xr, Yr);
    std::vector<double> YIr(Yr.size());
    for (size_t k = 0; k < Yr.size(); ++k) YIr[k] = std::sqrt(1 - Yr[k] * Yr[k]);
    const int lut = 1024, nl = 3 * lut + 3;
    std::vector<double> Xc(nl), Yc(nl);
    const int order = nbands - 1;
    const double cst = std::pow(2.0, 2 * order) * factorial(order) * factorial(order) / (nbands * factorial(2 * order));
    for (int k = 0; k < nl; ++k) {
        Xc[k] = kPi * (double)(k - (2 * lut + 1)) / lut;
        double alpha = std::fmod(Xc[k] + kPi, 2 * kPi);
        if (alpha < 0) alpha += 2 * kPi;
        alpha -= kPi;
        Yc[k] = 2 * std::sqrt(cst) * std::pow(std::cos(Xc[k]), order) * (std::fabs(alpha) < kPi / 2 ? 1.0 : 0.0);
    }
    // (-i)^(nbands-1): exact values, the power cycles with period 4
    static const double fre[4] = {1, 0, -1, 0}, fim[4] = {0, -1, 0, 1};
    const double cr = fre[order & 3], ci = fim[order & 3];

    // emit: real multiplier `mk` on the shifted n x n grid -> complex table in FFT order
    auto emit = [&](const std::vector<double>& mk, int n, double re, double im, int is_complex) {
        ScfOutput o;
        o.side = n;
        o.is_complex = is_complex;
        o.table.assign((size_t)n * n * 2, 0.0);
        const double norm = 1.0 / ((double)n * n);
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                const int u = (a + n / 2) % n, v = (b + n / 2) % n;  // fftshift (either parity: the DC sample lands on n / 2)
                const double m = mk[(size_t)u * n + v] * norm;
                o.table[((size_t)a * n + b) * 2] = m * re;
                o.table[((size_t)a * n + b) * 2 + 1] = m * im;
            }
        outs.push_back(std::move(o));
    };

    int n = n0;
    std::vector<double> lo((size_t)n * n), tmp((size_t)n * n);
    for (size_t k = 0; k < lo.size(); ++k) {
        lo[k] = interp(g.log_rad[k], Xr, YIr);   // lo0mask
        tmp[k] = interp(g.log_rad[k], Xr, Yr);   // hi0mask
    }
    emit(tmp, n, 1.0, 0.0, 0);
    for (int l = 1; l <= height - 2; ++l) {
        for (auto& x : Xr) x -= std::log2((double)scale_factor);
        for (int band = 0; band < nbands; ++band) {
            std::vector<double> Xs(nl);
            for (int k = 0; k < nl; ++k) Xs[k] = Xc[k] + kPi * band / nbands;
            tmp.resize((size_t)n * n);
            for (size_t k = 0; k < tmp.size(); ++k)
                tmp[k] = lo[k] * interp(g.angle[k], Xs, Yc) * interp(g.log_rad[k], Xr, Yr);
            emit(tmp, n, cr, ci, 1);
        }
        int s, e;
        crop_bounds(n, s, e);
        const int m = e - s;
        if (m <= 0) return MM_ERR_UNSUPPORTED;
        // (odd grids: batch_fftshift2d rolls by n//2 + 1 and batch_ifftshift2d by n//2, math_utils.py:33-47, i.e. the standard
        //  shifts -- FFT index a sits at shifted index (a + n/2) % n for either parity, which is what `emit` uses; the crop keeps
        //  the DC sample at m/2: s + m/2 == n/2 for the reference's bounds)
        if (s + m / 2 != n / 2) return MM_ERR_UNSUPPORTED;
        Grid g2;
        g2.n = m;
        g2.log_rad.resize((size_t)m * m);
        g2.angle.resize((size_t)m * m);
        std::vector<double> lo2((size_t)m * m);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                const size_t src = (size_t)(i + s) * n + (j + s), dst = (size_t)i * m + j;
                g2.log_rad[dst] = g.log_rad[src];
                g2.angle[dst] = g.angle[src];
                lo2[dst] = lo[src] * interp(g.log_rad[src], Xr, YIr);
            }
        g = g2;
        lo.swap(lo2);
        n = m;
    }
    emit(lo, n, 1.0, 0.0, 0);
    return MM_OK;
}

int build_pyramid_tables(const PyramidConfig& c, PyramidTables& t) {
    const int S = c.size, N = 2 * S;  // 48, 96
    t.dct.resize((size_t)S * S);
    t.ec.resize((size_t)S * S);
    t.es.resize((size_t)S * S);
    for (int f = 0; f < S; ++f)
        for (int m = 0; m < S; ++m) {
            t.dct[(size_t)f * S + m] = (float)(2.0 * std::cos(kPi * f * (2 * m + 1) / N));
            const int r = (f * m) % N;  // exact argument reduction
            t.ec[(size_t)f * S + m] = (float)std::cos(2.0 * kPi * r / N);
            t.es[(size_t)f * S + m] = (float)std::sin(2.0 * kPi * r / N);
        }
    for (int level = 1; level <= 2; ++level) {
        const int n = N >> (level - 1);  // 96, 48   (grid side)
        const int h = n / 2;             // 48, 24   (half-plane width / kept quadrant)
        for (int band = 0; band < 2; ++band) {
            std::vector<double> mk;
            int side, crop[2];
            int rc = host_level_mask(c, level, band, mk, side, crop);
            if (rc != MM_OK) return rc;
            std::vector<float>& dst = (level == 1 ? t.m1 : t.m2)[band];
            dst.assign((size_t)n * h * 2, 0.f);
            const double norm = 1.0 / ((double)n * n);  // ifft normalisation of this level (quirk Q13)
            for (int u = 0; u < n; ++u)
                for (int v = 0; v < n; ++v) {
                    const int fu = u - h, fv = v - h;  // signed frequency at this level == at level 0
                    // half-plane kept by the kernels: band 0 -> fv in [0,h), band 1 -> fu in [0,h)
                    const int fk = band == 0 ? fv : fu;
                    if (fk < 0) continue;
                    const double m = mk[(size_t)u * n + v];
                    // (-i)^(nbands-1) * exp(i pi (fu+fv)/N): phase of the mirrored-image DFT (see pyramid.hip)
                    const double ph = kPi * (fu + fv) / N - kPi / 2;
                    const double re = m * norm * std::cos(ph), im = m * norm * std::sin(ph);
                    size_t idx = band == 0 ? ((size_t)u * h + fv) : ((size_t)fu * n + v);
                    dst[idx * 2] = (float)re;
                    dst[idx * 2 + 1] = (float)im;
                }
        }
    }
    return MM_OK;
}

}  // namespace mm
