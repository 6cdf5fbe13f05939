// The reference's carrier interpolator is scipy.optimize.curve_fit on 7 points with 2
// parameters (carrier_sync.py:179-194), i.e. MINPACK's lmdif with SciPy's defaults
// (ftol = xtol = 1.49012e-8, gtol = 0, factor = 100, forward-difference Jacobian with
// eps = sqrt(machine epsilon), diag from the column norms).  Where the fit is well
// conditioned any least-squares solver lands on the same minimum; where it is not (short
// templates: the 7 points sit on a flat main lobe) the answer is whatever lmdif's path and
// stopping rules produce.  To stay on the reference's answer in both regimes this is lmdif
// itself -- restated from the published MINPACK-1 algorithm (lmdif, fdjac2, qrfac, lmpar,
// qrsolv, enorm; More', Garbow, Hillstrom 1980) for m = 7, n = 2 -- not a solver of our own.
// A scalar Python restatement of the same steps, kept with the test infrastructure
// (tests/test_lmdif_restatement.py), agrees with curve_fit to the last bit on random problems.
//
// Parallel layout: lane j (0..6) of an 8-lane group owns data point j (lane 7 carries
// zeros): function values, Jacobian rows and Householder vectors are lane-distributed,
// every sum over points is an 8-lane DPP reduction that is bitwise identical in all lanes,
// and the 2 x 2 part (R, lmpar, qrsolv, trust-region logic) is computed redundantly by all
// lanes from identical inputs -- so control flow stays uniform.  Sums over points are
// pairwise here and sequential in MINPACK (differences ~1e-16 relative).
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_util.hpp"

namespace thr {

// ---- 8-lane group primitives -------------------------------------------------------
// Sum over the 8-lane group, bitwise identical in all 8 lanes, on the DPP path
// (quad_perm xor 1, xor 2, then row_half_mirror: i <-> 7-i).  Each step adds the same two
// operands in both partners, so all lanes agree bit for bit PROVIDED nothing gets
// contracted into the adds (with -ffp-contract=fast the caller's `x*x` would be fused into
// the first add on one side only, partner lanes would differ by an ulp, and an
// accept/reject decision would eventually flip in some lanes only).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = dpp_u32<CTRL, 0xf>(0u, (unsigned)u);
    const unsigned hi = dpp_u32<CTRL, 0xf>(0u, (unsigned)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double group8_sum(double v) {
#pragma clang fp contract(off)
    asm volatile("" : "+v"(v));  // materialise the operand: nothing upstream may fuse into the adds
    v = v + dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_f64<0x141>(v);  // row_half_mirror
    return v;
}
// value of lane `src` (0..7) of this lane's group
__device__ __forceinline__ double group8_get(double v, int src) {
    const int lane = (threadIdx.x & 63 & ~7) | src;
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __shfl((unsigned)u, lane, 64), hi = __shfl((unsigned)(u >> 32), lane, 64);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

namespace lm {

constexpr double EPSMCH = 2.220446049250313e-16;   // dpmpar(1)
constexpr double DWARF = 2.2250738585072014e-308;  // dpmpar(2)

// enorm of a lane-distributed vector / of two scalars.  MINPACK's enorm rescales only
// outside [3.8e-20, 1.3e19 / n]; magnitudes here (FFT magnitudes, unit-free steps) stay inside.
__device__ __forceinline__ double enorm_lanes(double v) {
#pragma clang fp contract(off)
    return sqrt(group8_sum(v * v));
}
__device__ __forceinline__ double enorm2(double a, double b) {
#pragma clang fp contract(off)
    return sqrt(a * a + b * b);
}

// Model residual of this lane's point: A * |D(x - o)| - y with
// D(u) = sin(pi W u / N) / sin(pi u / N) / W, D(0) = 1 (carrier_sync.py:121-132,179-182;
// operation order as NumPy evaluates `np.pi*W*xdata/N`)
struct Point {
    double x, y;
    bool live;
    double n, w;
};
__device__ __forceinline__ double residual(const Point& p, double amp, double off) {
#pragma clang fp contract(off)
    if (!p.live) return 0.0;
    const double pi = 3.141592653589793;
    const double u = p.x - off;
    const double num = sin(((pi * p.w) * u) / p.n), den = sin((pi * u) / p.n);
    double d = (num / den) / p.w;
    if (d != d) d = 1.0;   // 0/0 at u == 0
    return amp * fabs(d) - p.y;
}

// qrsolv for n = 2 (R upper triangular as r00 r01 r11, column permutation ipvt, diagonal
// d[], right-hand side qtb[]): returns x[] and sdiag[] and the strict lower part s10.
__device__ __forceinline__ void qrsolv2(double r00, double r01, double r11, const int* ipvt,
                                        const double* d, const double* qtb, double* x,
                                        double* sdiag, double& s10) {
#pragma clang fp contract(off)
    // copy R^T into the lower triangle (r[i][j] = r[j][i]); save the diagonal in x
    double rr[2][2] = {{r00, r01}, {r01, r11}};
    double wa[2] = {qtb[0], qtb[1]};
    x[0] = rr[0][0];
    x[1] = rr[1][1];
    for (int j = 0; j < 2; ++j) {
        const int l = ipvt[j];
        if (d[l] != 0.0) {
            for (int k = j; k < 2; ++k) sdiag[k] = 0.0;
            sdiag[j] = d[l];
            double qtbpj = 0.0;
            for (int k = j; k < 2; ++k) {
                if (sdiag[k] == 0.0) continue;
                double cs, sn;
                if (fabs(rr[k][k]) < fabs(sdiag[k])) {
                    const double cotan = rr[k][k] / sdiag[k];
                    sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                    cs = sn * cotan;
                } else {
                    const double tn = sdiag[k] / rr[k][k];
                    cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                    sn = cs * tn;
                }
                rr[k][k] = cs * rr[k][k] + sn * sdiag[k];
                const double temp = cs * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + cs * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < 2; ++i) {
                    const double t2 = cs * rr[i][k] + sn * sdiag[i];
                    sdiag[i] = -sn * rr[i][k] + cs * sdiag[i];
                    rr[i][k] = t2;
                }
            }
        }
        sdiag[j] = rr[j][j];
        rr[j][j] = x[j];
    }
    int nsing = 2;
    for (int j = 0; j < 2; ++j) {
        if (sdiag[j] == 0.0 && nsing == 2) nsing = j;
        if (nsing < 2) wa[j] = 0.0;
    }
    for (int k = 0; k < nsing; ++k) {
        const int j = nsing - k - 1;
        double sum = 0.0;
        for (int i = j + 1; i < nsing; ++i) sum += rr[i][j] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < 2; ++j) x[ipvt[j]] = wa[j];
    s10 = rr[1][0];
}

// lmpar for n = 2: Levenberg-Marquardt parameter and step x[] for trust radius delta
__device__ __forceinline__ void lmpar2(double r00, double r01, double r11, const int* ipvt,
                                       const double* diag, const double* qtb, double delta,
                                       double& par, double* x) {
#pragma clang fp contract(off)
    const double r[2][2] = {{r00, r01}, {0.0, r11}};
    double wa1[2], wa2[2], sdiag[2] = {0.0, 0.0};
    int nsing = 2;
    for (int j = 0; j < 2; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == 0.0 && nsing == 2) nsing = j;
        if (nsing < 2) wa1[j] = 0.0;
    }
    for (int k = 0; k < nsing; ++k) {
        const int j = nsing - k - 1;
        wa1[j] /= r[j][j];
        const double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[i][j] * temp;
    }
    for (int j = 0; j < 2; ++j) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < 2; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm2(wa2[0], wa2[1]);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0.0;   // iter == 0
        return;
    }
    double parl = 0.0;
    if (nsing >= 2) {
        for (int j = 0; j < 2; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < 2; ++j) {
            double sum = 0.0;
            for (int i = 0; i < j; ++i) sum += r[i][j] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j][j];
        }
        const double temp = enorm2(wa1[0], wa1[1]);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < 2; ++j) {
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += r[i][j] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = enorm2(wa1[0], wa1[1]);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = DWARF / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (par == 0.0) par = fmax(DWARF, 0.001 * paru);
        double temp = sqrt(par);
        for (int j = 0; j < 2; ++j) wa1[j] = temp * diag[j];
        double s10;
        qrsolv2(r00, r01, r11, ipvt, wa1, qtb, x, sdiag, s10);
        for (int j = 0; j < 2; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm2(wa2[0], wa2[1]);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < 2; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        // solve with the lower-triangular S^T: s00 = sdiag[0], s10, s11 = sdiag[1]
        wa1[0] /= sdiag[0];
        wa1[1] -= s10 * wa1[0];
        wa1[1] /= sdiag[1];
        temp = enorm2(wa1[0], wa1[1]);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

}  // namespace lm

// curve_fit(model, x = -3..3, y, p0 = (y[3], 0)) -> fitted offset.  `j` = this lane's point.
__device__ inline double lmdif_dirichlet8(float yj, float y_peak, int j, double n, double w) {
#pragma clang fp contract(off)
    using namespace lm;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, factor = 100.0;
    const int maxfev = 600;   // 200 * (n + 1)
    Point pt{double(j - 3), double(yj), j < 7, n, w};
    double x[2] = {double(y_peak), 0.0};
    double fvec = residual(pt, x[0], x[1]);
    int nfev = 1;
    double fnorm = enorm_lanes(fvec);
    double par = 0.0, delta = 0.0, xnorm = 0.0;
    double diag[2] = {0.0, 0.0};
    int iter = 1, info = 0;
    const double eps = sqrt(EPSMCH);   // epsfcn = machine epsilon
    for (;;) {
        // ---- fdjac2: forward differences, one column per parameter
        double a[2];   // this lane's Jacobian row
        for (int c = 0; c < 2; ++c) {
            const double temp = x[c];
            double h = eps * fabs(temp);
            if (h == 0.0) h = eps;
            x[c] = temp + h;
            const double wa = residual(pt, x[0], x[1]);
            x[c] = temp;
            a[c] = (wa - fvec) / h;
        }
        nfev += 2;
        // ---- qrfac with column pivoting (rows = lanes)
        int ipvt[2] = {0, 1};
        double acnorm[2] = {enorm_lanes(a[0]), enorm_lanes(a[1])};
        double rdiag[2] = {acnorm[0], acnorm[1]}, wa_n[2] = {acnorm[0], acnorm[1]};
        const double acn[2] = {acnorm[0], acnorm[1]};   // wa2 of lmdif: norms in ORIGINAL column order
        for (int c = 0; c < 2; ++c) {
            int kmax = c;
            for (int k = c; k < 2; ++k)
                if (rdiag[k] > rdiag[kmax]) kmax = k;
            if (kmax != c) {   // (only c == 0, kmax == 1 can happen)
                const double t = a[c];
                a[c] = a[kmax];
                a[kmax] = t;
                rdiag[kmax] = rdiag[c];
                wa_n[kmax] = wa_n[c];
                const int ti = ipvt[c];
                ipvt[c] = ipvt[kmax];
                ipvt[kmax] = ti;
            }
            const bool in_rows = j >= c;   // Householder acts on rows c..m-1
            double ajnorm = enorm_lanes(in_rows ? a[c] : 0.0);
            if (ajnorm != 0.0) {
                const double ajj = group8_get(a[c], c);
                if (ajj < 0.0) ajnorm = -ajnorm;
                if (in_rows) a[c] /= ajnorm;
                if (j == c) a[c] += 1.0;
                for (int k = c + 1; k < 2; ++k) {
                    const double sum = group8_sum(in_rows ? a[c] * a[k] : 0.0);
                    const double temp = sum / group8_get(a[c], c);
                    if (in_rows) a[k] -= temp * a[c];
                    if (rdiag[k] != 0.0) {
                        const double t2 = group8_get(a[k], c) / rdiag[k];
                        rdiag[k] *= sqrt(fmax(0.0, 1.0 - t2 * t2));
                        const double q = rdiag[k] / wa_n[k];
                        if (0.05 * (q * q) <= EPSMCH) {
                            rdiag[k] = enorm_lanes(j > c ? a[k] : 0.0);
                            wa_n[k] = rdiag[k];
                        }
                    }
                }
            }
            rdiag[c] = -ajnorm;
        }
        // ---- first iteration: scaling and trust radius
        if (iter == 1) {
            for (int c = 0; c < 2; ++c) diag[c] = acn[c] != 0.0 ? acn[c] : 1.0;
            xnorm = enorm2(diag[0] * x[0], diag[1] * x[1]);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        // ---- (Q^T fvec)[0..1], R
        double wa4 = fvec, qtf[2];
        for (int c = 0; c < 2; ++c) {
            const double acc = group8_get(a[c], c);
            if (acc != 0.0) {
                const double sum = group8_sum(j >= c ? a[c] * wa4 : 0.0);
                const double temp = -sum / acc;
                if (j >= c) wa4 += a[c] * temp;
            }
            qtf[c] = group8_get(wa4, c);
        }
        const double r00 = rdiag[0], r01 = group8_get(a[1], 0), r11 = rdiag[1];
        // ---- scaled gradient norm
        double gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int c = 0; c < 2; ++c) {
                const int l = ipvt[c];
                if (acn[l] != 0.0) {
                    double sum = 0.0;
                    if (c == 0) sum = r00 * (qtf[0] / fnorm);
                    else sum = r01 * (qtf[0] / fnorm) + r11 * (qtf[1] / fnorm);
                    gnorm = fmax(gnorm, fabs(sum / acn[l]));
                }
            }
        }
        if (gnorm <= 0.0) {   // gtol = 0
            info = 4;
            break;
        }
        for (int c = 0; c < 2; ++c) diag[c] = fmax(diag[c], acn[c]);
        // ---- inner loop: step, ratio, trust-region update
        double ratio = 0.0;
        do {
            double p[2];
            lmpar2(r00, r01, r11, ipvt, diag, qtf, delta, par, p);
            const double wa1[2] = {-p[0], -p[1]};
            const double wa2[2] = {x[0] + wa1[0], x[1] + wa1[1]};
            const double pnorm = enorm2(diag[0] * wa1[0], diag[1] * wa1[1]);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double fnew = residual(pt, wa2[0], wa2[1]);
            ++nfev;
            const double fnorm1 = enorm_lanes(fnew);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1.0 - q * q;
            }
            // predicted reduction: R * P^T * step
            double wa3[2] = {0.0, 0.0};
            for (int c = 0; c < 2; ++c) {
                const double temp = wa1[ipvt[c]];
                if (c == 0) wa3[0] += r00 * temp;
                else {
                    wa3[0] += r01 * temp;
                    wa3[1] += r11 * temp;
                }
            }
            const double temp1 = enorm2(wa3[0], wa3[1]) / fnorm;
            const double temp2 = (sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + (temp2 * temp2) / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = prered != 0.0 ? actred / prered : 0.0;
            if (ratio <= 0.25) {
                double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par = par / temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
                x[0] = wa2[0];
                x[1] = wa2[1];
                fvec = fnew;
                xnorm = enorm2(diag[0] * x[0], diag[1] * x[1]);
                fnorm = fnorm1;
                ++iter;
            }
            const bool conv_f = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (conv_f) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (conv_f && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= EPSMCH * xnorm) info = 7;
            if (gnorm <= EPSMCH) info = 8;
            if (info != 0) break;
        } while (ratio < 1e-4);
        if (info != 0) break;
    }
    return x[1];
}

}  // namespace thr
