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

// Two-element vectors indexed by a run-time 0/1 (the column permutation): a select instead of
// an indexed private array, which the compiler would place in scratch memory.
struct V2 {
    double a, b;
    __device__ __forceinline__ double get(int i) const { return i ? b : a; }
};

// qrsolv for n = 2 (R upper triangular as r00 r01 r11, column permutation p0 -> {p0, 1 - p0},
// diagonal d, right-hand side qtb): returns x, sdiag and the strict lower part s10.
__device__ __forceinline__ void qrsolv2(double r00, double r01, double r11, int p0, V2 d, V2 qtb,
                                        V2& x, V2& sdiag, double& s10) {
#pragma clang fp contract(off)
    // MINPACK copies R^T into the lower triangle, saves the diagonal, then eliminates the
    // diagonal matrix D row by row with Givens rotations.  Written out for n = 2.
    double rkk0 = r00, rkk1 = r11, r10 = r01;   // r[0][0], r[1][1], r[1][0] (= r[0][1] transposed)
    double wa0 = qtb.a, wa1 = qtb.b;
    const double x0 = r00, x1 = r11;
    // MINPACK's two branches (cotangent / tangent form) do the same arithmetic on swapped
    // roles; written with selects so that cs / sn stay in registers
    auto givens = [](double rkk, double sd, double& cs, double& sn) {
        const bool small = fabs(rkk) < fabs(sd);
        const double ratio = small ? rkk / sd : sd / rkk;
        const double first = 0.5 / sqrt(0.25 + 0.25 * (ratio * ratio));
        const double second = first * ratio;
        sn = small ? first : second;
        cs = small ? second : first;
    };
    double sd0 = 0.0, sd1 = 0.0;
    // ---- j = 0: l = ipvt[0]
    {
        const double dl = d.get(p0);
        if (dl != 0.0) {
            double s0 = dl, s1 = 0.0, qtbpj = 0.0;
            // k = 0
            if (s0 != 0.0) {
                double cs, sn;
                givens(rkk0, s0, cs, sn);
                rkk0 = cs * rkk0 + sn * s0;
                const double temp = cs * wa0 + sn * qtbpj;
                qtbpj = -sn * wa0 + cs * qtbpj;
                wa0 = temp;
                const double t2 = cs * r10 + sn * s1;
                s1 = -sn * r10 + cs * s1;
                r10 = t2;
            }
            // k = 1
            if (s1 != 0.0) {
                double cs, sn;
                givens(rkk1, s1, cs, sn);
                rkk1 = cs * rkk1 + sn * s1;
                const double temp = cs * wa1 + sn * qtbpj;
                qtbpj = -sn * wa1 + cs * qtbpj;
                wa1 = temp;
            }
        }
        sd0 = rkk0;
        rkk0 = x0;
    }
    // ---- j = 1: l = ipvt[1]
    {
        const double dl = d.get(1 - p0);
        if (dl != 0.0) {
            double s1 = dl, qtbpj = 0.0;
            if (s1 != 0.0) {
                double cs, sn;
                givens(rkk1, s1, cs, sn);
                rkk1 = cs * rkk1 + sn * s1;
                const double temp = cs * wa1 + sn * qtbpj;
                qtbpj = -sn * wa1 + cs * qtbpj;
                wa1 = temp;
            }
        }
        sd1 = rkk1;
        rkk1 = x1;
    }
    (void)rkk0;
    (void)rkk1;
    int nsing = 2;
    if (sd0 == 0.0) nsing = 0;
    if (nsing < 2) wa0 = 0.0;
    if (sd1 == 0.0 && nsing == 2) nsing = 1;
    if (nsing < 2) wa1 = 0.0;
    // back substitution on the nsing x nsing leading block (lower part r10 holds s[1][0])
    if (nsing == 2) {
        wa1 = (wa1 - 0.0) / sd1;
        wa0 = (wa0 - r10 * wa1) / sd0;
    } else if (nsing == 1) {
        wa0 = (wa0 - 0.0) / sd0;
    }
    x = V2{p0 ? wa1 : wa0, p0 ? wa0 : wa1};   // x[ipvt[j]] = wa[j]
    sdiag = V2{sd0, sd1};
    s10 = r10;
}

// lmpar for n = 2: Levenberg-Marquardt parameter and step x for trust radius delta
__device__ __forceinline__ void lmpar2(double r00, double r01, double r11, int p0, V2 diag, V2 qtb,
                                       double delta, double& par, V2& x) {
#pragma clang fp contract(off)
    const int p1 = 1 - p0;
    // Gauss-Newton direction (rank-deficient R handled as MINPACK does)
    double wa0 = qtb.a, wa1 = qtb.b;
    int nsing = 2;
    if (r00 == 0.0) nsing = 0;
    if (nsing < 2) wa0 = 0.0;
    if (r11 == 0.0 && nsing == 2) nsing = 1;
    if (nsing < 2) wa1 = 0.0;
    if (nsing == 2) {
        wa1 /= r11;
        wa0 -= r01 * wa1;
        wa0 /= r00;
    } else if (nsing == 1) {
        wa0 /= r00;
    }
    x = V2{p0 ? wa1 : wa0, p0 ? wa0 : wa1};   // x[ipvt[j]] = wa1[j]
    int iter = 0;
    V2 wa2{diag.a * x.a, diag.b * x.b};
    double dxnorm = enorm2(wa2.a, wa2.b);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0.0;   // iter == 0
        return;
    }
    double parl = 0.0;
    if (nsing >= 2) {
        double u0 = diag.get(p0) * (wa2.get(p0) / dxnorm), u1 = diag.get(p1) * (wa2.get(p1) / dxnorm);
        u0 = (u0 - 0.0) / r00;
        u1 = (u1 - r01 * u0) / r11;
        const double temp = enorm2(u0, u1);
        parl = ((fp / delta) / temp) / temp;
    }
    const double g0 = (r00 * qtb.a) / diag.get(p0);
    const double g1 = (r01 * qtb.a + r11 * qtb.b) / diag.get(p1);
    const double gnorm = enorm2(g0, g1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = DWARF / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (par == 0.0) par = fmax(DWARF, 0.001 * paru);
        double temp = sqrt(par);
        const V2 dd{temp * diag.a, temp * diag.b};
        V2 sdiag;
        double s10;
        qrsolv2(r00, r01, r11, p0, dd, qtb, x, sdiag, s10);
        wa2 = V2{diag.a * x.a, diag.b * x.b};
        dxnorm = enorm2(wa2.a, wa2.b);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        double u0 = diag.get(p0) * (wa2.get(p0) / dxnorm), u1 = diag.get(p1) * (wa2.get(p1) / dxnorm);
        u0 /= sdiag.a;
        u1 -= s10 * u0;
        u1 /= sdiag.b;
        temp = enorm2(u0, u1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

}  // namespace lm

// curve_fit(model, x = -3..3, y, p0 = (y[3], 0)) -> fitted offset.  `j` = this lane's point.
// (Everything indexed by the column permutation is a V2 / an explicit pair: no private arrays
// with run-time indices, which would live in scratch memory.)
// *info_out (optional): MINPACK's exit code -- 1 .. 4 converged; 5 (600 evaluations) .. 8 are what
// curve_fit turns into RuntimeError("Optimal parameters not found") (scipy/optimize/_minpack_py.py)
__device__ inline double lmdif_dirichlet8(float yj, float y_peak, int j, double n, double w,
                                          int* info_out = nullptr) {
#pragma clang fp contract(off)
    using namespace lm;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, factor = 100.0;
    const int maxfev = 600;   // 200 * (n + 1)
    Point pt{double(j - 3), double(yj), j < 7, n, w};
    V2 x{double(y_peak), 0.0};   // (amplitude, offset)
    double fvec = residual(pt, x.a, x.b);
    int nfev = 1;
    double fnorm = enorm_lanes(fvec);
    double par = 0.0, delta = 0.0, xnorm = 0.0;
    V2 diag{0.0, 0.0};
    int iter = 1, info = 0;
    const double eps = sqrt(EPSMCH);   // epsfcn = machine epsilon
    for (;;) {
        // ---- fdjac2: forward differences, one column per parameter (this lane's Jacobian row)
        double a0, a1;
        {
            double h = eps * fabs(x.a);
            if (h == 0.0) h = eps;
            a0 = (residual(pt, x.a + h, x.b) - fvec) / h;
            h = eps * fabs(x.b);
            if (h == 0.0) h = eps;
            a1 = (residual(pt, x.a, x.b + h) - fvec) / h;
        }
        nfev += 2;
        // ---- qrfac with column pivoting (rows = lanes); acn = column norms in ORIGINAL order
        const double acn0 = enorm_lanes(a0), acn1 = enorm_lanes(a1);
        double rd0, rd1 = acn1, wn1 = acn1;
        int p0 = 0;   // ipvt = {p0, 1 - p0}
        if (acn1 > acn0) {
            const double t = a0;
            a0 = a1;
            a1 = t;
            rd1 = acn0;
            wn1 = acn0;
            p0 = 1;
        }
        {   // column 0: Householder over rows 0..6, applied to column 1
            double ajnorm = enorm_lanes(a0);
            if (ajnorm != 0.0) {
                if (group8_get(a0, 0) < 0.0) ajnorm = -ajnorm;
                a0 /= ajnorm;
                if (j == 0) a0 += 1.0;
                const double sum = group8_sum(a0 * a1);
                const double temp = sum / group8_get(a0, 0);
                a1 -= temp * a0;
                if (rd1 != 0.0) {
                    const double t2 = group8_get(a1, 0) / rd1;
                    rd1 *= sqrt(fmax(0.0, 1.0 - t2 * t2));
                    const double q = rd1 / wn1;
                    if (0.05 * (q * q) <= EPSMCH) {
                        rd1 = enorm_lanes(j > 0 ? a1 : 0.0);
                        wn1 = rd1;
                    }
                }
            }
            rd0 = -ajnorm;
        }
        {   // column 1: Householder over rows 1..6
            double ajnorm = enorm_lanes(j >= 1 ? a1 : 0.0);
            if (ajnorm != 0.0) {
                if (group8_get(a1, 1) < 0.0) ajnorm = -ajnorm;
                if (j >= 1) a1 /= ajnorm;
                if (j == 1) a1 += 1.0;
            }
            rd1 = -ajnorm;
        }
        // ---- first iteration: scaling and trust radius
        if (iter == 1) {
            diag = V2{acn0 != 0.0 ? acn0 : 1.0, acn1 != 0.0 ? acn1 : 1.0};
            xnorm = enorm2(diag.a * x.a, diag.b * x.b);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        // ---- (Q^T fvec)[0..1], R
        double wa4 = fvec;
        V2 qtf;
        {
            const double acc = group8_get(a0, 0);
            if (acc != 0.0) {
                const double sum = group8_sum(a0 * wa4);
                const double temp = -sum / acc;
                wa4 += a0 * temp;
            }
            qtf.a = group8_get(wa4, 0);
        }
        {
            const double acc = group8_get(a1, 1);
            if (acc != 0.0) {
                const double sum = group8_sum(j >= 1 ? a1 * wa4 : 0.0);
                const double temp = -sum / acc;
                if (j >= 1) wa4 += a1 * temp;
            }
            qtf.b = group8_get(wa4, 1);
        }
        const double r00 = rd0, r01 = group8_get(a1, 0), r11 = rd1;
        // ---- scaled gradient norm
        const double acn_p0 = p0 ? acn1 : acn0, acn_p1 = p0 ? acn0 : acn1;
        double gnorm = 0.0;
        if (fnorm != 0.0) {
            if (acn_p0 != 0.0) gnorm = fmax(gnorm, fabs((r00 * (qtf.a / fnorm)) / acn_p0));
            if (acn_p1 != 0.0)
                gnorm = fmax(gnorm, fabs((r01 * (qtf.a / fnorm) + r11 * (qtf.b / fnorm)) / acn_p1));
        }
        if (gnorm <= 0.0) {   // gtol = 0
            info = 4;
            break;
        }
        diag = V2{fmax(diag.a, acn0), fmax(diag.b, acn1)};
        // ---- inner loop: step, ratio, trust-region update
        double ratio = 0.0;
        do {
            V2 p;
            lmpar2(r00, r01, r11, p0, diag, qtf, delta, par, p);
            const V2 wa1{-p.a, -p.b};
            const V2 wa2{x.a + wa1.a, x.b + wa1.b};
            const double pnorm = enorm2(diag.a * wa1.a, diag.b * wa1.b);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double fnew = residual(pt, wa2.a, wa2.b);
            ++nfev;
            const double fnorm1 = enorm_lanes(fnew);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1.0 - q * q;
            }
            // predicted reduction: R * P^T * step
            const double s0 = wa1.get(p0), s1 = wa1.get(1 - p0);
            const double w30 = (0.0 + r00 * s0) + r01 * s1, w31 = r11 * s1;
            const double temp1 = enorm2(w30, w31) / fnorm;
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
                x = wa2;
                fvec = fnew;
                xnorm = enorm2(diag.a * x.a, diag.b * x.b);
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
    if (info_out != nullptr) *info_out = info;
    return x.b;
}

}  // namespace thr
