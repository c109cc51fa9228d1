// Forward-mode automatic differentiation for the device model functors.
// The reference differentiates its plugins exactly (pydrake.forwarddiff.jacobian,
// quadrotor_dynamics.py:136-138; symbolic Jacobian, pendulum_dynamics.py:25-26);
// a K-wide dual number does the same on device.  Everything is force-inlined and
// fully unrolled so that partials seeded with literal 0/1 fold away.
#pragma once
#include <hip/hip_runtime.h>

#define IRS_HD __host__ __device__ __forceinline__

template <typename T, int K>
struct Dual {
    T v;
    T d[K];
};

template <typename T> struct scalar_of { using type = T; };
// underlying value (for branch conditions of non-smooth models)
IRS_HD float irs_value(float x) { return x; }
IRS_HD double irs_value(double x) { return x; }
template <typename T, int K> IRS_HD T irs_value(const Dual<T, K>& x) { return x.v; }
template <typename T, int K> struct scalar_of<Dual<T, K>> { using type = T; };

// f32 sine/cosine for the sample pass: branch-free (no large-argument slow path, so the
// compiler can batch the sample loads around it), ~20 VALU instructions for the pair.
// Quadrant reduction r = x - q*pi/2 by two FMAs against a hi/lo split of pi/2, then
// the minimax polynomials on [-pi/4, pi/4].  Absolute error <= ~1.5e-7 for |x| < 1e3
// (grows like 6e-8*|x|/1e3 beyond); the sample pass is held to an f32 tolerance.
// sin/cos of q*pi/2 + r for |r| <= pi/4 (+ rounding), q an integer quadrant count.
IRS_HD void irs_sincos_quadrant(float r, int qi, float& s, float& c);

IRS_HD void irs_sincos(float x, float& s, float& c) {
    const float q = rintf(x * 0.63661977236758134f);           // 2/pi
    float r = fmaf(q, -1.57079637050628662f, x);                // pi/2 hi
    r = fmaf(q, 4.37113900018624283e-8f, r);                    // -(pi/2 lo)
    irs_sincos_quadrant(r, (int)q, s, c);
}

IRS_HD void irs_sincos_quadrant(float r, int qi, float& s, float& c) {
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    ps = fmaf(ps * r2, r, r);                                   // sin(r)
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    pc = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));              // cos(r)
    const float ss = (qi & 1) ? pc : ps;
    const float cc = (qi & 1) ? ps : pc;
    s = (qi & 2) ? -ss : ss;
    c = ((qi + 1) & 2) ? -cc : cc;
}
// f64 pair for the sequential chains (rollouts, nominal point): the same quadrant
// reduction with a hi/lo split of pi/2 and the fdlibm kernel polynomials (k_sin.c /
// k_cos.c coefficients, < 1 ulp on [-pi/4, pi/4]).  ~35 dependent f64 operations instead
// of two ocml calls with large-argument paths: the rollout chain is latency-bound on this.
// Accurate to ~1 ulp for |x| < 1e6.
IRS_HD void irs_sincos(double x, double& s, double& c) {
    const double q = rint(x * 0.63661977236758134308);
    double r = fma(q, -1.57079632679489655800, x);              // pi/2 hi
    r = fma(q, -6.12323399573676603587e-17, r);                 // pi/2 lo
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    ps = fma(ps * z, r, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    pc = fma(pc * z, z, fma(z, -0.5, 1.0));
    const int qi = (int)q;
    const double ss = (qi & 1) ? pc : ps;
    const double cc = (qi & 1) ? ps : pc;
    s = (qi & 2) ? -ss : ss;
    c = ((qi + 1) & 2) ? -cc : cc;
}
IRS_HD float irs_sin(float x) { float s, c; irs_sincos(x, s, c); return s; }
IRS_HD float irs_cos(float x) { float s, c; irs_sincos(x, s, c); return c; }
IRS_HD double irs_sin(double x) { double s, c; irs_sincos(x, s, c); return s; }
IRS_HD double irs_cos(double x) { double s, c; irs_sincos(x, s, c); return c; }

template <typename T, int K> IRS_HD Dual<T, K> make_const(T v) {
    Dual<T, K> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = T(0);
    return r;
}
template <typename T, int K> IRS_HD Dual<T, K> make_var(T v, int idx) {
    Dual<T, K> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = (i == idx) ? T(1) : T(0);
    return r;
}

#define IRS_DUAL_TK template <typename T, int K> IRS_HD
IRS_DUAL_TK Dual<T, K> operator+(const Dual<T, K>& a, const Dual<T, K>& b) {
    Dual<T, K> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
IRS_DUAL_TK Dual<T, K> operator-(const Dual<T, K>& a, const Dual<T, K>& b) {
    Dual<T, K> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
IRS_DUAL_TK Dual<T, K> operator-(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -a.d[i];
    return r;
}
IRS_DUAL_TK Dual<T, K> operator*(const Dual<T, K>& a, const Dual<T, K>& b) {
    Dual<T, K> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
IRS_DUAL_TK Dual<T, K> operator/(const Dual<T, K>& a, const Dual<T, K>& b) {
    Dual<T, K> r;
    T inv = T(1) / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
// mixed with the underlying scalar
IRS_DUAL_TK Dual<T, K> operator+(const Dual<T, K>& a, T b) { Dual<T, K> r = a; r.v = a.v + b; return r; }
IRS_DUAL_TK Dual<T, K> operator+(T b, const Dual<T, K>& a) { return a + b; }
IRS_DUAL_TK Dual<T, K> operator-(const Dual<T, K>& a, T b) { Dual<T, K> r = a; r.v = a.v - b; return r; }
IRS_DUAL_TK Dual<T, K> operator-(T b, const Dual<T, K>& a) { return (-a) + b; }
IRS_DUAL_TK Dual<T, K> operator*(const Dual<T, K>& a, T b) {
    Dual<T, K> r; r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b;
    return r;
}
IRS_DUAL_TK Dual<T, K> operator*(T b, const Dual<T, K>& a) { return a * b; }
IRS_DUAL_TK Dual<T, K> operator/(const Dual<T, K>& a, T b) { return a * (T(1) / b); }
IRS_DUAL_TK Dual<T, K> operator/(T b, const Dual<T, K>& a) { return make_const<T, K>(b) / a; }

IRS_DUAL_TK void irs_sincos(const Dual<T, K>& a, Dual<T, K>& s, Dual<T, K>& c) {
    irs_sincos(a.v, s.v, c.v);
#pragma unroll
    for (int i = 0; i < K; ++i) { s.d[i] = c.v * a.d[i]; c.d[i] = -(s.v * a.d[i]); }
}
IRS_DUAL_TK Dual<T, K> irs_sin(const Dual<T, K>& a) { Dual<T, K> s, c; irs_sincos(a, s, c); return s; }
IRS_DUAL_TK Dual<T, K> irs_cos(const Dual<T, K>& a) { Dual<T, K> s, c; irs_sincos(a, s, c); return c; }
IRS_DUAL_TK Dual<T, K> irs_sqrt(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = sqrt(a.v);
    const T g = T(0.5) / r.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = g * a.d[i];
    return r;
}
#undef IRS_DUAL_TK
IRS_HD float irs_sqrt(float x) { return sqrtf(x); }
IRS_HD double irs_sqrt(double x) { return sqrt(x); }
// value-based selection (piecewise-smooth models): the derivative follows the chosen branch
template <typename S> IRS_HD S irs_select(bool c, const S& a, const S& b) { return c ? a : b; }
