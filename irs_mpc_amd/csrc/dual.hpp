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
template <typename T, int K> struct scalar_of<Dual<T, K>> { using type = T; };

IRS_HD float irs_sin(float x) { return sinf(x); }
IRS_HD float irs_cos(float x) { return cosf(x); }
IRS_HD double irs_sin(double x) { return sin(x); }
IRS_HD double irs_cos(double x) { return cos(x); }

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

IRS_DUAL_TK Dual<T, K> irs_sin(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = irs_sin(a.v);
    T c = irs_cos(a.v);
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = c * a.d[i];
    return r;
}
IRS_DUAL_TK Dual<T, K> irs_cos(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = irs_cos(a.v);
    T s = -irs_sin(a.v);
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = s * a.d[i];
    return r;
}
#undef IRS_DUAL_TK
