// Device functors for the reference's DynamicalSystem plugins
// (irs_lqr/dynamical_system.py:1-66).  One templated `step` per model serves every
// scalar type: float (sample evaluation), double (nominal point, rollouts) and
// Dual<.,n+m> (Jacobians).  Adding a plugin = adding a struct here + a case in
// IRS_DISPATCH_MODEL (irs_common.hpp).
#pragma once
#include "dual.hpp"

#define IRS_MAX_PARAMS 12
struct ModelParams {
    double v[IRS_MAX_PARAMS];
};

// examples/pendulum/pendulum_dynamics.py:46-60 -- semi-implicit Euler.
struct PendulumModel {
    static constexpr int NX = 2, NU = 1, NPARAMS = 1;
    static constexpr bool HAS_JACOBIAN = true;
    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x, const S* u, S* xn) {
        using T = typename scalar_of<S>::type;
        const T h = T(p.v[0]);
        S next_speed = x[1] + h * (u[0] - irs_sin(x[0]));
        xn[0] = x[0] + h * next_speed;
        xn[1] = next_speed;
    }
};

// examples/quadrotor/quadrotor_dynamics.py:40-77 -- explicit Euler on an rpy rigid
// body; x = [xyz, rpy, xyz_d, rpy_d], u = 4 rotor commands.
// params = {h, m, L, g, Ixx, Iyy, Izz, kF, kM} (quadrotor_dynamics.py:22-37).
struct QuadrotorModel {
    static constexpr int NX = 12, NU = 4, NPARAMS = 9;
    static constexpr bool HAS_JACOBIAN = true;
    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x, const S* u, S* xn) {
        using T = typename scalar_of<S>::type;
        const T h = T(p.v[0]), mass = T(p.v[1]), L = T(p.v[2]), g = T(p.v[3]);
        const T Ixx = T(p.v[4]), Iyy = T(p.v[5]), Izz = T(p.v[6]);
        const T kF = T(p.v[7]), kM = T(p.v[8]);
        // :43-49 rotor forces and body moments
        S uF0 = kF * u[0], uF1 = kF * u[1], uF2 = kF * u[2], uF3 = kF * u[3];
        S Fz = uF0 + uF1 + uF2 + uF3;
        S M0 = L * (uF2 + uF3 - uF0 - uF1);
        S M1 = L * (uF1 + uF2 - uF0 - uF3);
        S M2 = kM * (u[1] + u[3] - u[0] - u[2]);
        S sr, cr, sp, cp, sy, cy;
        irs_sincos(x[3], sr, cr);
        irs_sincos(x[4], sp, cp);
        irs_sincos(x[5], sy, cy);
        const S rd0 = x[9], rd1 = x[10], rd2 = x[11];
        // :56 xyz_dd = (R_WB F + Fg)/m; only the third column of Rz Ry Rx (:177-183) matters
        T inv_m = T(1) / mass;
        S ax = (cy * sp * cr + sy * sr) * Fz * inv_m;
        S ay = (sy * sp * cr - cy * sr) * Fz * inv_m;
        S az = cp * cr * Fz * inv_m - g;
        // :59-61 body rates pqr = PhiInv(rpy) rpy_d (:188-199), Euler's equation
        S pb = rd0 - sp * rd2;
        S qb = cr * rd1 + sr * cp * rd2;
        S rb = cr * cp * rd2 - sr * rd1;
        S pd = (M0 - (qb * (Izz * rb) - rb * (Iyy * qb))) * (T(1) / Ixx);
        S qd = (M1 - (rb * (Ixx * pb) - pb * (Izz * rb))) * (T(1) / Iyy);
        S rdd = (M2 - (pb * (Iyy * qb) - qb * (Ixx * pb))) * (T(1) / Izz);
        // :69-71 rpy_dd = Phi pqr_d + (Phi_d . rpy_d) pqr  (Phi :201-212, Phi_d :215-231)
        S icp = T(1) / cp;
        S tp = sp * icp;
        S icp2 = icp * icp;
        S E01 = cr * tp * rd0 + sr * icp2 * rd1;
        S E02 = cr * icp2 * rd1 - sr * tp * rd0;
        S E11 = -(sr * rd0);
        S E12 = -(cr * rd0);
        S E21 = cr * icp * rd0 + sr * sp * icp2 * rd1;
        S E22 = cr * sp * icp2 * rd1 - sr * icp * rd0;
        S rr0 = pd + sr * tp * qd + cr * tp * rdd + E01 * qb + E02 * rb;
        S rr1 = cr * qd - sr * rdd + E11 * qb + E12 * rb;
        S rr2 = sr * icp * qd + cr * icp * rdd + E21 * qb + E22 * rb;
        // :73-77 x + h xdot
#pragma unroll
        for (int i = 0; i < 6; ++i) xn[i] = x[i] + h * x[6 + i];
        xn[6] = x[6] + h * ax;
        xn[7] = x[7] + h * ay;
        xn[8] = x[8] + h * az;
        xn[9] = x[9] + h * rr0;
        xn[10] = x[10] + h * rr1;
        xn[11] = x[11] + h * rr2;
    }

    // ---- the step and the SAMPLE-DEPENDENT part of its Jacobian, derived by hand ---------------------------------
    // (for the first-order sample pass, csrc/smooth.hip: forward-mode duals carry 12 x 16 partials through the step --
    // ~250 live registers, two waves per SIMD, and the pass waits for its own loads; of the 192 entries only these
    // depend on the sample, and the 4 control columns of a row are +-combinations of at most three numbers.)
    // Compact layout, every entry already the entry of d xn / d (x, u) (identity and factor h included):
    //   Jc[4 k + {0,1,2}]  row 6+k (acceleration k = x, y, z): columns roll, pitch, yaw;  Jc[4 k + 3]: its four control
    //                      columns (all equal)
    //   Jc[12 + 5 k + v]   row 9+k (rpy_dd): columns v = roll, pitch, rpy_d[0..2]
    //   Jc[27 + 2 k + {0,1}]  row 9+k, control columns:  J[9+k][12+j] = a_k s0_j + Jc[27+2k] s1_j + Jc[28+2k] s2_j,
    //                      s0 = (-1,-1,+1,+1), s1 = (-1,+1,+1,-1), s2 = (-1,+1,-1,+1), a_0 = h L kF / Ixx, a_1 = a_2 = 0
    // Everything else of the Jacobian is constant: I, and h in the velocity columns of the first six rows.
    static constexpr int NJ = 33;
    template <typename T>
    IRS_HD static void step_jac(const ModelParams& p, const T* x, const T* u, T* xn, T* Jc) {
        const T h = T(p.v[0]), mass = T(p.v[1]), L = T(p.v[2]), g = T(p.v[3]);
        const T Ixx = T(p.v[4]), Iyy = T(p.v[5]), Izz = T(p.v[6]);
        const T kF = T(p.v[7]), kM = T(p.v[8]);
        const T uF0 = kF * u[0], uF1 = kF * u[1], uF2 = kF * u[2], uF3 = kF * u[3];
        const T Fz = uF0 + uF1 + uF2 + uF3;
        const T M0 = L * (uF2 + uF3 - uF0 - uF1);
        const T M1 = L * (uF1 + uF2 - uF0 - uF3);
        const T M2 = kM * (u[1] + u[3] - u[0] - u[2]);
        T sr, cr, sp, cp, sy, cy;
        irs_sincos(x[3], sr, cr);
        irs_sincos(x[4], sp, cp);
        irs_sincos(x[5], sy, cy);
        const T rd0 = x[9], rd1 = x[10], rd2 = x[11];
        const T inv_m = T(1) / mass;
        // accelerations: a = c(rpy) F, F = Fz / m
        const T cax = cy * sp * cr + sy * sr, cay = sy * sp * cr - cy * sr, caz = cp * cr;
        const T F = Fz * inv_m, hF = h * F, hkfm = h * kF * inv_m;
        const T ax = cax * Fz * inv_m, ay = cay * Fz * inv_m, az = caz * Fz * inv_m - g;
        Jc[0] = hF * (sy * cr - cy * sp * sr);
        Jc[1] = hF * (cy * cp * cr);
        Jc[2] = -hF * cay;
        Jc[3] = hkfm * cax;
        Jc[4] = -hF * (sy * sp * sr + cy * cr);
        Jc[5] = hF * (sy * cp * cr);
        Jc[6] = hF * cax;
        Jc[7] = hkfm * cay;
        Jc[8] = -hF * (cp * sr);
        Jc[9] = -hF * (sp * cr);
        Jc[10] = T(0);
        Jc[11] = hkfm * caz;
        // body rates and their partials with respect to v = (roll, pitch, rpy_d0, rpy_d1, rpy_d2)
        const T pb = rd0 - sp * rd2;
        const T qb = cr * rd1 + sr * cp * rd2;
        const T rb = cr * cp * rd2 - sr * rd1;
        const T dpb[5] = {T(0), -cp * rd2, T(1), T(0), -sp};
        const T dqb[5] = {rb, -sr * sp * rd2, T(0), cr, sr * cp};
        const T drb[5] = {-qb, -cr * sp * rd2, T(0), -sr, cr * cp};
        // Euler's equation: pd = (M0 - (Izz - Iyy) qb rb) / Ixx, ... (values in the step's own order of operations)
        const T iIxx = T(1) / Ixx, iIyy = T(1) / Iyy, iIzz = T(1) / Izz;
        const T pd = (M0 - (qb * (Izz * rb) - rb * (Iyy * qb))) * iIxx;
        const T qd = (M1 - (rb * (Ixx * pb) - pb * (Izz * rb))) * iIyy;
        const T rdd = (M2 - (pb * (Iyy * qb) - qb * (Ixx * pb))) * iIzz;
        const T c1 = (Izz - Iyy) * iIxx, c2 = (Ixx - Izz) * iIyy, c3 = (Iyy - Ixx) * iIzz;
        // kinematics: rpy_dd = Phi pqr_d + (Phi_d . rpy_d) pqr
        const T icp = T(1) / cp;
        const T tp = sp * icp;
        const T icp2 = icp * icp;
        const T E01 = cr * tp * rd0 + sr * icp2 * rd1;
        const T E02 = cr * icp2 * rd1 - sr * tp * rd0;
        const T E11 = -(sr * rd0);
        const T E12 = -(cr * rd0);
        const T E21 = cr * icp * rd0 + sr * sp * icp2 * rd1;
        const T E22 = cr * sp * icp2 * rd1 - sr * icp * rd0;
        const T a0 = sr * tp, b0 = cr * tp, a2 = sr * icp, b2 = cr * icp;      // (a1 = cr, b1 = -sr)
        const T rr0 = pd + a0 * qd + b0 * rdd + E01 * qb + E02 * rb;
        const T rr1 = cr * qd - sr * rdd + E11 * qb + E12 * rb;
        const T rr2 = a2 * qd + b2 * rdd + E21 * qb + E22 * rb;
        const T q2 = sp * icp2, tpicp = tp * icp, d_icp2 = T(2) * icp2 * tp, dq2 = icp * (T(1) + T(2) * tp * tp);
        const T dE01[5] = {E02, cr * icp2 * rd0 + sr * d_icp2 * rd1, cr * tp, sr * icp2, T(0)};
        const T dE02[5] = {-E01, cr * d_icp2 * rd1 - sr * icp2 * rd0, -(sr * tp), cr * icp2, T(0)};
        const T dE11[5] = {E12, T(0), -sr, T(0), T(0)};
        const T dE12[5] = {-E11, T(0), -cr, T(0), T(0)};
        const T dE21[5] = {E22, cr * tpicp * rd0 + sr * dq2 * rd1, cr * icp, sr * q2, T(0)};
        const T dE22[5] = {-E21, cr * dq2 * rd1 - sr * tpicp * rd0, -(sr * icp), cr * q2, T(0)};
        const T da0[2] = {cr * tp, sr * icp2}, db0[2] = {-(sr * tp), cr * icp2};
        const T da1[2] = {-sr, T(0)}, db1[2] = {-cr, T(0)};
        const T da2[2] = {cr * icp, sr * tpicp}, db2[2] = {-(sr * icp), cr * tpicp};
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const T dpd = -c1 * (dqb[v] * rb + qb * drb[v]);
            const T dqd = -c2 * (drb[v] * pb + rb * dpb[v]);
            const T drdd = -c3 * (dpb[v] * qb + pb * dqb[v]);
            T r0 = dpd + a0 * dqd + b0 * drdd + dE01[v] * qb + E01 * dqb[v] + dE02[v] * rb + E02 * drb[v];
            T r1 = cr * dqd - sr * drdd + dE11[v] * qb + E11 * dqb[v] + dE12[v] * rb + E12 * drb[v];
            T r2 = a2 * dqd + b2 * drdd + dE21[v] * qb + E21 * dqb[v] + dE22[v] * rb + E22 * drb[v];
            if (v < 2) {
                r0 += da0[v] * qd + db0[v] * rdd;
                r1 += da1[v] * qd + db1[v] * rdd;
                r2 += da2[v] * qd + db2[v] * rdd;
            }
            Jc[12 + v] = (v == 2 ? T(1) : T(0)) + h * r0;
            Jc[17 + v] = (v == 3 ? T(1) : T(0)) + h * r1;
            Jc[22 + v] = (v == 4 ? T(1) : T(0)) + h * r2;
        }
        const T hl = h * L * kF * iIyy, hm = h * kM * iIzz;
        Jc[27] = hl * a0;  Jc[28] = hm * b0;
        Jc[29] = hl * cr;  Jc[30] = -hm * sr;
        Jc[31] = hl * a2;  Jc[32] = hm * b2;
#pragma unroll
        for (int i = 0; i < 6; ++i) xn[i] = x[i] + h * x[6 + i];
        xn[6] = x[6] + h * ax;
        xn[7] = x[7] + h * ay;
        xn[8] = x[8] + h * az;
        xn[9] = x[9] + h * rr0;
        xn[10] = x[10] + h * rr1;
        xn[11] = x[11] + h * rr2;
    }

    // entry q = i d + c of the n x d Jacobian sum of `cnt` samples, from the sums sJc of their compact entries
    template <typename T>
    IRS_HD static T expand_entry(const ModelParams& p, const T* sJc, T cnt, int q) {
        constexpr int d = NX + NU;
        const int i = q / d, c = q - i * d;
        const T h = T(p.v[0]);
        if (i < 6) return c == i ? cnt : (c == 6 + i ? cnt * h : T(0));
        if (i < 9) {
            const int k = i - 6;
            if (c == i) return cnt;
            if (c >= 3 && c < 6) return sJc[4 * k + (c - 3)];
            if (c >= 12) return sJc[4 * k + 3];
            return T(0);
        }
        const int k = i - 9;
        if (c == 3 || c == 4) return sJc[12 + 5 * k + (c - 3)];
        if (c >= 9 && c < 12) return sJc[12 + 5 * k + 2 + (c - 9)];
        if (c >= 12) {
            const int j = c - 12;
            const T s0 = j < 2 ? T(-1) : T(1), s1 = (j == 1 || j == 2) ? T(1) : T(-1), s2 = (j & 1) ? T(1) : T(-1);
            const T a0 = k == 0 ? cnt * (h * T(p.v[2]) * T(p.v[7]) / T(p.v[4])) : T(0);
            return a0 * s0 + sJc[27 + 2 * k] * s1 + sJc[28 + 2 * k] * s2;
        }
        return T(0);
    }

    // the full n x d Jacobian sum of `cnt` samples from the sums of their compact entries (sumJ[q] += ...)
    template <typename T>
    IRS_HD static void expand_jac(const ModelParams& p, const T* sJc, T cnt, T* sumJ) {
        constexpr int d = NX + NU;
        const T h = T(p.v[0]);
#pragma unroll
        for (int q = 0; q < NX * d; ++q) sumJ[q] = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) sumJ[i * d + i] = cnt;
#pragma unroll
        for (int i = 0; i < 6; ++i) sumJ[i * d + 6 + i] = cnt * h;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) sumJ[(6 + k) * d + 3 + c] = sJc[4 * k + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) sumJ[(6 + k) * d + 12 + j] = sJc[4 * k + 3];
            sumJ[(9 + k) * d + 3] = sJc[12 + 5 * k];
            sumJ[(9 + k) * d + 4] = sJc[12 + 5 * k + 1];
#pragma unroll
            for (int c = 0; c < 3; ++c) sumJ[(9 + k) * d + 9 + c] = sJc[12 + 5 * k + 2 + c];
        }
        const T a0 = cnt * (h * T(p.v[2]) * T(p.v[7]) / T(p.v[4]));
        const T s0[4] = {T(-1), T(-1), T(1), T(1)}, s1[4] = {T(-1), T(1), T(1), T(-1)}, s2[4] = {T(-1), T(1), T(-1), T(1)};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sumJ[(9 + k) * d + 12 + j] = (k == 0 ? a0 * s0[j] : T(0)) + sJc[27 + 2 * k] * s1[j] + sJc[28 + 2 * k] * s2[j];
    }
};

// examples/bicycle/bicycle_dynamics.py:47-64 -- explicit Euler on a kinematic bicycle;
// x = [x, y, heading, speed, steering angle], u = [acceleration, steering velocity].
struct BicycleModel {
    static constexpr int NX = 5, NU = 2, NPARAMS = 1;
    static constexpr bool HAS_JACOBIAN = true;
    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x, const S* u, S* xn) {
        using T = typename scalar_of<S>::type;
        const T h = T(p.v[0]);
        S sh, ch, ss, cs;
        irs_sincos(x[2], sh, ch);
        irs_sincos(x[4], ss, cs);
        xn[0] = x[0] + h * (x[3] * ch);
        xn[1] = x[1] + h * (x[3] * sh);
        xn[2] = x[2] + h * (x[3] * (ss / cs));
        xn[3] = x[3] + h * u[0];
        xn[4] = x[4] + h * u[1];
    }
};

// examples/three_cart/three_cart_dynamics.py:22-107 -- three carts on a line, perfectly
// inelastic contact resolved by branching; x = [q1,q2,q3,v1,v2,v3], u = [u1,u3],
// params = {h, d (cart width)}.  The scalar `dynamics` is followed (penetration split in
// halves); the reference's `dynamics_batch` (:175-188) moves each cart by the full depth.
// Non-smooth: Jacobians (dual numbers) are those of the active branch.
struct ThreeCartModel {
    static constexpr int NX = 6, NU = 2, NPARAMS = 2;
    static constexpr bool HAS_JACOBIAN = true;
    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x, const S* u, S* xn) {
        using T = typename scalar_of<S>::type;
        const T h = T(p.v[0]), d = T(p.v[1]);
        // :32-40 semi-implicit velocity then position update
        S v1 = x[3] + h * u[0], v2 = x[4], v3 = x[5] + h * u[1];
        S q1 = x[0] + h * v1, q2 = x[1] + h * v2, q3 = x[2] + h * v3;
        const bool c12 = irs_value(q2 - q1) < d;
        const bool c23 = irs_value(q3 - q2) < d;
        if (c12 && c23) {                       // :48-62 all three stick together
            S qm = (q1 + q2 + q3) * T(1.0 / 3.0);
            S vm = (v1 + v2 + v3) * T(1.0 / 3.0);
            xn[0] = qm - d; xn[1] = qm; xn[2] = qm + d;
            xn[3] = vm; xn[4] = vm; xn[5] = vm;
        } else if (c12) {                       // :64-78 carts 1-2 collide
            S pen = d - (q2 - q1);
            S vm = (v1 + v2) * T(0.5);
            xn[0] = q1 - T(0.5) * pen; xn[1] = q2 + T(0.5) * pen; xn[2] = q3;
            xn[3] = vm; xn[4] = vm; xn[5] = v3;
        } else if (c23) {                       // :80-94 carts 2-3 collide
            S pen = d - (q3 - q2);
            S vm = (v2 + v3) * T(0.5);
            xn[0] = q1; xn[1] = q2 - T(0.5) * pen; xn[2] = q3 + T(0.5) * pen;
            xn[3] = v1; xn[4] = vm; xn[5] = vm;
        } else {                                // :96-104 free motion
            xn[0] = q1; xn[1] = q2; xn[2] = q3;
            xn[3] = v1; xn[4] = v2; xn[5] = v3;
        }
    }
};

#include "contact_models.hpp"

// J (n x (n+m), row-major) = d step / d [x,u] at (x,u), T = float or double.  Analytic models are
// differentiated by dual numbers (the reference: symbolic / forward-mode AD,
// examples/quadrotor/quadrotor_dynamics.py:136-138); contact models through the active constraints of
// their step QP (the reference: q_sim.get_Dq_nextDq / get_Dq_nextDqa_cmd, quasistatic_dynamics.py:184-191).
template <class Model, typename T>
IRS_HD void model_jacobian(const ModelParams& p, const T* x, const T* u, T* xn, T* J) {
    constexpr int n = Model::NX, m = Model::NU, d = n + m;
    if constexpr (Model::HAS_JACOBIAN) {
        using D = Dual<T, d>;
        D xd[n], ud[m], out[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xd[i] = make_var<T, d>(x[i], i);
#pragma unroll
        for (int j = 0; j < m; ++j) ud[j] = make_var<T, d>(u[j], n + j);
        Model::template step<D>(p, xd, ud, out);
#pragma unroll
        for (int i = 0; i < n; ++i) {
            xn[i] = out[i].v;
#pragma unroll
            for (int j = 0; j < d; ++j) J[i * d + j] = out[i].d[j];
        }
    } else {
        T A[n * n], B[n * m];
        irs_contact_step_grad<Model, T, true>(p, x, u, xn, B, A);
#pragma unroll
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int k = 0; k < n; ++k) J[i * d + k] = A[i * n + k];
#pragma unroll
            for (int k = 0; k < m; ++k) J[i * d + n + k] = B[i * m + k];
        }
    }
}
