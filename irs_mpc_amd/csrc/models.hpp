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
