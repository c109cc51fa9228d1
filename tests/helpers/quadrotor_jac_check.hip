// Host-side check of QuadrotorModel::step_jac (the hand-derived, compact Jacobian of csrc/models.hpp) against the
// forward-mode dual-number Jacobian of the same step, in f64 on random states.  No HIP API call: runs without a GPU.
//   hipcc -std=c++17 --offload-arch=gfx950 -I irs_mpc_amd/csrc -o /tmp/qjc tests/helpers/quadrotor_jac_check.hip && /tmp/qjc
#include <cmath>
#include <cstdio>
#include <random>

#include "models.hpp"

int main() {
    constexpr int n = QuadrotorModel::NX, m = QuadrotorModel::NU, d = n + m;
    ModelParams p;
    const double pv[9] = {0.05, 0.5, 0.25, 9.81, 0.0023, 0.0023, 0.004, 1.0, 0.0245};
    for (int i = 0; i < 9; ++i) p.v[i] = pv[i];
    std::mt19937_64 gen(7);
    std::normal_distribution<double> nd(0.0, 1.0);
    double worst = 0.0, worst_f = 0.0, scale = 0.0;
    for (int trial = 0; trial < 2000; ++trial) {
        double x[n], u[m];
        for (int i = 0; i < n; ++i) x[i] = nd(gen) * (i >= 3 && i < 6 ? 0.6 : 1.5);
        for (int j = 0; j < m; ++j) u[j] = 2.0 + nd(gen);
        double f1[n], J1[n * d];
        model_jacobian<QuadrotorModel, double>(p, x, u, f1, J1);
        double f2[n], Jc[QuadrotorModel::NJ], J2[n * d];
        QuadrotorModel::step_jac<double>(p, x, u, f2, Jc);
        QuadrotorModel::expand_jac<double>(p, Jc, 1.0, J2);
        for (int q = 0; q < n * d; ++q) {
            worst = std::fmax(worst, std::fabs(J1[q] - J2[q]) / (1.0 + std::fabs(J1[q])));
            scale = std::fmax(scale, std::fabs(J1[q]));
        }
        for (int i = 0; i < n; ++i) worst_f = std::fmax(worst_f, std::fabs(f1[i] - f2[i]) / (1.0 + std::fabs(f1[i])));
    }
    std::printf("max rel |J_dual - J_hand| = %.3e (largest entry %.3e), max rel |f - f| = %.3e\n", worst, scale, worst_f);
    return (worst < 1e-11 && worst_f < 1e-13) ? 0 : 1;
}
