// Shared between the two bounded TV-LQR kernels (boxqp.hip: ADMM; ctrlbox.hip: active set).
#pragma once
#include <type_traits>

#include "irs_common.hpp"

struct BoxArgs {
    ModelParams p;
    const double *At, *Bt, *ct, *Q, *Qd, *R, *xd, *x0;
    // bounds: row t at ptr + t * stride (stride 0 = one constant row); null = unbounded; +-inf ok
    const double *xlo, *xhi;                 // on x_t, t = 0..T   (row 0 unused: x_0 is fixed)
    const double *ulo, *uhi;                 // on u_t, t = 0..T-1
    const double *dlo, *dhi;                 // DU only: on u_t - u_{t-1}
    int sx, su, sd;
    double *x_new, *u_new, *cost;            // cost may be null
    int* info;                               // [0] Hessian not PD at t+1, [1] max ADMM iterations used,
                                             // [2] number of tail problems that hit max_iter
    double alpha, rho, relax, eps;
    int T, max_iter;
    // active-set solver only (may be null): (T,m) in {-1 at lo, 0 free, +1 at hi}.  In: the active set
    // the FIRST tail starts from (zeros = cold start); out: the set that tail converged to -- what the
    // next iLQR iteration's descent should start from
    double* act_io;
    // ADMM kernel only: 1 = solve the FIRST tail problem alone and return its plan (x*, u*) in x_new / u_new --
    // the stand-alone solve_tvlqr (irs_lqr/tv_lqr.py:30-145); no true-dynamics step is taken
    int single_tail;
    // ADMM kernel only (may be null): DEV int; the kernel returns at once, touching nothing, when *run_flag == 0 --
    // the fused iterate (iterate.hip) enqueues the bounded descent behind the test "does any tail's unconstrained plan
    // leave the box", without a host round trip
    const int* run_flag;
};

// position-controlled models expose indices_u_into_x (quasistatic_dynamics.py:57-65)
template <class M, class = void>
struct has_u_into_x : std::false_type {};
template <class M>
struct has_u_into_x<M, std::void_t<decltype(M::u_into_x(0))>> : std::true_type {};


// ctrlbox.hip: active-set solver for the quasistatic descent with ONE control box.
// kind 0: bounds on u_t (a.ulo/a.uhi), kind 1: bounds on u_t - u_{t-1} (a.dlo/a.dhi); a bound pair
// may be null (unbounded).  Returns IRS_ERR_UNSUPPORTED (message set) if the model is not position
// controlled or the horizon does not fit LDS.
int irs_ctrlbox_launch(int model, const BoxArgs& a, int kind, hipStream_t st);
size_t irs_ctrlbox_lds_bytes(int model, int T);

// ctrlbox_mfma.hip: the same active-set method with every step riding in one 16 x 16 matrix-core tile.
// record_bytes: size of the per-step records (0 = the model does not fit the tile); lds_bytes: LDS needed
// to keep them on chip (beyond the CU's 160 KB the caller supplies `ws`, >= record_bytes, in HBM).
size_t irs_ctrlbox_mfma_record_bytes(int model, int T);
size_t irs_ctrlbox_mfma_lds_bytes(int model, int T);
int irs_ctrlbox_mfma_launch(int model, const BoxArgs& a, int kind, double* ws, size_t ws_bytes, hipStream_t st);
