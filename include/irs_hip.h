/*
 * irs_hip.h -- C ABI of libirs_hip.so, the MI355X (gfx950) implementation of the
 * iRS-LQR inner loop of hjsuh94/irs_mpc.
 *
 * The reference has no FFI: its hot path is Python calling NumPy / Drake.  Each
 * entry point below names the reference interface (file:line, relative to the
 * reference repo root) it replaces; INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *  - Plain C: pointers and sizes only.  No allocation, no retained pointers, no
 *    host synchronisation inside any call: every call enqueues kernels on
 *    `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 *    returns.  All calls are hipGraph-capturable.
 *  - Pointers marked DEV are device (HBM) pointers owned by the caller; pointers
 *    marked HOST are read during the call and not retained.
 *  - All matrices are C-contiguous row-major.  Trajectories, matrices, gains and
 *    costs are float64 (the reference computes in float64); the sample tensors
 *    dx/du, which carry all the bytes, are float32 and the per-sample dynamics
 *    evaluation runs in float32 (`dtype` of the path: f32).
 *  - Return value: IRS_OK (0) or a negative irs_status; irs_last_error() returns
 *    a thread-local message.  Numerical failures inside kernels (non-SPD Gram or
 *    Hessian) are reported through the DEV `info` arrays, LAPACK style.
 *  - `params` HOST: model constants, params[0] = h (step size); see irs_model_info.
 */
#ifndef IRS_HIP_H
#define IRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRS_ABI_VERSION 1

typedef enum irs_status {
    IRS_OK = 0,
    IRS_ERR_INVALID_ARG = -1,
    IRS_ERR_HIP = -2,
    IRS_ERR_UNSUPPORTED = -3,
    IRS_ERR_WORKSPACE = -4
} irs_status;

/* Device models = the reference's DynamicalSystem plugins that exist as device
 * functors (irs_lqr/dynamical_system.py:1-66).                                  */
typedef enum irs_model_id {
    IRS_MODEL_PENDULUM = 0,   /* examples/pendulum/pendulum_dynamics.py:8-127; params = {h}            */
    IRS_MODEL_QUADROTOR = 1,  /* examples/quadrotor/quadrotor_dynamics.py:15-231;
                                 params = {h, m, L, g, Ixx, Iyy, Izz, kF, kM}                          */
    IRS_MODEL_BICYCLE = 2,    /* examples/bicycle/bicycle_dynamics.py:8-132; params = {h}              */
    IRS_MODEL_THREE_CART = 3, /* examples/three_cart/three_cart_dynamics.py:8-107 (scalar `dynamics`:
                                 contact by branching); params = {h, d}                                */
    IRS_MODEL_PLANAR_HAND = 4, /* examples/planar_hand (QuasistaticDynamics over the external simulator,
                                 irs_lqr/quasistatic_dynamics.py:136-164): planar quasi-dynamic contact,
                                 Anitescu convex step; x = [xo, ql1, qr1, yo, ql2, qr2, th] (the reference's order,
                                 planar_hand_analysis.py:61-67), u = [ql1, ql2, qr1, qr2] joint commands;
                                 params = {h, g, mass, R, mu, kp1, kp2, l1, l2, r_link, base_x, pgs_iters};
                                 Jacobian = the step QP's active-set derivative (the simulator's Dq_nextDq |
                                 Dq_nextDqa_cmd, quasistatic_dynamics.py:184-191); the sample-pass modes
                                 ZERO_ORDER_B / FIRST_ORDER perturb u only and return the decoupled (A,B) of
                                 irs_lqr_quasistatic.py:275-284.  PARITY UNPINNED.                       */
    IRS_MODEL_BOX_PIVOT = 5   /* examples/box_pivoting: a 1 m square box on the ground pivoted by a position-
                                 controlled disc; x = [x_h, x_b, y_h, y_b, th_b] (box_pivoting_analysis.py:53-64),
                                 u = commanded hand position; params = {h, g, mass, half, mu, kp, r_hand,
                                 pgs_iters}; same contact scheme and restrictions as the planar hand.       */
    , IRS_MODEL_BOX_ON_BOX = 6 /* examples/box_pushing/analysis/box_on_box.py:11-20: the reference's 1-D
                                 statement of the quasi-dynamic step (x = [x_a, x_u], u = commanded x_a;
                                 params = {h, m, k, pgs_iters}); pins the contact QP code the functors share */
    , IRS_MODEL_BOX_PUSH = 7  /* examples/box_pushing (box_pushing_setup.py:6-19): the box and disc of
                                 box_pivoting seen from above (no gravity, no ground, Kp = 500); x, u as there;
                                 params = {h, mass, inertia, half, mu, kp, r_hand, pgs_iters}.  PINNED by the
                                 simulator data examples/box_pushing/analysis/{xu,dxdu}_quasistatic.npy        */
    , IRS_MODEL_PLANAR_HAND_EXACT = 8 /* IRS_MODEL_PLANAR_HAND with the step QP solved EXACTLY (dual active-set
                                 method, csrc/contact_models.hpp) -- what the reference's simulator does (Gurobi) --
                                 instead of by pgs_iters projected sweeps; same params (pgs_iters ignored)      */
    , IRS_MODEL_BOX_PIVOT_EXACT = 9 /* IRS_MODEL_BOX_PIVOT with the step QP solved exactly, likewise           */
    , IRS_MODEL_BOX_PUSH_EXACT = 10 /* IRS_MODEL_BOX_PUSH with the step QP solved exactly, likewise            */
} irs_model_id;

/* Smoothing estimators.                                                         */
typedef enum irs_smooth_mode {
    IRS_SMOOTH_ZERO_ORDER_AB = 0, /* irs_lqr/irs_lqr_zero_order.py:38-63 (LSQ fit through N one-step evals) */
    IRS_SMOOTH_FIRST_ORDER = 1,   /* irs_lqr/irs_lqr_first_order.py:28-54 (mean of N sampled Jacobians); on a
                                     contact model: calc_AB_first_order, irs_lqr/quasistatic_dynamics.py:193-208
                                     (u-only noise, dx may be NULL; sums = (T, n*m): the B blocks; decoupled)  */
    IRS_SMOOTH_ZERO_ORDER_B = 2   /* irs_lqr/quasistatic_dynamics.py:242-266 (u-only noise: B by LSQ,
                                     A = exact Jacobian at the nominal point)                               */
} irs_smooth_mode;

int irs_abi_version(void);
const char *irs_last_error(void);

/* dim_x, dim_u and the number of model constants (`h, dim_x, dim_u` attributes of
 * irs_lqr/dynamical_system.py:8-10).                                             */
int irs_model_info(int model, int *dim_x, int *dim_u, int *n_params);

/* ---- DynamicalSystem plugin surface (irs_lqr/dynamical_system.py:12-66) ------ */

/* dynamics_batch (:24-38): Xn[b] = f(X[b], U[b]).  X (B,n), U (B,m), Xn (B,n) DEV f64. */
int irs_dynamics_batch(int model, const double *params, int n_params,
                       const double *X, const double *U, int B, double *Xn, void *stream);

/* jacobian_xu_batch (:53-66): J[b] = df/d[x,u] at (X[b],U[b]); J (B,n,n+m) DEV f64.
 * Forward-mode AD of the device functor, like the reference's forwarddiff/symbolic
 * Jacobians (quadrotor_dynamics.py:136-148, pendulum_dynamics.py:110-127); contact models:
 * the derivative of the step QP through its active constraints, contact geometry fixed =
 * QuasistaticDynamics.jacobian_xu (irs_lqr/quasistatic_dynamics.py:184-191).           */
int irs_jacobian_xu_batch(int model, const double *params, int n_params,
                          const double *X, const double *U, int B, double *J, void *stream);

/* Per-sample view of calc_AB_first_order on a contact model (irs_lqr/quasistatic_dynamics.py:193-208: the loop
 * body, one u-perturbation per lane): from the nominal x (n), u (m) DEV f64 and du (B,m) DEV f32, exactly what a
 * lane of the FIRST_ORDER sample pass evaluates in f32 -- Xn (B,n) the step from ((float)x, (float)u + du[b]),
 * Bs (B,n,m) the block Dq_nextDqa_cmd of its active-set derivative, active_mask (B) i32 bit i = contact row i
 * active.  Diagnostics: lets a test separate the samples whose active set differs between the f32 lanes and an
 * f64 evaluation from the rest.  IRS_ERR_UNSUPPORTED for analytic models.                                   */
int irs_contact_samples_f32(int model, const double *params, int n_params, const double *x, const double *u,
                            const float *du, int B, float *Xn, float *Bs, int *active_mask, void *stream);

/* IrsLqr.rollout + evaluate_cost (irs_lqr/irs_lqr.py:105-119, :121-137).
 * x0 (n), u_trj (T,m), Q (n,n), R (m,m), xd_trj (T+1,n) DEV f64 in;
 * x_trj (T+1,n), cost (1) DEV f64 out.  Terminal term uses Q (reference :135-136). */
int irs_rollout_cost(int model, const double *params, int n_params, int T,
                     const double *x0, const double *u_trj, const double *Q,
                     const double *R, const double *xd_trj, double *x_trj,
                     double *cost, void *stream);

/* IrsLqr.evaluate_cost (irs_lqr/irs_lqr.py:121-137) of a GIVEN (x_trj, u_trj) pair:
 * sum_t (x_t-xd_t)'Q(x_t-xd_t) + u_t'R u_t  +  (x_T-xd_T)'Q(x_T-xd_T).  Any n<=32, m<=16. */
int irs_evaluate_cost(int n, int m, int T, const double *x_trj, const double *u_trj,
                      const double *Q, const double *R, const double *xd_trj,
                      double *cost, void *stream);

/* ---- Randomised-smoothing linearisation (get_TV_matrices) -------------------- */

/* Length P of one timestep's sufficient statistics (z = the perturbed components, d = n+m):
 *   ZERO_ORDER_AB: d(d+1)/2 (upper Gram of z=[dx,du]) + d*n (z (f(x+dx,u+du)-f(x,u))')
 *   FIRST_ORDER  : n*d      (sum of Jacobians)
 *   ZERO_ORDER_B : m(m+1)/2 + m*n                                  (z = du)
 * Contact models (IRS_MODEL_PLANAR_HAND, IRS_MODEL_BOX_PIVOT; expensive step) use the layout
 *   [Gram | z (f(x+dx,u+du) - xb)' | sum of z],  P as above + d (resp. m),
 * xb = the f32-rounded x_t: the finalize step subtracts (sum z)(f(x,u) - xb)' with f(x,u) in f64, so
 * no lane of the sample pass evaluates the nominal step.  Sums of shards add in either layout.    */
int irs_sums_len(int model, int mode);

/* Bytes of DEV scratch the irs_smooth* calls need for (T, N) (T <= 1024).        */
size_t irs_smooth_workspace_bytes(int model, int mode, int T, int N);

/* Zeroes the arrival counters at the head of a freshly allocated workspace.  Call
 * once per allocation; every irs_smooth* call leaves them zero again.  One call in
 * flight per workspace at a time.                                                */
int irs_workspace_init(void *workspace, size_t workspace_bytes, void *stream);

/* Whole get_TV_matrices in ONE launch (single-GPU path): sample pass + reduction +
 * least-squares solve.  Samples SUPPLIED: dx (T,N,n), du (T,N,m) DEV f32 -- what the
 * reference's `sampling(x_t,u_t,iter)` closure returned at each t, e.g.
 * examples/pendulum/pendulum_zero_order.py:38-43 ("identical seeds").
 * Outputs as irs_smooth_accumulate (sums) + irs_smooth_finalize (At,Bt,ct,info).
 * Kernels: csrc/smooth.hip (general sample pass), except IRS_MODEL_PLANAR_HAND_EXACT in the u-only modes
 * (IRS_SMOOTH_ZERO_ORDER_B, IRS_SMOOTH_FIRST_ORDER) -- quasistatic_dynamics.py:193-266, the state is not perturbed --
 * which run csrc/smooth_ug.hip: one dual Hessian per timestep, all 2^8 active-set maps tabulated in LDS, every
 * sample's step QP solved exactly by table rows.  Same arguments, statistics layout and results (to f32 rounding);
 * the environment variable IRS_UG=0 (read per call) selects the general kernel for comparison runs.             */
int irs_smooth(int model, const double *params, int n_params, int mode, int T, int N,
               const double *x_trj, const double *u_trj, const float *dx, const float *du,
               double *sums, double *At, double *Bt, double *ct, int *info,
               void *workspace, size_t workspace_bytes, void *stream);

/* Same with on-device Philox draws (see irs_smooth_accumulate_rng).              */
int irs_smooth_rng(int model, const double *params, int n_params, int mode, int T, int N,
                   const double *x_trj, const double *u_trj, const double *std_x,
                   const double *std_u, uint64_t seed, uint32_t iter, double *sums,
                   double *At, double *Bt, double *ct, int *info,
                   void *workspace, size_t workspace_bytes, void *stream);

/* Sample pass, samples SUPPLIED (parity mode; "identical seeds" = the host draws
 * them exactly as the reference's `sampling(x_t,u_t,iter)` closure does, e.g.
 * examples/pendulum/pendulum_zero_order.py:38-43).
 *   x_trj (T+1,n), u_trj (T,m) DEV f64; dx (T,N,n), du (T,N,m) DEV f32
 *   (dx may be NULL for ZERO_ORDER_B); sums (T,P) DEV f64 out.
 * N is the number of samples per timestep held by THIS device; sums of several
 * devices add (one all-reduce of `sums` replaces zmq_parallel_cmp/array_io.py:6-26). */
int irs_smooth_accumulate(int model, const double *params, int n_params, int mode,
                          int T, int N, const double *x_trj, const double *u_trj,
                          const float *dx, const float *du, double *sums,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Sample pass with on-device Philox4x32-10 + Box-Muller draws (throughput mode):
 * z_c ~ N(0, std_c), std = [std_x | std_u] HOST (n+m); the stream is a pure function
 * of (seed, iter, t, sample_offset + i), so results do not depend on how samples
 * are split over devices.  Specification: oracle/irs_oracle.py:device_gaussian_samples. */
int irs_smooth_accumulate_rng(int model, const double *params, int n_params, int mode,
                              int T, int N, const double *x_trj, const double *u_trj,
                              const double *std_x, const double *std_u, uint64_t seed,
                              uint32_t iter, uint64_t sample_offset, double *sums,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Debug/verification: writes the samples irs_smooth_accumulate_rng would draw.   */
int irs_rng_samples(int n, int m, int T, int N, const double *std_x, const double *std_u,
                    uint64_t seed, uint32_t iter, uint64_t sample_offset,
                    float *dx, float *du, void *stream);

/* Solve: sums (T,P) -> At (T,n,n), Bt (T,n,m), ct (T,n) DEV f64,
 * c_t = f(x_t,u_t) - A_t x_t - B_t u_t (irs_lqr_zero_order.py:59-62).
 * N_total = samples per timestep summed over all devices.
 * info (T) DEV int32: 0 ok, j>0 = Gram matrix not positive definite at pivot j.
 * ZERO_ORDER_AB solves the normal equations of compute_least_squares
 * (irs_lqr_zero_order.py:27-36) by Jacobi-scaled Cholesky in f64.                */
int irs_smooth_finalize(int model, const double *params, int n_params, int mode,
                        int T, long long N_total, const double *x_trj, const double *u_trj,
                        const double *sums, double *At, double *Bt, double *ct,
                        int *info, void *stream);

/* irs_smooth_finalize that may reuse work of the preceding irs_smooth_accumulate[_rng] call:
 * `workspace` = the workspace that call was given (same model, mode, T, x_trj, u_trj), or NULL.  For
 * contact models the accumulate launch leaves f(x_t,u_t) (f64, 400 serial PGS updates each) there,
 * evaluated by workgroup 0 of every timestep while the others sample; finalize then skips them.    */
int irs_smooth_finalize_ws(int model, const double *params, int n_params, int mode, int T,
                           long long N_total, const double *x_trj, const double *u_trj,
                           const double *sums, double *At, double *Bt, double *ct, int *info,
                           const void *workspace, size_t workspace_bytes, void *stream);

/* IrsLqrExact.get_TV_matrices (irs_lqr/irs_lqr_exact.py:15-31).                  */
int irs_exact_linearize(int model, const double *params, int n_params, int T,
                        const double *x_trj, const double *u_trj,
                        double *At, double *Bt, double *ct, void *stream);

/* ---- TV-LQR (irs_lqr/tv_lqr.py:30-145, bounds inactive) ---------------------- */

/* Backward Riccati pass of the QP solve_tvlqr poses:
 *   min sum_t (x_t-xd_t)'Q(x_t-xd_t) + alpha_R u_t'R u_t + (x_T-xd_T)'Qd(x_T-xd_T)
 *   s.t. x_{t+1} = A_t x_t + B_t u_t + c_t
 * alpha_R = 0.5 reproduces Drake's AddQuadraticCost (tv_lqr.py:110).
 * Any n <= 32, m <= 16.  K (T,m,n), k (T,m) DEV f64 out: u_t = K_t x_t + k_t.
 * info (1) DEV int32: 0 ok, t+1 = Hessian not positive definite at step t.       */
int irs_tvlqr_riccati(int n, int m, int T, const double *At, const double *Bt,
                      const double *ct, const double *Q, const double *Qd,
                      const double *R, double alpha_R, const double *xd_trj,
                      double *K, double *k, int *info, void *stream);

/* solve_tvlqr's return value (x*, u*): the policy rolled out on the LINEAR model
 * from x0.  x_star (T+1,n), u_star (T,m) DEV f64 out.                            */
int irs_tvlqr_linear_rollout(int n, int m, int T, const double *At, const double *Bt,
                             const double *ct, const double *K, const double *k,
                             const double *x0, double *x_star, double *u_star,
                             void *stream);

/* The forward loop of IrsLqr.local_descent (irs_lqr/irs_lqr.py:169-184): u_t =
 * K_t x_t + k_t (== the re-solved QP's first control), x_{t+1} = f(x_t,u_t) on the
 * TRUE dynamics, plus evaluate_cost of the result.                                */
int irs_closed_loop_rollout(int model, const double *params, int n_params, int T,
                            const double *K, const double *k, const double *x0,
                            const double *Q, const double *R, const double *xd_trj,
                            double *x_new, double *u_new, double *cost, void *stream);

/* irs_tvlqr_riccati + irs_closed_loop_rollout in ONE launch: everything
 * IrsLqr.local_descent does after get_TV_matrices (irs_lqr/irs_lqr.py:169-184) plus
 * evaluate_cost (:121-137) of the new trajectory.  x0 (n) DEV.                    */
int irs_tvlqr_descent(int model, const double *params, int n_params, int T,
                      const double *At, const double *Bt, const double *ct,
                      const double *Q, const double *Qd, const double *R, double alpha_R,
                      const double *xd_trj, const double *x0, double *K, double *k,
                      double *x_new, double *u_new, double *cost, int *info, void *stream);

/* IrsLqr.local_descent (irs_lqr/irs_lqr.py:148-186) with ACTIVE absolute box bounds
 * (solve_tvlqr's x_bound_abs / u_bound_abs, irs_lqr/tv_lqr.py:112-123): T tail QPs, each
 * re-solved from the realised state, first control applied to the true dynamics.  The QPs
 * (OSQP in the reference) are solved by ADMM on the box around one shared Riccati
 * factorisation, warm started from tail to tail.  xlo,xhi (n), ulo,uhi (m) DEV f64, +-inf =
 * unbounded, applied to x_1..x_T and u_0..u_{T-1}.  rho > 0 ADMM penalty, 0 < relax < 2
 * over-relaxation, stop at max(primal, dual residual) < eps or max_iter.
 * info (3) DEV int32: [0] t+1 of a non-PD Hessian (0 ok), [1] most ADMM iterations any tail
 * needed, [2] number of tails that stopped at max_iter.  One launch; the factorisation lives
 * in LDS, so T is limited (irs_tvlqr_box_lds_bytes(model,T) <= ~160 KB).                 */
int irs_tvlqr_box_descent(int model, const double *params, int n_params, int T,
                          const double *At, const double *Bt, const double *ct,
                          const double *Q, const double *Qd, const double *R, double alpha_R,
                          const double *xd_trj, const double *x0,
                          const double *xlo, const double *xhi, const double *ulo, const double *uhi,
                          double rho, double relax, int max_iter, double eps,
                          double *x_new, double *u_new, int *info, void *stream);
size_t irs_tvlqr_box_lds_bytes(int model, int T);

/* IrsLqrQuasistatic.local_descent after get_TV_matrices (irs_lqr/irs_lqr_quasistatic.py:286-345)
 * for a position-controlled model (one with indices_u_into_x, e.g. IRS_MODEL_PLANAR_HAND): T tail
 * re-solves of solve_tvlqr(..., indices_u_into_x, x_bound_abs, u_bound_abs, u_bound_rel)
 * (irs_lqr/tv_lqr.py:30-137), first control applied to the true dynamics, plus eval_cost
 * (:153-194) of the new trajectory.  The input cost is du_t'R du_t with du_t = u_t - u_{t-1},
 * du_0 = u_0 - x_0[indices_u_into_x] (tv_lqr.py:98-108); inside each tail re-solve "u_{-1}" is
 * the realised actuated position.  Solved as the box-LQR of the augmented state [x; u_prev] with
 * the ADMM of irs_tvlqr_box_descent.
 * Bounds, all DEV f64, NULL = absent, +-inf entries allowed:
 *   x_lo,x_hi (T+1,n): ABSOLUTE bounds on x_t   (the caller adds the nominal trajectory to the
 *   u_lo,u_hi (T,m)  : ABSOLUTE bounds on u_t    reference's trust-region offsets, :303-314)
 *   du_lo,du_hi (T,m): bounds on u_t - u_{t-1}  (u_bounds_rel, :321-325)
 * x_bound_rel is not supported ("should be rarely used", :315-319).
 * solver: 1 = the ADMM of irs_tvlqr_box_descent on the augmented problem (any combination of
 *   bounds; rho, relax, max_iter, eps as there);
 *   2 = exact active-set method (primal-dual active set with a primal active-set safeguard) on the
 *   control-box form of the QP: needs x_lo == NULL and at most ONE of the u / du pairs (every
 *   example of the reference); the active set and the backward sweep are carried from tail to
 *   tail; eps = tolerance on bounds and multipliers, max_iter = cap on safeguard iterations per
 *   tail, rho/relax unused; one wave, every quantity of all T steps in LDS (planar hand: T <= 52);
 *   3 = the same method with every step riding in one 16 x 16 matrix-core tile of homogeneous
 *   coordinates (csrc/ctrlbox_mfma.hip; models with n + 2 m + 1 <= 16 after padding: all contact
 *   models here): ~5x faster, and no horizon limit -- per-step records stay in LDS when they fit (planar
 *   hand T <= 53, box pivoting T <= 123) and
 *   otherwise live in the workspace of irs_quasistatic_box_descent_wsx;
 *   0 = 3 where it applies (on chip, or a workspace was given), else 2 where it fits LDS, else 1.
 * cost (1) DEV (may be NULL); info (3): [0] t+1 of a non-PD Hessian, [1] most iterations any tail
 * needed, [2] number of tails that did not converge.                                          */
int irs_quasistatic_box_descent(int model, const double *params, int n_params, int T,
                                const double *At, const double *Bt, const double *ct,
                                const double *Q, const double *Qd, const double *R,
                                const double *xd_trj, const double *x0,
                                const double *x_lo, const double *x_hi,
                                const double *u_lo, const double *u_hi,
                                const double *du_lo, const double *du_hi,
                                int solver, double rho, double relax, int max_iter, double eps,
                                double *x_new, double *u_new, double *cost, int *info, void *stream);
/* The same with the active set of the FIRST tail handed from one iLQR iteration to the next.
 * act_io (T,m) DEV f64 in {-1: at the lower bound, 0: free, +1: at the upper bound}, may be NULL
 * (= irs_quasistatic_box_descent).  In: the set the first tail's solve starts from (all zeros = cold
 * start; any content is valid, the QP's solution does not depend on it); out: the set that tail
 * converged to.  Consecutive iterations of IrsLqrQuasistatic.iterate (irs_lqr_quasistatic.py:349-385)
 * bind nearly the same bounds, so the cold start of tail 0 -- up to 40 % of a descent -- shrinks to a
 * few sweeps.  Used by the active-set solver only (ADMM ignores it).                                  */
int irs_quasistatic_box_descent_ws(int model, const double *params, int n_params, int T,
                                   const double *At, const double *Bt, const double *ct,
                                   const double *Q, const double *Qd, const double *R,
                                   const double *xd_trj, const double *x0,
                                   const double *x_lo, const double *x_hi,
                                   const double *u_lo, const double *u_hi,
                                   const double *du_lo, const double *du_hi,
                                   int solver, double rho, double relax, int max_iter, double eps,
                                   double *x_new, double *u_new, double *cost, int *info,
                                   double *act_io, void *stream);
size_t irs_quasistatic_box_lds_bytes(int model, int T, int solver);
/* solve_tvlqr stand-alone (irs_lqr/tv_lqr.py:30-145): ONE QP, possibly bounded, its plan returned -- what a
 * caller outside the MPC loops gets from the reference function.  `model` only selects the compiled (n, m)
 * (and, position_controlled = 1: indices_u_into_x, tv_lqr.py:96-108, cost on du_t = u_t - u_{t-1} with
 * du_0 = u_0 - x0[idx], full R); its dynamics are not used.  Bounds: per-time rows, DEV f64, NULL = absent,
 * +-inf allowed: x_lo,x_hi (T+1,n) [x_bound_abs, :112-115; row 0 is not enforced: x_0 is data], u_lo,u_hi (T,m)
 * [u_bound_abs, :116-118], du_lo,du_hi (T,m) [u_bound_rel, :122-125; position-controlled form only].
 * x_bound_rel (:119-121) is not supported.  alpha_R: 1/2 for the plain branch (Drake's AddQuadraticCost(R, 0, u),
 * :110), 1 for the position-controlled one (:107).  ADMM around one Riccati factorisation (csrc/boxqp.hip);
 * x_star (T+1,n), u_star (T,m) DEV out; info (3) as for irs_tvlqr_box_descent.                              */
int irs_tvlqr_box_solve(int model, const double *params, int n_params, int T,
                        const double *At, const double *Bt, const double *ct,
                        const double *Q, const double *Qd, const double *R, double alpha_R,
                        const double *xd_trj, const double *x0, int position_controlled,
                        const double *x_lo, const double *x_hi, const double *u_lo, const double *u_hi,
                        const double *du_lo, const double *du_hi,
                        double rho, double relax, int max_iter, double eps,
                        double *x_star, double *u_star, int *info, void *stream);
/* IrsLqrZeroOrder.compute_least_squares (irs_lqr/irs_lqr_zero_order.py:27-36) stand-alone: dxdu (N, n+m),
 * deltaf (N, n) DEV f64 -> A (n,n), B (n,m) with [A | B] = lstsq(dxdu, deltaf)[0]'.  Normal equations in f64,
 * Jacobi-scaled Cholesky (the solve the sample pass ends with).  info (1): 0, the failed pivot (1-based: a
 * rank-deficient design), or n+m+1 for non-finite data.  n <= 32, m <= 16.                                   */
int irs_least_squares(int n, int m, int N, const double *dxdu, const double *deltaf, double *A, double *B,
                      int *info, void *stream);
/* The same with a device workspace for horizons whose per-step records do not fit the 160 KB of LDS
 * (solver 3 / 0): `workspace` DEV, at least irs_quasistatic_descent_workspace_bytes(model, T, solver)
 * bytes (0 = none needed: pass NULL); uninitialised scratch, no state is kept in it between calls.  The
 * reference has no horizon limit (irs_lqr_quasistatic.py:325-345).                                    */
int irs_quasistatic_box_descent_wsx(int model, const double *params, int n_params, int T,
                                    const double *At, const double *Bt, const double *ct,
                                    const double *Q, const double *Qd, const double *R,
                                    const double *xd_trj, const double *x0,
                                    const double *x_lo, const double *x_hi,
                                    const double *u_lo, const double *u_hi,
                                    const double *du_lo, const double *du_hi,
                                    int solver, double rho, double relax, int max_iter, double eps,
                                    double *x_new, double *u_new, double *cost, int *info,
                                    double *act_io, void *workspace, size_t workspace_bytes, void *stream);
size_t irs_quasistatic_descent_workspace_bytes(int model, int T, int solver);

/* ---- Cross-entropy-method baseline (irs_lqr/cem.py:151-184) -------------------- */

/* Steps 1-2 of CrossEntropyMethod.local_descent (cem.py:163-168): roll out each of the
 * B candidate control sequences u_cand (B,T,m) DEV f64 from x0 on the true dynamics and
 * evaluate its cost (evaluate_cost, cem.py:121-140) -> costs (B) DEV f64.             */
int irs_cem_rollout_costs(int model, const double *params, int n_params, int T, int B,
                          const double *u_cand, const double *x0, const double *Q,
                          const double *R, const double *xd_trj, double *costs, void *stream);

/* Steps 1-2 of CrossEntropyMethodQuasistatic.local_descent (irs_lqr/cem_quasistatic.py:186-200)
 * for a position-controlled model: as irs_cem_rollout_costs, but every candidate is priced with
 * the quasistatic eval_cost (:124-165): state error with Q, TERMINAL Qd, input cost on
 * u_t - u_{t-1} (u_{-1} = x_0[indices_u_into_x]).                                              */
int irs_cem_rollout_costs_quasistatic(int model, const double *params, int n_params, int T, int B,
                                      const double *u_cand, const double *x0, const double *Q,
                                      const double *Qd, const double *R, const double *xd_trj,
                                      double *costs, void *stream);

/* Steps 3-4 (cem.py:173-180): the n_elite cheapest candidates (np.argpartition) ->
 * elite_idx (n_elite) DEV int32 (cheaper-than-threshold candidates in increasing index
 * order, then threshold ties, lowest index first: deterministic), their mean -> u_new (T,m) and
 * population standard deviation -> std_new (T,m), DEV f64.  Any m, T.               */
int irs_cem_refit(int T, int m, int B, int n_elite, const double *u_cand, const double *costs,
                  int *elite_idx, double *u_new, double *std_new, void *stream);

/* ---- Pre-marshalled calls ------------------------------------------------------
 * The same operations with their arguments packed in a caller-owned struct, so that a
 * host loop (IrsLqr.iterate, irs_lqr/irs_lqr.py:188-218) pays one pointer-sized FFI
 * call per launch instead of marshalling ~20 scalars.  Fields have the meaning of the
 * like-named parameters above.                                                      */
typedef struct irs_smooth_call {
    int model, n_params;
    double params[12];
    int mode, T, N;
    const double *x_trj, *u_trj;     /* DEV */
    const float *dx, *du;            /* DEV; used when use_rng == 0 */
    int use_rng;                     /* 1: draw on device from std/seed/iter/sample_offset */
    uint32_t iter;
    double std_x[32], std_u[16];
    uint64_t seed, sample_offset;
    double *sums;                    /* DEV (T,P) out */
    double *At, *Bt, *ct;            /* DEV out; all NULL = accumulate only */
    int *info;                       /* DEV (T) out (when At != NULL) */
    long long n_total;               /* samples per timestep over all devices (solve) */
    void *workspace;
    size_t workspace_bytes;
} irs_smooth_call;

/* irs_smooth / irs_smooth_rng / irs_smooth_accumulate[_rng], by struct.           */
int irs_smooth_run(const irs_smooth_call *call, void *stream);

typedef struct irs_descent_call {
    int model, n_params;
    double params[12];
    int T;
    double alpha_R;
    const double *At, *Bt, *ct, *Q, *Qd, *R, *xd_trj, *x0;   /* DEV */
    double *K, *k, *x_new, *u_new, *cost;                     /* DEV out */
    int *info;                                                /* DEV (1) out */
} irs_descent_call;

/* irs_tvlqr_descent, by struct.                                                    */
int irs_descent_run(const irs_descent_call *call, void *stream);

/* ---- IrsLqr.iterate as one call (csrc/iterate.hip) ---------------------------------------------------------
 * irs_lqr/irs_lqr.py:188-218: `n_descents` descents (iterate(k) performs k + 1), each = get_TV_matrices (device-drawn
 * samples: irs_smooth_rng; or IRS_ITERATE_EXACT: irs_exact_linearize) -> irs_tvlqr_descent -> with box bounds:
 * irs_tvlqr_plan_within_bounds and, behind its device-side flag, irs_tvlqr_box_descent_if -- all enqueued back to back
 * on `stream`, no host synchronisation between phases or iterations.  Descent i linearises around the trajectory
 * descent i-1 wrote into x_hist / u_hist (descent 0: x_trj0 / u_trj0); the caller reads the histories back once.
 * std_x (n_descents, n) / std_u (n_descents, m): HOST, the sampling schedule evaluated by the caller (e.g.
 * sigma / iter^p, examples/pendulum/pendulum_zero_order.py:38-43); Philox iteration counter of descent i = iter0 + i.
 * info_hist (n_descents, 8) DEV int32 per descent: [0] Riccati info (t+1 of a non-PD Hessian), [1] number of
 * timesteps whose smoothing solve failed, [2] 1 = some tail's unconstrained plan left the box (the bounded descent
 * ran), [3..5] its info (irs_tvlqr_box_descent), [6] 1 = it was needed but the horizon does not fit its kernel.
 * scratch: DEV, >= irs_iterate_scratch_bytes(model, mode, T, N).  timing (optional, HOST out): per-phase device
 * time summed over the descents -- timing mode synchronises after every descent; NULL = fully asynchronous.   */
#define IRS_ITERATE_EXACT 3
typedef struct irs_iterate_call {
    int model, n_params;
    double params[12];
    int mode, T, N, n_descents;
    const double *std_x, *std_u;               /* HOST (n_descents, n) / (n_descents, m); std_x NULL in u-only modes */
    uint64_t seed;
    uint32_t iter0;
    const double *Q, *Qd, *R, *xd_trj;         /* DEV */
    double alpha_R;
    const double *xlo, *xhi, *ulo, *uhi;       /* DEV (n) / (m), all four or none */
    double qp_rho, qp_relax, qp_eps;           /* bounded descent (ADMM); <= 0: defaults 10, 1.6, 1e-8 */
    int qp_max_iter;                           /* <= 0: 5000 */
    const double *x_trj0, *u_trj0;             /* DEV (T+1, n), (T, m): the nominal trajectory of descent 0 */
    double *x_hist, *u_hist, *cost_hist;       /* DEV out (n_descents, T+1, n), (n_descents, T, m), (n_descents) */
    int *info_hist;                            /* DEV out (n_descents, 8) */
    void *scratch;
    size_t scratch_bytes;
} irs_iterate_call;

/* what SURVEY 8(b) calls irs_get_timing: filled by irs_iterate when asked for */
typedef struct irs_timing {
    double linearise_ms, descent_ms, bounds_ms;   /* device time per phase, summed over the descents */
    int descents;
    double sample_steps;                          /* N x T x descents one-step evaluations */
    double sample_bytes;                          /* the f32 perturbations those would stream if supplied */
} irs_timing;

size_t irs_iterate_scratch_bytes(int model, int mode, int T, int N);
int irs_iterate(const irs_iterate_call *call, irs_timing *timing, void *stream);

/* *flag (DEV int32) = 1 iff the unconstrained plan of SOME tail QP (start t, realised state x_new[t], policy K, k rolled
 * out on the linear model) violates the box xlo <= x <= xhi (n), ulo <= u <= uhi (m): only then do the bounded QPs
 * of irs_lqr/tv_lqr.py:112-123 differ from the Riccati descent.  n <= 32, m <= 16.                              */
int irs_tvlqr_plan_within_bounds(int n, int m, int T, const double *At, const double *Bt, const double *ct,
                                 const double *K, const double *k, const double *x_new,
                                 const double *xlo, const double *xhi, const double *ulo, const double *uhi,
                                 int *flag, void *stream);

/* irs_tvlqr_box_descent that (a) also returns evaluate_cost of its result (cost, may be NULL) and (b) returns at
 * once, writing nothing, when *run_flag (DEV int32, may be NULL = always run) is 0.                            */
int irs_tvlqr_box_descent_if(int model, const double *params, int n_params, int T,
                             const double *At, const double *Bt, const double *ct,
                             const double *Q, const double *Qd, const double *R, double alpha_R,
                             const double *xd_trj, const double *x0,
                             const double *xlo, const double *xhi, const double *ulo, const double *uhi,
                             double rho, double relax, int max_iter, double eps,
                             double *x_new, double *u_new, double *cost, int *info, const int *run_flag,
                             void *stream);

/* ---- the multi-GPU smoothing step inside the library (csrc/collective.hip) ------------------------------
 * One process per GPU; samples sharded over the ranks; per step: sample pass -> ONE all-reduce of the
 * (T,P) f64 statistics (RCCL over xGMI) -> solve (every rank, redundantly).  Replaces the reference's ZeroMQ
 * worker pool (zmq_parallel_cmp/array_io.py:6-26, irs_lqr/irs_lqr_quasistatic.py:245-263).  RCCL is bound at
 * run time (the copy torch.distributed's "nccl" backend loaded, if any).
 *   irs_comm_available   a rank-local, non-collective pre-check a launcher makes before the collective create
 *   irs_comm_unique_id   rank 0 fills 128 bytes (ncclGetUniqueId); the host distributes them to all ranks
 *   irs_comm_create      collective: every rank, with its device current (ncclCommInitRank)
 *   irs_allreduce_sums   in-place f64 SUM all-reduce of `count` doubles on `stream`
 *   irs_smooth_step_collective   the three enqueues (accumulate, all-reduce, solve) on `stream`; `call` must
 *                        carry sums AND At/Bt/ct/info, n_total = samples per timestep over all ranks, and, for
 *                        device-drawn samples, sample_offset = this rank's first global sample; comm may be NULL
 *                        (one rank: no collective)
 *   irs_step_graph_*     the same three enqueues captured ONCE into a HIP graph (on `stream`, which must not be
 *                        the default stream) and replayed with one call per step: the step is ~70-100 us of
 *                        device work in three launches, which a host issuing them one by one cannot keep fed */
int irs_comm_available(void);   /* 1 if RCCL could be bound in this process (no communicator is made), else 0 */
int irs_comm_unique_id(void *id128);
int irs_comm_create(const void *id128, int nranks, int rank, void **comm);
int irs_comm_destroy(void *comm);
int irs_allreduce_sums(void *comm, double *sums, size_t count, void *stream);
int irs_smooth_step_collective(const irs_smooth_call *call, void *comm, void *stream);
int irs_step_graph_create(const irs_smooth_call *call, void *comm, void *stream, void **graph_exec);
int irs_step_graph_launch(void *graph_exec, void *stream);
int irs_step_graph_destroy(void *graph_exec);

/* ---- the same step with the exchange done WITHOUT a collective library (csrc/collective.hip; opt-in) ----------
 * Replaces the same ZeroMQ fan-in as above.  Every rank publishes its (T,P) statistics in an exchange region of
 * its own device memory and reads the other ranks' regions, mapped by IPC handle, directly over xGMI: one small
 * launch between the sample pass and the solve (publish with system-scope stores -> release the step flag -> wait,
 * bounded, for every rank's flag -> sum the blocks in RANK ORDER: the same bits on every rank).  Two alternating
 * slots per region; a peer that does not arrive within IRS_PEER_TIMEOUT_MS (default 2000) poisons the statistics
 * (NaN), which the solve reports through `info` -- the job fails, the GPU does not hang.
 *   irs_peer_alloc       `region` = device memory for `count` doubles per slot (zeroed), `handle64` = its 64-byte
 *                        IPC handle; the host gathers the handles of all ranks (any rendezvous)
 *   irs_peer_create      maps the other ranks' regions; handles = nranks x 64 bytes in rank order (own entry unused)
 *   irs_peer_allreduce_sums   the exchange launch alone: `sums` (count doubles) in place
 *   irs_smooth_step_peer / irs_step_graph_create_peer   as irs_smooth_step_collective / irs_step_graph_create
 *   irs_peer_status      launches and timeouts so far (synchronising read)
 *   irs_peer_destroy     unmaps, frees the counters and, if given, the region (call on every rank after a barrier)
 * Every rank must issue the same sequence of exchange launches.  Not yet run on more than one physical GPU. */
int irs_peer_alloc(size_t count, void **region, void *handle64);
int irs_peer_create(int nranks, int rank, void *region, size_t count, const void *handles, void **peer);
int irs_peer_destroy(void *peer, void *region);
int irs_peer_status(void *peer, unsigned long long *launches, unsigned long long *timeouts);
int irs_peer_allreduce_sums(void *peer, double *sums, size_t count, void *stream);
int irs_smooth_step_peer(const irs_smooth_call *call, void *peer, void *stream);
int irs_step_graph_create_peer(const irs_smooth_call *call, void *peer, void *stream, void **graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* IRS_HIP_H */
