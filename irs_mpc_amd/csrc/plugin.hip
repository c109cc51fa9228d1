// Device-side DynamicalSystem plugin surface (irs_lqr/dynamical_system.py:12-66):
// dynamics_batch and jacobian_xu_batch in float64, one lane per batch element.
// Also owns the library-wide error string and model registry queries.
#include <cstdarg>
#include "irs_common.hpp"

static thread_local char g_err[512] = "";

void irs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

// 64-thread workgroups: a contact step holds hundreds of f64 registers per lane (no spills at 512)
template <class Model>
__global__ __launch_bounds__(64) void dynamics_batch_kernel(ModelParams p, const double* X, const double* U, int B, double* Xn) {
    constexpr int n = Model::NX, m = Model::NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x[n], u[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = X[(size_t)b * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) u[j] = U[(size_t)b * m + j];
    Model::template step<double>(p, x, u, xn);
#pragma unroll
    for (int i = 0; i < n; ++i) Xn[(size_t)b * n + i] = xn[i];
}

template <class Model>
__global__ __launch_bounds__(64) void jacobian_batch_kernel(ModelParams p, const double* X, const double* U, int B, double* J) {
    constexpr int n = Model::NX, m = Model::NU, d = n + m;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x[n], u[m], xn[n], Jl[n * d];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = X[(size_t)b * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) u[j] = U[(size_t)b * m + j];
    model_jacobian<Model, double>(p, x, u, xn, Jl);
#pragma unroll
    for (int q = 0; q < n * d; ++q) J[(size_t)b * n * d + q] = Jl[q];
}

// One f32 lane per u-perturbation, exactly what the FIRST_ORDER sample pass of a contact model evaluates per
// sample (smooth.hip): the step from (float)x, (float)u + du and its active-set derivative block B, plus the
// active set itself -- the per-sample view the parity tests classify borderline samples with.
template <class Model>
__global__ __launch_bounds__(64) void contact_samples_f32_kernel(ModelParams p, const double* x, const double* u,
                                                                 const float* du, int B, float* Xn, float* Bs,
                                                                 int* mask) {
    if constexpr (!Model::HAS_JACOBIAN) {
        constexpr int n = Model::NX, m = Model::NU;
        const int b = blockIdx.x * blockDim.x + threadIdx.x;
        if (b >= B) return;
        float xs[n], us[m], fx[n], Bl[n * m];
#pragma unroll
        for (int i = 0; i < n; ++i) xs[i] = (float)x[i];
#pragma unroll
        for (int j = 0; j < m; ++j) us[j] = (float)u[j] + du[(size_t)b * m + j];
        unsigned mk = 0u;
        irs_contact_step_grad<Model, float, false>(p, xs, us, fx, Bl, nullptr, &mk);
#pragma unroll
        for (int i = 0; i < n; ++i) Xn[(size_t)b * n + i] = fx[i];
#pragma unroll
        for (int q = 0; q < n * m; ++q) Bs[(size_t)b * n * m + q] = Bl[q];
        mask[b] = (int)mk;
    }
}

}  // namespace

extern "C" {

int irs_abi_version(void) { return IRS_ABI_VERSION; }

const char* irs_last_error(void) { return g_err; }

int irs_model_info(int model, int* dim_x, int* dim_u, int* n_params) {
    IRS_DISPATCH_MODEL(model, {
        if (dim_x) *dim_x = Model::NX;
        if (dim_u) *dim_u = Model::NU;
        if (n_params) *n_params = Model::NPARAMS;
    });
    return IRS_OK;
}

int irs_dynamics_batch(int model, const double* params, int n_params, const double* X,
                       const double* U, int B, double* Xn, void* stream) {
    IRS_CHECK_ARG(B > 0 && X && U && Xn, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, {
        hipLaunchKernelGGL((dynamics_batch_kernel<Model>), dim3((B + 63) / 64), dim3(64), 0, st, p, X, U, B, Xn);
    });
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_contact_samples_f32(int model, const double* params, int n_params, const double* x, const double* u,
                            const float* du, int B, float* Xn, float* Bs, int* active_mask, void* stream) {
    IRS_CHECK_ARG(B > 0 && x && u && du && Xn && Bs && active_mask, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    bool contact = false;
    IRS_DISPATCH_MODEL(model, {
        contact = !Model::HAS_JACOBIAN;
        if (contact)
            hipLaunchKernelGGL((contact_samples_f32_kernel<Model>), dim3((B + 63) / 64), dim3(64), 0, st, p, x, u, du,
                               B, Xn, Bs, active_mask);
    });
    if (!contact) {
        irs_set_error("irs_contact_samples_f32: model %d is not a contact model", model);
        return IRS_ERR_UNSUPPORTED;
    }
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_jacobian_xu_batch(int model, const double* params, int n_params, const double* X,
                          const double* U, int B, double* J, void* stream) {
    IRS_CHECK_ARG(B > 0 && X && U && J, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, {
        hipLaunchKernelGGL((jacobian_batch_kernel<Model>), dim3((B + 63) / 64), dim3(64), 0, st, p, X, U, B, J);
    });
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

}  // extern "C"
