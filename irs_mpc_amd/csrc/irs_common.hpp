// Shared host-side helpers for the C ABI implementation files.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/irs_hip.h"
#include "models.hpp"

void irs_set_error(const char* fmt, ...);

// tvlqr.hip, for iterate.hip: irs_tvlqr_descent whose launch also writes the fused iterate's info row (row may be null)
int irs_tvlqr_descent_row(int model, const double* params, int n_params, int T, const double* At, const double* Bt,
                          const double* ct, const double* Q, const double* Qd, const double* R, double alpha_R,
                          const double* xd_trj, const double* x0, double* K, double* k, double* x_new, double* u_new,
                          double* cost, int* info, const int* smooth_info, int* row, void* stream);

#define IRS_CHECK_ARG(cond, msg)                               \
    do {                                                       \
        if (!(cond)) {                                         \
            irs_set_error("%s: %s", __func__, msg);            \
            return IRS_ERR_INVALID_ARG;                        \
        }                                                      \
    } while (0)

#define IRS_CHECK_LAUNCH()                                                   \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            irs_set_error("%s: HIP error %s", __func__, hipGetErrorString(e_)); \
            return IRS_ERR_HIP;                                              \
        }                                                                    \
    } while (0)

// Expands BODY once per registered model with `Model` bound to its functor type.
#define IRS_DISPATCH_MODEL(model_id, ...)                                       \
    switch (model_id) {                                                           \
        case IRS_MODEL_PENDULUM: { using Model = PendulumModel; __VA_ARGS__; } break;    \
        case IRS_MODEL_QUADROTOR: { using Model = QuadrotorModel; __VA_ARGS__; } break;  \
        case IRS_MODEL_BICYCLE: { using Model = BicycleModel; __VA_ARGS__; } break;      \
        case IRS_MODEL_THREE_CART: { using Model = ThreeCartModel; __VA_ARGS__; } break; \
        case IRS_MODEL_PLANAR_HAND: { using Model = PlanarHandModel; __VA_ARGS__; } break; \
        case IRS_MODEL_BOX_PIVOT: { using Model = BoxPivotModel; __VA_ARGS__; } break; \
        case IRS_MODEL_BOX_ON_BOX: { using Model = BoxOnBoxModel; __VA_ARGS__; } break; \
        case IRS_MODEL_BOX_PUSH: { using Model = BoxPushModel; __VA_ARGS__; } break; \
        case IRS_MODEL_PLANAR_HAND_EXACT: { using Model = PlanarHandExactModel; __VA_ARGS__; } break; \
        case IRS_MODEL_BOX_PIVOT_EXACT: { using Model = BoxPivotExactModel; __VA_ARGS__; } break; \
        case IRS_MODEL_BOX_PUSH_EXACT: { using Model = BoxPushExactModel; __VA_ARGS__; } break; \
        default:                                                                  \
            irs_set_error("%s: unknown model id %d", __func__, (int)(model_id));  \
            return IRS_ERR_UNSUPPORTED;                                           \
    }

static inline int irs_load_params(int model, const double* params, int n_params, ModelParams* out) {
    int need = -1;
    switch (model) {
        case IRS_MODEL_PENDULUM: need = PendulumModel::NPARAMS; break;
        case IRS_MODEL_QUADROTOR: need = QuadrotorModel::NPARAMS; break;
        case IRS_MODEL_BICYCLE: need = BicycleModel::NPARAMS; break;
        case IRS_MODEL_THREE_CART: need = ThreeCartModel::NPARAMS; break;
        case IRS_MODEL_PLANAR_HAND: need = PlanarHandModel::NPARAMS; break;
        case IRS_MODEL_BOX_PIVOT: need = BoxPivotModel::NPARAMS; break;
        case IRS_MODEL_BOX_ON_BOX: need = BoxOnBoxModel::NPARAMS; break;
        case IRS_MODEL_BOX_PUSH: need = BoxPushModel::NPARAMS; break;
        case IRS_MODEL_PLANAR_HAND_EXACT: need = PlanarHandExactModel::NPARAMS; break;
        case IRS_MODEL_BOX_PIVOT_EXACT: need = BoxPivotExactModel::NPARAMS; break;
        case IRS_MODEL_BOX_PUSH_EXACT: need = BoxPushExactModel::NPARAMS; break;
        default: irs_set_error("unknown model id %d", model); return IRS_ERR_UNSUPPORTED;
    }
    if (params == nullptr || n_params != need) {
        irs_set_error("model %d expects %d params, got %d", model, need, n_params);
        return IRS_ERR_INVALID_ARG;
    }
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < need; ++i) out->v[i] = params[i];
    return IRS_OK;
}
