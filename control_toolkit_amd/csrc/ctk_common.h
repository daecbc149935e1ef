// ctk_common.h — types shared by the host C-ABI (ctk_api.hip) and the device kernels.
// gfx950 only: wave = 64 lanes, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ctk_hip.h"

constexpr int CTK_S = 4;          // CartPole: num_states (position, positionD, angle, angleD) — the hand-tuned kernels
constexpr int CTK_C = 1;          // CartPole: num_control_inputs     (other environments: csrc/ctk_env.h, ctk_generic.hip)
constexpr int CTK_WAVE = 64;
constexpr int CTK_MLP_IN = 5, CTK_MLP_H = 32, CTK_MLP_OUT = 4;
constexpr int CTK_MLP_NW = CTK_MLP_IN * CTK_MLP_H + CTK_MLP_H + CTK_MLP_H * CTK_MLP_H + CTK_MLP_H +
                           CTK_MLP_H * CTK_MLP_OUT + CTK_MLP_OUT;  // 1380
constexpr int CTK_MLP_TRAJ_PER_WAVE = 16;   // MFMA 16x16x4: trajectories are the 16 columns

// Derived fp32 constants of the cart-pole step and of the cost.  Computed on the host in double
// and rounded once — the same expressions as oracle/ctk_oracle.py:derived_constants.
struct EnvK {
    float dt, u_max, g, M_fric, inv_mt, k_ml, k_jf, k43l, k_mpl_mt;            // dynamics
    float inv_xs, ep_c, ccR, target_position, dd_weight, ekp_weight, ccrc_weight, terminal_weight;  // cost
    float dd_c;               // dd_weight / x_scale^2 (the recurrence's folded form)
    int intermediate_steps;
};

inline EnvK derive_constants(const float* p /* CTK_P_COUNT primary params (fp32) */, float dt, int isteps) {
    auto d = [&](int id) { return (double)p[id]; };
    EnvK k;
    double inv_mt = 1.0 / (d(CTK_P_M_CART) + d(CTK_P_M_POLE));
    double ml = d(CTK_P_M_POLE) * d(CTK_P_L);
    k.dt = (float)((double)dt / isteps);
    k.u_max = (float)d(CTK_P_U_MAX);
    k.g = (float)d(CTK_P_G);
    k.M_fric = (float)d(CTK_P_M_FRIC);
    k.inv_mt = (float)inv_mt;
    k.k_ml = (float)ml;
    k.k_jf = (float)(d(CTK_P_J_FRIC) / ml);
    k.k43l = (float)(d(CTK_P_L) * (4.0 / 3.0));
    k.k_mpl_mt = (float)(ml * inv_mt);
    k.inv_xs = (float)(1.0 / d(CTK_P_X_SCALE));
    k.dd_c = (float)(d(CTK_P_DD_WEIGHT) / (d(CTK_P_X_SCALE) * d(CTK_P_X_SCALE)));
    k.ep_c = (float)(d(CTK_P_EP_WEIGHT) * d(CTK_P_TARGET_EQUILIBRIUM) * 0.25);
    k.ccR = (float)(d(CTK_P_CC_WEIGHT) * d(CTK_P_R));
    k.target_position = p[CTK_P_TARGET_POSITION];
    k.dd_weight = p[CTK_P_DD_WEIGHT];
    k.ekp_weight = p[CTK_P_EKP_WEIGHT];
    k.ccrc_weight = p[CTK_P_CCRC_WEIGHT];
    k.terminal_weight = p[CTK_P_TERMINAL_WEIGHT];
    k.intermediate_steps = isteps;
    return k;
}

// Column t of the reference's interpolation matrix (others/Interpolator.py:53-77) has at most
// two non-zeros: u[t] = y[i0]*w0 + y[i0+1]*w1.  Built on the host exactly like the matrix
// (fp32 (p-j)/p, including the closing-row quirk) — see ctk_api.hip:build_interp_table.
struct InterpEntry {
    int i0;
    float w0, w1;
};

// MPPI scalars (optimizer_mppi.py:154-168), pre-combined in the reference's evaluation order.
struct MppiK {
    float stdev;        // SQRTRHOINV / sqrt(dt)                       (:130)
    float k_dd;         // (0.5*(1-1/NU))*R   multiplies delta_u^2     (:155)
    float R;            //                    multiplies u*delta_u
    float k_uu;         // 0.5*R              multiplies u^2
    float cc;           // cc_weight
    float neg_inv_lbd;  // (float)(-1.0/LBD)                           (:165)
};

// Arguments every rollout kernel takes by value (no H2D copy for s / u_prev).
struct RolloutArgs {
    float s0[CTK_MAX_STATES];
    float u_prev[CTK_MAX_INPUTS];   // used when u_prev_dev == nullptr
    const float* u_prev_dev;        // optimizer's own last output [C], device resident
    float lo[CTK_MAX_INPUTS], hi[CTK_MAX_INPUTS];   // control limits per input
    int C;                          // num_control_inputs of the environment (1 in the CartPole kernels)
    int N, H, P;
    uint32_t p_magic;        // ceil(2^32 / P): flat / P == umulhi(flat, p_magic) for flat*P < 2^32 (P >= 2)
    int identity_interp;     // period == 1: u[t] = y[t] (column t of the matrix is e_t)
    float inv_Hp1;           // 1/(H+1): mean over [H stage costs | terminal], Cost_Functions/__init__.py:92
    const InterpEntry* interp;  // [H]
    float* J;                // [N]
    float* Q_out;            // [N,H] (u_run), nullable
    float* traj_out;         // [N,H+1,4], nullable
    // counter-based RNG (samples == nullptr)
    uint32_t seed_lo, seed_hi, call, stream_id;
    int global_row0;
    int fast_cos_ok;         // network predictors: every angle the cost will see is inside the unchecked cos range (host-side bound)
    unsigned long long* stamps;   // diagnostic builds only (-DCTK_STAMPS): 8 s_memtime stamps per block
};

// The caller-supplied previous input (wave-uniform), through an SGPR: u_prev_dev is read with a vector load, and a loop
// that first uses the value inherits the load's s_waitcnt vmcnt(0) — which then also waits for the loop's own stores.
__device__ __forceinline__ float uniform_u_prev0(const RolloutArgs& a) {
    const float v = a.u_prev_dev ? *a.u_prev_dev : a.u_prev[0];
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// s0[g] for a per-lane g in 0..3 by selects: a dynamically indexed kernarg array is read with a VECTOR load (through a
// scratch copy), and the s_waitcnt vmcnt(0) guarding that one load lands INSIDE the consumer's loop, where it also waits
// for every store the loop issued (seen in ctk_rpgd_mlp_wide: each step stalled on its tape stores, 900 vs 600 ns).
__device__ __forceinline__ float lane_state4(const RolloutArgs& a, int g, int base = 0) {
    const float lo = g == 0 ? a.s0[base] : a.s0[base + 1], hi = g == 2 ? a.s0[base + 2] : a.s0[base + 3];
    return g < 2 ? lo : hi;
}
