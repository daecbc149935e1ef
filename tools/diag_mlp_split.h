// diag_mlp_split.h — the two-wave form of the MLP step (an experiment of round 2, measured by tools/diag_mlp_step.hip and, wired into
// ctk_mppi_rollout, at BASELINE cfg5's shard: 99 us against 94 us for the one-wave form) — kept with the
// diagnostic tool, not in the product.
#pragma once
#include "ctk_mlp.h"

// =============================================================================================
// Two-wave form of the same network: the 16-trajectory tile is shared by a PAIR of waves (two SIMDs of one CU), wave m
// owning hidden-unit tile m (units 16m .. 16m+15) of both hidden layers:
//   layer 1   2 MFMAs  -> tanh (4 units per lane)            — swap the halves through LDS (16 B per lane each way)
//   layer 2   8 MFMAs over all 32 inputs, own output tile    -> tanh
//   layer 3   4 MFMAs over the wave's own 16 inputs (split K) — swap the partial outputs, add in a fixed order
// 14 MFMAs + 8 tanh per wave and step instead of 28 + 16, two barriers.  A lone wave's step is a dependent chain
// (layer n+1 needs layer n), so when the launch has fewer tiles than the chip has SIMDs (BASELINE cfg5's 8192-rollout
// shard: 512 tiles on 1024 SIMDs; cfg4: 16 tiles) halving the chain per wave is what shortens the kernel.
// Inputs are general: the layer-1 B operand of k-step 0 / 1 is the value of network input 0+g / 4+g (lane group g), i.e.
// any environment with S + C <= 8 inputs and S <= 8 outputs fits; outputs come back as (component g, component 4+g).
// Both waves end a step with bit-identical outputs (same instructions on the same data in the same order).
// =============================================================================================
struct MlpFwdWS {      // wave m's share of the forward operands (same per-lane table as MlpFwdW)
    float w1[2];       // [k-step]
    float w2[8];       // [k-step]   output tile m
    float w3[4];       // [k-step within input tile m]
    f32x4 b1, b2, b3;
};

CTK_DEV MlpFwdWS mlp_load_fwd_split(const float* __restrict__ wperm, int m) {
    const float* p = wperm + (threadIdx.x & 63) * MLP_FWD_PER_LANE;
    MlpFwdWS w;
    w.w1[0] = p[2 * m]; w.w1[1] = p[2 * m + 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) w.w2[j] = p[4 + 8 * m + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) w.w3[j] = p[20 + 4 * m + j];
#pragma unroll
    for (int r = 0; r < 4; ++r) { w.b1[r] = p[28 + 4 * m + r]; w.b2[r] = p[36 + 4 * m + r]; w.b3[r] = p[44 + r]; }
    return w;
}

// LDS of one pair: h1 halves [2][64] float4, partial outputs [2][64] float2
constexpr int MLP_PAIR_EX_FLOATS = 2 * 64 * 4 + 2 * 64 * 2;

// One predictor step by a pair of waves (both call it; m = wave's tile; ex = the pair's exchange slots).  x0 / x1:
// layer-1 B operands of k-steps 0 / 1.  keep_h1 / keep_h2: the wave's OWN activation tiles (reverse mode tapes them).
CTK_DEV MlpPair mlp_step_split(const MlpFwdWS& w, float x0, float x1, float* ex, int m, f32x4* keep_h1 = nullptr, f32x4* keep_h2 = nullptr) {
    const int lane = threadIdx.x & 63;
    float4* ex_h = reinterpret_cast<float4*>(ex);
    float2* ex_o = reinterpret_cast<float2*>(ex + 2 * 64 * 4);
    f32x4 a = w.b1;
    a = CTK_MFMA(w.w1[0], x0, a);
    a = CTK_MFMA(w.w1[1], x1, a);
    const f32x4 h1m = ctk_tanhf4(a);
    ex_h[m * 64 + lane] = make_float4(h1m[0], h1m[1], h1m[2], h1m[3]);
    __syncthreads();
    const float4 o4 = ex_h[(m ^ 1) * 64 + lane];
    const f32x4 h1o = f32x4{o4.x, o4.y, o4.z, o4.w};
    const f32x4 h1a = m == 0 ? h1m : h1o, h1b = m == 0 ? h1o : h1m;      // hidden tiles 0 and 1
    // layer 2, own output tile: two interleaved accumulation chains (even / odd k-steps)
    f32x4 c0 = w.b2, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        c0 = CTK_MFMA(w.w2[j], (j >> 2) ? h1b[j & 3] : h1a[j & 3], c0);
        c1 = CTK_MFMA(w.w2[j + 1], ((j + 1) >> 2) ? h1b[(j + 1) & 3] : h1a[(j + 1) & 3], c1);
    }
    const f32x4 h2m = ctk_tanhf4(c0 + c1);
    // layer 3 over the wave's own 16 inputs
    f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = p0;
    p0 = CTK_MFMA(w.w3[0], h2m[0], p0);
    p1 = CTK_MFMA(w.w3[1], h2m[1], p1);
    p0 = CTK_MFMA(w.w3[2], h2m[2], p0);
    p1 = CTK_MFMA(w.w3[3], h2m[3], p1);
    const f32x4 pm = p0 + p1;
    ex_o[m * 64 + lane] = make_float2(pm[0], pm[1]);
    __syncthreads();
    const float2 po = ex_o[(m ^ 1) * 64 + lane];
    if (keep_h1) *keep_h1 = h1m;
    if (keep_h2) *keep_h2 = h2m;
    const float2 t0 = m == 0 ? make_float2(pm[0], pm[1]) : po, t1 = m == 0 ? po : make_float2(pm[0], pm[1]);   // tile 0 + tile 1, in that order
    return MlpPair{(t0.x + t1.x) + w.b3[0], (t0.y + t1.y) + w.b3[1]};
}

// CartPole rollout of one 16-trajectory tile by a pair of waves (wave = index inside the pair's workgroup; m = wave & 1).
// Returns J of trajectory c in every lane of the pair's wave 0 (m == 0 carries the cost terms and the trajectory stores);
// the other wave returns 0.
template <bool WRITE_TRAJ, bool INPUT_COST, bool CHECKED, class UFn>
CTK_DEV float rollout_mlp_split_impl(const RolloutArgs& a, const EnvK& k, const MlpFwdWS& w, float* ex, int m, int traj0, UFn&& ufn,
                                     float* amax) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int n = traj0 + c;
    const bool valid = n < a.N;
    const MlpCostK ck = mlp_cost_coeffs(k, g, INPUT_COST);
    float sv = a.s0[g];
    float uprev = a.u_prev_dev ? *a.u_prev_dev : a.u_prev[0];
    float csum = 0.0f, am = 0.0f;
    const int H = a.H;
    float u_next = ufn(0);
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = ufn(h + 1);
        if (m == 0) {
            csum += mlp_stage_cost_share<CHECKED>(k, ck, sv, u, uprev);
            if constexpr (!CHECKED) am = fmaxf(am, fabsf(sv));
            if constexpr (WRITE_TRAJ) {
                if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + h) * CTK_S + g] = sv;
            }
        }
        sv = mlp_step_split(w, sv, g == 0 ? u : 0.0f, ex, m).lo;
        uprev = u;
    }
    *amax = am;
    if (m != 0) return 0.0f;
    if constexpr (WRITE_TRAJ) {
        if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + H) * CTK_S + g] = sv;
    }
    csum += mlp_terminal_cost_share(k, ck, g, sv);
    return sum_over_groups(csum) * a.inv_Hp1;
}

// every wave of the workgroup must call this (barriers inside); `redo` is a workgroup-shared flag word
template <bool WRITE_TRAJ, bool INPUT_COST = true, class UFn>
CTK_DEV float rollout_mlp_split(const RolloutArgs& a, const EnvK& k, const MlpFwdWS& w, float* ex, int m, int traj0, int* redo, UFn&& ufn) {
    float amax;
    float J = rollout_mlp_split_impl<WRITE_TRAJ, INPUT_COST, false>(a, k, w, ex, m, traj0, ufn, &amax);
    // an angle beyond the unchecked cos's range anywhere in the WORKGROUP (never in practice): all waves redo, checked
    if (__builtin_amdgcn_ballot_w64(!(amax <= CTK_SINCOS_FAST_LIMIT)) != 0 && (threadIdx.x & 63) == 0) atomicOr(redo, 1);
    __syncthreads();
    if (__builtin_expect(*redo != 0, 0))
        J = rollout_mlp_split_impl<WRITE_TRAJ, INPUT_COST, true>(a, k, w, ex, m, traj0, ufn, &amax);
    return J;
}

// ---- reverse mode, two-wave form --------------------------------------------------------------------------------------
struct MlpBwdWS {
    float w3t[2];      // [k-step]  A = W3^T rows of hidden tile m (k = output component 4*ks + g)
    float w2t[8];      // [k-step]  A = W2^T rows hidden_in tile m
    float w1t[4];      // [k-step within hidden tile m]  A = W1^T (rows = network inputs at row 4*(k%4) + k/4)
};

CTK_DEV MlpBwdWS mlp_load_bwd_split(const float* __restrict__ wperm, int m) {
    const float* p = wperm + 64 * MLP_FWD_PER_LANE + (threadIdx.x & 63) * MLP_BWD_PER_LANE;
    MlpBwdWS w;
    w.w3t[0] = p[2 * m]; w.w3t[1] = p[2 * m + 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) w.w2t[j] = p[4 + 8 * m + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) w.w1t[j] = p[20 + 4 * m + j];
    return w;
}

// lam0 / lam1: adjoints of the NEXT state's components g / 4+g.  h1m / h2m: the wave's own activation tiles of this step.
// Returns the adjoints w.r.t. the network inputs g and 4+g (state components, then control inputs).
CTK_DEV MlpPair mlp_step_vjp_split(const MlpBwdWS& w, f32x4 h1m, f32x4 h2m, float lam0, float lam1, float* ex, int m) {
    const int lane = threadIdx.x & 63;
    float4* ex_h = reinterpret_cast<float4*>(ex);
    float2* ex_o = reinterpret_cast<float2*>(ex + 2 * 64 * 4);
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 t = CTK_MFMA(w.w3t[0], lam0, z);
    t = CTK_MFMA(w.w3t[1], lam1, t);
    f32x4 d2m;
#pragma unroll
    for (int r = 0; r < 4; ++r) d2m[r] = t[r] * (1.0f - h2m[r] * h2m[r]);
    ex_h[m * 64 + lane] = make_float4(d2m[0], d2m[1], d2m[2], d2m[3]);
    __syncthreads();
    const float4 o4 = ex_h[(m ^ 1) * 64 + lane];
    const f32x4 d2o = f32x4{o4.x, o4.y, o4.z, o4.w};
    const f32x4 d2a = m == 0 ? d2m : d2o, d2b = m == 0 ? d2o : d2m;
    f32x4 s0 = z, s1 = z;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        s0 = CTK_MFMA(w.w2t[j], (j >> 2) ? d2b[j & 3] : d2a[j & 3], s0);
        s1 = CTK_MFMA(w.w2t[j + 1], ((j + 1) >> 2) ? d2b[(j + 1) & 3] : d2a[(j + 1) & 3], s1);
    }
    const f32x4 s = s0 + s1;
    f32x4 d1m;
#pragma unroll
    for (int r = 0; r < 4; ++r) d1m[r] = s[r] * (1.0f - h1m[r] * h1m[r]);
    f32x4 p0 = z, p1 = z;
    p0 = CTK_MFMA(w.w1t[0], d1m[0], p0);
    p1 = CTK_MFMA(w.w1t[1], d1m[1], p1);
    p0 = CTK_MFMA(w.w1t[2], d1m[2], p0);
    p1 = CTK_MFMA(w.w1t[3], d1m[3], p1);
    const f32x4 pm = p0 + p1;
    ex_o[m * 64 + lane] = make_float2(pm[0], pm[1]);
    __syncthreads();
    const float2 po = ex_o[(m ^ 1) * 64 + lane];
    const float2 t0 = m == 0 ? make_float2(pm[0], pm[1]) : po, t1 = m == 0 ? po : make_float2(pm[0], pm[1]);
    return MlpPair{t0.x + t1.x, t0.y + t1.y};
}
