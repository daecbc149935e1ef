// ctk_gru.h — 2x32 GRU + dense(32->4) predictor on the fp32 matrix cores, forward path.
// (Build-defined network, PyTorch gate convention; oracle/ctk_oracle.py:gru_cell.  The reference only
// hints at it: network name 'GRU-6IN-32H1-32H2-5OUT-0', and predictor.update(s, Q0) = the hidden-state
// advance at optimizer_mppi.py:195-197.)
//
// Same operand-layout trick as ctk_mlp.h: products are formed transposed, Z[gate neuron, traj] =
// W[neuron, k] * X[k, traj]; a hidden vector lives in the accumulator (D) layout — lane (c, g), tile m,
// register r  <->  unit 16m + 4g + r of trajectory c — which IS the B-operand layout of the next
// product with the k order permuted (hid(j, g)); the weights are pre-permuted per lane on the host.
// 164 A operands + 68 accumulator-init (bias) values per lane: too many to pin in registers next to 8
// live accumulators, so they sit in LDS as [index][lane] (conflict-free: lane-contiguous rows), shared by
// the block's waves (every wave needs the same per-lane values), and stream into the MFMAs.
#pragma once
#include "ctk_mlp.h"

constexpr int GRU_NW_RAW = (96 * 5 + 96 * 32 + 192) + (96 * 32 + 96 * 32 + 192) + (4 * 32 + 4);   // 10212
constexpr int GRU_W_L1_IH = 0, GRU_W_L1_HH = 12, GRU_W_L2_IH = 60, GRU_W_L2_HH = 108, GRU_W_OUT = 156, GRU_W_BIAS = 164;
constexpr int GRU_LANE_ENTRIES = 232;                      // 164 weights + 2 layers * 32 biases + 4
constexpr int GRU_LDS_FLOATS = GRU_LANE_ENTRIES * 64;      // 59 392 B
constexpr int GRU_HIDDEN_FLOATS = 64;                      // carried state: h1[32] h2[32]

struct GruState {
    f32x4 h1[2], h2[2];
};

CTK_DEV float ctk_sigmoidf(float x) {   // 1 / (1 + exp(-x)) via v_exp_f32 + v_rcp_f32
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// cooperative copy of the per-lane table into LDS (all threads of the block)
template <int THREADS>
CTK_DEV void gru_stage_weights(float* w_s, const float* __restrict__ wperm) {
    const float4* src = reinterpret_cast<const float4*>(wperm);
    float4* dst = reinterpret_cast<float4*>(w_s);
    for (int i = threadIdx.x; i < GRU_LDS_FLOATS / 4; i += THREADS) dst[i] = src[i];
}

CTK_DEV GruState gru_load_state(const float* __restrict__ h0, int g) {
    GruState st;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st.h1[m][r] = h0[16 * m + 4 * g + r];
            st.h2[m][r] = h0[32 + 16 * m + 4 * g + r];
        }
    return st;
}

// one GRU layer for the wave's 16 trajectories.  wl = w_s + lane; x given as a callable B(ks) for KS k-steps.
template <int KS, class BFn>
CTK_DEV void gru_layer(const float* wl, int ih_base, int hh_base, int bias_base, BFn&& xb, f32x4 (&h)[2]) {
    auto W = [&](int i) { return wl[i * 64]; };
    auto B4 = [&](int i) { return f32x4{wl[(GRU_W_BIAS + i) * 64], wl[(GRU_W_BIAS + i + 1) * 64], wl[(GRU_W_BIAS + i + 2) * 64],
                                        wl[(GRU_W_BIAS + i + 3) * 64]}; };
    // accumulators: [gate r,z][tile] fed by input AND hidden products; n gate split into its input and hidden parts
    f32x4 ar[2] = {B4(bias_base + 0), B4(bias_base + 4)}, az[2] = {B4(bias_base + 8), B4(bias_base + 12)};
    f32x4 ani[2] = {B4(bias_base + 16), B4(bias_base + 20)}, anh[2] = {B4(bias_base + 24), B4(bias_base + 28)};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {          // eight independent accumulation chains interleave
        const float b = xb(ks);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ar[m] = CTK_MFMA(W(ih_base + ((0 * 2 + m) * KS + ks)), b, ar[m]);
            az[m] = CTK_MFMA(W(ih_base + ((1 * 2 + m) * KS + ks)), b, az[m]);
            ani[m] = CTK_MFMA(W(ih_base + ((2 * 2 + m) * KS + ks)), b, ani[m]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = h[j >> 2][j & 3];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ar[m] = CTK_MFMA(W(hh_base + ((0 * 2 + m) * 8 + j)), b, ar[m]);
            az[m] = CTK_MFMA(W(hh_base + ((1 * 2 + m) * 8 + j)), b, az[m]);
            anh[m] = CTK_MFMA(W(hh_base + ((2 * 2 + m) * 8 + j)), b, anh[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rr = ctk_sigmoidf(ar[m][r]);
            const float zz = ctk_sigmoidf(az[m][r]);
            const float nn = ctk_tanhf(ani[m][r] + rr * anh[m][r]);
            h[m][r] = (1.0f - zz) * nn + zz * h[m][r];
        }
}

// one predictor step: next state component g of trajectory c; st advanced in place
CTK_DEV float gru_step(const float* wl, GruState& st, float sv, float u, int g) {
    const float x1 = (g == 0) ? u : 0.0f;
    gru_layer<2>(wl, GRU_W_L1_IH, GRU_W_L1_HH, 0, [&](int ks) { return ks == 0 ? sv : x1; }, st.h1);
    const f32x4 h1a = st.h1[0], h1b = st.h1[1];
    gru_layer<8>(wl, GRU_W_L2_IH, GRU_W_L2_HH, 32, [&](int j) { return (j >> 2) ? h1b[j & 3] : h1a[j & 3]; }, st.h2);
    f32x4 o0 = f32x4{wl[(GRU_W_BIAS + 64) * 64], wl[(GRU_W_BIAS + 65) * 64], wl[(GRU_W_BIAS + 66) * 64], wl[(GRU_W_BIAS + 67) * 64]};
    f32x4 o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        o0 = CTK_MFMA(wl[(GRU_W_OUT + j) * 64], st.h2[j >> 2][j & 3], o0);
        o1 = CTK_MFMA(wl[(GRU_W_OUT + j + 1) * 64], st.h2[(j + 1) >> 2][(j + 1) & 3], o1);
    }
    return o0[0] + o1[0];
}

// Rolls the wave's 16 trajectories from the carried hidden state h0; same contract as rollout_mlp.
template <bool WRITE_Q, bool WRITE_TRAJ, bool INPUT_COST, class UFn>
CTK_DEV float rollout_gru(const RolloutArgs& a, const EnvK& k, const float* w_s, const float* __restrict__ h0, int traj0, UFn&& ufn) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int n = traj0 + c;
    const bool valid = n < a.N;
    const float* wl = w_s + lane;
    const MlpCostK ck = mlp_cost_coeffs(k, g, INPUT_COST);
    GruState st = gru_load_state(h0, g);
    float sv = a.s0[g];
    float uprev = a.u_prev_dev ? *a.u_prev_dev : a.u_prev;
    float csum = 0.0f;
    const int H = a.H;
    float u_next = ufn(0);
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = ufn(h + 1);
        csum += mlp_stage_cost_share<true>(k, ck, sv, u, uprev);
        if constexpr (WRITE_TRAJ) {
            if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + h) * CTK_S + g] = sv;
        }
        if constexpr (WRITE_Q) {
            if (valid && g == 0) a.Q_out[(size_t)n * H + h] = u;
        }
        sv = gru_step(wl, st, sv, u, g);
        uprev = u;
    }
    if constexpr (WRITE_TRAJ) {
        if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + H) * CTK_S + g] = sv;
    }
    csum += mlp_terminal_cost_share(k, ck, g, sv);
    return sum_over_groups(csum) * a.inv_Hp1;
}
