// ctk_gru.h — 2x32 GRU + dense(32->4) predictor on the fp32 matrix cores, forward path.
// (Build-defined network, PyTorch gate convention; oracle/ctk_oracle.py:gru_cell.  The reference only
// hints at it: network name 'GRU-6IN-32H1-32H2-5OUT-0', and predictor.update(s, Q0) = the hidden-state
// advance at optimizer_mppi.py:195-197.)
//
// Operand-layout trick of ctk_mlp.h: products are formed transposed, Z[gate neuron, traj] = W[neuron, k] *
// X[k, traj]; a hidden vector lives in the accumulator (D) layout — lane (c, g), tile m, register r <-> unit
// 16m + 4g + r of trajectory c — which IS the B-operand layout of the next product with the k order permuted
// (hid(j, g)); the weights are pre-permuted per lane on the host.
//
// A GRU step is 164 dependent-ish MFMAs (32 cycles each on a SIMD) + 48 gate nonlinearities per lane; on one
// wave that is ~7.8 k cycles per step.  The step is therefore spread over the FOUR waves (= four SIMDs) of a
// workgroup that owns 16 trajectories: wave (m, q) owns hidden-unit tile m (units 16m..16m+15) and
//   q = 0: the r-gate rows (input + recurrent products) and the input half of the n-gate,
//   q = 1: the z-gate rows                              and the recurrent half of the n-gate
// (18 + 24 MFMAs per step instead of 60 + 96).  Per layer the waves meet twice through LDS: (1) partners
// (m, 0) <-> (m, 1) swap the pre-activation halves the other one turns into gates — each wave applies the
// nonlinearities to 2 of a lane's 4 units of tile m; (2) everybody publishes its 2 new hidden values per lane
// and reads the full 2x4 back, so that all four waves hold the identical hidden vector (bitwise: same
// instructions on the same inputs) as B operand of the next product.  The 8-MFMA output layer is done
// redundantly by every wave (cheaper than a third exchange).  A wave's 50 A operands + 20 bias values stay in
// registers for the whole rollout.
#pragma once
#include "ctk_mlp.h"

constexpr int GRU_NW_RAW = (96 * 5 + 96 * 32 + 192) + (96 * 32 + 96 * 32 + 192) + (4 * 32 + 4);   // 10212
constexpr int GRU_TRAJ = 16;                               // trajectories per workgroup (4 waves)
constexpr int GRU_BLOCK = 256;
constexpr int GRU_W_PER_LANE = 72;                         // 18 float4: see GruW
constexpr int GRU_TABLE_FLOATS = 4 * 64 * GRU_W_PER_LANE;  // [wave][lane][72]
constexpr int GRU_HIDDEN_FLOATS = 64;                      // carried state: h1[32] h2[32]
constexpr int GRU_EX_FLOATS = (2 * 4 + 2 * 2) * 64 * 4;    // LDS exchange: per layer [4 waves][64] + [2 tiles][64] float4

// per-lane operands of wave (m, q), in table order
struct GruW {
    float l1[18];   // layer 1: A.ih[2] A.hh[8] B[8]   (A = r or z rows; B = n rows: ih (q = 0, 2 used) or hh (q = 1))
    float l2[24];   // layer 2: A.ih[8] A.hh[8] B[8]
    float out[8];   // dense 32 -> 4 (rows 0, 4, 8, 12 of a 16-row tile)
    f32x4 bA1, bB1, bA2, bB2, bo;   // accumulator initial values (biases in D layout)
};

struct GruState {
    f32x4 h1[2], h2[2];
};

CTK_DEV float ctk_sigmoidf(float x) {   // 1 / (1 + exp(-x)) via v_exp_f32 + v_rcp_f32
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

CTK_DEV GruW gru_load_weights(const float* __restrict__ table, int wave, int lane) {
    const float4* p = reinterpret_cast<const float4*>(table + (size_t)(wave * 64 + lane) * GRU_W_PER_LANE);
    float f[GRU_W_PER_LANE];
#pragma unroll
    for (int i = 0; i < GRU_W_PER_LANE / 4; ++i) {
        const float4 v = p[i];
        f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
    }
    GruW w;
#pragma unroll
    for (int i = 0; i < 18; ++i) w.l1[i] = f[i];
#pragma unroll
    for (int i = 0; i < 24; ++i) w.l2[i] = f[18 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) w.out[i] = f[42 + i];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w.bA1[r] = f[50 + r]; w.bB1[r] = f[54 + r]; w.bA2[r] = f[58 + r]; w.bB2[r] = f[62 + r]; w.bo[r] = f[66 + r];
    }
    return w;
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

// what wave (m, q) leaves for the adjoint of its two units (registers 2q, 2q+1 of tile m): ctk_gru4.hip
struct GruPairTape {
    float r[2], z[2], n[2], ghn[2], hp[2];
};

// One GRU layer for the workgroup's 16 trajectories; called by all four waves.  wv: [KS] A.ih, [8] A.hh, [8] B.
// ex1: [4 waves][64 lanes] float4, ex2: [2 tiles][64 lanes] float4 (this layer's exchange slots).
// Slot reuse needs no extra barrier: a slot is rewritten one full step later, after at least one workgroup
// barrier that every reader of the old value reaches only after consuming it.
template <int KS, class XFn>
CTK_DEV void gru_layer(const float* wv, f32x4 biasA, f32x4 biasB, XFn&& xb, f32x4 (&h)[2], float* ex1, float* ex2,
                       int wave, int lane, GruPairTape* tp = nullptr) {
    const int m = wave >> 1, q = wave & 1;
    f32x4 a = biasA, b = biasB;
    if (q == 0) {   // r rows: input + recurrent products; n rows: input products
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float x = xb(ks);
            a = CTK_MFMA(wv[ks], x, a);
            b = CTK_MFMA(wv[KS + 8 + ks], x, b);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) a = CTK_MFMA(wv[KS + j], h[j >> 2][j & 3], a);
    } else {        // z rows: input + recurrent products; n rows: recurrent products
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = CTK_MFMA(wv[ks], xb(ks), a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float hv = h[j >> 2][j & 3];
            a = CTK_MFMA(wv[KS + j], hv, a);
            b = CTK_MFMA(wv[KS + 8 + j], hv, b);
        }
    }
    // (1) hand the partner (m, 1-q) the registers it turns into gates: {2(1-q), 2(1-q)+1} of both accumulators
    float4* e1 = reinterpret_cast<float4*>(ex1);
    e1[wave * 64 + lane] = q == 0 ? make_float4(a[2], a[3], b[2], b[3]) : make_float4(a[0], a[1], b[0], b[1]);
    __syncthreads();
    const float4 in = e1[(wave ^ 1) * 64 + lane];
    // my two units of tile m: registers 2q, 2q+1
    const float r0 = q == 0 ? a[0] : in.x, r1 = q == 0 ? a[1] : in.y;
    const float ni0 = q == 0 ? b[0] : in.z, ni1 = q == 0 ? b[1] : in.w;
    const float z0 = q == 0 ? in.x : a[2], z1 = q == 0 ? in.y : a[3];
    const float nh0 = q == 0 ? in.z : b[2], nh1 = q == 0 ? in.w : b[3];
    const f32x4 hm = m == 0 ? h[0] : h[1];
    const float ho0 = q == 0 ? hm[0] : hm[2], ho1 = q == 0 ? hm[1] : hm[3];
    const float rr0 = ctk_sigmoidf(r0), rr1 = ctk_sigmoidf(r1);
    const float zz0 = ctk_sigmoidf(z0), zz1 = ctk_sigmoidf(z1);
    const float nn0 = ctk_tanhf(ni0 + rr0 * nh0), nn1 = ctk_tanhf(ni1 + rr1 * nh1);
    const float hn0 = (1.0f - zz0) * nn0 + zz0 * ho0, hn1 = (1.0f - zz1) * nn1 + zz1 * ho1;
    if (tp) {
        tp->r[0] = rr0; tp->r[1] = rr1; tp->z[0] = zz0; tp->z[1] = zz1; tp->n[0] = nn0; tp->n[1] = nn1;
        tp->ghn[0] = nh0; tp->ghn[1] = nh1; tp->hp[0] = ho0; tp->hp[1] = ho1;
    }
    // (2) publish, then everybody reads the full new hidden vector of its lane: 2 tiles x 4 registers
    reinterpret_cast<float2*>(ex2)[(m * 64 + lane) * 2 + q] = make_float2(hn0, hn1);
    __syncthreads();
    const float4 t0 = reinterpret_cast<const float4*>(ex2)[lane], t1 = reinterpret_cast<const float4*>(ex2)[64 + lane];
    h[0] = f32x4{t0.x, t0.y, t0.z, t0.w};
    h[1] = f32x4{t1.x, t1.y, t1.z, t1.w};
}

// One predictor step, called by all four waves of the workgroup: next state component g of trajectory c
// (identical in every wave); st advanced in place.  ex: GRU_EX_FLOATS floats of LDS.
CTK_DEV float gru_step(const GruW& w, GruState& st, float sv, float u, int g, float* ex, int wave, int lane) {
    const float x1 = (g == 0) ? u : 0.0f;
    gru_layer<2>(w.l1, w.bA1, w.bB1, [&](int ks) { return ks == 0 ? sv : x1; }, st.h1, ex, ex + 2048, wave, lane);
    const f32x4 h1a = st.h1[0], h1b = st.h1[1];
    gru_layer<8>(w.l2, w.bA2, w.bB2, [&](int j) { return (j >> 2) ? h1b[j & 3] : h1a[j & 3]; }, st.h2, ex + 1024, ex + 2560,
                 wave, lane);
    f32x4 o0 = w.bo, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        o0 = CTK_MFMA(w.out[j], st.h2[j >> 2][j & 3], o0);
        o1 = CTK_MFMA(w.out[j + 1], st.h2[(j + 1) >> 2][(j + 1) & 3], o1);
    }
    return o0[0] + o1[0];
}

// Rolls the workgroup's 16 trajectories (first one = traj0) from the carried hidden state h0; called by all
// four waves.  ufn(h): input of trajectory c = lane & 15.  Returns J of trajectory c in every lane of WAVE 0
// (the other waves return 0; wave 0 carries the cost terms, wave 2 the trajectory stores).
template <bool WRITE_TRAJ, bool INPUT_COST, bool FASTCOS, class UFn>
CTK_DEV float rollout_gru_impl(const RolloutArgs& a, const EnvK& k, const GruW& w, const float* __restrict__ h0,
                               float* ex, int traj0, UFn&& ufn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int n = traj0 + c;
    const bool valid = n < a.N;
    const MlpCostK ck = mlp_cost_coeffs(k, g, INPUT_COST);
    GruState st = gru_load_state(h0, g);
    float sv = lane_state4(a, g);
    float uprev = uniform_u_prev0(a);
    float csum = 0.0f;
    const int H = a.H;
    float u_next = ufn(0);
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = ufn(h + 1);
        if (wave == 0) csum += mlp_stage_cost_share<!FASTCOS>(k, ck, sv, u, uprev);
        if constexpr (WRITE_TRAJ) {
            if (wave == 2 && valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + h) * CTK_S + g] = sv;
        }
        sv = gru_step(w, st, sv, u, g, ex, wave, lane);
        uprev = u;
    }
    if constexpr (WRITE_TRAJ) {
        if (wave == 2 && valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + H) * CTK_S + g] = sv;
    }
    if (wave != 0) return 0.0f;
    csum += mlp_terminal_cost_share(k, ck, g, sv);
    return sum_over_groups(csum) * a.inv_Hp1;
}

// FASTCOS (a.fast_cos_ok, decided on the host from a bound on the network's outputs): the unchecked cos in the
// per-step cost on wave 0 — it sits on the workgroup's critical path, the other waves wait for it at the next barrier
template <bool WRITE_TRAJ, bool INPUT_COST, class UFn>
CTK_DEV float rollout_gru(const RolloutArgs& a, const EnvK& k, const float* __restrict__ table, const float* __restrict__ h0,
                          float* ex, int traj0, UFn&& ufn) {
    const GruW w = gru_load_weights(table, threadIdx.x >> 6, threadIdx.x & 63);
    if (a.fast_cos_ok) return rollout_gru_impl<WRITE_TRAJ, INPUT_COST, true>(a, k, w, h0, ex, traj0, ufn);
    return rollout_gru_impl<WRITE_TRAJ, INPUT_COST, false>(a, k, w, h0, ex, traj0, ufn);
}
