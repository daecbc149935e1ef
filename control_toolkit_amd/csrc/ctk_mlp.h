// ctk_mlp.h — the 5-32-32-4 tanh MLP predictor on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate — bit-for-bit a k-ordered fmaf
// chain, so parity with the fp32 oracle is a summation-order matter only).
//
// One wave owns 16 trajectories for all H steps; nothing leaves registers between layers.
// The products are formed TRANSPOSED, Z[out, traj] = W[out, k] * X[k, traj], so that the
// accumulator layout of one layer IS the B-operand layout of the next:
//   lane l: c = l & 15 (trajectory), g = l >> 4
//   A operand: lane holds A[row = l&15][k = g]          (weights, pre-permuted per lane on the host)
//   B operand: lane holds B[k = g][col = c]             (activations of trajectory c)
//   C/D      : lane, reg r holds D[row = 4g + r][col = c]
// A D tile (rows = 16 neurons) therefore supplies, in its register r, the B operand of the k-step
// whose k-slot g is neuron 4g + r: the k order is permuted (hid(j, g) = 16(j>>2) + 4g + (j&3)),
// and the weights are stored in that same order.  The state lives as "lane (c, g) holds component
// g of trajectory c": it is the B operand of layer 1 as it stands, and layer 3's weight rows are
// placed at rows {0,4,8,12} so that its output register 0 is again component g.
#pragma once
#include "ctk_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_FWD_PER_LANE = 68;   // floats per lane, forward  (12 x float4 general + 5 x float4 thin-layer form, S = 4 / C = 1)
constexpr int MLP_BWD_PER_LANE = 28;   // floats per lane, backward ( 7 x float4)

__host__ __device__ inline int mlp_hid(int j, int g) { return 16 * (j >> 2) + 4 * g + (j & 3); }

struct MlpFwdW {
    float w1[2][3];   // [out tile m][k-step]; k-step 2 (network inputs 8 + g) only where S + C > 8
    float w2[2][8];   // [out tile][k-step]
    float w3[8];      // [k-step]
    f32x4 b1[2], b2[2], b3;
};

CTK_DEV MlpFwdW mlp_load_fwd(const float* __restrict__ wperm) {
    const float4* p = reinterpret_cast<const float4*>(wperm) + (threadIdx.x & 63) * (MLP_FWD_PER_LANE / 4);
    float4 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = p[i];
    MlpFwdW w;
    w.w1[0][0] = v[0].x; w.w1[0][1] = v[0].y; w.w1[1][0] = v[0].z; w.w1[1][1] = v[0].w;
    const float4 v16 = p[16];                 // slots 65, 66: the third k-step of layer 1 (zero where S + C <= 8)
    w.w1[0][2] = v16.y; w.w1[1][2] = v16.z;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mo = i >> 1, j0 = (i & 1) * 4;
        w.w2[mo][j0] = v[1 + i].x; w.w2[mo][j0 + 1] = v[1 + i].y; w.w2[mo][j0 + 2] = v[1 + i].z; w.w2[mo][j0 + 3] = v[1 + i].w;
    }
    w.w3[0] = v[5].x; w.w3[1] = v[5].y; w.w3[2] = v[5].z; w.w3[3] = v[5].w;
    w.w3[4] = v[6].x; w.w3[5] = v[6].y; w.w3[6] = v[6].z; w.w3[7] = v[6].w;
    w.b1[0] = f32x4{v[7].x, v[7].y, v[7].z, v[7].w};
    w.b1[1] = f32x4{v[8].x, v[8].y, v[8].z, v[8].w};
    w.b2[0] = f32x4{v[9].x, v[9].y, v[9].z, v[9].w};
    w.b2[1] = f32x4{v[10].x, v[10].y, v[10].z, v[10].w};
    w.b3 = f32x4{v[11].x, v[11].y, v[11].z, v[11].w};
    return w;
}

// tanh(x) = 1 - 2 / (exp(2x) + 1): v_exp_f32 + v_rcp_f32 (1 ulp each); exact limits +-1 for
// |x| large (exp -> inf / 0).  Absolute error <~ 2e-7 (cancellation near 0 costs relative, not
// absolute, accuracy; the activations feed fp32 sums of 32 terms).
CTK_DEV float ctk_tanhf(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);   // exp(2x) = 2^(2x*log2 e)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}

// two at a time: the multiplies/adds/fma pack into v_pk_*_f32 (the exp/rcp stay per value)
typedef float f32x2 __attribute__((ext_vector_type(2)));
CTK_DEV f32x2 ctk_tanhf2(f32x2 x) {
    const f32x2 y = x * 2.885390081777927f;
    f32x2 t;
    t.x = __builtin_amdgcn_exp2f(y.x); t.y = __builtin_amdgcn_exp2f(y.y);
    const f32x2 d = t + 1.0f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    return r * -2.0f + 1.0f;
}
CTK_DEV f32x4 ctk_tanhf4(f32x4 x) {
    const f32x2 lo = ctk_tanhf2(f32x2{x[0], x[1]}), hi = ctk_tanhf2(f32x2{x[2], x[3]});
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

#define CTK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct MlpAct {
    f32x4 h1[2], h2[2];
};

// ---------------------------------------------------------------------------------------------
// The CartPole network (4 states + 1 input -> 32 -> 32 -> 4) with its two THIN layers taken off the 16x16x4 tiles
// (tools/diag_mlp_l3.hip: 725 -> 630 ns per step of a lone wave, 4860 -> 4130 ns with the SIMDs four waves deep):
//   layer 1: the input's column enters the accumulator on the VALU (b1 + w1u * u, two v_pk_fma per tile), so the
//            matrix part is ONE k-step (the four state components) instead of two with 3 of 4 k-slots padding;
//   layer 3: 32 -> 4 as eight v_mfma_f32_4x4x1 (16 blocks of 4 lanes, 2 passes each instead of 8): block b = lane / 4
//            multiplies A[i] = W3[i][unit] (held by lane 4b + i) with B = the lane's own h2 value, so reg i of lane
//            (c, g) accumulates output i over the 8 hidden units lane group g holds.  The four lane groups' partial
//            sums meet in a reduce-scatter — v_permlane32_swap pairs (g, g+2), v_permlane16_swap pairs (g, g+1) —
//            that leaves component g in lane (c, g): the state layout layer 1 consumes.  12 of the 16 rows of the
//            16x16x4 form were padding.
// Same products; other summation order than the k-ordered chain (|diff| ~ 2e-6 over 20 steps, tools/diag_mlp_l3.hip).
// ---------------------------------------------------------------------------------------------
struct MlpFwdT {
    float w1s[2];     // [out tile]: k-step 0 of layer 1 (state components)
    float w2[2][8];   // [out tile][k-step]
    float w3n[8];     // [k-step]: A operand of the 4x4x1 blocks, output row lane % 4
    f32x4 b1[2], w1u[2], b2[2];
    float b3g;        // bias of output component g
};

CTK_DEV MlpFwdT mlp_load_fwd_thin(const float* __restrict__ wperm) {
    const float4* p = reinterpret_cast<const float4*>(wperm) + (threadIdx.x & 63) * (MLP_FWD_PER_LANE / 4);
    float4 v[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) v[i] = p[i];
    MlpFwdT w;
    w.w1s[0] = v[0].x; w.w1s[1] = v[0].z;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mo = i >> 1, j0 = (i & 1) * 4;
        w.w2[mo][j0] = v[1 + i].x; w.w2[mo][j0 + 1] = v[1 + i].y; w.w2[mo][j0 + 2] = v[1 + i].z; w.w2[mo][j0 + 3] = v[1 + i].w;
    }
    w.b1[0] = f32x4{v[7].x, v[7].y, v[7].z, v[7].w};
    w.b1[1] = f32x4{v[8].x, v[8].y, v[8].z, v[8].w};
    w.b2[0] = f32x4{v[9].x, v[9].y, v[9].z, v[9].w};
    w.b2[1] = f32x4{v[10].x, v[10].y, v[10].z, v[10].w};
    w.w3n[0] = v[12].x; w.w3n[1] = v[12].y; w.w3n[2] = v[12].z; w.w3n[3] = v[12].w;
    w.w3n[4] = v[13].x; w.w3n[5] = v[13].y; w.w3n[6] = v[13].z; w.w3n[7] = v[13].w;
    w.w1u[0] = f32x4{v[14].x, v[14].y, v[14].z, v[14].w};
    w.w1u[1] = f32x4{v[15].x, v[15].y, v[15].z, v[15].w};
    w.b3g = v[16].x;
    return w;
}

// a + b after exchanging halves: returns a[l] + a[l ^ 32] in the lower 32 lanes, b[l ^ 32] + b[l] in the upper 32.
// Written as asm: this compiler's __builtin_amdgcn_permlane32_swap / permlane16_swap fold r[0] + r[1] into
// 2 * r[0] (seen in the ISA, tools/diag_mlp_l3.hip); the s_nop is the two wait states the hazard recogniser puts
// between a VALU write of an operand and the swap, which it cannot see inside an asm statement.
CTK_DEV float swap_sum32(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// rows of 16 lanes: a[row] + a[row ^ 1] in the even rows, b[row ^ 1] + b[row] in the odd rows
CTK_DEV float swap_sum16(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// One predictor step for the wave's 16 trajectories.  sv: state component g of trajectory c;
// u: input of trajectory c (every lane group).  Returns the next state component.
CTK_DEV float mlp_step(const MlpFwdT& w, float sv, float u, int /*g*/, MlpAct* keep = nullptr) {
    f32x4 a0 = w.w1u[0] * u + w.b1[0], a1 = w.w1u[1] * u + w.b1[1];
    a0 = CTK_MFMA(w.w1s[0], sv, a0);
    a1 = CTK_MFMA(w.w1s[1], sv, a1);
    f32x4 h1[2];
    h1[0] = ctk_tanhf4(a0); h1[1] = ctk_tanhf4(a1);
    f32x4 c0 = w.b2[0], c1 = w.b2[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = h1[j >> 2][j & 3];
        c0 = CTK_MFMA(w.w2[0][j], b, c0);
        c1 = CTK_MFMA(w.w2[1][j], b, c1);
    }
    f32x4 h2[2];
    h2[0] = ctk_tanhf4(c0); h2[1] = ctk_tanhf4(c1);
    // layer 3: two interleaved accumulation chains of 4x4x1 blocks, then the reduce-scatter over the lane groups
    f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = p0;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[j], h2[j >> 2][j & 3], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[j + 1], h2[(j + 1) >> 2][(j + 1) & 3], p1, 0, 0, 0);
    }
    const f32x4 p = p0 + p1;
    const float s02 = swap_sum32(p[0], p[2]), s13 = swap_sum32(p[1], p[3]);
    if (keep) { keep->h1[0] = h1[0]; keep->h1[1] = h1[1]; keep->h2[0] = h2[0]; keep->h2[1] = h2[1]; }
    return swap_sum16(s02, s13) + w.b3g;
}

// The hidden activations of one step exactly as mlp_step_pair forms them (its wave 1 sums layer 2's k-steps 4..7 before 0..3), on ONE wave
// and without layer 3: a Jacobian worker of the one-launch RPGD descent (ctk_rpgd.hip) linearises the step at the very activations the
// forward pass had, from the state and input alone.
CTK_DEV void mlp_acts_as_pair(const MlpFwdT& w, float sv, float u, MlpAct* act) {
    f32x4 a0 = w.w1u[0] * u + w.b1[0], a1 = w.w1u[1] * u + w.b1[1];
    a0 = CTK_MFMA(w.w1s[0], sv, a0);
    a1 = CTK_MFMA(w.w1s[1], sv, a1);
    const f32x4 h10 = ctk_tanhf4(a0), h11 = ctk_tanhf4(a1);
    f32x4 c0 = w.b2[0], c1 = w.b2[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                      // each pair wave: its OWN hidden units' k-steps first ...
        c0 = CTK_MFMA(w.w2[0][j], h10[j], c0);
        c1 = CTK_MFMA(w.w2[1][4 + j], h11[j], c1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                      // ... then the other wave's
        c0 = CTK_MFMA(w.w2[0][4 + j], h11[j], c0);
        c1 = CTK_MFMA(w.w2[1][j], h10[j], c1);
    }
    act->h1[0] = h10; act->h1[1] = h11;
    act->h2[0] = ctk_tanhf4(c0); act->h2[1] = ctk_tanhf4(c1);
}

// ---------------------------------------------------------------------------------------------
// The step SHARED BY TWO WAVES (where a launch leaves SIMDs idle: a wave's matrix and vector time add up —
// profiles/r02_mlp_step_microbench.txt — so halving both per wave shortens the recurrence).  Wave m of the pair owns hidden
// units 16m .. 16m+15 of both layers: layer 1 is its own MFMA, it publishes its 4 tanh values per lane (LDS, one barrier) and
// starts layer 2 with its OWN half of the k-steps while the other half arrives; layer 3 is the four 4x4x1 blocks of its own
// units, reduce-scattered like mlp_step, and the two partial outputs meet through LDS (second barrier).  Both waves end
// with the same next state (same association of the two partial sums).
// ex: [2][64] float4 (h1 halves) + [2][64] float (partial outputs) per pair.  Two WORKGROUP barriers per step: every wave of
// the workgroup must take every step.
// ---------------------------------------------------------------------------------------------
constexpr int MLP_PAIR_EX = 2 * 64 * 4 + 2 * 64;

struct MlpFwdHalf {
    float w1s;              // layer 1, state k-step, own row tile
    float w2o[4], w2x[4];   // layer 2, own row tile: k-steps of the OWN / the OTHER wave's hidden units
    float w3n[4];           // layer 3 blocks, own hidden units
    f32x4 b1, w1u, b2;
    float b3g;
};

CTK_DEV MlpFwdHalf mlp_half_of(const MlpFwdT& w, int m) {
    MlpFwdHalf x;
    x.w1s = m ? w.w1s[1] : w.w1s[0];
    x.b1 = m ? w.b1[1] : w.b1[0]; x.w1u = m ? w.w1u[1] : w.w1u[0]; x.b2 = m ? w.b2[1] : w.b2[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x.w2o[j] = m ? w.w2[1][4 + j] : w.w2[0][j];
        x.w2x[j] = m ? w.w2[1][j] : w.w2[0][4 + j];
        x.w3n[j] = m ? w.w3n[4 + j] : w.w3n[j];
    }
    x.b3g = w.b3g;
    return x;
}

// Pins every operand to a register HERE.  In front of a loop that stores to memory without waiting: a value whose load (or reload from
// scratch) the compiler has not yet waited for carries that `s_waitcnt vmcnt(0)` to its first use, and inside the loop the same wait then
// also waits for the loop's own stores — once per step.
CTK_DEV void mlp_pin(MlpFwdHalf& w) {
    asm volatile("" : "+v"(w.w1s), "+v"(w.b3g));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(w.w2o[j]), "+v"(w.w2x[j]), "+v"(w.w3n[j]));
    asm volatile("" : "+v"(w.b1), "+v"(w.w1u), "+v"(w.b2));
}

struct MlpHalfAct {
    f32x4 h1m, h2m;         // this wave's halves of the activations
};

CTK_DEV float mlp_step_pair(const MlpFwdHalf& w, float sv, float u, int m, float* ex, MlpHalfAct* keep = nullptr) {
    const int lane = threadIdx.x & 63;
    float4* ex_h = reinterpret_cast<float4*>(ex);          // [2][64]
    float* ex_o = ex + 2 * 64 * 4;                         // [2][64]
    f32x4 a1 = w.w1u * u + w.b1;
    a1 = CTK_MFMA(w.w1s, sv, a1);
    const f32x4 h1m = ctk_tanhf4(a1);
    ex_h[m * 64 + lane] = make_float4(h1m[0], h1m[1], h1m[2], h1m[3]);
    __syncthreads();
    f32x4 c = w.b2;
#pragma unroll
    for (int j = 0; j < 4; ++j) c = CTK_MFMA(w.w2o[j], h1m[j], c);
    const float4 o4 = ex_h[(m ^ 1) * 64 + lane];
    const f32x4 h1x = f32x4{o4.x, o4.y, o4.z, o4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) c = CTK_MFMA(w.w2x[j], h1x[j], c);
    const f32x4 h2m = ctk_tanhf4(c);
    f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = p0;
    p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[0], h2m[0], p0, 0, 0, 0);
    p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[1], h2m[1], p1, 0, 0, 0);
    p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[2], h2m[2], p0, 0, 0, 0);
    p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[3], h2m[3], p1, 0, 0, 0);
    const f32x4 p = p0 + p1;
    const float part = swap_sum16(swap_sum32(p[0], p[2]), swap_sum32(p[1], p[3]));
    ex_o[m * 64 + lane] = part;
    if (keep) { keep->h1m = h1m; keep->h2m = h2m; }
    __syncthreads();
    const float other = ex_o[(m ^ 1) * 64 + lane];
    return (m == 0 ? part + other : other + part) + w.b3g;     // the same association in both waves
}

// Stage-cost share of lane group g (oracle Cost._get_stage_cost split by state component):
//   g = 0: dd(x)   g = 1: cc(u) + ccrc(u - u_prev)   g = 2: ep(angle)   g = 3: ekp(angleD)
// as per-lane coefficients, so every lane runs the same few instructions:
//   share = A*(sv - B)^2 + E*(1 - cos sv)^2 + I*(ccR u^2 + ccrc (u - u_prev)^2)
struct MlpCostK {
    float A, B, E, I;
};
CTK_DEV MlpCostK mlp_cost_coeffs(const EnvK& k, int g, bool with_input_cost) {
    MlpCostK c;
    c.A = g == 0 ? k.dd_weight * k.inv_xs * k.inv_xs : (g == 3 ? k.ekp_weight : 0.0f);
    c.B = g == 0 ? k.target_position : 0.0f;
    c.E = g == 2 ? k.ep_c : 0.0f;
    c.I = (g == 1 && with_input_cost) ? 1.0f : 0.0f;
    return c;
}

// cos only, unchecked (see ctk_sincosf_fast): same reduction and polynomials
CTK_DEV float ctk_cosf_fast(float x) {
    const float ax = fabsf(x);
    const float fn = rintf(ax * 0.636619772f);
    float r = fmaf(fn, -1.57079637e+00f, ax);
    r = fmaf(fn, 4.37113883e-08f, r);
    r = fmaf(fn, 1.71512489e-15f, r);
    const int n = (int)fn;
    const float r2 = r * r;
    float ps = fmaf(r2, -1.95152959e-04f, 8.33216087e-03f);
    ps = fmaf(r2, ps, -1.66666546e-01f);
    const float s = fmaf(r, r2 * ps, r);
    float pc = fmaf(r2, 2.44331571e-05f, -1.38873163e-03f);
    pc = fmaf(r2, pc, 4.16666456e-02f);
    pc = fmaf(r2, pc, -0.5f);
    const float c = fmaf(r2, pc, 1.0f);
    const unsigned cbits = __builtin_bit_cast(unsigned, (n & 1) ? s : c);
    return __builtin_bit_cast(float, cbits ^ ((unsigned)((n + 1) & 2) << 30));
}

template <bool CHECKED>
CTK_DEV float mlp_stage_cost_share(const EnvK& k, const MlpCostK& c, float sv, float u, float uprev) {
    const float d = sv - c.B;
    const float omc = 1.0f - (CHECKED ? cosf(sv) : ctk_cosf_fast(sv));
    const float du = u - uprev;
    return c.A * d * d + c.E * omc * omc + c.I * (k.ccR * u * u + k.ccrc_weight * du * du);
}

CTK_DEV float mlp_terminal_cost_share(const EnvK& k, const MlpCostK& c, int g, float sv) {
    const float d = sv - c.B;
    const float omc = 1.0f - cosf(sv);
    // terminal = terminal_weight * (dd + ep): the dd share lives on g == 0 (A there is the dd coefficient)
    return k.terminal_weight * ((g == 0 ? c.A * d * d : 0.0f) + c.E * omc * omc);
}

// sum over the 4 lane groups (lanes c, c+16, c+32, c+48): every lane ends with the trajectory total
CTK_DEV float sum_over_groups(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Rolls the wave's 16 trajectories (first one = traj0).  ufn(h) yields the input of trajectory c.
// Returns J of trajectory c in every lane.  INPUT_COST: include cc + ccrc (callers that sum the
// input-only terms off the recurrence pass false and add them themselves).
template <bool WRITE_Q, bool WRITE_TRAJ, bool INPUT_COST, bool CHECKED, class UFn>
CTK_DEV float rollout_mlp_impl(const RolloutArgs& a, const EnvK& k, const MlpFwdT& w, int traj0, UFn&& ufn, float* amax) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int n = traj0 + c;
    const bool valid = n < a.N;
    const MlpCostK ck = mlp_cost_coeffs(k, g, INPUT_COST);
    float sv = lane_state4(a, g);
    float uprev = uniform_u_prev0(a);
    float csum = 0.0f, am = 0.0f;
    const int H = a.H;
    float u_next = ufn(0);
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = ufn(h + 1);
        csum += mlp_stage_cost_share<CHECKED>(k, ck, sv, u, uprev);
        if constexpr (!CHECKED) am = fmaxf(am, fabsf(sv));
        if constexpr (WRITE_TRAJ) {
            if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + h) * CTK_S + g] = sv;
        }
        if constexpr (WRITE_Q) {
            if (valid && g == 0) a.Q_out[(size_t)n * H + h] = u;
        }
        sv = mlp_step(w, sv, u, g);
        uprev = u;
    }
    if constexpr (WRITE_TRAJ) {
        if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + H) * CTK_S + g] = sv;
    }
    csum += mlp_terminal_cost_share(k, ck, g, sv);
    *amax = am;
    return sum_over_groups(csum) * a.inv_Hp1;
}

template <bool WRITE_Q, bool WRITE_TRAJ, bool INPUT_COST = true, class UFn>
CTK_DEV float rollout_mlp(const RolloutArgs& a, const EnvK& k, const MlpFwdT& w, int traj0, UFn&& ufn) {
    float amax;
    float J = rollout_mlp_impl<WRITE_Q, WRITE_TRAJ, INPUT_COST, false>(a, k, w, traj0, ufn, &amax);
    // angle beyond the unchecked cos's range somewhere in the wave (never in practice): redo, checked
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(amax <= CTK_SINCOS_FAST_LIMIT)) != 0, 0))
        J = rollout_mlp_impl<WRITE_Q, WRITE_TRAJ, INPUT_COST, true>(a, k, w, traj0, ufn, &amax);
    return J;
}

// The same with a tile shared by two waves (mlp_step_pair): wave 0 of the pair carries the angle term of the cost (the cos) and
// the range check, wave 1 the quadratic and the input terms; the two partial costs meet through `ex` at the end.
// Every wave of the workgroup must call this (workgroup barriers per step and one at the end).  *amax: the pair's max |angle|.
template <bool WRITE_Q, bool WRITE_TRAJ, bool INPUT_COST, bool CHECKED, class UFn>
CTK_DEV float rollout_mlp_pair_impl(const RolloutArgs& a, const EnvK& k, const MlpFwdHalf& w, int traj0, int m, float* ex, UFn&& ufn, float* amax) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int n = traj0 + c;
    const bool valid = n < a.N;
    const MlpCostK ck = mlp_cost_coeffs(k, g, INPUT_COST);
    float sv = lane_state4(a, g);
    float uprev = uniform_u_prev0(a);
    float csum = 0.0f, am = 0.0f;
    const int H = a.H;
    float u_next = ufn(0);
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = ufn(h + 1);
        if (m == 0) {
            const float omc = 1.0f - (CHECKED ? cosf(sv) : ctk_cosf_fast(sv));
            csum += ck.E * omc * omc;
            if constexpr (!CHECKED) am = fmaxf(am, fabsf(sv));
            if constexpr (WRITE_TRAJ) {
                if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + h) * CTK_S + g] = sv;
            }
        } else {
            const float d = sv - ck.B, du = u - uprev;
            csum += ck.A * d * d + ck.I * (k.ccR * u * u + k.ccrc_weight * du * du);
            if constexpr (WRITE_Q) {
                if (valid && g == 0) a.Q_out[(size_t)n * H + h] = u;
            }
        }
        sv = mlp_step_pair(w, sv, u, m, ex);
        uprev = u;
    }
    if (m == 0) {
        if constexpr (WRITE_TRAJ) {
            if (valid && a.traj_out) a.traj_out[((size_t)n * (H + 1) + H) * CTK_S + g] = sv;
        }
        const float omc = 1.0f - cosf(sv);
        csum += k.terminal_weight * ck.E * omc * omc;
    } else {
        const float d = sv - ck.B;
        csum += g == 0 ? k.terminal_weight * ck.A * d * d : 0.0f;
    }
    const float part = sum_over_groups(csum);
    float* ex_j = ex;                      // [2][64] partial costs, [2][64] max |angle| (the step's exchange slots are free now)
    ex_j[m * 64 + lane] = part;
    ex_j[128 + m * 64 + lane] = am;
    __syncthreads();
    const float other = ex_j[(m ^ 1) * 64 + lane];
    *amax = fmaxf(am, ex_j[128 + (m ^ 1) * 64 + lane]);
    __syncthreads();                       // the slots are reused by a caller's next pass
    return (m == 0 ? part + other : other + part) * a.inv_Hp1;
}

// ---------------------------------------------------------------------------------------------
// reverse mode (RPGD): vector-Jacobian product of one step, same operand-layout trick.
// ---------------------------------------------------------------------------------------------
struct MlpBwdW {
    float w3t[2];      // [hidden tile]            A = W3^T (rows hidden, k = output component)
    float w2t[2][8];   // [hidden_in tile][k-step] A = W2^T (rows hidden_in, k = hidden_out permuted)
    float w1t[8];      // [k-step]                 A = W1^T (rows {0,4,8,12} -> state 0..3, row 1 -> input)
};

CTK_DEV MlpBwdW mlp_load_bwd(const float* __restrict__ wperm) {
    const float4* p = reinterpret_cast<const float4*>(wperm + 64 * MLP_FWD_PER_LANE) + (threadIdx.x & 63) * (MLP_BWD_PER_LANE / 4);
    float4 v[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = p[i];
    const float f[28] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w,
                         v[3].x, v[3].y, v[3].z, v[3].w, v[4].x, v[4].y, v[4].z, v[4].w, v[5].x, v[5].y, v[5].z, v[5].w,
                         v[6].x, v[6].y, v[6].z, v[6].w};
    // table: [0..3] w3t[tile][k-step] | [4..19] w2t[tile][k-step] | [20..27] w1t[k-step]   (ctk_api.hip:permute_mlp_weights);
    // CartPole has 4 outputs: only k-step 0 of W3^T is non-zero
    MlpBwdW w;
    w.w3t[0] = f[0]; w.w3t[1] = f[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w.w2t[0][j] = f[4 + j]; w.w2t[1][j] = f[12 + j]; w.w1t[j] = f[20 + j]; }
    return w;
}

// lam: adjoint of the NEXT state component g of trajectory c.  act: the activations of this step.
// Returns the adjoint w.r.t. this step's state component g; *du (valid on lane group 0) is the
// adjoint w.r.t. the step's input.
CTK_DEV float mlp_step_vjp(const MlpBwdW& w, const MlpAct& act, float lam, float* du) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 t0 = CTK_MFMA(w.w3t[0], lam, z), t1 = CTK_MFMA(w.w3t[1], lam, z);
    f32x4 d2[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d2[0][r] = t0[r] * (1.0f - act.h2[0][r] * act.h2[0][r]);
        d2[1][r] = t1[r] * (1.0f - act.h2[1][r] * act.h2[1][r]);
    }
    f32x4 s0 = z, s1 = z;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = d2[j >> 2][j & 3];
        s0 = CTK_MFMA(w.w2t[0][j], b, s0);
        s1 = CTK_MFMA(w.w2t[1][j], b, s1);
    }
    f32x4 d1[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d1[0][r] = s0[r] * (1.0f - act.h1[0][r] * act.h1[0][r]);
        d1[1][r] = s1[r] * (1.0f - act.h1[1][r] * act.h1[1][r]);
    }
    f32x4 o0 = z, o1 = z;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        o0 = CTK_MFMA(w.w1t[j], d1[j >> 2][j & 3], o0);
        o1 = CTK_MFMA(w.w1t[j + 1], d1[(j + 1) >> 2][(j + 1) & 3], o1);
    }
    *du = o0[1] + o1[1];
    return o0[0] + o1[0];
}

// ---------------------------------------------------------------------------------------------
// The same network for any environment with I = S + C <= 8 inputs and S <= 8 outputs (ctk_generic.hip): network input /
// output index k lives in lane group k % 4, k-step (inputs) or register (outputs) k / 4 — the per-lane tables are built for
// (S, C) by ctk_api.hip:permute_mlp_weights.  CartPole's own entry points above are the I = 5, S = 4 case with the second
// half unused.
// ---------------------------------------------------------------------------------------------
struct MlpPair {
    float lo, hi;      // network input / output (or adjoint) index g and 4+g of the lane's trajectory
    float ex = 0.0f;   // adjoints only: network input 8+g (environments with S + C > 8)
};

// x0 / x1 / x2: values of network inputs g, 4+g and (K3: S + C > 8) 8+g (0 beyond I).  Returns outputs g and 4+g.
template <bool K3 = false>
CTK_DEV MlpPair mlp_step2(const MlpFwdW& w, float x0, float x1, float x2 = 0.0f, MlpAct* keep = nullptr) {
    f32x4 a0 = w.b1[0], a1 = w.b1[1];
    a0 = CTK_MFMA(w.w1[0][0], x0, a0);
    a1 = CTK_MFMA(w.w1[1][0], x0, a1);
    a0 = CTK_MFMA(w.w1[0][1], x1, a0);
    a1 = CTK_MFMA(w.w1[1][1], x1, a1);
    if constexpr (K3) {
        a0 = CTK_MFMA(w.w1[0][2], x2, a0);
        a1 = CTK_MFMA(w.w1[1][2], x2, a1);
    }
    f32x4 h1[2];
    h1[0] = ctk_tanhf4(a0); h1[1] = ctk_tanhf4(a1);
    f32x4 c0 = w.b2[0], c1 = w.b2[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = h1[j >> 2][j & 3];
        c0 = CTK_MFMA(w.w2[0][j], b, c0);
        c1 = CTK_MFMA(w.w2[1][j], b, c1);
    }
    f32x4 h2[2];
    h2[0] = ctk_tanhf4(c0); h2[1] = ctk_tanhf4(c1);
    f32x4 o0 = w.b3, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        o0 = CTK_MFMA(w.w3[j], h2[j >> 2][j & 3], o0);
        o1 = CTK_MFMA(w.w3[j + 1], h2[(j + 1) >> 2][(j + 1) & 3], o1);
    }
    if (keep) { keep->h1[0] = h1[0]; keep->h1[1] = h1[1]; keep->h2[0] = h2[0]; keep->h2[1] = h2[1]; }
    return MlpPair{o0[0] + o1[0], o0[1] + o1[1]};
}

struct MlpBwdW2 {
    float w3t[2][2];   // [hidden tile][k-step]   A = W3^T (rows hidden, k = output component 4*ks + g)
    float w2t[2][8];
    float w1t[8];
};

CTK_DEV MlpBwdW2 mlp_load_bwd2(const float* __restrict__ wperm) {
    const float* p = wperm + 64 * MLP_FWD_PER_LANE + (threadIdx.x & 63) * MLP_BWD_PER_LANE;
    MlpBwdW2 w;
    w.w3t[0][0] = p[0]; w.w3t[0][1] = p[1]; w.w3t[1][0] = p[2]; w.w3t[1][1] = p[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w.w2t[0][j] = p[4 + j]; w.w2t[1][j] = p[12 + j]; w.w1t[j] = p[20 + j]; }
    return w;
}

// lam0 / lam1: adjoints of the NEXT state's components g / 4+g.  Returns the adjoints w.r.t. network inputs g, 4+g and 8+g (D register r
// of lane group g is tile row 4g + r = network input 4r + g: ctk_api.hip:permute_mlp_weights places W1^T's rows that way).
CTK_DEV MlpPair mlp_step_vjp2(const MlpBwdW2& w, const MlpAct& act, float lam0, float lam1) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 t0 = CTK_MFMA(w.w3t[0][0], lam0, z), t1 = CTK_MFMA(w.w3t[1][0], lam0, z);
    t0 = CTK_MFMA(w.w3t[0][1], lam1, t0);
    t1 = CTK_MFMA(w.w3t[1][1], lam1, t1);
    f32x4 d2[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d2[0][r] = t0[r] * (1.0f - act.h2[0][r] * act.h2[0][r]);
        d2[1][r] = t1[r] * (1.0f - act.h2[1][r] * act.h2[1][r]);
    }
    f32x4 s0 = z, s1 = z;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = d2[j >> 2][j & 3];
        s0 = CTK_MFMA(w.w2t[0][j], b, s0);
        s1 = CTK_MFMA(w.w2t[1][j], b, s1);
    }
    f32x4 d1[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d1[0][r] = s0[r] * (1.0f - act.h1[0][r] * act.h1[0][r]);
        d1[1][r] = s1[r] * (1.0f - act.h1[1][r] * act.h1[1][r]);
    }
    f32x4 o0 = z, o1 = z;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        o0 = CTK_MFMA(w.w1t[j], d1[j >> 2][j & 3], o0);
        o1 = CTK_MFMA(w.w1t[j + 1], d1[(j + 1) >> 2][(j + 1) & 3], o1);
    }
    return MlpPair{o0[0] + o1[0], o0[1] + o1[1], o0[2] + o1[2]};
}
