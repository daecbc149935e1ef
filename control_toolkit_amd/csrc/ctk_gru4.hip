// ctk_gru4.hip — the recurrent predictor of the template kernels with ONE 16-trajectory tile spread over the FOUR waves (= four SIMDs)
// of a workgroup, forward AND reverse (back-propagation through time), for any environment with S + C <= 8.
//
// Why: a GRU step on one wave (ctk_net.h: NetGru) is 164 dependent-ish MFMAs + 48 gate nonlinearities per lane forward and 172 MFMAs +
// the gate adjoints reverse, every A operand re-read from LDS — 6.3 ms for RPGD at N 256 / H 50 x 10 Adam iterations while 16 waves
// occupy 16 of the chip's 1 024 SIMDs.  The population is small exactly where the gradient-based optimizers run
// (optimizer_rpgd.py:306-338 differentiates through whatever predictor it is given), so the step is split instead:
//
//   forward  — ctk_gru.h: gru_layer.  Wave (m, q) owns hidden tile m; q = 0: r rows + the input half of the n rows, q = 1: z rows + the
//              recurrent half; 18 + 24 + 8 MFMAs per wave and step, operands in registers, two LDS exchanges per layer.  Each wave turns
//              TWO of a lane's four units of tile m into gates (registers 2q, 2q+1) and tapes exactly those (GruPairTape).
//   reverse  — the same ownership: wave (m, q) forms the gate adjoints of ITS two units element-wise from its own tape, and those are
//              k-steps (gate G, tile m, register 2q+i) of the transposed products W_i^T dgi, W_h^T dgh (ctk_net.h: layer_products) — the
//              contraction over the 96 gate neurons is split four ways, every wave accumulates partial tiles of ALL outputs (2 + 24 + 18
//              MFMAs per step), and the partial sums meet through LDS once per layer: two barriers per reverse step.
//   the cost — nothing of it rides on the recurrence: the forward pass leaves the states in LDS, and the stage / terminal cost, their
//              state gradients and the input-only gradient terms are evaluated afterwards by all 256 threads over (step, plan) pairs.
//
// Tables: the per-lane operand tables of NetGru as they are (ctk_net.h: GRUG_FWD / GRUG_BWD entry-major layouts); a wave picks its own
// entries once per launch.  Tape: 5 float4 per lane, wave and step in the L2-resident scratch (20 KiB per workgroup-step), read back
// one step ahead of its use.
#include "ctk_rollout.h"
#include "ctk_env.h"
#include "ctk_gru.h"
#include "ctk_net.h"
#include "ctk_adam.h"
#include "ctk_launch.h"
#include "ctk_mppi_merge.h"
#include <algorithm>

NetFuse ctk_net_fuse(const MppiFuse* fuse, int mode, const RolloutArgs& a, int C, const float* u_nom, int blocks, int cols);   // ctk_generic_net.hip

constexpr int G4_TRAJ = 16, G4_BLOCK = 256, G4_LD = G4_TRAJ + 1;
constexpr int G4_TAPE_F4 = 5;                       // float4 per lane, wave and step
constexpr int G4_EX_FWD = GRU_EX_FLOATS;            // 3072: gru_layer's exchange slots
constexpr int G4_EX_A = 4 * 4 * 64 * 4;             // reverse, layer 2: [4 waves][dx0 dx1 dhp0 dhp1][64] float4
constexpr int G4_EX_B = 4 * 3 * 64 * 4;             // reverse, layer 1: [4 waves][din dhp0 dhp1][64] float4
constexpr int G4_RED = 4 * 16 + 16;

// ---- operands ---------------------------------------------------------------------------------------------------------------------
// forward operands of wave (m, q) in GruW's order (ctk_gru.h) from the generic table (ctk_net.h: per layer Wi[gate][m][ks], Wh[gate][m][j],
// b_r b_z b_in b_hn [m][4]; then Wo[8], b_o[4])
CTK_DEV GruW gru4_load_fwd(const float* __restrict__ tab, int m, int q, int lane) {
    GruW w;
    const int G = q;                                 // q = 0: r rows, q = 1: z rows
    {
        const float* wi = tab;
        const float* wh = tab + 6 * 2 * 64;
        const float* bb = wh + 48 * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w.l1[ks] = wi[((G * 2 + m) * 2 + ks) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w.l1[2 + j] = wh[((G * 2 + m) * 8 + j) * 64 + lane];
            w.l1[10 + j] = q == 0 ? (j < 2 ? wi[((4 + m) * 2 + j) * 64 + lane] : 0.0f) : wh[((4 + m) * 8 + j) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w.bA1[r] = bb[((q == 0 ? 0 : 8) + m * 4 + r) * 64 + lane];
            w.bB1[r] = bb[((q == 0 ? 16 : 24) + m * 4 + r) * 64 + lane];
        }
    }
    {
        const float* wi = tab + GRUG_L1 * 64;
        const float* wh = wi + 6 * 8 * 64;
        const float* bb = wh + 48 * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w.l2[j] = wi[((G * 2 + m) * 8 + j) * 64 + lane];
            w.l2[8 + j] = wh[((G * 2 + m) * 8 + j) * 64 + lane];
            w.l2[16 + j] = q == 0 ? wi[((4 + m) * 8 + j) * 64 + lane] : wh[((4 + m) * 8 + j) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w.bA2[r] = bb[((q == 0 ? 0 : 8) + m * 4 + r) * 64 + lane];
            w.bB2[r] = bb[((q == 0 ? 16 : 24) + m * 4 + r) * 64 + lane];
        }
    }
    const float* wo = tab + (GRUG_L1 + GRUG_L2) * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) w.out[j] = wo[j * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) w.bo[r] = wo[(8 + r) * 64 + lane];
    return w;
}

// one predictor step by all four waves: network inputs (g, 4+g) -> outputs (g, 4+g) of trajectory c, identical in every wave
CTK_DEV MlpPair gru4_step(const GruW& w, GruState& st, float x0, float x1, float* ex, int wave, int lane, GruPairTape* t1, GruPairTape* t2) {
    gru_layer<2>(w.l1, w.bA1, w.bB1, [&](int ks) { return ks == 0 ? x0 : x1; }, st.h1, ex, ex + 2048, wave, lane, t1);
    const f32x4 a = st.h1[0], b = st.h1[1];
    gru_layer<8>(w.l2, w.bA2, w.bB2, [&](int j) { return (j >> 2) ? b[j & 3] : a[j & 3]; }, st.h2, ex + 1024, ex + 2560, wave, lane, t2);
    f32x4 o0 = w.bo, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        o0 = CTK_MFMA(w.out[j], st.h2[j >> 2][j & 3], o0);
        o1 = CTK_MFMA(w.out[j + 1], st.h2[(j + 1) >> 2][(j + 1) & 3], o1);
    }
    return MlpPair{o0[0] + o1[0], o0[1] + o1[1]};
}

// reverse operands of wave (m, q): k-steps (gate G, tile m, register 2q + i) of every transposed product, entry e = 2G + i
struct Gru4BwdW {
    float woT[2];
    float wi2[2][6], wh2[2][6];      // layer 2: -> h1' adjoint tiles, -> carried h2 adjoint tiles
    float wi1[6], wh1[2][6];         // layer 1: -> network input adjoints, -> carried h1 adjoint tiles
};

CTK_DEV Gru4BwdW gru4_load_bwd(const float* __restrict__ tab, int m, int q, int lane) {
    Gru4BwdW w;
    w.woT[0] = tab[(m * 2 + 0) * 64 + lane];
    w.woT[1] = tab[(m * 2 + 1) * 64 + lane];
    const float* wiT2 = tab + 4 * 64;
    const float* whT2 = tab + (4 + 48) * 64;
    const float* wiT1 = tab + (4 + 96) * 64;
    const float* whT1 = tab + (4 + 96 + 24) * 64;
#pragma unroll
    for (int G = 0; G < 3; ++G)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ks = G * 8 + m * 4 + 2 * q + i, e = 2 * G + i;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                w.wi2[t][e] = wiT2[(t * 24 + ks) * 64 + lane];
                w.wh2[t][e] = whT2[(t * 24 + ks) * 64 + lane];
                w.wh1[t][e] = whT1[(t * 24 + ks) * 64 + lane];
            }
            w.wi1[e] = wiT1[ks * 64 + lane];
        }
    return w;
}

struct Gru4Adj {            // adjoints of the hidden states handed to the EARLIER step: my two units of tile m
    float dh1[2], dh2[2];
};

// gate adjoints of two units (ctk_net.h: cell_adjoint): d = adjoint of h'.  bi[e] / bh[e], e = 2G + i: B operands of the input / recurrent products
CTK_DEV void gru4_cell_adjoint(const float (&d)[2], const float (&r)[2], const float (&z)[2], const float (&n)[2], const float (&ghn)[2],
                               const float (&hp)[2], float (&bi)[6], float (&bh)[6], float (&direct)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float dn = d[i] * (1.0f - z[i]);
        const float dz = d[i] * (hp[i] - n[i]);
        direct[i] = d[i] * z[i];
        const float a = dn * (1.0f - n[i] * n[i]);
        const float dar = (a * ghn[i]) * r[i] * (1.0f - r[i]);
        const float daz = dz * z[i] * (1.0f - z[i]);
        bi[0 + i] = dar; bi[2 + i] = daz; bi[4 + i] = a;
        bh[0 + i] = dar; bh[2 + i] = daz; bh[4 + i] = a * r[i];
    }
}

// adjoint of one predictor step by all four waves: (lam0, lam1) = adjoint of the step's outputs (g, 4+g) -> adjoint of its inputs
// (g, 4+g), identical in every wave; tp: the wave's tape of this step
CTK_DEV MlpPair gru4_vjp(const Gru4BwdW& w, Gru4Adj& ad, const float4 (&tp)[G4_TAPE_F4], float lam0, float lam1, float* exA, float* exB,
                         int wave, int lane) {
    const int m = wave >> 1, q = wave & 1;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 d2t = CTK_MFMA(w.woT[0], lam0, zero);
    d2t = CTK_MFMA(w.woT[1], lam1, d2t);
    const float d2[2] = {(q == 0 ? d2t[0] : d2t[2]) + ad.dh2[0], (q == 0 ? d2t[1] : d2t[3]) + ad.dh2[1]};
    float bi[6], bh[6], direct[2];
    {   // layer 2: tape words 12..19 + hp in 10, 11
        const float r[2] = {tp[3].x, tp[3].y}, z[2] = {tp[3].z, tp[3].w}, n[2] = {tp[4].x, tp[4].y}, ghn[2] = {tp[4].z, tp[4].w}, hp[2] = {tp[2].z, tp[2].w};
        gru4_cell_adjoint(d2, r, z, n, ghn, hp, bi, bh, direct);
    }
    f32x4 dx[2] = {zero, zero}, dhp[2] = {zero, zero};
#pragma unroll
    for (int e = 0; e < 6; ++e)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            dx[t] = CTK_MFMA(w.wi2[t][e], bi[e], dx[t]);
            dhp[t] = CTK_MFMA(w.wh2[t][e], bh[e], dhp[t]);
        }
    float4* A4 = reinterpret_cast<float4*>(exA);
    A4[(wave * 4 + 0) * 64 + lane] = st4(dx[0]); A4[(wave * 4 + 1) * 64 + lane] = st4(dx[1]);
    A4[(wave * 4 + 2) * 64 + lane] = st4(dhp[0]); A4[(wave * 4 + 3) * 64 + lane] = st4(dhp[1]);
    __syncthreads();
    float d1[2] = {ad.dh1[0], ad.dh1[1]};
    float c2[2] = {direct[0], direct[1]};
    const float2* A2 = reinterpret_cast<const float2*>(exA);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float2 x = A2[((s * 4 + m) * 64 + lane) * 2 + q], y = A2[((s * 4 + 2 + m) * 64 + lane) * 2 + q];
        d1[0] += x.x; d1[1] += x.y; c2[0] += y.x; c2[1] += y.y;
    }
    ad.dh2[0] = c2[0]; ad.dh2[1] = c2[1];
    {   // layer 1: tape words 0..9
        const float r[2] = {tp[0].x, tp[0].y}, z[2] = {tp[0].z, tp[0].w}, n[2] = {tp[1].x, tp[1].y}, ghn[2] = {tp[1].z, tp[1].w}, hp[2] = {tp[2].x, tp[2].y};
        gru4_cell_adjoint(d1, r, z, n, ghn, hp, bi, bh, direct);
    }
    f32x4 din = zero, dh[2] = {zero, zero};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        din = CTK_MFMA(w.wi1[e], bi[e], din);
#pragma unroll
        for (int t = 0; t < 2; ++t) dh[t] = CTK_MFMA(w.wh1[t][e], bh[e], dh[t]);
    }
    float4* B4 = reinterpret_cast<float4*>(exB);
    B4[(wave * 3 + 0) * 64 + lane] = st4(din); B4[(wave * 3 + 1) * 64 + lane] = st4(dh[0]); B4[(wave * 3 + 2) * 64 + lane] = st4(dh[1]);
    __syncthreads();
    float lo = 0.0f, hi = 0.0f;
    float c1[2] = {direct[0], direct[1]};
    const float2* B2 = reinterpret_cast<const float2*>(exB);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float2 x = B2[((s * 3 + 0) * 64 + lane) * 2], y = B2[((s * 3 + 1 + m) * 64 + lane) * 2 + q];
        lo += x.x; hi += x.y; c1[0] += y.x; c1[1] += y.y;
    }
    ad.dh1[0] = c1[0]; ad.dh1[1] = c1[1];
    return MlpPair{lo, hi};
}

// network operands of a step from the (component g, component 4+g) state layout and the step's inputs
template <int S, int C>
CTK_DEV void gru4_operands(float sv0, float sv1, const float (&u)[C], int g, float& x0, float& x1) {
    x0 = (g < S) ? sv0 : 0.0f;
    x1 = (4 + g < S) ? sv1 : 0.0f;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        x0 = (g == S + cc) ? u[cc] : x0;
        x1 = (4 + g == S + cc) ? u[cc] : x1;
    }
}

// ---- RPGD descent ----------------------------------------------------------------------------------------------------------------
// LDS: exchange slots | states xs[H+1][64][2] | cost-gradient terms gs[H+1][64][2] | plans q[HC][17] | gradients g[HC][17] | reductions
template <int ENV>
__global__ __launch_bounds__(G4_BLOCK) void ctk_g_rpgd_descent_gru4(RolloutArgs a, typename Env<ENV>::K k, AdamK ad, float* __restrict__ Q,
                                                                   float* __restrict__ mom, float* __restrict__ var,
                                                                   const float* __restrict__ bc_table, int bc_len, int t0, int iters,
                                                                   const float* __restrict__ wperm, const float* __restrict__ wperm_bwd,
                                                                   const float* __restrict__ hidden, float* __restrict__ scratch) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    static_assert(S + C <= 8, "the GRU's input tile holds 8 columns");
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C;
    float* ex = lds;
    float* exA = ex + G4_EX_FWD;
    float* exB = exA + G4_EX_A;
    float* red_s = exB + G4_EX_B;
    float* xs_s = red_s + G4_RED;
    float* gs_s = xs_s + (H + 1) * 128;
    float* q_s = gs_s + (H + 1) * 128;
    float* g_s = q_s + HC * G4_LD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4, m = wave >> 1, q = wave & 1;
    const int row0 = blockIdx.x * G4_TRAJ;
    const int rows = min(G4_TRAJ, a.N - row0);
    const int total = rows * HC;
    const size_t gbase = (size_t)row0 * HC;
    float4* tape = reinterpret_cast<float4*>(scratch) + ((size_t)blockIdx.x * H * 4 + wave) * G4_TAPE_F4 * 64 + lane;   // + h * 4 * 5 * 64 + i * 64
    const size_t tape_step = (size_t)4 * G4_TAPE_F4 * 64;

    for (int i = t; i < G4_TRAJ * HC; i += G4_BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        q_s[hc * G4_LD + r] = i < total ? Q[gbase + i] : 0.0f;
    }
    const GruW wf = gru4_load_fwd(wperm, m, q, lane);
    const Gru4BwdW wb = gru4_load_bwd(wperm_bwd, m, q, lane);
    float up0[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up0[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    const float inv = a.inv_Hp1;
    const float s00 = g < S ? lane_state4(a, g) : 0.0f, s01 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    const int pc = t & 15, part = t >> 4;                 // (plan, part) decomposition of the parallel passes: part = 4 * wave + g
    __syncthreads();

    auto forward = [&](auto taping) {
        constexpr bool TAPE = decltype(taping)::value;
        GruState st = gru_load_state(hidden, g);
        float sv0 = s00, sv1 = s01;
        for (int h = 0; h < H; ++h) {
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = q_s[(h * C + cc) * G4_LD + c];
            if (wave == (h & 3)) reinterpret_cast<float2*>(xs_s)[h * 64 + lane] = make_float2(sv0, sv1);
            float x0, x1;
            gru4_operands<S, C>(sv0, sv1, u, g, x0, x1);
            GruPairTape t1, t2;
            const MlpPair o = gru4_step(wf, st, x0, x1, ex, wave, lane, TAPE ? &t1 : nullptr, TAPE ? &t2 : nullptr);
            if constexpr (TAPE) {
                float4* tq = tape + (size_t)h * tape_step;
                tq[0 * 64] = make_float4(t1.r[0], t1.r[1], t1.z[0], t1.z[1]);
                tq[1 * 64] = make_float4(t1.n[0], t1.n[1], t1.ghn[0], t1.ghn[1]);
                tq[2 * 64] = make_float4(t1.hp[0], t1.hp[1], t2.hp[0], t2.hp[1]);
                tq[3 * 64] = make_float4(t2.r[0], t2.r[1], t2.z[0], t2.z[1]);
                tq[4 * 64] = make_float4(t2.n[0], t2.n[1], t2.ghn[0], t2.ghn[1]);
            }
            sv0 = o.lo; sv1 = o.hi;
        }
        if (wave == 0) reinterpret_cast<float2*>(xs_s)[H * 64 + lane] = make_float2(sv0, sv1);
        __syncthreads();
    };
    auto state_of = [&](int h, int p, float (&s)[S]) {
#pragma unroll
        for (int j = 0; j < S; ++j) s[j] = xs_s[(h * 64 + (j & 3) * 16 + p) * 2 + (j >> 2)];
    };
    // sum over the 16 parts of a plan, in a fixed order; valid in every thread
    auto sum_parts = [&](float v) {
        v = sum_over_groups(v);
        if (g == 0) red_s[wave * 16 + c] = v;
        __syncthreads();
        const float r = (red_s[c] + red_s[16 + c]) + (red_s[32 + c] + red_s[48 + c]);
        __syncthreads();
        return r;
    };

    for (int it = 0; it < iters; ++it) {
        forward(std::true_type{});
        // ---- everything of the gradient that does not ride on the adjoint chain, over (step, plan) pairs
        for (int idx = t; idx < (H + 1) * G4_TRAJ; idx += G4_BLOCK) {
            const int h = idx >> 4, p = idx & 15;
            float s[S], gs[S];
            state_of(h, p, s);
            if (h < H) E::stage_grad_state(k, s, gs); else E::terminal_grad(k, s, gs);
#pragma unroll
            for (int j = 0; j < 8; ++j) gs_s[(h * 64 + (j & 3) * 16 + p) * 2 + (j >> 2)] = j < S ? gs[j < S ? j : 0] * inv : 0.0f;
            if (h < H) {
                float u[C], upv[C], un[C], gu[C], gp[C], gu2[C], gpn[C];
#pragma unroll
                for (int cc = 0; cc < C; ++cc) {
                    u[cc] = q_s[(h * C + cc) * G4_LD + p];
                    upv[cc] = h > 0 ? q_s[((h - 1) * C + cc) * G4_LD + p] : up0[cc];
                    un[cc] = h + 1 < H ? q_s[((h + 1) * C + cc) * G4_LD + p] : 0.0f;
                    gpn[cc] = 0.0f;
                }
                E::input_grad(k, u, upv, gu, gp);
                if (h + 1 < H) E::input_grad(k, un, u, gu2, gpn);        // the next step's term in u_h (rate-of-change cost)
#pragma unroll
                for (int cc = 0; cc < C; ++cc) g_s[(h * C + cc) * G4_LD + p] = (gu[cc] + gpn[cc]) * inv;
            }
        }
        __syncthreads();
        // ---- reverse sweep
        {
            float2 lam = reinterpret_cast<const float2*>(gs_s)[H * 64 + lane];
            Gru4Adj adj{{0.f, 0.f}, {0.f, 0.f}};
            float4 nxt[G4_TAPE_F4];
#pragma unroll
            for (int i = 0; i < G4_TAPE_F4; ++i) nxt[i] = tape[(size_t)(H - 1) * tape_step + i * 64];
            for (int h = H - 1; h >= 0; --h) {
                float4 cur[G4_TAPE_F4];
#pragma unroll
                for (int i = 0; i < G4_TAPE_F4; ++i) cur[i] = nxt[i];
                if (h > 0) {
#pragma unroll
                    for (int i = 0; i < G4_TAPE_F4; ++i) nxt[i] = tape[(size_t)(h - 1) * tape_step + i * 64];
                }
                const MlpPair d = gru4_vjp(wb, adj, cur, lam.x, lam.y, exA, exB, wave, lane);
                const float2 gsv = reinterpret_cast<const float2*>(gs_s)[h * 64 + lane];
                if (wave == 0) {
#pragma unroll
                    for (int cc = 0; cc < C; ++cc) {
                        const int kk = S + cc;                         // network input index of control input cc
                        if (g == (kk & 3)) g_s[(h * C + cc) * G4_LD + c] += (kk >= 4 ? d.hi : d.lo);
                    }
                }
                lam.x = gsv.x + (g < S ? d.lo : 0.0f);
                lam.y = gsv.y + (4 + g < S ? d.hi : 0.0f);
            }
        }
        __syncthreads();
        // ---- per-plan clip_by_norm, Adam, clip
        float n2 = 0.0f;
        for (int hc = part; hc < HC; hc += 16) { const float x = g_s[hc * G4_LD + pc]; n2 += x * x; }
        n2 = sum_parts(n2);
        const float scl = ad.clip / fmaxf(sqrtf(n2), ad.clip);       // of plan pc = c
        if (wave == 0 && g == 0) red_s[64 + c] = scl;
        __syncthreads();
        const int ti = t0 + it + 1;
        const float bc1 = ti <= bc_len ? bc_table[2 * (ti - 1)] : 1.0f;
        const float bc2 = ti <= bc_len ? bc_table[2 * (ti - 1) + 1] : 1.0f;
        for (int i = t; i < total; i += G4_BLOCK) {
            const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC, cc = hc % C;
            float mm = 0.0f, vv = 0.0f;
            if (ad.rule != 2) { mm = mom[gbase + i]; vv = var[gbase + i]; }
            const float gg = g_s[hc * G4_LD + r] * red_s[64 + r];
            q_s[hc * G4_LD + r] = adam_update(ad, q_s[hc * G4_LD + r], gg, mm, vv, bc1, bc2, a.lo[cc], a.hi[cc]);
            if (ad.rule != 2) { mom[gbase + i] = mm; var[gbase + i] = vv; }
        }
        __syncthreads();
    }
    // ---- get_action's forward pass (optimizer_rpgd.py:342): costs of the refined plans
    forward(std::false_type{});
    float cs = 0.0f;
    for (int h = part; h < H; h += 16) {
        float s[S], u[C], upv[C];
        state_of(h, pc, s);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            u[cc] = q_s[(h * C + cc) * G4_LD + pc];
            upv[cc] = h > 0 ? q_s[((h - 1) * C + cc) * G4_LD + pc] : up0[cc];
        }
        cs += E::stage_cost(k, s, u, upv);
    }
    if (part == 0) {
        float s[S];
        state_of(H, pc, s);
        cs += E::terminal_cost(k, s);
    }
    cs = sum_parts(cs);
    if (wave == 0 && g == 0 && row0 + c < a.N) a.J[row0 + c] = cs * inv;
    for (int i = t; i < total; i += G4_BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        Q[gbase + i] = q_s[hc * G4_LD + r];
    }
}

// ---- rollout + cost (MPPI / affine modes of ctk_generic_net.hip: ctk_g_rollout_net) ---------------------------------------------------
// 16 trajectories per workgroup.  Inputs (interpolation, shifted nominal, clip, MPPI correction) are formed for all (step, trajectory)
// pairs before the recurrence, the costs from the states it leaves in LDS after it; the recurrence itself is gru4_step only.
// LDS: exchange slots | reductions | states xs[H+1][64][2] | inputs u[HC][17] | sample tile [16][ts] | e[16] | base, scale [HC] | interp tables
template <int ENV, int MODE, bool LOG>
__global__ __launch_bounds__(G4_BLOCK) void ctk_g_rollout_gru4(RolloutArgs a, typename Env<ENV>::K k, MppiK mk, const float* __restrict__ samples,
                                                              const float* __restrict__ base, const float* __restrict__ scale, int rng_kind,
                                                              const float* __restrict__ wperm, const float* __restrict__ hidden,
                                                              float* __restrict__ parts, NetFuse gz) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    static_assert(S + C <= 8, "the GRU's input tile holds 8 columns");
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C, cols = a.P, ts = tile_stride(cols);
    float* ex = lds;
    float* red_s = ex + G4_EX_FWD;
    float* xs_s = red_s + G4_RED;
    float* u_s = xs_s + (H + 1) * 128;
    float* tile = u_s + HC * G4_LD;
    float* e_s = tile + G4_TRAJ * ts;
    float* base_s = e_s + G4_TRAJ;
    float* scale_s = base_s + HC;
    float* w0_s = scale_s + HC;
    float* w1_s = w0_s + H;
    int* i0_s = reinterpret_cast<int*>(w1_s + H);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * G4_TRAJ;
    const int pc = t & 15, part = t >> 4;
    const int n = row0 + pc;
    const bool valid = n < a.N;

    const GruW wf = gru4_load_fwd(wperm, wave >> 1, wave & 1, lane);
    load_tile_early<G4_TRAJ, G4_BLOCK>(tile, samples, a, row0, MODE == CTK_G_MODE_MPPI ? mk.stdev : 1.0f, rng_kind, [&] {
        if constexpr (MODE == CTK_G_MODE_MPPI) {
            for (int h = t; h < H; h += G4_BLOCK) {
                const InterpEntry e = a.interp[h];
                i0_s[h] = e.i0; w0_s[h] = e.w0; w1_s[h] = e.w1;
            }
            for (int hc = t; hc < HC; hc += G4_BLOCK) {
                const int h = hc / C, cc = hc - h * C;
                base_s[hc] = base[min(h + 1, H - 1) * C + cc];
            }
        } else {
            for (int hc = t; hc < HC; hc += G4_BLOCK) { base_s[hc] = base[hc]; scale_s[hc] = scale[hc]; }
        }
    });
    float up0[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up0[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    const float s00 = g < S ? lane_state4(a, g) : 0.0f, s01 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    __syncthreads();

    auto sum_parts = [&](float v) {           // over the 16 parts of a trajectory (part = 4 * wave + g), fixed order; valid in every thread
        v = sum_over_groups(v);
        if (g == 0) red_s[wave * 16 + c] = v;
        __syncthreads();
        const float r = (red_s[c] + red_s[16 + c]) + (red_s[32 + c] + red_s[48 + c]);
        __syncthreads();
        return r;
    };

    // ---- inputs of all steps, and what of the cost depends on them only
    float corr = 0.0f;
    {
        const float* my = tile + pc * ts;
        const int Pm1 = cols / C - 1;
        for (int h = part; h < H; h += 16) {
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                float u;
                if constexpr (MODE == CTK_G_MODE_MPPI) {
                    const int i0 = i0_s[h], i1 = min(i0 + 1, Pm1);
                    const float du = my[i0 * C + cc] * w0_s[h] + my[i1 * C + cc] * w1_s[h];
                    u = fminf(fmaxf(base_s[h * C + cc] + du, a.lo[cc]), a.hi[cc]);
                    corr += mk.cc * (mk.k_dd * (du * du) + mk.R * u * du + mk.k_uu * (u * u));
                } else {
                    u = fminf(fmaxf(base_s[h * C + cc] + my[h * C + cc] * scale_s[h * C + cc], a.lo[cc]), a.hi[cc]);
                }
                u_s[(h * C + cc) * G4_LD + pc] = u;
                if constexpr (LOG || MODE == CTK_G_MODE_AFFINE) {
                    if (valid && a.Q_out) a.Q_out[(size_t)n * HC + h * C + cc] = u;
                }
            }
        }
        if constexpr (MODE == CTK_G_MODE_MPPI) corr = sum_parts(corr);
        else __syncthreads();
    }
    // ---- the recurrence
    {
        GruState st = gru_load_state(hidden, g);
        float sv0 = s00, sv1 = s01;
        for (int h = 0; h < H; ++h) {
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = u_s[(h * C + cc) * G4_LD + c];
            if (wave == (h & 3)) reinterpret_cast<float2*>(xs_s)[h * 64 + lane] = make_float2(sv0, sv1);
            float x0, x1;
            gru4_operands<S, C>(sv0, sv1, u, g, x0, x1);
            const MlpPair o = gru4_step(wf, st, x0, x1, ex, wave, lane, nullptr, nullptr);
            sv0 = o.lo; sv1 = o.hi;
        }
        if (wave == 0) reinterpret_cast<float2*>(xs_s)[H * 64 + lane] = make_float2(sv0, sv1);
        __syncthreads();
    }
    // ---- costs from the states
    float cs = 0.0f;
    for (int h = part; h <= H; h += 16) {
        float s[S];
#pragma unroll
        for (int j = 0; j < S; ++j) s[j] = xs_s[(h * 64 + (j & 3) * 16 + pc) * 2 + (j >> 2)];
        if constexpr (LOG) {
            if (valid && a.traj_out) {
#pragma unroll
                for (int j = 0; j < S; ++j) a.traj_out[((size_t)n * (H + 1) + h) * S + j] = s[j];
            }
        }
        if (h < H) {
            float u[C], upv[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                u[cc] = u_s[(h * C + cc) * G4_LD + pc];
                upv[cc] = h > 0 ? u_s[((h - 1) * C + cc) * G4_LD + pc] : up0[cc];
            }
            cs += E::stage_cost(k, s, u, upv);
        } else {
            cs += E::terminal_cost(k, s);
        }
    }
    cs = sum_parts(cs);
    const float J = cs * a.inv_Hp1 + corr;                      // of trajectory c = pc, in every thread
    if (wave == 0 && g == 0 && valid) a.J[n] = J;

    if constexpr (MODE == CTK_G_MODE_MPPI) {
        float rho = valid ? J : INFINITY;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) rho = fminf(rho, __shfl_xor(rho, o, 64));
        const float e = valid ? expf(mk.neg_inv_lbd * (J - rho)) : 0.0f;
        float aw = e;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) aw += __shfl_xor(aw, o, 64);
        if (wave == 0 && g == 0) e_s[c] = e;
        __syncthreads();
        float* rec = parts + (size_t)blockIdx.x * (2 + cols);
        const bool use_ll = gz.mode != 0;            // kernel-argument uniform: the records are handed over inside this launch
        unsigned long long* llr = gz.ll + (size_t)blockIdx.x * (2 + cols);
        if (t == 0) {
            if (use_ll) { ll_store(llr, rho, gz.up.seq); ll_store(llr + 1, aw, gz.up.seq); }
            else { rec[0] = rho; rec[1] = aw; }
        }
        for (int p = t; p < cols; p += G4_BLOCK) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < G4_TRAJ; ++r) acc += e_s[r] * tile[r * ts + p];
            if (use_ll) ll_store(llr + 2 + p, acc, gz.up.seq);
            else rec[2 + p] = acc;
        }
        if (use_ll && blockIdx.x == 0) {             // block 0 gathers every block's words, merges, updates / emits the shard record
            __syncthreads();
            mppi_ll_tail<C>(lds, gz.ll, (int)gridDim.x, cols, mk.neg_inv_lbd, gz.mode, gz.out_rec, gz.up);
        }
    }
}

// ---- predictor.update(s, Q0) (optimizer_mppi.py:195-197): the carried hidden state advanced by the measured state and the applied
// input — one workgroup, the same four-wave step (all 16 MFMA columns carry the same values; column 0 writes back), operands straight
// from the tables into registers (the one-wave form stages 59 KiB in LDS with 64 threads: 27 us behind every MPPI step)
template <int ENV>
__global__ __launch_bounds__(G4_BLOCK) void ctk_g_gru_advance4(RolloutArgs a, const float* __restrict__ u_dev, const float* __restrict__ wperm,
                                                              float* __restrict__ hidden) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    static_assert(S + C <= 8, "the GRU's input tile holds 8 columns");
    __shared__ float ex[G4_EX_FWD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    const GruW wf = gru4_load_fwd(wperm, wave >> 1, wave & 1, lane);
    GruState st = gru_load_state(hidden, g);
    float u[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) u[cc] = u_dev ? u_dev[cc] : a.u_prev[cc];
    const float sv0 = g < S ? lane_state4(a, g) : 0.0f, sv1 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    float x0, x1;
    gru4_operands<S, C>(sv0, sv1, u, g, x0, x1);
    (void)gru4_step(wf, st, x0, x1, ex, wave, lane, nullptr, nullptr);      // its barriers order every lane's read of `hidden` before the write below
    if (wave == 0 && c == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { hidden[16 * m + 4 * g + r] = st.h1[m][r]; hidden[32 + 16 * m + 4 * g + r] = st.h2[m][r]; }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static uint32_t g4_magic_of(int d) { return d >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; }

size_t ctk_g_rpgd_descent_gru4_lds(int H, int C) {
    return (size_t)(G4_EX_FWD + G4_EX_A + G4_EX_B + G4_RED + 2 * (H + 1) * 128 + 2 * H * C * G4_LD) * sizeof(float);
}

// the four-wave form: while a 16-plan workgroup per tile still leaves SIMDs idle (<= 256 CUs x 2 workgroups) and its LDS fits
bool ctk_g_rpgd_gru4_ok(int env, int N, int H) {
    static const bool off = getenv("CTK_RPGD_GRU_ONE_WAVE") != nullptr;      // diagnostic switch (A/B measurements)
    int S = 0, C = 0;
    CTK_FOR_ENV(env, EV, { S = Env<EV>::S; C = Env<EV>::C; });
    return !off && S + C <= 8 && N <= 8192 && ctk_g_rpgd_descent_gru4_lds(H, C) <= 160 * 1024;
}

size_t ctk_g_rpgd_scratch_floats_gru4(int N, int H) { return (size_t)((N + G4_TRAJ - 1) / G4_TRAJ) * H * 4 * G4_TAPE_F4 * 64 * 4; }

const char* ctk_g_rpgd_descent_gru4_name(int env) { return ctk_kernel_name("ctk_g_rpgd_descent_gru4<%d>", env); }

hipError_t ctk_launch_g_rpgd_descent_gru4(hipStream_t st, int env, const RolloutArgs& a_in, const float* params, float dt, int isteps,
                                          const AdamK& ad, float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters,
                                          const float* wperm, const float* wperm_bwd, const float* hidden, float* scratch,
                                          hipEvent_t e0, hipEvent_t e1) {
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        if constexpr (E::S + E::C <= 8) {
            RolloutArgs a = a_in;
            a.C = E::C; a.p_magic = g4_magic_of(a.H * E::C);
            const typename E::K k = E::derive(params, dt, isteps);
            const dim3 grid((a.N + G4_TRAJ - 1) / G4_TRAJ), block(G4_BLOCK);
            const size_t lds = ctk_g_rpgd_descent_gru4_lds(a.H, E::C);
            CTK_LAUNCH((ctk_g_rpgd_descent_gru4<EV>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wperm_bwd, hidden, scratch);
        } else {
            return hipErrorInvalidValue;
        }
    });
    return hipGetLastError();
}

size_t ctk_g_rollout_gru4_lds(int cols, int H, int C) {
    return (size_t)(G4_EX_FWD + G4_RED + (H + 1) * 128 + H * C * G4_LD + G4_TRAJ * tile_stride(cols) + G4_TRAJ + 2 * H * C + 3 * H) * sizeof(float);
}

// the four-wave rollout: while 16-trajectory workgroups leave the chip room (one-wave tiles: 4 per workgroup of ctk_g_rollout_net)
bool ctk_g_rollout_gru4_ok(int env, int N, int H, int cols) {
    static const bool off = getenv("CTK_GRU_ONE_WAVE") != nullptr;            // diagnostic switch (A/B measurements)
    int S = 0, C = 0;
    CTK_FOR_ENV(env, EV, { S = Env<EV>::S; C = Env<EV>::C; });
    return !off && S + C <= 8 && N <= 8192 && ctk_g_rollout_gru4_lds(cols, H, C) <= 160 * 1024;
}

int ctk_g_rollout_gru4_blocks(int N) { return (N + G4_TRAJ - 1) / G4_TRAJ; }

const char* ctk_g_rollout_gru4_name(int env, int mode, bool log) { return ctk_kernel_name("ctk_g_rollout_gru4<%d, %d, %4$s>", env, mode, 0, log ? "true" : "false"); }

hipError_t ctk_launch_g_rollout_gru4(hipStream_t st, int env, int mode, const RolloutArgs& a_in, const float* params, float dt, int isteps,
                                     const MppiK& mk, const float* samples, const float* base, const float* scale, int rng_kind,
                                     const float* wperm, const float* hidden, float* parts, bool log, hipEvent_t e0, hipEvent_t e1,
                                     const MppiFuse* fuse) {
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        if constexpr (E::S + E::C <= 8) {
            RolloutArgs a = a_in;
            const int cols = (mode == CTK_G_MODE_MPPI ? a_in.P : a_in.H) * E::C;
            a.P = cols; a.p_magic = g4_magic_of(cols); a.C = E::C;
            const typename E::K k = E::derive(params, dt, isteps);
            const dim3 grid(ctk_g_rollout_gru4_blocks(a.N)), block(G4_BLOCK);
            const NetFuse gz = ctk_net_fuse(fuse, mode, a, E::C, base, (int)grid.x, cols);
            const size_t lds = std::max(ctk_g_rollout_gru4_lds(cols, a.H, E::C), gz.mode ? merge_lds_staged(cols, (int)grid.x) : 0);
            if (mode == CTK_G_MODE_MPPI) {
                if (log) CTK_LAUNCH((ctk_g_rollout_gru4<EV, CTK_G_MODE_MPPI, true>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
                else CTK_LAUNCH((ctk_g_rollout_gru4<EV, CTK_G_MODE_MPPI, false>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
            } else {
                if (log) CTK_LAUNCH((ctk_g_rollout_gru4<EV, CTK_G_MODE_AFFINE, true>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
                else CTK_LAUNCH((ctk_g_rollout_gru4<EV, CTK_G_MODE_AFFINE, false>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
            }
        } else {
            return hipErrorInvalidValue;
        }
    });
    return hipGetLastError();
}

hipError_t ctk_launch_g_gru_advance4(hipStream_t st, int env, const RolloutArgs& a, const float* u_dev, const float* wperm, float* hidden) {
    CTK_FOR_ENV(env, EV, {
        if constexpr (Env<EV>::S + Env<EV>::C <= 8) hipLaunchKernelGGL((ctk_g_gru_advance4<EV>), dim3(1), dim3(G4_BLOCK), 0, st, a, u_dev, wperm, hidden);
        else return hipErrorInvalidValue;
    });
    return hipGetLastError();
}
