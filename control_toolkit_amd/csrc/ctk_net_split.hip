// ctk_net_split.hip — the network predictors of the template kernels with ONE 16-trajectory MFMA tile spread over SEVERAL waves (= SIMDs)
// of a workgroup, forward AND reverse, for small populations: the chip has 1 024 SIMDs, an MPC population of 1 024 rollouts is 64
// tiles, and on gfx950 a wave's fp32 MFMA and VALU time add up — so the step's matrix AND vector work is divided.
//
//   SplitGru   2 x 32 GRU + dense over FOUR waves (network inputs beyond 8 — control inputs — folded into the layer-1 biases).  One wave per tile (ctk_net.h: NetGru) is 164 dependent-ish MFMAs + 48
//              gate nonlinearities per lane forward, 172 MFMAs + the gate adjoints reverse, every A operand re-read from LDS: 6.3 ms for
//              RPGD at N 256 / H 50 x 10 Adam iterations (optimizer_rpgd.py:306-338 differentiates through whatever predictor it gets).
//     forward  ctk_gru.h: gru_layer.  Wave (m, q) owns hidden tile m; q = 0: r rows + the input half of the n rows, q = 1: z rows + the
//              recurrent half; 18 + 24 + 8 MFMAs per wave and step, operands in registers, two LDS exchanges per layer.  Each wave turns
//              TWO of a lane's four units of tile m into gates (registers 2q, 2q+1) and tapes exactly those (GruPairTape).
//     reverse  the same ownership: wave (m, q) forms the gate adjoints of ITS two units element-wise from its own tape, and those are
//              k-steps (gate G, tile m, register 2q+i) of the transposed products W_i^T dgi, W_h^T dgh (ctk_net.h: layer_products) — the
//              contraction over the 96 gate neurons is split four ways, every wave accumulates partial tiles of ALL outputs (2 + 24 + 18
//              MFMAs per step), and the partial sums meet through LDS once per layer: two barriers per reverse step.
//   SplitMlp   (S+C)-32-32-S tanh MLP over TWO waves (any environment; K3: a third layer-1 k-step where S + C > 8).  Wave m owns hidden
//              units 16m .. 16m+15 of both layers (ctk_mlp.h: mlp_step_pair is CartPole's thin-layer version of this): forward 2-3 + 8 + 4
//              MFMAs and 8 tanh per wave (one wave: 28-30 and 16), h1 halves and partial outputs meet through LDS; reverse 2 + 8 + 4:
//              W2^T and W1^T contract over the wave's OWN hidden units, partial tiles meet through LDS — two barriers each way.
//   the cost   nothing of it rides on the recurrence: the forward pass leaves the states in LDS, and the stage / terminal cost, their
//              state gradients, the inputs (interpolation, clip, MPPI correction) and the input-only gradient terms are evaluated
//              before / after it by all threads over (step, trajectory) pairs.
//
// Tables: the per-lane operand tables of NetGru / NetMlp as they are (ctk_net.h, ctk_mlp.h); a wave picks its own entries once per
// launch into registers.  Tape: TAPE_F4 float4 per lane, wave and step in the L2-resident scratch, read back one step ahead of its use.
#include "ctk_rollout.h"
#include "ctk_env.h"
#include "ctk_gru.h"
#include "ctk_net.h"
#include "ctk_adam.h"
#include "ctk_launch.h"
#include <atomic>
#include "ctk_mppi_merge.h"
#include <algorithm>

NetFuse ctk_net_fuse(const MppiFuse* fuse, int mode, const RolloutArgs& a, int C, const float* u_nom, int blocks, int cols);   // ctk_generic_net.hip

constexpr int G4_TRAJ = 16, G4_LD = G4_TRAJ + 1;
constexpr int G4_TAPE_F4 = 5;                       // float4 per lane, wave and step
constexpr int G4_EX_FWD = GRU_EX_FLOATS;            // 3072: gru_layer's exchange slots
constexpr int G4_EX_A = 4 * 4 * 64 * 4;             // reverse, layer 2: [4 waves][dx0 dx1 dhp0 dhp1][64] float4
constexpr int G4_EX_B = 4 * 3 * 64 * 4;             // reverse, layer 1: [4 waves][din dhp0 dhp1][64] float4
constexpr int G4_RED = 4 * 16 + 16;
constexpr int M2_EX_FWD = 2 * 64 * 4 + 2 * 64 * 2;   // SplitMlp forward: [2 waves][64] float4 h1 halves + [2][64] float2 partial outputs
constexpr int M2_EX_BWD = 2 * 64 * 4 + 2 * 64 * 4;   // SplitMlp reverse: [2][64] float4 W2^T partials + [2][64] float4 W1^T partials

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
    float lo = 0.0f, hi = 0.0f, exi = 0.0f;
    float c1[2] = {direct[0], direct[1]};
    const float2* B2 = reinterpret_cast<const float2*>(exB);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float2 x = B2[((s * 3 + 0) * 64 + lane) * 2], y = B2[((s * 3 + 1 + m) * 64 + lane) * 2 + q];
        lo += x.x; hi += x.y; c1[0] += y.x; c1[1] += y.y;
        exi += B2[((s * 3 + 0) * 64 + lane) * 2 + 1].x;                // row 4g + 2 of the input tile: network input 8 + g (zero rows where S + C <= 8)
    }
    ad.dh1[0] = c1[0]; ad.dh1[1] = c1[1];
    return MlpPair{lo, hi, exi};
}

// network operands of a step (inputs g, 4+g, 8+g) from the (component g, component 4+g) state layout and the step's inputs
template <int S, int C>
CTK_DEV void split_operands(float sv0, float sv1, const float (&u)[C], int g, float& x0, float& x1, float& x2) {
    x0 = (g < S) ? sv0 : 0.0f;
    x1 = (4 + g < S) ? sv1 : 0.0f;
    x2 = 0.0f;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        x0 = (g == S + cc) ? u[cc] : x0;
        x1 = (4 + g == S + cc) ? u[cc] : x1;
        x2 = (8 + g == S + cc) ? u[cc] : x2;
    }
}

// ---- the split policies: WAVES per tile; Fwd {load, begin, step<TAPE>} / Bwd {load, begin, vjp}; exchange slots in floats ----------------
struct SplitGru {
    static constexpr int WAVES = 4, TAPE_F4 = G4_TAPE_F4, EX_FWD = G4_EX_FWD, EX_BWD = G4_EX_A + G4_EX_B, NET = NET_GRU;
    struct Fwd {
        GruW w;
        GruState st;
        const float* tab3;                  // layer 1's third k-step, Wi3[gate][tile] (ctk_net.h: behind the reverse table)
        int wv;
        // More than 8 network inputs (round 4): inputs 8 + f are control inputs (S <= 8), known before the step — their columns enter
        // layer 1 as a rank-NF update of the accumulators' initial values (this wave's r or z rows; the n rows' input half on q = 0), on
        // the VALU, like SplitMlp's folded columns: no third k-step on the recurrence.
        f32x4 xA[4], xB[4], bA0, bB0;
        CTK_DEV void load(const float* __restrict__ tab, int wave, int lane) {
            w = gru4_load_fwd(tab, wave >> 1, wave & 1, lane);
            tab3 = tab + (size_t)(GRUG_FWD + GRUG_BWD) * 64; wv = wave;
        }
        CTK_DEV void begin(const float* __restrict__ hidden, int g) { st = gru_load_state(hidden, g); }
        template <int S, int C> static constexpr int fold_n() { return S + C > 8 ? S + C - 8 : 0; }
        template <int S, int C>
        CTK_DEV void fold_load(int lane) {
            constexpr int NF = fold_n<S, C>();
            if constexpr (NF > 0) {
                const int m = wv >> 1, q = wv & 1, g = lane >> 4;
                const float a3 = tab3[(q * 2 + m) * 64 + lane], b3 = tab3[(2 * 2 + m) * 64 + lane];      // A operands: lane (i, k) holds W_i[row i][8 + k]
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        xA[f][r] = __shfl(a3, 4 * g + r + 16 * f, 64);
                        xB[f][r] = q == 0 ? __shfl(b3, 4 * g + r + 16 * f, 64) : 0.0f;
                    }
                bA0 = w.bA1; bB0 = w.bB1;
            }
        }
        template <int S, int C>
        CTK_DEV void fold_inputs(const float (&u)[C]) {
            constexpr int NF = fold_n<S, C>();
            if constexpr (NF > 0) {
                f32x4 a = bA0, b = bB0;
#pragma unroll
                for (int f = 0; f < NF; ++f) { a += xA[f] * u[8 - S + f]; b += xB[f] * u[8 - S + f]; }
                w.bA1 = a; w.bB1 = b;
            }
        }
        // tq: this wave's tape of the step (+ i * 64 per float4), used when TAPE
        template <bool TAPE, int = 0, int = 0>
        CTK_DEV MlpPair step(float x0, float x1, float, float* ex, int wave, int lane, float4* tq) {
            GruPairTape t1, t2;
            const MlpPair o = gru4_step(w, st, x0, x1, ex, wave, lane, TAPE ? &t1 : nullptr, TAPE ? &t2 : nullptr);
            if constexpr (TAPE) {
                tq[0 * 64] = make_float4(t1.r[0], t1.r[1], t1.z[0], t1.z[1]);
                tq[1 * 64] = make_float4(t1.n[0], t1.n[1], t1.ghn[0], t1.ghn[1]);
                tq[2 * 64] = make_float4(t1.hp[0], t1.hp[1], t2.hp[0], t2.hp[1]);
                tq[3 * 64] = make_float4(t2.r[0], t2.r[1], t2.z[0], t2.z[1]);
                tq[4 * 64] = make_float4(t2.n[0], t2.n[1], t2.ghn[0], t2.ghn[1]);
            }
            return o;
        }
    };
    struct Bwd {
        Gru4BwdW w;
        Gru4Adj adj;
        CTK_DEV void load(const float* __restrict__ tab, int wave, int lane) { w = gru4_load_bwd(tab, wave >> 1, wave & 1, lane); }
        CTK_DEV void begin() { adj = Gru4Adj{{0.f, 0.f}, {0.f, 0.f}}; }
        CTK_DEV MlpPair vjp(const float4 (&tp)[TAPE_F4], float lam0, float lam1, float* ex, int wave, int lane) {
            return gru4_vjp(w, adj, tp, lam0, lam1, ex, ex + G4_EX_A, wave, lane);
        }
    };
};

// ---- the 64-unit MLP (ctk_mlp_wide.h: hidden widths 33..64) over FOUR waves, forward only (round 4) ---------------------------------------
// One wave per tile runs 12 + 64 + 16 matrix products per step (2.2 us: MPPI N 1 024 / H 50 = 116 us against 45 for the 32-unit network).
// Here wave m owns hidden row tile m of both layers: layer 1 is its own KS products, the four waves exchange their h1 tiles (LDS, one
// barrier), layer 2 is 16 products (its own tile's k-steps first, the others' as they arrive), layer 3 the four k-steps of its own h2
// units, and the four partial outputs meet through LDS (second barrier) — 2..3 + 16 + 4 products per wave and step, two exchanges like
// SplitMlp.  Same operand table as NetMlpWideT (ctk_mlp_wide.h); every wave ends with the same next state (one association).
// The reverse mode (RPGD) keeps the one-wave kernels for this width.
constexpr int M4_EX_FWD = 4 * 64 * 4 + 4 * 64 * 2;
template <bool K3>
struct SplitMlp64 {
    static constexpr int WAVES = 4, TAPE_F4 = 2, EX_FWD = M4_EX_FWD, EX_BWD = 0, NET = NET_MLP64;
    struct Fwd {
        float w1[3], w2[MLPW_KH], w3[4];
        f32x4 b1, b2;
        float b3lo, b3hi;
        CTK_DEV void load(const float* __restrict__ tab, int m, int lane) {
            const float* p = tab + (size_t)lane * MLPW_FWD_PER_LANE;
            // per lane: w1[T][3] | w2[T][KH] | w3[KH] | b1[T][4] | b2[T][4] | b3[4]
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) w1[ks] = p[m * 3 + ks];
#pragma unroll
            for (int o = 0; o < 4; ++o)                                    // in the order of use: the own tile's k-steps, then tiles m + 1, m + 2, m + 3
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) w2[o * 4 + jj] = p[MLPW_T * 3 + m * MLPW_KH + ((m + o) & 3) * 4 + jj];
#pragma unroll
            for (int j = 0; j < 4; ++j) w3[j] = p[MLPW_T * 3 + MLPW_T * MLPW_KH + 4 * m + j];
            const int ob = MLPW_T * 3 + MLPW_T * MLPW_KH + MLPW_KH;
#pragma unroll
            for (int r = 0; r < 4; ++r) { b1[r] = p[ob + m * 4 + r]; b2[r] = p[ob + 4 * MLPW_T + m * 4 + r]; }
            b3lo = p[ob + 8 * MLPW_T]; b3hi = p[ob + 8 * MLPW_T + 1];
        }
        CTK_DEV void begin(const float*, int) {}
        template <int S, int C> CTK_DEV void fold_load(int) {}
        template <int S, int C> CTK_DEV void fold_inputs(const float (&)[C]) {}
        template <int TAPE, int = 0, int = 0>
        CTK_DEV MlpPair step(float x0, float x1, float x2, float* ex, int m, int lane, float4*) {
            static_assert(TAPE == 0, "forward only");
            float4* ex_h = reinterpret_cast<float4*>(ex);                  // [4][64]
            float2* ex_o = reinterpret_cast<float2*>(ex + 4 * 64 * 4);     // [4][64]
            f32x4 a = CTK_MFMA(w1[0], x0, b1);
            a = CTK_MFMA(w1[1], x1, a);
            if constexpr (K3) a = CTK_MFMA(w1[2], x2, a);
            const f32x4 h1m = ctk_tanhf4(a);
            ex_h[m * 64 + lane] = st4(h1m);
            __syncthreads();
            f32x4 c = b2;
#pragma unroll
            for (int o = 0; o < 4; ++o) {                                  // own tile first: the others are still arriving
                const int mm = (m + o) & 3;
                const f32x4 hx = o == 0 ? h1m : ld4(ex_h + mm * 64 + lane);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) c = CTK_MFMA(w2[o * 4 + jj], hx[jj], c);
            }
            const f32x4 h2m = ctk_tanhf4(c);
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 p0 = CTK_MFMA(w3[0], h2m[0], z), p1 = CTK_MFMA(w3[1], h2m[1], z);
            p0 = CTK_MFMA(w3[2], h2m[2], p0);
            p1 = CTK_MFMA(w3[3], h2m[3], p1);
            ex_o[m * 64 + lane] = make_float2(p0[0] + p1[0], p0[1] + p1[1]);
            __syncthreads();
            const float2 q0 = ex_o[lane], q1 = ex_o[64 + lane], q2 = ex_o[128 + lane], q3 = ex_o[192 + lane];
            return MlpPair{((q0.x + q1.x) + (q2.x + q3.x)) + b3lo, ((q0.y + q1.y) + (q2.y + q3.y)) + b3hi};   // the same association in every wave
        }
    };
    struct Bwd {};      // (not built: RPGD with this width runs ctk_generic_net.hip's one-wave kernels)
};

template <bool K3>
struct SplitMlp {
    static constexpr int WAVES = 2, TAPE_F4 = 2, EX_FWD = M2_EX_FWD, EX_BWD = M2_EX_BWD, NET = NET_MLP;
    struct Fwd {
        float w1[3], w2o[4], w2x[4], w3[4];     // own row tile m: layer 1 k-steps; layer 2 k-steps of the OWN / the OTHER wave's units; layer 3 k-steps of the own units
        f32x4 b1, b2;
        float b3lo, b3hi;
        CTK_DEV void load(const float* __restrict__ tab, int m, int) {
            const MlpFwdW f = mlp_load_fwd(tab);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) w1[ks] = m ? f.w1[1][ks] : f.w1[0][ks];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w2o[j] = m ? f.w2[1][4 + j] : f.w2[0][j];
                w2x[j] = m ? f.w2[1][j] : f.w2[0][4 + j];
                w3[j] = m ? f.w3[4 + j] : f.w3[j];
            }
            b1 = m ? f.b1[1] : f.b1[0]; b2 = m ? f.b2[1] : f.b2[0];
            b3lo = f.b3[0]; b3hi = f.b3[1];
        }
        CTK_DEV void begin(const float*, int) {}
        // Where the LAST layer-1 k-step holds control inputs only (S <= 4 * (k-steps - 1): CartPole 4 + 1, Hover 7 + 3) its matrix product is
        // a rank-NF update that does not depend on the state: b1u = b1 + sum_f W1[:, input f] * u_f is formed as soon as the inputs are
        // known and opens the step's first product as its accumulator — one dependent MFMA less per step (what CartPole's own thin
        // layer 1 does, ctk_mlp.h).  S = C = 0 (the default of step): no folding.
        static constexpr int KS = K3 ? 3 : 2;
        template <int S, int C> static constexpr int fold_n() { return (S + C > 4 * (KS - 1) && S <= 4 * (KS - 1)) ? S + C - 4 * (KS - 1) : 0; }
        f32x4 w1u[4], b1u;
        template <int S, int C>
        CTK_DEV void fold_load(int lane) {                                 // after load(): this lane's accumulator rows of the folded columns
            constexpr int NF = fold_n<S, C>();
            const int g = lane >> 4;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) w1u[f][r] = __shfl(w1[KS - 1], 4 * g + r + 16 * f, 64);    // A operand: lane (i, k) holds W1[16m + i][4 (KS-1) + k]
            b1u = b1;
        }
        template <int S, int C>
        CTK_DEV void fold_inputs(const float (&u)[C]) {
            constexpr int NF = fold_n<S, C>(), CC0 = 4 * (KS - 1) - S;      // the first folded input channel
            if constexpr (NF > 0) {
                f32x4 b = b1;
#pragma unroll
                for (int f = 0; f < NF; ++f) b += w1u[f] * u[CC0 + f];
                b1u = b;
            }
        }
        template <int TAPE, int S = 0, int C = 0>   // TAPE 0: none; 1: float4 tape at tq[0], tq[64]; 2: the same, stored through to memory from where the wave waits anyway
        CTK_DEV MlpPair step(float x0, float x1, float x2, float* ex, int m, int lane, float4* tq) {
            float4* ex_h = reinterpret_cast<float4*>(ex);                  // [2][64]
            float2* ex_o = reinterpret_cast<float2*>(ex + 2 * 64 * 4);     // [2][64]
            constexpr bool FOLD = fold_n<S, C>() > 0;
            f32x4 a = CTK_MFMA(w1[0], x0, FOLD ? b1u : b1);
            if constexpr (!(FOLD && !K3)) a = CTK_MFMA(w1[1], x1, a);
            if constexpr (K3 && !FOLD) a = CTK_MFMA(w1[2], x2, a);
            const f32x4 h1m = ctk_tanhf4(a);
            ex_h[m * 64 + lane] = st4(h1m);
            __syncthreads();
            f32x4 c = b2;
#pragma unroll
            for (int j = 0; j < 4; ++j) c = CTK_MFMA(w2o[j], h1m[j], c);   // own half first: the other one is still arriving
            const f32x4 h1x = ld4(ex_h + (m ^ 1) * 64 + lane);
            if constexpr (TAPE == 2) st4_through(tq, st4(h1m));
#pragma unroll
            for (int j = 0; j < 4; ++j) c = CTK_MFMA(w2x[j], h1x[j], c);
            const f32x4 h2m = ctk_tanhf4(c);
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 p0 = CTK_MFMA(w3[0], h2m[0], z), p1 = CTK_MFMA(w3[1], h2m[1], z);
            p0 = CTK_MFMA(w3[2], h2m[2], p0);
            p1 = CTK_MFMA(w3[3], h2m[3], p1);
            if constexpr (TAPE == 2) st4_through(tq + 64, st4(h2m));
            const float mylo = p0[0] + p1[0], myhi = p0[1] + p1[1];
            ex_o[m * 64 + lane] = make_float2(mylo, myhi);
            if constexpr (TAPE == 1) { tq[0] = st4(h1m); tq[64] = st4(h2m); }
            __syncthreads();
            const float2 o = ex_o[(m ^ 1) * 64 + lane];
            // the same association in both waves: (units 0..15) + (units 16..31) + bias
            return MlpPair{(m == 0 ? mylo + o.x : o.x + mylo) + b3lo, (m == 0 ? myhi + o.y : o.y + myhi) + b3hi};
        }
    };
    struct Bwd {
        float w3t[2], w2t[2][4], w1t[4];        // own hidden tile m: W3^T k-steps (outputs g, 4+g); W2^T [input tile][own k-steps]; W1^T own k-steps
        CTK_DEV void load(const float* __restrict__ tab, int m, int) {
            const MlpBwdW2 b = mlp_load_bwd2(tab);
            w3t[0] = m ? b.w3t[1][0] : b.w3t[0][0]; w3t[1] = m ? b.w3t[1][1] : b.w3t[0][1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w2t[0][j] = m ? b.w2t[0][4 + j] : b.w2t[0][j];
                w2t[1][j] = m ? b.w2t[1][4 + j] : b.w2t[1][j];
                w1t[j] = m ? b.w1t[4 + j] : b.w1t[j];
            }
        }
        CTK_DEV void begin() {}
        // tp: (h1 half, h2 half) of this wave.  Returns the adjoints of network inputs g, 4+g, 8+g — identical in both waves
        CTK_DEV MlpPair vjp(const float4 (&tp)[TAPE_F4], float lam0, float lam1, float* ex, int m, int lane) {
            float4* exA = reinterpret_cast<float4*>(ex);                   // [2][64]
            float4* exB = exA + 2 * 64;                                    // [2][64]
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 h1m = ld4(&tp[0]), h2m = ld4(&tp[1]);
            f32x4 t = CTK_MFMA(w3t[0], lam0, z);
            t = CTK_MFMA(w3t[1], lam1, t);
            f32x4 s0 = z, s1 = z;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d2 = t[r] * (1.0f - h2m[r] * h2m[r]);
                s0 = CTK_MFMA(w2t[0][r], d2, s0);
                s1 = CTK_MFMA(w2t[1][r], d2, s1);
            }
            exA[m * 64 + lane] = st4(m ? s0 : s1);                          // the OTHER wave's tile
            __syncthreads();
            const f32x4 so = ld4(exA + (m ^ 1) * 64 + lane), sm = m ? s1 : s0;
            f32x4 o0 = z, o1 = z;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float da = (sm[r] + so[r]) * (1.0f - h1m[r] * h1m[r]), db = (sm[r + 1] + so[r + 1]) * (1.0f - h1m[r + 1] * h1m[r + 1]);
                o0 = CTK_MFMA(w1t[r], da, o0);
                o1 = CTK_MFMA(w1t[r + 1], db, o1);
            }
            const f32x4 mine = o0 + o1;
            exB[m * 64 + lane] = st4(mine);
            __syncthreads();
            const f32x4 oth = ld4(exB + (m ^ 1) * 64 + lane);
            return m == 0 ? MlpPair{mine[0] + oth[0], mine[1] + oth[1], mine[2] + oth[2]} : MlpPair{oth[0] + mine[0], oth[1] + mine[1], oth[2] + mine[2]};
        }
    };
};

// ---- RPGD descent ----------------------------------------------------------------------------------------------------------------
// LDS: exchange slots | states xs[H+1][64][2] | cost-gradient terms gs[H+1][64][2] | plans q[HC][17] | gradients g[HC][17] | reductions
template <int ENV, class SP>
__global__ __launch_bounds__(64 * SP::WAVES) void ctk_g_rpgd_descent_split(RolloutArgs a, typename Env<ENV>::K k, AdamK ad, float* __restrict__ Q,
                                                                   float* __restrict__ mom, float* __restrict__ var,
                                                                   const float* __restrict__ bc_table, int bc_len, int t0, int iters,
                                                                   const float* __restrict__ wperm, const float* __restrict__ wperm_bwd,
                                                                   const float* __restrict__ hidden, float* __restrict__ scratch) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C, WAVES = SP::WAVES, BLOCK = 64 * WAVES, NPARTS = BLOCK / G4_TRAJ, TAPE_F4 = SP::TAPE_F4;
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C;
    float* ex = lds;
    float* exR = ex + SP::EX_FWD;
    float* red_s = exR + SP::EX_BWD;
    float* xs_s = red_s + G4_RED;
    float* gs_s = xs_s + (H + 1) * 128;
    float* q_s = gs_s + (H + 1) * 128;
    float* g_s = q_s + HC * G4_LD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * G4_TRAJ;
    const int rows = min(G4_TRAJ, a.N - row0);
    const int total = rows * HC;
    const size_t gbase = (size_t)row0 * HC;
    float4* tape = reinterpret_cast<float4*>(scratch) + ((size_t)blockIdx.x * H * WAVES + wave) * TAPE_F4 * 64 + lane;   // + h * tape_step + i * 64
    const size_t tape_step = (size_t)WAVES * TAPE_F4 * 64;

    for (int i = t; i < G4_TRAJ * HC; i += BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        q_s[hc * G4_LD + r] = i < total ? Q[gbase + i] : 0.0f;
    }
    typename SP::Fwd nf;
    typename SP::Bwd nb;
    nf.load(wperm, wave, lane);
    nf.template fold_load<S, C>(lane);
    nb.load(wperm_bwd, wave, lane);
    float up0[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up0[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    const float inv = a.inv_Hp1;
    const float s00 = g < S ? lane_state4(a, g) : 0.0f, s01 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    const int pc = t & 15, part = t >> 4;                 // (plan, part) decomposition of the parallel passes: part = 4 * wave + g
    __syncthreads();

    auto forward = [&](auto taping) {
        constexpr bool TAPE = decltype(taping)::value;
        nf.begin(hidden, g);
        float sv0 = s00, sv1 = s01;
        for (int h = 0; h < H; ++h) {
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = q_s[(h * C + cc) * G4_LD + c];
            if (wave == (h & (WAVES - 1))) reinterpret_cast<float2*>(xs_s)[h * 64 + lane] = make_float2(sv0, sv1);
            float x0, x1, x2;
            split_operands<S, C>(sv0, sv1, u, g, x0, x1, x2);
            nf.template fold_inputs<S, C>(u);
            const MlpPair o = nf.template step<TAPE, S, C>(x0, x1, x2, ex, wave, lane, tape + (size_t)h * tape_step);
            sv0 = o.lo; sv1 = o.hi;
        }
        if (wave == 0) reinterpret_cast<float2*>(xs_s)[H * 64 + lane] = make_float2(sv0, sv1);
        __syncthreads();
    };
    auto state_of = [&](int h, int p, float (&s)[S]) {
#pragma unroll
        for (int j = 0; j < S; ++j) s[j] = xs_s[(h * 64 + (j & 3) * 16 + p) * 2 + (j >> 2)];
    };
    // sum over the NPARTS parts of a plan, in a fixed order; valid in every thread
    auto sum_parts = [&](float v) {
        v = sum_over_groups(v);
        if (g == 0) red_s[wave * 16 + c] = v;
        __syncthreads();
        float r = red_s[c] + red_s[16 + c];
        if constexpr (WAVES == 4) r += red_s[32 + c] + red_s[48 + c];
        __syncthreads();
        return r;
    };

    for (int it = 0; it < iters; ++it) {
        forward(std::true_type{});
        // ---- everything of the gradient that does not ride on the adjoint chain, over (step, plan) pairs
        for (int idx = t; idx < (H + 1) * G4_TRAJ; idx += BLOCK) {
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
            nb.begin();
            float4 nxt[TAPE_F4];
#pragma unroll
            for (int i = 0; i < TAPE_F4; ++i) nxt[i] = tape[(size_t)(H - 1) * tape_step + i * 64];
            for (int h = H - 1; h >= 0; --h) {
                float4 cur[TAPE_F4];
#pragma unroll
                for (int i = 0; i < TAPE_F4; ++i) cur[i] = nxt[i];
                if (h > 0) {
#pragma unroll
                    for (int i = 0; i < TAPE_F4; ++i) nxt[i] = tape[(size_t)(h - 1) * tape_step + i * 64];
                }
                const MlpPair d = nb.vjp(cur, lam.x, lam.y, exR, wave, lane);
                const float2 gsv = reinterpret_cast<const float2*>(gs_s)[h * 64 + lane];
                if (wave == 0) {
#pragma unroll
                    for (int cc = 0; cc < C; ++cc) {
                        const int kk = S + cc;                         // network input index of control input cc
                        if (g == (kk & 3)) g_s[(h * C + cc) * G4_LD + c] += (kk >= 8 ? d.ex : (kk >= 4 ? d.hi : d.lo));
                    }
                }
                lam.x = gsv.x + (g < S ? d.lo : 0.0f);
                lam.y = gsv.y + (4 + g < S ? d.hi : 0.0f);
            }
        }
        __syncthreads();
        // ---- per-plan clip_by_norm, Adam, clip
        float n2 = 0.0f;
        for (int hc = part; hc < HC; hc += NPARTS) { const float x = g_s[hc * G4_LD + pc]; n2 += x * x; }
        n2 = sum_parts(n2);
        const float scl = ad.clip / fmaxf(sqrtf(n2), ad.clip);       // of plan pc = c
        if (wave == 0 && g == 0) red_s[64 + c] = scl;
        __syncthreads();
        const int ti = t0 + it + 1;
        const float bc1 = ti <= bc_len ? bc_table[2 * (ti - 1)] : 1.0f;
        const float bc2 = ti <= bc_len ? bc_table[2 * (ti - 1) + 1] : 1.0f;
        for (int i = t; i < total; i += BLOCK) {
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
    for (int h = part; h < H; h += NPARTS) {
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
    for (int i = t; i < total; i += BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        Q[gbase + i] = q_s[hc * G4_LD + r];
    }
}

// ---- RPGD descent with the MLP predictor, WIDE form (the template counterpart of ctk_rpgd.hip: ctk_rpgd_mlp_wide / _jacobians) -----------
// The reverse sweep of ctk_g_rpgd_descent_split is a chain of H dependent network products per tile (two barriers per step) while the chip
// idles.  But lambda_h = c_h + J_h^T lambda_{h+1} is LINEAR in lambda once the step Jacobians J_h = d s_{h+1} / d (s_h, u_h) — S x (S + C)
// — are known, and those depend on the taped activations only:
//   ctk_g_rpgd_jac_split    one wave per (16-plan tile, step), grid-wide: S + C forward-mode tangent passes through the network on the matrix
//                           cores (layer 1 of tangent j is column j of W1 — one MFMA per tile against an indicator B operand; 2 + 16 + 8
//                           MFMAs per tangent); lane (c, g) ends with rows g and 4+g of column j for plan c and stores them into the record
//                           [tile][step][plan 16][column 16][row 8];
//   ctk_g_rpgd_wide_split   the phase launch (16 plans per workgroup, two waves): [update from the previous launch's states and Jacobians:
//                           cost gradients over (step, plan) pairs, the adjoint chain — 8 lanes per plan, lane i owns columns i and 8+i,
//                           lambda passed round by DPP quad broadcasts, records read through a register ring —, clip_by_norm, Adam] then
//                           [forward with tape (SplitMlp) | final cost pass].
// iters + 1 phase launches and iters Jacobian launches per MPC step, in stream order.  Scratch per tile: activations [H][2][2][64] float4,
// states [H+1][64][2], records [H][16][16][8].
constexpr int GW_REC = 16 * 16 * 8;                                  // floats of one (tile, step) record
__host__ __device__ inline size_t gw_xs_off(int H) { return (size_t)H * (2 * 2 * 64 * 4); }
__host__ __device__ inline size_t gw_rec_off(int H) { return gw_xs_off(H) + (size_t)(H + 1) * 128; }
__host__ __device__ inline size_t gw_gs_off(int H) { return gw_rec_off(H) + (size_t)H * GW_REC; }                 // cost-gradient terms gs[H+1][64][2]
__host__ __device__ inline size_t gw_gd_off(int H) { return gw_gs_off(H) + (size_t)(H + 1) * 128; }               // direct input-gradient terms g[HC][17]
__host__ __device__ inline size_t gw_gd_floats(int H, int C) { return ((size_t)H * C * G4_LD + 3) & ~(size_t)3; }
__host__ __device__ inline size_t gw_flag_off(int H, int C) { return gw_gd_off(H) + gw_gd_floats(H, C); }     // hand-off flags [H+1][2 waves] (launch sequence numbers)
__host__ __device__ inline size_t gw_tile_floats(int H, int C) { return gw_flag_off(H, C) + (((size_t)(H + 1) * 2 + 3) & ~(size_t)3); }

// S + C forward-mode tangents through the network (j = j0, j0 + jstep, ...): column j of d s' / d (s, u) of plan c, rows g and 4 + g, into rec
template <int IO>
CTK_DEV void jac_tangents(const MlpFwdW& w, const f32x4 (&d1)[2], const f32x4 (&d2)[2], float* rec, int g, int j0, int jstep) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < IO; ++j) {
        if (jstep != 1 && (j % jstep) != j0) continue;                  // (wave-uniform)
        const float ind = (g == (j & 3)) ? 1.0f : 0.0f;                 // B = e_j: input j lives in k-step j / 4, k-slot j % 4
        f32x4 t1[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) t1[m] = CTK_MFMA(w.w1[m][j >> 2], ind, z) * d1[m];
        f32x4 z0 = z, z1 = z;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float b = t1[q >> 2][q & 3];
            z0 = CTK_MFMA(w.w2[0][q], b, z0);
            z1 = CTK_MFMA(w.w2[1][q], b, z1);
        }
        const f32x4 t2[2] = {z0 * d2[0], z1 * d2[1]};
        f32x4 o0 = z, o1 = z;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            o0 = CTK_MFMA(w.w3[q], t2[q >> 2][q & 3], o0);
            o1 = CTK_MFMA(w.w3[q + 1], t2[(q + 1) >> 2][(q + 1) & 3], o1);
        }
        rec[j * 8 + g] = o0[0] + o1[0];                                 // d s'_g / d x_j, d s'_{4+g} / d x_j of plan c
        rec[j * 8 + 4 + g] = o0[1] + o1[1];
    }
}

template <int ENV, bool K3>
__global__ __launch_bounds__(64) void ctk_g_rpgd_jac_split(RolloutArgs a, typename Env<ENV>::K k, const float* __restrict__ Q, const float* __restrict__ wperm,
                                                          float* __restrict__ scratch) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C, IO = S + C;
    static_assert(IO <= (K3 ? 12 : 8), "network inputs");
    const int H = a.H;
    const int tile = blockIdx.x / H, h = blockIdx.x - tile * H;
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const MlpFwdW w = mlp_load_fwd(wperm);
    float* base = scratch + (size_t)tile * gw_tile_floats(H, C);
    // what of the gradient does not ride on the adjoint chain, for this step and (lanes g = 1 of the last step's wave) the terminal one:
    // the cost's state gradient of plan c and the input-only terms — here, where the whole chip works, not in the phase launch's 16
    // workgroups; written in the phase launch's LDS layout, which copies them in with LDS-DMA.  Operands are loaded here, with the
    // activations; the arithmetic follows the tangent passes
    const int hh = g == 0 ? h : H;
    const bool costs = g == 0 || (g == 1 && h == H - 1);
    float cs_s[S], cs_u[C], cs_up[C], cs_un[C];
    {
        const float* xs_g = base + gw_xs_off(H);
#pragma unroll
        for (int j = 0; j < S; ++j) cs_s[j] = xs_g[(hh * 64 + (j & 3) * 16 + c) * 2 + (j >> 2)];
        const int HC = H * C, n = tile * G4_TRAJ + c;
        const bool live = n < a.N;                                       // (plans beyond N read as zeros, as the phase launch holds them)
        const float* q = Q + (size_t)(live ? n : 0) * HC;
        const int h0 = min(hh, H - 1);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const float qu = q[h0 * C + cc], qp = q[max(h0 - 1, 0) * C + cc], qn = q[min(h0 + 1, H - 1) * C + cc];
            cs_u[cc] = live ? qu : 0.0f;
            cs_up[cc] = h0 > 0 ? (live ? qp : 0.0f) : (a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc]);
            cs_un[cc] = h0 + 1 < H && live ? qn : 0.0f;
        }
    }
    const float4* act = reinterpret_cast<const float4*>(base) + (size_t)h * (2 * 2 * 64) + lane;     // [wave m][h1 | h2][64]
    f32x4 d1[2], d2[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f32x4 h1 = ld4(act + (m * 2 + 0) * 64), h2 = ld4(act + (m * 2 + 1) * 64);
        d1[m] = 1.0f - h1 * h1;
        d2[m] = 1.0f - h2 * h2;
    }
    jac_tangents<IO>(w, d1, d2, base + gw_rec_off(H) + (size_t)h * GW_REC + c * 128, g, 0, 1);
    if (costs) {
        float gs[S];
        if (hh < H) E::stage_grad_state(k, cs_s, gs); else E::terminal_grad(k, cs_s, gs);
        float* gs_g = base + gw_gs_off(H);
#pragma unroll
        for (int j = 0; j < 8; ++j) gs_g[(hh * 64 + (j & 3) * 16 + c) * 2 + (j >> 2)] = j < S ? gs[j < S ? j : 0] * a.inv_Hp1 : 0.0f;
        if (hh < H) {
            float gu[C], gp[C], gu2[C], gpn[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) gpn[cc] = 0.0f;
            E::input_grad(k, cs_u, cs_up, gu, gp);
            if (hh + 1 < H) E::input_grad(k, cs_un, cs_u, gu2, gpn);
            float* gd_g = base + gw_gd_off(H);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) gd_g[(hh * C + cc) * G4_LD + c] = (gu[cc] + gpn[cc]) * a.inv_Hp1;
        }
    }
}

// A Jacobian workgroup INSIDE the phase launch (ovl_seq != 0: workgroups tiles.. of the grid, one per (step, tile), step-major).  The forward
// pass stores every step's activations and state through to memory as it goes (st4_through, in the shadow of its matrix products) and, two
// steps later, raises that step's flag = this launch's sequence number behind a counted wait (flag_through: no drain on the recurrence);
// this workgroup polls the flags, reads the step with loads that bypass its own L2, and does what ctk_g_rpgd_jac_split does: the tangents
// (split over its two waves) and the cost's state gradient.  When the forward pass ends only the last steps' records are outstanding: the
// separate Jacobian launch and its boundary (~8 us per Adam iteration) are off the iteration's path.
// Producers never wait for workers, and the launch's LDS size keeps a CU to one workgroup, so workers never sit on a producer's SIMDs.
// A poll that runs out is an ERROR, not a hang and not a silent NaN: the worker raises the error word behind {u, seq} (code 3: finish_step
// returns CTK_ERR_STATE) and leaves NaN records; the tile's next phase launch sees a non-finite gradient norm and SKIPS its update, so
// the plans and the Adam moments stay as they were (a NaN through `fminf(fmaxf(q - lr m / (sqrt(v) + eps), lo), hi)` would come out as
// `lo` with NaN moments left behind for every later step).
template <int ENV, bool K3>
CTK_DEV void rpgd_jac_worker(const RolloutArgs& a, const typename Env<ENV>::K& k, const float* __restrict__ wperm, float* __restrict__ scratch,
                             int tile, int h, uint32_t seq, uint32_t* __restrict__ err_word) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C, IO = S + C;
    const int H = a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const MlpFwdW w = mlp_load_fwd(wperm);
    float* base = scratch + (size_t)tile * gw_tile_floats(H, C);
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(base + gw_flag_off(H, C));
    // the error word's neighbours take what a poll that ran out saw (h_u words 8..11: which flag of which (step, tile), the value it last
    // held, the sequence number awaited, and the wait in microseconds) — the message of the failed step quotes them
    auto gave_up = [&](int i, uint32_t v, unsigned long long t_begin) {
        if (err_word == nullptr || (threadIdx.x & 63) != 0) return;
        const unsigned long long us = (wall_clock64() - t_begin) / 100u;                       // (the 100 MHz constant clock)
        __hip_atomic_store(err_word + 6, ((uint32_t)i << 20) | ((uint32_t)h << 10) | (uint32_t)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(err_word + 7, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(err_word + 8, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(err_word + 9, (uint32_t)(us > 0xffffffffull ? 0xffffffffull : us), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(err_word, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto await = [&](int i, int nap) {                                 // flag i; false: gave up
        uint32_t v = load_flag_wave(flags + i);                        // (wave-uniform by construction: ctk_device.h)
        if (v == seq) return true;
        const unsigned long long t_begin = wall_clock64();
        for (int spin = 0; v != seq && spin < (1 << 17); ++spin) {
            if (nap) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(2);
            v = load_flag_wave(flags + i);
        }
        if (v != seq) gave_up(i, v, t_begin);
        return v == seq;
    };
    bool ok = true;
    if (h >= CTK_HANDOFF_LAG + 2) ok &= await((h - CTK_HANDOFF_LAG - 2) * 2 + 1, 1);   // far from its step: seldom
    ok &= await(h * 2, 0);
    ok &= await(h * 2 + 1, 0);
    // acquire (agent scope) between the flags and the data they publish: the producer's data stores are ordered before its flag store
    // (flag_through: counted wait, then the store), and nothing below may be satisfied from before the flag was seen.  The cost is on this
    // worker, not on the recurrence.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const unsigned long long* act = reinterpret_cast<const unsigned long long*>(base) + ((size_t)h * (2 * 2 * 64) + lane) * 2;   // [wave m][h1 | h2][64] float4
    f32x4 d1[2], d2[2];
    {
        unsigned long long v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __hip_atomic_load(act + (size_t)(q >> 1) * 128 + (q & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 h1, h2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h1[r] = __builtin_bit_cast(float, (uint32_t)(v[m * 4 + (r >> 1)] >> ((r & 1) * 32)));
                h2[r] = __builtin_bit_cast(float, (uint32_t)(v[m * 4 + 2 + (r >> 1)] >> ((r & 1) * 32)));
            }
            d1[m] = 1.0f - h1 * h1;
            d2[m] = 1.0f - h2 * h2;
        }
    }
    if (!__all(ok)) {
        const float nan = __builtin_nanf("");
#pragma unroll
        for (int m = 0; m < 2; ++m) { d1[m] = f32x4{nan, nan, nan, nan}; d2[m] = d1[m]; }
    }
    jac_tangents<IO>(w, d1, d2, base + gw_rec_off(H) + (size_t)h * GW_REC + c * 128, g, wave, 2);
    // the cost's state gradient of this step (wave 0) and, with the last step, of the terminal state (wave 1), plan c = lanes g == 0
    const int hh = wave == 0 ? h : H;
    if (wave == 0 || h == H - 1) {
        const bool got = hh < H ? __all(ok) : await(H * 2, 0);
        if (hh == H) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (g == 0) {
            const uint32_t* xs_u = reinterpret_cast<const uint32_t*>(base + gw_xs_off(H));
            float sx[S], gs[S];
#pragma unroll
            for (int j = 0; j < S; ++j)
                sx[j] = __builtin_bit_cast(float, __hip_atomic_load(xs_u + (hh * 64 + (j & 3) * 16 + c) * 2 + (j >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (hh < H) E::stage_grad_state(k, sx, gs); else E::terminal_grad(k, sx, gs);
            float* gs_g = base + gw_gs_off(H);
#pragma unroll
            for (int j = 0; j < 8; ++j) gs_g[(hh * 64 + (j & 3) * 16 + c) * 2 + (j >> 2)] = !got ? __builtin_nanf("") : j < S ? gs[j < S ? j : 0] * a.inv_Hp1 : 0.0f;
        }
    }
}

// LDS: exchange slots | reductions | states xs[H+1][64][2] | cost-gradient terms gs[H+1][64][2] | gradients g[HC][17] | plans q[HC][17]
template <int ENV, bool K3>
__global__ __launch_bounds__(128) void ctk_g_rpgd_wide_split(RolloutArgs a, typename Env<ENV>::K k, AdamK ad, float* __restrict__ Q,
                                                            float* __restrict__ mom, float* __restrict__ var, const float* __restrict__ bc_table,
                                                            int bc_len, int ti, const float* __restrict__ wperm, float* __restrict__ scratch,
                                                            int do_update, int last, uint32_t ovl_seq, uint32_t* __restrict__ err_word,
                                                            int diag_withhold) {
    using E = Env<ENV>;
    using SP = SplitMlp<K3>;
    constexpr int S = E::S, C = E::C, IO = S + C, BLOCK = 128, NPARTS = BLOCK / G4_TRAJ;
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C;
    {
        const int tiles = (a.N + G4_TRAJ - 1) / G4_TRAJ;
        if ((int)blockIdx.x >= tiles) {                                 // (ovl_seq != 0 and not the last launch: the grid is tiles * (1 + H))
            const int idx = (int)blockIdx.x - tiles, h = idx / tiles;
            rpgd_jac_worker<ENV, K3>(a, k, wperm, scratch, idx - h * tiles, h, ovl_seq, err_word);
            return;
        }
    }
    float* ex = lds;
    float* red_s = ex + SP::EX_FWD;
    float* xs_s = red_s + G4_RED;
    float* gs_s = xs_s + (H + 1) * 128;
    float* g_s = gs_s + (H + 1) * 128;                                  // (16-byte aligned like gs: both are LDS-DMA destinations)
    float* q_s = g_s + gw_gd_floats(H, C);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * G4_TRAJ;
    const int rows = min(G4_TRAJ, a.N - row0);
    const int total = rows * HC;
    const size_t gbase = (size_t)row0 * HC;
    float* tbase = scratch + (size_t)blockIdx.x * gw_tile_floats(H, C);
    float4* tape = reinterpret_cast<float4*>(tbase) + wave * 2 * 64 + lane;          // + h * 256 + i * 64
    float* xs_g = tbase + gw_xs_off(H);
    const float* rec_g = tbase + gw_rec_off(H);
    // The adjoint chain's operands (below): lane (plan cp = t / 8, ci = t % 8) reads rows < S of column ci (and of column 8 + ci where the
    // network has more than 8 inputs) of every step's record.  The records were written by waves on every XCD, so a load is a trip to
    // memory (~2 us) against ~0.1 us of chain work per step: they travel through a register ring RING steps deep that is filled HERE, before
    // anything else, and refilled slot by slot as the chain consumes it.
    // Depth: as many steps as ~32 float4 of registers hold, trimmed to what divides the usual horizons with little left over — the walk
    // goes in whole blocks (below), so H = 50 at depth 32 walks 64 steps, at depth 25 exactly 50 (CartPole 496 -> 490 us; 16 -> 17:
    // Quad2D 548 -> 540; 8 -> 10: Hover 637 -> 627)
    constexpr int RW = S > 4 ? 2 : 1, W4 = RW * (IO > 8 ? 2 : 1), RING = W4 == 1 ? 25 : W4 == 2 ? 17 : 10;
    float4 ring[RING][W4];
    const int cp = t >> 3, ci = t & 7;
    auto rec_fetch = [&](int h, float4 (&dst)[W4]) {
        const float4* r4 = reinterpret_cast<const float4*>(rec_g + (size_t)h * GW_REC + cp * 128);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            dst[r] = r4[ci * 2 + r];
            if constexpr (IO > 8) dst[RW + r] = r4[(8 + ci) * 2 + r];
        }
    };
    auto rec_dot = [&](const float4 (&v)[W4], int o, const float (&l)[8]) {
        float d = (v[o].x * l[0] + v[o].y * l[1]) + (v[o].z * l[2] + v[o].w * l[3]);
        if constexpr (RW == 2) d += (v[o + 1].x * l[4] + v[o + 1].y * l[5]) + (v[o + 1].z * l[6] + v[o + 1].w * l[7]);
        return d;
    };
    if (do_update) {
#pragma unroll
        for (int d = 0; d < RING; ++d) rec_fetch(max(H - 1 - d, 0), ring[d]);
    }

    // Everything this launch reads was written by other launches, on every XCD: a dependent trip to memory costs ~1-2 us, so the loads go
    // out in batches of AB per thread with nothing waited for in between (a plain copy loop waits once per element): the chain's ring
    // (above), the Adam moments of this thread's first AB elements, the plans, the state tape
    constexpr int AB = 8;
    float mm0[AB], vv0[AB];
    float bc1 = 1.0f, bc2 = 1.0f;
    if (do_update) {
        if (ti <= bc_len) { bc1 = bc_table[2 * (ti - 1)]; bc2 = bc_table[2 * (ti - 1) + 1]; }
#pragma unroll
        for (int j = 0; j < AB; ++j) {
            const int i = min(t + j * BLOCK, total - 1);
            mm0[j] = 0.0f; vv0[j] = 0.0f;
            if (ad.rule != 2) { mm0[j] = mom[gbase + i]; vv0[j] = var[gbase + i]; }
        }
    }
    for (int b = t; b < G4_TRAJ * HC; b += AB * BLOCK) {
        float qv[AB];
#pragma unroll
        for (int j = 0; j < AB; ++j) qv[j] = Q[gbase + min(b + j * BLOCK, total - 1)];
#pragma unroll
        for (int j = 0; j < AB; ++j) {
            const int i = b + j * BLOCK;
            if (i < G4_TRAJ * HC) {
                const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
                q_s[hc * G4_LD + r] = i < total ? qv[j] : 0.0f;
            }
        }
    }
    if (do_update) {
        // the cost-gradient terms the Jacobian launch left (gs, then the direct input-gradient terms g: one span in memory and in LDS):
        // LDS-DMA, 1 KiB per wave-instruction, no registers; the barrier below waits for it
        const int n4 = (int)(((size_t)(H + 1) * 128 + gw_gd_floats(H, C)) / 4);
        const float4* src = reinterpret_cast<const float4*>(tbase + gw_gs_off(H));
        float4* dst = reinterpret_cast<float4*>(gs_s);
        for (int b = wave * 64; b < n4; b += BLOCK)
            if (b + lane < n4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b + lane),
                                                 (__attribute__((address_space(3))) void*)(dst + b), 16, 0, 0);
    }
    typename SP::Fwd nf;
    nf.load(wperm, wave, lane);
    nf.template fold_load<S, C>(lane);
    float up0[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up0[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    const float inv = a.inv_Hp1;
    const float s00 = g < S ? lane_state4(a, g) : 0.0f, s01 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    const int pc = t & 15, part = t >> 4;
    __syncthreads();
    auto state_of = [&](int h, int p, float (&s)[S]) {
#pragma unroll
        for (int j = 0; j < S; ++j) s[j] = xs_s[(h * 64 + (j & 3) * 16 + p) * 2 + (j >> 2)];
    };
    auto sum_parts = [&](float v) {
        v = sum_over_groups(v);
        if (g == 0) red_s[wave * 16 + c] = v;
        __syncthreads();
        const float r = red_s[c] + red_s[16 + c];
        __syncthreads();
        return r;
    };

    if (do_update) {
        // ---- the adjoint chain: plan p = t / 8 (its 8 lanes are neighbours), lane i owns column i (a state component: lambda_i; or an
        //      input: its gradient) and, where the network has more than 8 inputs, column 8 + i
        {
            const int p = cp, i = ci;
            // the lane's own LDS word of step h: a state lane's cost-gradient term gs[h][i][p]; an input lane's direct gradient term
            // g[h][i - S][p], which it overwrites with the total.  Read BEFORE the shuffles are waited for, written without a wait: the
            // step's dependent path is shuffle -> dot -> add only
            const bool is_state = i < S, is_input = i >= S && i < IO;
            const int w0 = is_state ? (int)(gs_s - lds) + ((i & 3) * 16 + p) * 2 + (i >> 2) : is_input ? (int)(g_s - lds) + (i - S) * G4_LD + p : (int)(gs_s - lds);
            const int wstep = is_state ? 128 : is_input ? C * G4_LD : 0;
            float lam = is_state ? lds[w0 + H * 128] : 0.0f;
            // the walk: step h uses ring slot (H - 1 - h) % RING and refills it with step h - RING.  Whole blocks of RING steps with nothing
            // conditional in them (the steps below 0 of the last block run on clamped addresses and write nothing; refills nobody uses re-read
            // step 0): the compiler then knows how many loads are in flight at every use and waits for that one alone — with a guard per step
            // it waits for ALL of them at the top of every block, a trip to memory each
            const int nblk = (H + RING - 1) / RING;
            float o0 = lds[w0 + (H - 1) * wstep], o1 = lds[w0 + max(H - 2, 0) * wstep];
            for (int blk = 0, htop = H - 1; blk < nblk; ++blk, htop -= RING) {
#pragma unroll
                for (int d = 0; d < RING; ++d) {
                    const int hr = htop - d, h = max(hr, 0);
                    {
                        const int w = w0 + h * wstep;
                        const float own = o0;
                        o0 = o1;
                        o1 = lds[w0 + max(hr - 2, 0) * wstep];         // two steps ahead: an LDS read takes longer than a step
                        __builtin_amdgcn_sched_barrier(0);             // (issued here, not after the dot product)
                        // lambda_r of the plan sits in lane r of its 8: a quad broadcast of lamA gives components 0..3, of lamB 4..7 — DPP
                        // operands of the multiplies themselves, no trip through the LDS crossbar
                        // (bank masks, not a select: a select becomes a branch, and DPP reads nothing from lanes the branch switched off)
                        const int lami = __builtin_bit_cast(int, lam);
                        const float lamA = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(lami, lami, 0x114, 0xF, 0xA, false));   // lanes 4..7 <- 0..3 (row_shr:4)
                        float lamB = 0.0f;
                        if constexpr (S > 4) lamB = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(lami, lami, 0x104, 0xF, 0x5, false));   // lanes 0..3 <- 4..7 (row_shl:4)
                        float l[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) l[r] = 0.0f;
                        l[0] = dpp_mov<0x00>(lamA);
                        if constexpr (S > 1) l[1] = dpp_mov<0x55>(lamA);
                        if constexpr (S > 2) l[2] = dpp_mov<0xAA>(lamA);
                        if constexpr (S > 3) l[3] = dpp_mov<0xFF>(lamA);
                        if constexpr (S > 4) l[4] = dpp_mov<0x00>(lamB);
                        if constexpr (S > 5) l[5] = dpp_mov<0x55>(lamB);
                        if constexpr (S > 6) l[6] = dpp_mov<0xAA>(lamB);
                        if constexpr (S > 7) l[7] = dpp_mov<0xFF>(lamB);
                        const float dA = rec_dot(ring[d], 0, l);
                        float dB = 0.0f;
                        if constexpr (IO > 8) dB = rec_dot(ring[d], RW, l);
                        rec_fetch(max(hr - RING, 0), ring[d]);
                        const float v = own + dA;
                        if (is_input && hr >= 0) lds[w] = v;
                        lam = is_state ? v : 0.0f;
                        if constexpr (IO > 8) {
                            if (8 + i < IO && hr >= 0) g_s[(h * C + (8 + i - S)) * G4_LD + p] += dB;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- per-plan clip_by_norm, Adam, clip
        float n2 = 0.0f;
        for (int hc = part; hc < HC; hc += NPARTS) { const float x = g_s[hc * G4_LD + pc]; n2 += x * x; }
        n2 = sum_parts(n2);
        // a plan whose gradient norm is not finite (a Jacobian record that never arrived is NaN; so is a rollout that diverged): the
        // whole tile keeps its plans and moments, and the step reports CTK_ERR_STATE
        const int tile_bad = __syncthreads_or(!(n2 <= 3.0e38f));
        if (tile_bad && t == 0 && err_word != nullptr) __hip_atomic_store(err_word + 1, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (h_u word 3)
        const float scl = ad.clip / fmaxf(sqrtf(n2), ad.clip);
        if (wave == 0 && g == 0) red_s[64 + c] = scl;
        __syncthreads();
        auto adam_element = [&](int i, float mm, float vv) {
            const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC, cc = hc % C;
            const float gg = g_s[hc * G4_LD + r] * red_s[64 + r];
            q_s[hc * G4_LD + r] = adam_update(ad, q_s[hc * G4_LD + r], gg, mm, vv, bc1, bc2, a.lo[cc], a.hi[cc]);
            if (ad.rule != 2) { mom[gbase + i] = mm; var[gbase + i] = vv; }
        };
#pragma unroll
        for (int j = 0; j < AB; ++j)
            if (!tile_bad && t + j * BLOCK < total) adam_element(t + j * BLOCK, mm0[j], vv0[j]);
        for (int b = t + AB * BLOCK; !tile_bad && b < total; b += AB * BLOCK) {
            float mm[AB], vv[AB];
#pragma unroll
            for (int j = 0; j < AB; ++j) {
                const int i = min(b + j * BLOCK, total - 1);
                mm[j] = 0.0f; vv[j] = 0.0f;
                if (ad.rule != 2) { mm[j] = mom[gbase + i]; vv[j] = var[gbase + i]; }
            }
#pragma unroll
            for (int j = 0; j < AB; ++j)
                if (b + j * BLOCK < total) adam_element(b + j * BLOCK, mm[j], vv[j]);
        }
        __syncthreads();
    }
    // ---- forward: with tape for the next launch's update, or get_action's cost pass (optimizer_rpgd.py:342)
    {
        const bool ovl = ovl_seq != 0 && !last;
        constexpr int LAG = CTK_HANDOFF_LAG;
        uint32_t* flags = reinterpret_cast<uint32_t*>(tbase + gw_flag_off(H, C));
        nf.begin(nullptr, g);
        float sv0 = s00, sv1 = s01;
        for (int h = 0; h < H; ++h) {
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = q_s[(h * C + cc) * G4_LD + c];
            if (wave == (h & 1)) {
                reinterpret_cast<float2*>(xs_s)[h * 64 + lane] = make_float2(sv0, sv1);
                if (ovl) st2_through(xs_g + (h * 64 + lane) * 2, sv0, sv1);
            }
            float x0, x1, x2;
            split_operands<S, C>(sv0, sv1, u, g, x0, x1, x2);
            nf.template fold_inputs<S, C>(u);
            const MlpPair o = last ? nf.template step<0, S, C>(x0, x1, x2, ex, wave, lane, nullptr)
                              : ovl ? nf.template step<2, S, C>(x0, x1, x2, ex, wave, lane, tape + (size_t)h * (2 * 2 * 64))
                                    : nf.template step<1, S, C>(x0, x1, x2, ex, wave, lane, tape + (size_t)h * (2 * 2 * 64));
            sv0 = o.lo; sv1 = o.hi;
            // step h - LAG is in memory once at most the 2 * LAG activation stores of the steps since are outstanding
            if (ovl && h >= LAG && h - LAG != diag_withhold) flag_through<2 * LAG>(flags + (h - LAG) * 2 + wave, ovl_seq);
        }
        if (wave == 0) {
            reinterpret_cast<float2*>(xs_s)[H * 64 + lane] = make_float2(sv0, sv1);
            if (ovl) st2_through(xs_g + (H * 64 + lane) * 2, sv0, sv1);
        }
        if (ovl) {
            for (int hq = max(H - LAG, 0); hq < H; ++hq)
                if (hq != diag_withhold) flag_through<0>(flags + hq * 2 + wave, ovl_seq);
            if (wave == 0) flag_through<0>(flags + H * 2, ovl_seq);
        }
        __syncthreads();
    }
    if (!last && ovl_seq != 0) {
        // the input-only gradient terms of the new plans, for the next launch (the Jacobian workgroups take the state terms)
        float* gd_g = tbase + gw_gd_off(H);
        for (int idx = t; idx < H * G4_TRAJ; idx += BLOCK) {
            const int h = idx >> 4, p = idx & 15;
            float u[C], upv[C], un[C], gu[C], gp[C], gu2[C], gpn[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                u[cc] = q_s[(h * C + cc) * G4_LD + p];
                upv[cc] = h > 0 ? q_s[((h - 1) * C + cc) * G4_LD + p] : up0[cc];
                un[cc] = h + 1 < H ? q_s[((h + 1) * C + cc) * G4_LD + p] : 0.0f;
                gpn[cc] = 0.0f;
            }
            E::input_grad(k, u, upv, gu, gp);
            if (h + 1 < H) E::input_grad(k, un, u, gu2, gpn);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) gd_g[(h * C + cc) * G4_LD + p] = (gu[cc] + gpn[cc]) * inv;
        }
    } else if (!last) {
        for (int i = t; i < (H + 1) * 32; i += BLOCK) reinterpret_cast<float4*>(xs_g)[i] = reinterpret_cast<const float4*>(xs_s)[i];
    } else {
        float cs = 0.0f;
        for (int h = part; h < H; h += NPARTS) {
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
    }
    if (do_update) {
        for (int i = t; i < total; i += BLOCK) {
            const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
            Q[gbase + i] = q_s[hc * G4_LD + r];
        }
    }
}

// ---- RPGD + MLP, the wide form as ONE launch per MPC step (round 4; the template counterpart of ctk_rpgd.hip: ctk_rpgd_mlp_persistent) ----------
// The first `tiles` workgroups are the producers of ctk_g_rpgd_wide_split (one 16-plan tile each, two waves; plans, Adam moments and
// weights resident for all `iters` iterations), the rest Jacobian WORKERS that stay for the whole step:
//   forward pass : every step's network inputs leave as 8-byte {value, seq} words — {state g}, {state 4 + g}, {input g} per lane (value and
//                  number in one store: who sees the number has the value; nothing waits on the recurrence, no activation tape);
//   worker       : next (iteration, step, tile) ticket, polls the step's words, recomputes the step's activations bit for bit as
//                  SplitMlp::Fwd::step forms them (the folded layer-1 bias, wave 1's k-step order), S + C tangents over its four waves into an
//                  LDS image of the record, the cost's state gradient, everything stored through to memory, each wave waits, the step's flag;
//   update       : both producer waves poll the tile's H flags, acquire, copy the cost-gradient terms in (LDS-DMA) and walk the chain on
//                  plain loads of the records — then clip_by_norm and Adam on LDS-resident moments.
// Same arithmetic as the phase launches, statement for statement: tests/test_gpu_env.py holds the two bit for bit.  Every poll is bounded
// (ctk_rpgd.hip: RP_POLL_TICKS' 200 ms); seq = seq0 + iteration.  H <= 64, iters <= 63, up to 32 tiles.
constexpr unsigned long long GP_POLL_TICKS = 20000000ull;
struct GPersistK {
    uint32_t seq0, ticket_base;
    uint32_t* err_word;
    int tiles, withhold;
};
__host__ __device__ inline size_t gp_ticket_off(int tiles, int H, int C) { return (size_t)tiles * gw_tile_floats(H, C); }

CTK_DEV void gp_gave_up(uint32_t* err_word, int what, int h, int tile, uint32_t seen, uint32_t want, unsigned long long t_begin) {
    if (err_word == nullptr) return;
    const unsigned long long us = (wall_clock64() - t_begin) / 100u;
    __hip_atomic_store(err_word + 6, ((uint32_t)what << 20) | ((uint32_t)h << 10) | (uint32_t)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word + 7, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word + 8, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word + 9, (uint32_t)(us > 0xffffffffull ? 0xffffffffull : us), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// S + C tangents through the 64-unit network (the worker of the WIDE instantiation below; ctk_mlp_wide.h's operand table)
template <int IO, bool K3>
CTK_DEV void jac_tangents64(const typename NetMlpWideT<K3>::Fwd& w, const f32x4 (&d1)[MLPW_T], const f32x4 (&d2)[MLPW_T], float* rec, int g, int j0, int jstep) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < IO; ++j) {
        if (jstep != 1 && (j % jstep) != j0) continue;                  // (wave-uniform)
        const float ind = (g == (j & 3)) ? 1.0f : 0.0f;
        f32x4 t1[MLPW_T], zz[MLPW_T];
#pragma unroll
        for (int m = 0; m < MLPW_T; ++m) { t1[m] = CTK_MFMA(w.w1[m][j >> 2], ind, z) * d1[m]; zz[m] = z; }
#pragma unroll
        for (int q = 0; q < MLPW_KH; ++q) {
            const float bq = t1[q >> 2][q & 3];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) zz[m] = CTK_MFMA(w.w2[m][q], bq, zz[m]);
        }
        f32x4 o0 = z, o1 = z;
#pragma unroll
        for (int q = 0; q < MLPW_KH; q += 2) {
            o0 = CTK_MFMA(w.w3[q], zz[q >> 2][q & 3] * d2[q >> 2][q & 3], o0);
            o1 = CTK_MFMA(w.w3[q + 1], zz[(q + 1) >> 2][(q + 1) & 3] * d2[(q + 1) >> 2][(q + 1) & 3], o1);
        }
        rec[j * 8 + g] = o0[0] + o1[0];
        rec[j * 8 + 4 + g] = o0[1] + o1[1];
    }
}

// WIDE: the 64-unit network (SplitMlp64 forward over four producer waves; no phase-launch counterpart exists for it — it is held to the
// oracle and to the one-wave kernels, tests/test_gpu_net_shapes.py)
template <int ENV, bool K3, bool WIDE>
__global__ __launch_bounds__(256) void ctk_g_rpgd_persist(RolloutArgs a, typename Env<ENV>::K k, AdamK ad, float* __restrict__ Q, float* __restrict__ mom,
                                                         float* __restrict__ var, const float* __restrict__ bc_table, int bc_len, int t0, int iters,
                                                         const float* __restrict__ wperm, float* __restrict__ scratch, GPersistK pk) {
    using E = Env<ENV>;
    using SP = std::conditional_t<WIDE, SplitMlp64<K3>, SplitMlp<K3>>;
    constexpr int S = E::S, C = E::C, IO = S + C, PW = SP::WAVES, BLOCK = 64 * PW, NPARTS = BLOCK / G4_TRAJ;
    constexpr int KS = K3 ? 3 : 2, NF = WIDE ? 0 : SplitMlp<K3>::Fwd::template fold_n<S, C>(), CC0 = 4 * (KS - 1) - S;
    constexpr bool FOLD = NF > 0;
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    uint32_t* ticket = reinterpret_cast<uint32_t*>(scratch + gp_ticket_off(pk.tiles, H, C));
    if ((int)blockIdx.x >= pk.tiles) {
        // ------------------------------------------------------------------------------------------ a Jacobian worker (four waves)
        using WT = std::conditional_t<WIDE, typename NetMlpWideT<K3>::Fwd, MlpFwdW>;
        WT w;
        if constexpr (WIDE) w.load(wperm, nullptr); else w = mlp_load_fwd(wperm);
        f32x4 w1u[2][NF > 0 ? NF : 1];                                   // the folded layer-1 columns at this lane's accumulator rows (SplitMlp::fold_load)
        if constexpr (FOLD) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w1u[m][f][r] = __shfl(w.w1[m][KS - 1], 4 * g + r + 16 * f, 64);
        }
        uint32_t* slot_s = reinterpret_cast<uint32_t*>(lds);             // [2] ticket hand-down
        float* rec_s = lds + 16;                                         // [16 plans][16 columns][8 rows]: the record's image
        const int per_it = pk.tiles * H;
        const uint32_t total = (uint32_t)per_it * (uint32_t)iters;
        uint32_t pending = 0;
        if (t == 0) pending = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int n = 0;; ++n) {
            if (t == 0) slot_s[n & 1] = pending - pk.ticket_base;
            __syncthreads();
            const uint32_t job = slot_s[n & 1];
            if (job >= total) break;
            if (t == 0) pending = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int it = (int)(job / (uint32_t)per_it), r0 = (int)job - it * per_it, h = r0 / pk.tiles, tile = r0 - h * pk.tiles;
            const uint32_t seq = pk.seq0 + (uint32_t)it;
            float* base = scratch + (size_t)tile * gw_tile_floats(H, C);
            const unsigned long long* pw = reinterpret_cast<const unsigned long long*>(base) + (size_t)h * 192 + lane;      // [3][64] words of step h
            // wave 1 of the last step's job also takes the terminal state (the words of "step" H: states only)
            const bool term = wave == 1 && h == H - 1;
            unsigned long long v0, v1, v2, z0 = 0, z1 = 0;
            bool got;
            const unsigned long long t_begin = wall_clock64();
            for (;;) {
                v0 = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v1 = __hip_atomic_load(pw + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v2 = __hip_atomic_load(pw + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = (uint32_t)(v0 >> 32) == seq && (uint32_t)(v1 >> 32) == seq && (uint32_t)(v2 >> 32) == seq;
                if (term) {
                    z0 = __hip_atomic_load(pw + 192, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    z1 = __hip_atomic_load(pw + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    got = got && (uint32_t)(z0 >> 32) == seq && (uint32_t)(z1 >> 32) == seq;
                }
                if (__all(got) || wall_clock64() - t_begin > GP_POLL_TICKS) break;
                __builtin_amdgcn_s_sleep(2);
            }
            const bool ok = __all(got);
            if (!ok && lane == 0) gp_gave_up(pk.err_word, 1, h, tile, (uint32_t)(v0 >> 32), seq, t_begin);
            const float sv0 = __builtin_bit_cast(float, (uint32_t)v0), sv1 = __builtin_bit_cast(float, (uint32_t)v1), ug = __builtin_bit_cast(float, (uint32_t)v2);
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = __shfl(ug, c + 16 * cc, 64);
            float x0, x1, x2;
            split_operands<S, C>(sv0, sv1, u, g, x0, x1, x2);
            // the step's activations, every row tile on this one wave, as the waves of the forward pass form them
            if constexpr (WIDE) {
                f32x4 h1[MLPW_T], d1[MLPW_T], d2[MLPW_T];
#pragma unroll
                for (int m = 0; m < MLPW_T; ++m) {
                    f32x4 acc = CTK_MFMA(w.w1[m][0], x0, w.b1[m]);
                    acc = CTK_MFMA(w.w1[m][1], x1, acc);
                    if constexpr (K3) acc = CTK_MFMA(w.w1[m][2], x2, acc);
                    h1[m] = ctk_tanhf4(acc);
                }
#pragma unroll
                for (int m = 0; m < MLPW_T; ++m) {
                    f32x4 cc2 = w.b2[m];
#pragma unroll
                    for (int o = 0; o < 4; ++o)                            // SplitMlp64::Fwd::step: tile m's own k-steps, then tiles m + 1, m + 2, m + 3
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) cc2 = CTK_MFMA(w.w2[m][((m + o) & 3) * 4 + jj], h1[(m + o) & 3][jj], cc2);
                    const f32x4 h2 = ctk_tanhf4(cc2);
                    d1[m] = 1.0f - h1[m] * h1[m]; d2[m] = 1.0f - h2 * h2;
                }
                if (!ok) {
                    const float nan = __builtin_nanf("");
#pragma unroll
                    for (int m = 0; m < MLPW_T; ++m) { d1[m] = f32x4{nan, nan, nan, nan}; d2[m] = d1[m]; }
                }
                jac_tangents64<IO, K3>(w, d1, d2, rec_s + c * 128, g, wave, 4);
            } else {
                f32x4 h1[2], h2[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 b = w.b1[m];
                    if constexpr (FOLD) {
#pragma unroll
                        for (int f = 0; f < NF; ++f) b += w1u[m][f] * u[CC0 + f];
                    }
                    f32x4 acc = CTK_MFMA(w.w1[m][0], x0, b);
                    if constexpr (!(FOLD && !K3)) acc = CTK_MFMA(w.w1[m][1], x1, acc);
                    if constexpr (K3 && !FOLD) acc = CTK_MFMA(w.w1[m][2], x2, acc);
                    h1[m] = ctk_tanhf4(acc);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 cc2 = w.b2[m];
#pragma unroll
                    for (int j = 0; j < 4; ++j) cc2 = CTK_MFMA(w.w2[m][m ? 4 + j : j], h1[m][j], cc2);              // own half first ...
#pragma unroll
                    for (int j = 0; j < 4; ++j) cc2 = CTK_MFMA(w.w2[m][m ? j : 4 + j], h1[m ^ 1][j], cc2);          // ... then the other wave's
                    h2[m] = ctk_tanhf4(cc2);
                }
                f32x4 d1[2], d2[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) { d1[m] = 1.0f - h1[m] * h1[m]; d2[m] = 1.0f - h2[m] * h2[m]; }
                if (!ok) {
                    const float nan = __builtin_nanf("");
#pragma unroll
                    for (int m = 0; m < 2; ++m) { d1[m] = f32x4{nan, nan, nan, nan}; d2[m] = d1[m]; }
                }
                jac_tangents<IO>(w, d1, d2, rec_s + c * 128, g, wave, 4);
            }
            // the cost's state gradient of this step (wave 0) and, with the last step, of the terminal state (wave 1): plan c = lanes g == 0
            if (wave == 0 || term) {
                const float a0 = term ? __builtin_bit_cast(float, (uint32_t)z0) : sv0, a1 = term ? __builtin_bit_cast(float, (uint32_t)z1) : sv1;
                float sx[S], gs[S];
#pragma unroll
                for (int j = 0; j < S; ++j) sx[j] = __shfl(j < 4 ? a0 : a1, c + 16 * (j & 3), 64);
                const int hh = term ? H : h;
                if (g == 0) {
                    if (hh < H) E::stage_grad_state(k, sx, gs); else E::terminal_grad(k, sx, gs);
                    uint32_t* gs_g = reinterpret_cast<uint32_t*>(base + gw_gs_off(H));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float val = !ok ? __builtin_nanf("") : j < S ? gs[j < S ? j : 0] * a.inv_Hp1 : 0.0f;
                        __hip_atomic_store(gs_g + (hh * 64 + (j & 3) * 16 + c) * 2 + (j >> 2), __builtin_bit_cast(uint32_t, val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();                                             // the record's image is complete
            {
                const float4* src = reinterpret_cast<const float4*>(rec_s);
                float4* dst = reinterpret_cast<float4*>(base + gw_rec_off(H) + (size_t)h * GW_REC);
                st4_through(dst + t, src[t]);
                st4_through(dst + 256 + t, src[256 + t]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's stores have reached memory
            __syncthreads();                                             // ... and the other waves' (and the image may be rewritten)
            if (t == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(base + gw_flag_off(H, C)) + h, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------- a producer (one tile; two waves, four for the 64-unit network)
    if (wave >= PW) return;                                              // (a finished wave no longer counts at the workgroup's barriers)
    float* ex = lds;
    float* red_s = ex + SP::EX_FWD;
    float* xs_s = red_s + G4_RED;
    float* gs_s = xs_s + (H + 1) * 128;
    float* g_s = gs_s + (H + 1) * 128;                                  // (16-byte aligned like gs: LDS-DMA destinations)
    float* q_s = g_s + gw_gd_floats(H, C);
    float* m_s = q_s + HC * G4_LD;                                      // Adam moments, resident for the step
    float* v_s = m_s + HC * G4_LD;
    const int row0 = blockIdx.x * G4_TRAJ;
    const int rows = min(G4_TRAJ, a.N - row0);
    const int total = rows * HC;
    const size_t gbase = (size_t)row0 * HC;
    float* tbase = scratch + (size_t)blockIdx.x * gw_tile_floats(H, C);
    unsigned long long* pub = reinterpret_cast<unsigned long long*>(tbase);                          // [H + 1][3][64] words
    const float* rec_g = tbase + gw_rec_off(H);
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(tbase + gw_flag_off(H, C));
    constexpr int RW = S > 4 ? 2 : 1, W4 = RW * (IO > 8 ? 2 : 1), RING = W4 == 1 ? 25 : W4 == 2 ? 17 : 10;
    const int cp = t >> 3, ci = t & 7;
    auto rec_fetch = [&](int h, float4 (&dst)[W4]) {
        const float4* r4 = reinterpret_cast<const float4*>(rec_g + (size_t)h * GW_REC + cp * 128);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            dst[r] = r4[ci * 2 + r];
            if constexpr (IO > 8) dst[RW + r] = r4[(8 + ci) * 2 + r];
        }
    };
    auto rec_dot = [&](const float4 (&v)[W4], int o, const float (&l)[8]) {
        float d = (v[o].x * l[0] + v[o].y * l[1]) + (v[o].z * l[2] + v[o].w * l[3]);
        if constexpr (RW == 2) d += (v[o + 1].x * l[4] + v[o + 1].y * l[5]) + (v[o + 1].z * l[6] + v[o + 1].w * l[7]);
        return d;
    };
    for (int i = t; i < G4_TRAJ * HC; i += BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        const bool in = i < total;
        q_s[hc * G4_LD + r] = in ? Q[gbase + i] : 0.0f;
        m_s[hc * G4_LD + r] = in && ad.rule != 2 ? mom[gbase + i] : 0.0f;
        v_s[hc * G4_LD + r] = in && ad.rule != 2 ? var[gbase + i] : 0.0f;
    }
    typename SP::Fwd nf;
    nf.load(wperm, wave, lane);
    nf.template fold_load<S, C>(lane);
    float up0[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up0[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    const float inv = a.inv_Hp1;
    const float s00 = g < S ? lane_state4(a, g) : 0.0f, s01 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    const int pc = t & 15, part = t >> 4;
    __syncthreads();
    auto state_of = [&](int h, int p, float (&s)[S]) {
#pragma unroll
        for (int j = 0; j < S; ++j) s[j] = xs_s[(h * 64 + (j & 3) * 16 + p) * 2 + (j >> 2)];
    };
    auto sum_parts = [&](float v) {
        v = sum_over_groups(v);
        if (g == 0) red_s[wave * 16 + c] = v;
        __syncthreads();
        float r = red_s[c] + red_s[16 + c];
        if constexpr (PW == 4) r += red_s[32 + c] + red_s[48 + c];
        __syncthreads();
        return r;
    };
    // the forward pass; PUBLISH: the step's network inputs leave as {value, seq} words (pass `it`); otherwise the states stay in LDS for the cost pass
    auto forward = [&](bool publish, uint32_t seq, int withhold) {
        const unsigned long long hi = (unsigned long long)seq << 32;
        uint32_t word = (uint32_t)lane;
        asm volatile("" : "+v"(word));                                   // (pinned: ctk_rpgd.hip, rpgd_forward_mlp_publish_pair)
        const int wv = __builtin_amdgcn_readfirstlane(wave);             // (a scalar: the per-step choice below is a branch, not an exec mask)
        nf.begin(nullptr, g);
        float sv0 = s00, sv1 = s01;
        for (int h = 0; h < H; ++h) {
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = q_s[(h * C + cc) * G4_LD + c];
            if (publish) {
                if (h != withhold) {
                    unsigned long long* p = pub + (uint32_t)(h * 192) + word;
                    if (wv >= 2) {
                    } else if (wv == 0) {
                        float ug = 0.0f;
#pragma unroll
                        for (int cc = 0; cc < C; ++cc) ug = g == cc ? u[cc] : ug;
                        __hip_atomic_store(p, hi | (unsigned long long)__builtin_bit_cast(uint32_t, sv0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p + 128, hi | (unsigned long long)__builtin_bit_cast(uint32_t, ug), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        __hip_atomic_store(p + 64, hi | (unsigned long long)__builtin_bit_cast(uint32_t, sv1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else if (wave == (h & (PW - 1))) {
                reinterpret_cast<float2*>(xs_s)[h * 64 + lane] = make_float2(sv0, sv1);
            }
            float x0, x1, x2;
            split_operands<S, C>(sv0, sv1, u, g, x0, x1, x2);
            nf.template fold_inputs<S, C>(u);
            const MlpPair o = nf.template step<0, S, C>(x0, x1, x2, ex, wave, lane, nullptr);
            sv0 = o.lo; sv1 = o.hi;
        }
        if (publish) {                                                   // the terminal state: words of "step" H
            unsigned long long* p = pub + (uint32_t)(H * 192) + word;
            if (wv == 0) __hip_atomic_store(p, hi | (unsigned long long)__builtin_bit_cast(uint32_t, sv0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (wv == 1) __hip_atomic_store(p + 64, hi | (unsigned long long)__builtin_bit_cast(uint32_t, sv1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (wave == 0) {
            reinterpret_cast<float2*>(xs_s)[H * 64 + lane] = make_float2(sv0, sv1);
        }
        __syncthreads();
    };
    for (int it = 0; it < iters; ++it) {
        const uint32_t seq = pk.seq0 + (uint32_t)it;
        const int ti = t0 + it + 1;
        float bc1 = 1.0f, bc2 = 1.0f;
        if (ti <= bc_len) { bc1 = bc_table[2 * (ti - 1)]; bc2 = bc_table[2 * (ti - 1) + 1]; }
        forward(true, seq, it == 0 ? pk.withhold : -1);
        // the input-only gradient terms of these plans, straight into the chain's LDS words (the phase launches pass them through memory)
        for (int idx = t; idx < H * G4_TRAJ; idx += BLOCK) {
            const int h = idx >> 4, p = idx & 15;
            float u[C], upv[C], un[C], gu[C], gp[C], gu2[C], gpn[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                u[cc] = q_s[(h * C + cc) * G4_LD + p];
                upv[cc] = h > 0 ? q_s[((h - 1) * C + cc) * G4_LD + p] : up0[cc];
                un[cc] = h + 1 < H ? q_s[((h + 1) * C + cc) * G4_LD + p] : 0.0f;
                gpn[cc] = 0.0f;
            }
            E::input_grad(k, u, upv, gu, gp);
            if (h + 1 < H) E::input_grad(k, un, u, gu2, gpn);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) g_s[(h * C + cc) * G4_LD + p] = (gu[cc] + gpn[cc]) * inv;
        }
        // ---- the tile's H records: both waves poll the flags (lane h <- flag h) and acquire — each then reads with plain loads of its own
        {
            uint32_t f = lane < H ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
            if (!__all(f == seq)) {
                const unsigned long long t_begin = wall_clock64();
                while (!__all(f == seq)) {
                    if (wall_clock64() - t_begin > 2 * GP_POLL_TICKS) break;
                    __builtin_amdgcn_s_sleep(1);
                    f = lane < H ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
                }
                if (!__all(f == seq) && wave == 0) {
                    const unsigned long long missing = __builtin_amdgcn_ballot_w64(f != seq);
                    if (lane == (int)__builtin_ctzll(missing)) gp_gave_up(pk.err_word, 2, lane, (int)blockIdx.x, f, seq, t_begin);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        float4 ring[RING][W4];
        if (t < 128) {
#pragma unroll
            for (int d = 0; d < RING; ++d) rec_fetch(max(H - 1 - d, 0), ring[d]);
        }
        {   // the cost-gradient terms the workers left: LDS-DMA, 1 KiB per wave-instruction; the barrier below waits for it
            const int n4 = (H + 1) * 32;
            const float4* src = reinterpret_cast<const float4*>(tbase + gw_gs_off(H));
            float4* dst = reinterpret_cast<float4*>(gs_s);
            for (int b = wave * 64; b < n4; b += BLOCK)
                if (b + lane < n4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b + lane),
                                                     (__attribute__((address_space(3))) void*)(dst + b), 16, 0, 0);
        }
        __syncthreads();
        // ---- the adjoint chain (ctk_g_rpgd_wide_split: plan p = t / 8, lane i owns column i and, beyond 8 inputs, column 8 + i)
        if (t < 128) {
            const int p = cp, i = ci;
            const bool is_state = i < S, is_input = i >= S && i < IO;
            const int w0 = is_state ? (int)(gs_s - lds) + ((i & 3) * 16 + p) * 2 + (i >> 2) : is_input ? (int)(g_s - lds) + (i - S) * G4_LD + p : (int)(gs_s - lds);
            const int wstep = is_state ? 128 : is_input ? C * G4_LD : 0;
            float lam = is_state ? lds[w0 + H * 128] : 0.0f;
            const int nblk = (H + RING - 1) / RING;
            float o0 = lds[w0 + (H - 1) * wstep], o1 = lds[w0 + max(H - 2, 0) * wstep];
            for (int blk = 0, htop = H - 1; blk < nblk; ++blk, htop -= RING) {
#pragma unroll
                for (int d = 0; d < RING; ++d) {
                    const int hr = htop - d, h = max(hr, 0);
                    const int wd = w0 + h * wstep;
                    const float own = o0;
                    o0 = o1;
                    o1 = lds[w0 + max(hr - 2, 0) * wstep];
                    __builtin_amdgcn_sched_barrier(0);
                    const int lami = __builtin_bit_cast(int, lam);
                    const float lamA = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(lami, lami, 0x114, 0xF, 0xA, false));
                    float lamB = 0.0f;
                    if constexpr (S > 4) lamB = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(lami, lami, 0x104, 0xF, 0x5, false));
                    float l[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) l[r] = 0.0f;
                    l[0] = dpp_mov<0x00>(lamA);
                    if constexpr (S > 1) l[1] = dpp_mov<0x55>(lamA);
                    if constexpr (S > 2) l[2] = dpp_mov<0xAA>(lamA);
                    if constexpr (S > 3) l[3] = dpp_mov<0xFF>(lamA);
                    if constexpr (S > 4) l[4] = dpp_mov<0x00>(lamB);
                    if constexpr (S > 5) l[5] = dpp_mov<0x55>(lamB);
                    if constexpr (S > 6) l[6] = dpp_mov<0xAA>(lamB);
                    if constexpr (S > 7) l[7] = dpp_mov<0xFF>(lamB);
                    const float dA = rec_dot(ring[d], 0, l);
                    float dB = 0.0f;
                    if constexpr (IO > 8) dB = rec_dot(ring[d], RW, l);
                    rec_fetch(max(hr - RING, 0), ring[d]);
                    const float v = own + dA;
                    if (is_input && hr >= 0) lds[wd] = v;
                    lam = is_state ? v : 0.0f;
                    if constexpr (IO > 8) {
                        if (8 + i < IO && hr >= 0) g_s[(h * C + (8 + i - S)) * G4_LD + p] += dB;
                    }
                }
            }
        }
        __syncthreads();
        // ---- per-plan clip_by_norm, Adam, clip
        float n2 = 0.0f;
        for (int hc = part; hc < HC; hc += NPARTS) { const float x = g_s[hc * G4_LD + pc]; n2 += x * x; }
        n2 = sum_parts(n2);
        const int tile_bad = __syncthreads_or(!(n2 <= 3.0e38f));
        if (tile_bad && t == 0 && pk.err_word != nullptr) __hip_atomic_store(pk.err_word + 1, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const float scl = ad.clip / fmaxf(sqrtf(n2), ad.clip);
        if (wave == 0 && g == 0) red_s[64 + c] = scl;                    // (G4_RED = 80 floats: [4 waves][16] + [16])
        __syncthreads();
        for (int i = t; !tile_bad && i < total; i += BLOCK) {
            const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC, cc = hc % C;
            const float gg = g_s[hc * G4_LD + r] * red_s[64 + r];
            float mm = m_s[hc * G4_LD + r], vv = v_s[hc * G4_LD + r];
            q_s[hc * G4_LD + r] = adam_update(ad, q_s[hc * G4_LD + r], gg, mm, vv, bc1, bc2, a.lo[cc], a.hi[cc]);
            m_s[hc * G4_LD + r] = mm; v_s[hc * G4_LD + r] = vv;
        }
        __syncthreads();
    }
    // ---- get_action's cost pass (optimizer_rpgd.py:342), then the plans and moments back to memory
    forward(false, 0u, -1);
    {
        float cs = 0.0f;
        for (int h = part; h < H; h += NPARTS) {
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
    }
    for (int i = t; i < total; i += BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        Q[gbase + i] = q_s[hc * G4_LD + r];
        if (ad.rule != 2) { mom[gbase + i] = m_s[hc * G4_LD + r]; var[gbase + i] = v_s[hc * G4_LD + r]; }
    }
}

// floats of one tile's LDS carve in ctk_g_rollout_split (a multiple of 4: the second tile of a workgroup starts 16-byte aligned)
__host__ __device__ inline size_t split_tile_floats(int ex_fwd, int cols, int H, int C) {
    const size_t n = (size_t)ex_fwd + G4_RED + (size_t)(H + 1) * 128 + (size_t)H * C * G4_LD + (size_t)G4_TRAJ * tile_stride(cols) + G4_TRAJ + 2 * (size_t)H * C + 3 * (size_t)H;
    return (n + 3) & ~(size_t)3;
}

// ---- rollout + cost (MPPI / affine modes of ctk_generic_net.hip: ctk_g_rollout_net) ---------------------------------------------------
// 16 trajectories per workgroup.  Inputs (interpolation, shifted nominal, clip, MPPI correction) are formed for all (step, trajectory)
// pairs before the recurrence, the costs from the states it leaves in LDS after it; the recurrence itself is gru4_step only.
// LDS: exchange slots | reductions | states xs[H+1][64][2] | inputs u[HC][17] | sample tile [16][ts] | e[16] | base, scale [HC] | interp tables
// (the Philox path of load_tile_early spreads a row's column blocks over BLOCK / 16 threads)
template <int ENV, class SP, int MODE, bool LOG, int TILES>
__global__ __launch_bounds__(64 * SP::WAVES * TILES) void ctk_g_rollout_split(RolloutArgs a, typename Env<ENV>::K k, MppiK mk, const float* __restrict__ samples,
                                                              const float* __restrict__ base, const float* __restrict__ scale, int rng_kind,
                                                              const float* __restrict__ wperm, const float* __restrict__ hidden,
                                                              float* __restrict__ parts, NetFuse gz) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C, WAVES = SP::WAVES, BLOCK = 64 * WAVES, NPARTS = BLOCK / G4_TRAJ;
    static_assert(TILES == 1 || WAVES == 2, "two tiles per workgroup: the two-wave policies");
    extern __shared__ float lds_all[];
    const int H = a.H, HC = H * C, cols = a.P, ts = tile_stride(cols);
    // TILES == 2: the workgroup is two independent halves (sub = 0, 1) of BLOCK threads, each with its own tile, LDS carve and record;
    // they only share the barriers (same control flow: every tile is full, the host sees to it).  Why: a CU gives the four waves of ONE
    // workgroup four different SIMDs, but puts the waves of two 2-wave workgroups where it likes — measured at N = 8 192 (512 tiles,
    // two per CU): one SIMD holds a wave of each and one idles, and the later workgroup runs its 100 steps in 78 us against 60
    const int sub = TILES > 1 ? (int)threadIdx.x / BLOCK : 0;
    const int vb = (int)blockIdx.x * TILES + sub;                       // the tile = the "block" of the one-tile form
    float* lds = lds_all + (size_t)sub * split_tile_floats(SP::EX_FWD, cols, H, C);
    float* ex = lds;
    float* red_s = ex + SP::EX_FWD;
    float* xs_s = red_s + G4_RED;
    float* u_s = xs_s + (H + 1) * 128;
    float* tile = u_s + HC * G4_LD;
    float* e_s = tile + G4_TRAJ * ts;
    float* base_s = e_s + G4_TRAJ;
    float* scale_s = base_s + HC;
    float* w0_s = scale_s + HC;
    float* w1_s = w0_s + H;
    int* i0_s = reinterpret_cast<int*>(w1_s + H);
    const int t = (int)threadIdx.x - sub * BLOCK, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    const int row0 = vb * G4_TRAJ;
    const int pc = t & 15, part = t >> 4;
    const int n = row0 + pc;
    const bool valid = n < a.N;

    typename SP::Fwd nf;
    nf.load(wperm, wave, lane);
    nf.template fold_load<S, C>(lane);
    load_tile_early<G4_TRAJ, BLOCK>(tile, samples, a, row0, MODE == CTK_G_MODE_MPPI ? mk.stdev : 1.0f, rng_kind, [&] {
        if constexpr (MODE == CTK_G_MODE_MPPI) {
            for (int h = t; h < H; h += BLOCK) {
                const InterpEntry e = a.interp[h];
                i0_s[h] = e.i0; w0_s[h] = e.w0; w1_s[h] = e.w1;
            }
            for (int hc = t; hc < HC; hc += BLOCK) {
                const int h = hc / C, cc = hc - h * C;
                base_s[hc] = base[min(h + 1, H - 1) * C + cc];
            }
        } else {
            for (int hc = t; hc < HC; hc += BLOCK) { base_s[hc] = base[hc]; scale_s[hc] = scale[hc]; }
        }
    }, t);
    float up0[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up0[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    const float s00 = g < S ? lane_state4(a, g) : 0.0f, s01 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    __syncthreads();

    auto sum_parts = [&](float v) {           // over the 16 parts of a trajectory (part = 4 * wave + g), fixed order; valid in every thread
        v = sum_over_groups(v);
        if (g == 0) red_s[wave * 16 + c] = v;
        __syncthreads();
        float r = red_s[c] + red_s[16 + c];
        if constexpr (WAVES == 4) r += red_s[32 + c] + red_s[48 + c];
        __syncthreads();
        return r;
    };

    // ---- inputs of all steps, and what of the cost depends on them only
    float corr = 0.0f;
    {
        const float* my = tile + pc * ts;
        const int Pm1 = cols / C - 1;
        for (int h = part; h < H; h += NPARTS) {
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
        nf.begin(hidden, g);
        float sv0 = s00, sv1 = s01;
        for (int h = 0; h < H; ++h) {                      // (no input fold here: measured no gain in this kernel, CartPole 45.6 / 73.5 us either way)
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = u_s[(h * C + cc) * G4_LD + c];
            if (wave == (h & (WAVES - 1))) reinterpret_cast<float2*>(xs_s)[h * 64 + lane] = make_float2(sv0, sv1);
            float x0, x1, x2;
            split_operands<S, C>(sv0, sv1, u, g, x0, x1, x2);
            if constexpr (SP::NET == NET_GRU) nf.template fold_inputs<S, C>(u);     // (the GRU's inputs beyond 8: it has no third k-step to take them)
            const MlpPair o = nf.template step<false>(x0, x1, x2, ex, wave, lane, nullptr);
            sv0 = o.lo; sv1 = o.hi;
        }
        if (wave == 0) reinterpret_cast<float2*>(xs_s)[H * 64 + lane] = make_float2(sv0, sv1);
        __syncthreads();
    }
    // ---- costs from the states
    float cs = 0.0f;
    for (int h = part; h <= H; h += NPARTS) {
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
        float* rec = parts + (size_t)vb * (2 + cols);
        const bool use_ll = gz.mode != 0;            // kernel-argument uniform: the records are handed over inside this launch
        unsigned long long* llr = gz.ll + (size_t)vb * (2 + cols);
        if (t == 0) {
            if (use_ll) { ll_store(llr, rho, gz.up.seq); ll_store(llr + 1, aw, gz.up.seq); }
            else { rec[0] = rho; rec[1] = aw; }
        }
        for (int p = t; p < cols; p += BLOCK) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < G4_TRAJ; ++r) acc += e_s[r] * tile[r * ts + p];
            if (use_ll) ll_store(llr + 2 + p, acc, gz.up.seq);
            else rec[2 + p] = acc;
        }
        if constexpr (TILES == 1) {
            if (use_ll && blockIdx.x == 0) {         // block 0 gathers every block's words, merges, updates / emits the shard record
                __syncthreads();
                mppi_ll_tail<C>(lds, gz.ll, (int)gridDim.x, cols, mk.neg_inv_lbd, gz.mode, gz.out_rec, gz.up);
            }
        }
    }
}

// ---- predictor.update(s, Q0) (optimizer_mppi.py:195-197): the carried hidden state advanced by the measured state and the applied
// input — one workgroup, the same four-wave step (all 16 MFMA columns carry the same values; column 0 writes back), operands straight
// from the tables into registers (the one-wave form stages 59 KiB in LDS with 64 threads: 27 us behind every MPPI step)
template <int ENV>
__global__ __launch_bounds__(256) void ctk_g_gru_advance4(RolloutArgs a, const float* __restrict__ u_dev, const float* __restrict__ wperm,
                                                              float* __restrict__ hidden) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    __shared__ float ex[G4_EX_FWD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    SplitGru::Fwd nf;
    nf.load(wperm, wave, lane);
    nf.template fold_load<S, C>(lane);
    nf.begin(hidden, g);
    float u[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) u[cc] = u_dev ? u_dev[cc] : a.u_prev[cc];
    const float sv0 = g < S ? lane_state4(a, g) : 0.0f, sv1 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    float x0, x1, x2;
    split_operands<S, C>(sv0, sv1, u, g, x0, x1, x2);
    nf.template fold_inputs<S, C>(u);
    (void)nf.template step<false>(x0, x1, x2, ex, wave, lane, nullptr);     // its barriers order every lane's read of `hidden` before the write below
    if (wave == 0 && c == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { hidden[16 * m + 4 * g + r] = nf.st.h1[m][r]; hidden[32 + 16 * m + 4 * g + r] = nf.st.h2[m][r]; }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static uint32_t g4_magic_of(int d) { return d >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; }
static int split_waves(int net) { return net == NET_GRU ? SplitGru::WAVES : net == NET_MLP64 ? SplitMlp64<false>::WAVES : SplitMlp<false>::WAVES; }
static int split_ex_fwd(int net) { return net == NET_GRU ? SplitGru::EX_FWD : net == NET_MLP64 ? SplitMlp64<false>::EX_FWD : SplitMlp<false>::EX_FWD; }
static int split_ex_bwd(int net) { return net == NET_GRU ? SplitGru::EX_BWD : SplitMlp<false>::EX_BWD; }
static int split_tape_f4(int net) { return net == NET_GRU ? SplitGru::TAPE_F4 : SplitMlp<false>::TAPE_F4; }
static void env_dims(int env, int* S, int* C) { CTK_FOR_ENV(env, EV, { *S = Env<EV>::S; *C = Env<EV>::C; }); }
static const char* split_policy_name(int env, int net) {
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    if (net == NET_MLP64) return S + C > 8 ? "SplitMlp64<true>" : "SplitMlp64<false>";
    return net == NET_GRU ? "SplitGru" : (S + C > 8 ? "SplitMlp<true>" : "SplitMlp<false>");
}
// the split forms serve the populations that leave SIMDs idle with one wave per tile (N <= 8 192: 512 tiles on 1 024 SIMDs); larger
// ones keep ctk_generic_net.hip's kernels.  CTK_NET_ONE_WAVE / CTK_RPGD_NET_ONE_WAVE: diagnostic switches (A/B measurements, tests)
static bool split_env_ok(int env, int net) {
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    return net == NET_MLP || (net == NET_GRU && S + C <= 12);      // (beyond 8 inputs the GRU's extra ones are folded into its layer-1 biases: SplitGru::Fwd)
}

size_t ctk_g_rpgd_descent_split_lds(int net, int H, int C) {
    return (size_t)(split_ex_fwd(net) + split_ex_bwd(net) + G4_RED + 2 * (H + 1) * 128 + 2 * H * C * G4_LD) * sizeof(float);
}
bool ctk_g_rpgd_split_ok(int env, int net, int N, int H) {
    static const bool off = getenv("CTK_RPGD_NET_ONE_WAVE") != nullptr;
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    return !off && split_env_ok(env, net) && N <= 8192 && ctk_g_rpgd_descent_split_lds(net, H, C) <= 160 * 1024;
}
size_t ctk_g_rpgd_scratch_floats_split(int net, int N, int H) {
    return (size_t)((N + G4_TRAJ - 1) / G4_TRAJ) * H * split_waves(net) * split_tape_f4(net) * 64 * 4;
}
const char* ctk_g_rpgd_descent_split_name(int env, int net) { return ctk_kernel_name("ctk_g_rpgd_descent_split<%d, %4$s>", env, 0, 0, split_policy_name(env, net)); }

template <int EV, class SP>
static void launch_descent_split(hipStream_t st, const RolloutArgs& a_in, const float* params, float dt, int isteps, const AdamK& ad, float* Q, float* m,
                                 float* v, const float* bc_table, int bc_len, int t0, int iters, const float* wperm, const float* wperm_bwd,
                                 const float* hidden, float* scratch, hipEvent_t e0, hipEvent_t e1) {
    using E = Env<EV>;
    RolloutArgs a = a_in;
    a.C = E::C; a.p_magic = g4_magic_of(a.H * E::C);
    const typename E::K k = E::derive(params, dt, isteps);
    const dim3 grid((a.N + G4_TRAJ - 1) / G4_TRAJ), block(64 * SP::WAVES);
    const size_t lds = ctk_g_rpgd_descent_split_lds(SP::NET, a.H, E::C);
    CTK_LAUNCH((ctk_g_rpgd_descent_split<EV, SP>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wperm_bwd, hidden, scratch);
}

hipError_t ctk_launch_g_rpgd_descent_split(hipStream_t st, int env, int net, const RolloutArgs& a, const float* params, float dt, int isteps,
                                           const AdamK& ad, float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters,
                                           const float* wperm, const float* wperm_bwd, const float* hidden, float* scratch,
                                           hipEvent_t e0, hipEvent_t e1) {
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        if (net == NET_GRU) {
            launch_descent_split<EV, SplitGru>(st, a, params, dt, isteps, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wperm_bwd, hidden, scratch, e0, e1);
        } else {
            launch_descent_split<EV, SplitMlp<(E::S + E::C > 8)>>(st, a, params, dt, isteps, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wperm_bwd, hidden, scratch, e0, e1);
        }
    });
    return hipGetLastError();
}

// the wide form of the MLP descent: populations that leave the chip idle under the chain (as CartPole's own, ctk_rpgd.hip: N <= 4 096)
bool ctk_g_rpgd_wide_ok(int env, int net, int N, int H) {
    static const bool narrow = getenv("CTK_RPGD_NARROW") != nullptr;      // diagnostic switch: A/B the two forms (shared with ctk_rpgd.hip)
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    // (more than 8 network inputs — S + C tangent passes, two record columns per lane — was slower than the split chain before the register
    // ring, the DPP chain and the in-launch Jacobian workgroups: 760 vs 742 us; now 637)
    return !narrow && net == NET_MLP && S + C <= 12 && N <= 4096 && ctk_g_rpgd_split_ok(env, net, N, H);
}
size_t ctk_g_rpgd_scratch_floats_wide(int N, int H) { return (size_t)((N + G4_TRAJ - 1) / G4_TRAJ) * gw_tile_floats(H, 4) + 16; }   // (C <= 4; + the one-launch form's ticket counter)
// up to 32 tiles and H <= 64 the whole descent is ONE launch (ctk_g_rpgd_persist: producers + resident Jacobian workers)
bool ctk_g_rpgd_persist_ok(int env, int net, int N, int H) {
    static const bool off = getenv("CTK_RPGD_NO_PERSISTENT") != nullptr;   // diagnostic switch: A/B the two forms (shared with ctk_rpgd.hip)
    return !off && ctk_g_rpgd_wide_ok(env, net, N, H) && N <= 32 * G4_TRAJ && H <= 64;
}
const char* ctk_g_rpgd_wide_name(int env, int N, int H) {
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    if (H > 0 && ctk_g_rpgd_persist_ok(env, NET_MLP, N, H)) return ctk_kernel_name("ctk_g_rpgd_persist<%d, %4$s>", env, 0, 0, S + C > 8 ? "true" : "false");
    return ctk_kernel_name("ctk_g_rpgd_wide_split<%d, %4$s> + ctk_g_rpgd_jac_split", env, 0, 0, S + C > 8 ? "true" : "false");
}

template <int EV>
static void launch_wide_split(hipStream_t st, const RolloutArgs& a_in, const float* params, float dt, int isteps, const AdamK& ad, float* Q, float* m,
                              float* v, const float* bc_table, int bc_len, int t0, int iters, const float* wperm, float* scratch, hipEvent_t e0,
                              hipEvent_t e1, uint32_t* err_word, RpgdPersist* pers) {
    using E = Env<EV>;
    constexpr bool K3 = E::S + E::C > 8;
    // diagnostic switch, read once per process (tests/test_gpu_rpgd.py: the time-out path): CTK_DIAG_RPGD_WITHHOLD_FLAG=<step> makes the
    // forward passes of the FIRST overlapped phase launch of the process never raise that step's flags
    static const int diag_step = getenv("CTK_DIAG_RPGD_WITHHOLD_FLAG") ? atoi(getenv("CTK_DIAG_RPGD_WITHHOLD_FLAG")) : -1;
    static std::atomic<int> diag_armed{diag_step >= 0 ? 1 : 0};
    RolloutArgs a = a_in;
    a.C = E::C; a.p_magic = g4_magic_of(a.H * E::C);
    const typename E::K k = E::derive(params, dt, isteps);
    const int tiles = (a.N + G4_TRAJ - 1) / G4_TRAJ;
    size_t lds = (size_t)(SplitMlp<K3>::EX_FWD + G4_RED + 2 * (a.H + 1) * 128 + gw_gd_floats(a.H, E::C) + a.H * E::C * G4_LD) * sizeof(float);
    if (pers != nullptr && iters >= 1 && iters <= 63 && ctk_g_rpgd_persist_ok(EV, NET_MLP, a.N, a.H)) {
        // ONE launch: producers (a tile each) + Jacobian workers that stay for all iterations; more than half a CU's LDS per workgroup, so
        // that no worker sits on a producer's SIMDs
        const int per_it = tiles * a.H, W = std::min(240, per_it);
        if (pers->seq0 < 64u || pers->seq0 > 0xffffff00u) pers->seq0 = 64u;
        GPersistK pk{pers->seq0, pers->ticket_base, err_word, tiles, diag_armed.exchange(0) ? diag_step : -1};
        pers->seq0 += 64u;
        pers->ticket_base += (uint32_t)per_it * (uint32_t)iters + (uint32_t)W;      // every job + one ticket past the end per worker workgroup
        const size_t plds = std::max(lds + (size_t)2 * a.H * E::C * G4_LD * sizeof(float), (size_t)84 * 1024);
        if (e0 || e1)
            hipExtLaunchKernelGGL((ctk_g_rpgd_persist<EV, K3, false>), dim3(tiles + W), dim3(256), plds, st, e0, e1, 0, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm,
                                  scratch, pk);
        else
            hipLaunchKernelGGL((ctk_g_rpgd_persist<EV, K3, false>), dim3(tiles + W), dim3(256), plds, st, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, scratch, pk);
        return;
    }
    // the Jacobian work rides inside the phase launch (rpgd_jac_worker); CTK_RPGD_NO_OVERLAP: its own launch after each phase launch
    static const bool no_overlap = getenv("CTK_RPGD_NO_OVERLAP") != nullptr;
    static std::atomic<uint32_t> launch_seq{0};
    const bool ovl = !no_overlap && tiles <= CTK_HANDOFF_MAX_TILES;   // (the Jacobian workgroups need CUs of their own while the forward passes run)
    if (ovl) lds = std::max(lds, (size_t)84 * 1024);      // more than half a CU's LDS: one workgroup per CU, no worker beside a forward pass
    for (int it = 0; it <= iters; ++it) {
        const bool last = it == iters;
        hipEvent_t s0 = it == 0 ? e0 : nullptr, s1 = last ? e1 : nullptr;
        uint32_t seq = 0;
        if (ovl && !last) { seq = ++launch_seq; if (seq == 0) seq = ++launch_seq; }
        const dim3 grid(ovl && !last ? tiles * (1 + a.H) : tiles);
        const int withhold = (ovl && !last && diag_armed.exchange(0)) ? diag_step : -1;
        if (s0 || s1)
            hipExtLaunchKernelGGL((ctk_g_rpgd_wide_split<EV, K3>), grid, dim3(128), lds, st, s0, s1, 0, a, k, ad, Q, m, v, bc_table, bc_len, t0 + it, wperm,
                                  scratch, it > 0 ? 1 : 0, last ? 1 : 0, seq, err_word, withhold);
        else
            hipLaunchKernelGGL((ctk_g_rpgd_wide_split<EV, K3>), grid, dim3(128), lds, st, a, k, ad, Q, m, v, bc_table, bc_len, t0 + it, wperm, scratch,
                               it > 0 ? 1 : 0, last ? 1 : 0, seq, err_word, withhold);
        if (!last && !ovl) hipLaunchKernelGGL((ctk_g_rpgd_jac_split<EV, K3>), dim3(tiles * a.H), dim3(64), 0, st, a, k, Q, wperm, scratch);
    }
}

// the 64-unit network's descent as one launch (ctk_g_rpgd_persist<., ., true>): up to 32 tiles, H <= 64, 1..63 iterations
bool ctk_g_rpgd_persist64_ok(int env, int N, int H) {
    static const bool off = getenv("CTK_RPGD_NO_PERSISTENT") != nullptr || getenv("CTK_RPGD_NET_ONE_WAVE") != nullptr;
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    const size_t lds = (size_t)(M4_EX_FWD + G4_RED + 2 * (H + 1) * 128 + gw_gd_floats(H, C) + 3 * H * C * G4_LD) * sizeof(float);
    return !off && S + C <= 12 && N <= 32 * G4_TRAJ && H <= 64 && lds <= 160 * 1024;
}
const char* ctk_g_rpgd_persist64_name(int env) {
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    return ctk_kernel_name("ctk_g_rpgd_persist<%d, %4$s, true>", env, 0, 0, S + C > 8 ? "true" : "false");
}
hipError_t ctk_launch_g_rpgd_persist64(hipStream_t st, int env, const RolloutArgs& a_in, const float* params, float dt, int isteps, const AdamK& ad,
                                       float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters, const float* wperm,
                                       float* scratch, hipEvent_t e0, hipEvent_t e1, uint32_t* err_word, RpgdPersist* pers) {
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        constexpr bool K3 = E::S + E::C > 8;
        RolloutArgs a = a_in;
        a.C = E::C; a.p_magic = g4_magic_of(a.H * E::C);
        const typename E::K k = E::derive(params, dt, isteps);
        const int tiles = (a.N + G4_TRAJ - 1) / G4_TRAJ, per_it = tiles * a.H, W = std::min(240, per_it);
        if (pers->seq0 < 64u || pers->seq0 > 0xffffff00u) pers->seq0 = 64u;
        GPersistK pk{pers->seq0, pers->ticket_base, err_word, tiles, -1};
        pers->seq0 += 64u;
        pers->ticket_base += (uint32_t)per_it * (uint32_t)iters + (uint32_t)W;
        const size_t lds = std::max((size_t)(M4_EX_FWD + G4_RED + 2 * (a.H + 1) * 128 + gw_gd_floats(a.H, E::C) + 3 * a.H * E::C * G4_LD) * sizeof(float), (size_t)84 * 1024);
        if (e0 || e1)
            hipExtLaunchKernelGGL((ctk_g_rpgd_persist<EV, K3, true>), dim3(tiles + W), dim3(256), lds, st, e0, e1, 0, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm,
                                  scratch, pk);
        else
            hipLaunchKernelGGL((ctk_g_rpgd_persist<EV, K3, true>), dim3(tiles + W), dim3(256), lds, st, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, scratch, pk);
    });
    return hipGetLastError();
}

hipError_t ctk_launch_g_rpgd_wide_split(hipStream_t st, int env, const RolloutArgs& a, const float* params, float dt, int isteps, const AdamK& ad,
                                        float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters, const float* wperm,
                                        float* scratch, hipEvent_t e0, hipEvent_t e1, uint32_t* err_word, RpgdPersist* pers) {
    CTK_FOR_ENV(env, EV, { launch_wide_split<EV>(st, a, params, dt, isteps, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, scratch, e0, e1, err_word, pers); });
    return hipGetLastError();
}

// two tiles per workgroup (kernel comment): the two-wave policies, more than one tile per CU, every tile full
static bool rollout_split_two_tiles(int net, int N, int H, int cols, int C) {
    const int tiles = (N + G4_TRAJ - 1) / G4_TRAJ;
    return net == NET_MLP && tiles > 256 && tiles % 2 == 0 && N % G4_TRAJ == 0 &&
           2 * split_tile_floats(split_ex_fwd(net), cols, H, C) * sizeof(float) <= 160 * 1024 && getenv("CTK_SPLIT_ONE_TILE") == nullptr;
}
size_t ctk_g_rollout_split_lds(int net, int cols, int H, int C) { return split_tile_floats(split_ex_fwd(net), cols, H, C) * sizeof(float); }
bool ctk_g_rollout_split_ok(int env, int net, int N, int H, int cols) {
    static const bool off = getenv("CTK_NET_ONE_WAVE") != nullptr;
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    // (the 64-unit MLP: forward only, so the rollout kernels take it and the RPGD descent does not)
    // (... up to 256 tiles: its four waves per tile fill the chip's 1 024 SIMDs there; beyond, one wave per tile is faster — 233 against 333 us at 512)
    return !off && (split_env_ok(env, net) || net == NET_MLP64) && N <= (net == NET_MLP64 ? 4096 : 8192) && ctk_g_rollout_split_lds(net, cols, H, C) <= 160 * 1024;
}
int ctk_g_rollout_split_blocks(int N) { return (N + G4_TRAJ - 1) / G4_TRAJ; }
const char* ctk_g_rollout_split_name(int env, int net, int mode, bool log, int N, int H, int cols) {
    int S = 0, C = 0;
    env_dims(env, &S, &C);
    return ctk_kernel_name("ctk_g_rollout_split<%d, %4$s, %d, %5$s, %3$d>", env, mode, rollout_split_two_tiles(net, N, H, cols, C) ? 2 : 1,
                           split_policy_name(env, net), log ? "true" : "false");
}

template <int EV, class SP>
static void launch_rollout_split(hipStream_t st, int mode, const RolloutArgs& a_in, const float* params, float dt, int isteps, const MppiK& mk,
                                 const float* samples, const float* base, const float* scale, int rng_kind, const float* wperm, const float* hidden,
                                 float* parts, bool log, hipEvent_t e0, hipEvent_t e1, const MppiFuse* fuse) {
    using E = Env<EV>;
    RolloutArgs a = a_in;
    const int cols = (mode == CTK_G_MODE_MPPI ? a_in.P : a_in.H) * E::C;
    a.P = cols; a.p_magic = g4_magic_of(cols); a.C = E::C;
    const typename E::K k = E::derive(params, dt, isteps);
    const int tiles = ctk_g_rollout_split_blocks(a.N);
    const NetFuse gz = ctk_net_fuse(SP::WAVES == 4 ? fuse : nullptr, mode, a, E::C, base, tiles, cols);   // mppi_ll_tail: 256-thread workgroups
    const size_t lds1 = std::max(ctk_g_rollout_split_lds(SP::NET, cols, a.H, E::C), gz.mode ? merge_lds_staged(cols, tiles) : 0);
    auto go = [&](auto tiles_per_wg) {
        constexpr int T = decltype(tiles_per_wg)::value;
        const dim3 grid(tiles / T), block(64 * SP::WAVES * T);
        const size_t lds = T == 1 ? lds1 : (size_t)T * ctk_g_rollout_split_lds(SP::NET, cols, a.H, E::C);
        if (mode == CTK_G_MODE_MPPI) {
            if (log) CTK_LAUNCH((ctk_g_rollout_split<EV, SP, CTK_G_MODE_MPPI, true, T>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
            else CTK_LAUNCH((ctk_g_rollout_split<EV, SP, CTK_G_MODE_MPPI, false, T>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
        } else {
            if (log) CTK_LAUNCH((ctk_g_rollout_split<EV, SP, CTK_G_MODE_AFFINE, true, T>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
            else CTK_LAUNCH((ctk_g_rollout_split<EV, SP, CTK_G_MODE_AFFINE, false, T>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, parts, gz);
        }
    };
    if constexpr (SP::WAVES == 2) {
        if (rollout_split_two_tiles(SP::NET, a.N, a.H, cols, E::C)) { go(std::integral_constant<int, 2>{}); return; }
    }
    go(std::integral_constant<int, 1>{});
}

hipError_t ctk_launch_g_rollout_split(hipStream_t st, int env, int net, int mode, const RolloutArgs& a, const float* params, float dt, int isteps,
                                      const MppiK& mk, const float* samples, const float* base, const float* scale, int rng_kind,
                                      const float* wperm, const float* hidden, float* parts, bool log, hipEvent_t e0, hipEvent_t e1,
                                      const MppiFuse* fuse) {
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        if (net == NET_GRU) {
            launch_rollout_split<EV, SplitGru>(st, mode, a, params, dt, isteps, mk, samples, base, scale, rng_kind, wperm, hidden, parts, log, e0, e1, fuse);
        } else if (net == NET_MLP64) {
            launch_rollout_split<EV, SplitMlp64<(E::S + E::C > 8)>>(st, mode, a, params, dt, isteps, mk, samples, base, scale, rng_kind, wperm, hidden, parts, log, e0, e1, fuse);
        } else {
            launch_rollout_split<EV, SplitMlp<(E::S + E::C > 8)>>(st, mode, a, params, dt, isteps, mk, samples, base, scale, rng_kind, wperm, hidden, parts, log, e0, e1, fuse);
        }
    });
    return hipGetLastError();
}

hipError_t ctk_launch_g_gru_advance4(hipStream_t st, int env, const RolloutArgs& a, const float* u_dev, const float* wperm, float* hidden) {
    CTK_FOR_ENV(env, EV, {
        hipLaunchKernelGGL((ctk_g_gru_advance4<EV>), dim3(1), dim3(256), 0, st, a, u_dev, wperm, hidden);
    });
    return hipGetLastError();
}
