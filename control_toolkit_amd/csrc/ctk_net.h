// ctk_net.h — the network predictors as policies of the template kernels (ctk_generic_net.hip): one wave owns a 16-trajectory
// MFMA tile; lane (c, g) holds network input / output index g and 4+g of trajectory c (inputs = S state components then C control
// inputs, I = S + C <= 8; outputs = S <= 8 state components), see ctk_mlp.h.  A policy supplies
//     Fwd  per-thread forward object :  load(table, lds)  begin(hidden)  step(x0, x1, tape | nullptr) -> (out g, out 4+g)
//     Bwd  per-thread reverse object :  load(table, lds)  begin()        vjp(tape, lam g, lam 4+g)    -> (d input g, d input 4+g)
//     TAPE   floats a step's forward leaves per lane for its adjoint (float4-aligned)
//     LDS_FWD / LDS_BWD   floats of LDS the per-lane operand tables take (0: operands live in registers)
//   NetMlp   (S+C)-32-32-S tanh MLP: operands in registers (ctk_mlp.h)
//   NetGru   2 x 32 GRU + dense 32 -> S, PyTorch gate convention (oracle/ctk_oracle.py: gru_cell_fwd / gru_cell_bwd): 164 MFMAs
//            per step forward, 172 reverse (back-propagation through time: the hidden-state adjoints of both layers are carried
//            along the horizon); the per-lane A operands (232 + 172 values) are staged in LDS once per launch.
#pragma once
#include "ctk_mlp.h"

constexpr int NET_MLP = 1, NET_GRU = 2;     // == CTK_PRED_MLP, CTK_PRED_GRU
constexpr int NET_MLP64 = 4;                // kernel-variant id (handles created with hidden widths 33..64): the 64-unit MLP of ctk_mlp_wide.h, one-wave kernels only

CTK_DEV f32x4 ld4(const float4* p) { const float4 v = *p; return f32x4{v.x, v.y, v.z, v.w}; }
CTK_DEV float4 st4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

// ---------------------------------------------------------------------------------------------------------------
// K3: the environment has more than 8 network inputs (S + C > 8): layer 1 takes a third k-step (inputs 8 + g)
template <bool K3>
struct NetMlpT {
    static constexpr int TAPE = 20, LDS_FWD = 0, LDS_BWD = 0, HIDDEN = 0;
    static constexpr bool THREE_KSTEPS = K3;
    struct Fwd {
        MlpFwdW w;
        CTK_DEV void load(const float* __restrict__ table, float*) { w = mlp_load_fwd(table); }
        CTK_DEV void begin(const float*) {}
        CTK_DEV MlpPair step(float x0, float x1, float x2, float4* tape) {
            if (tape == nullptr) return mlp_step2<K3>(w, x0, x1, x2);
            MlpAct act;
            const MlpPair o = mlp_step2<K3>(w, x0, x1, x2, &act);
            tape[0] = make_float4(x0, x1, x2, 0.f);
            tape[1] = st4(act.h1[0]); tape[2] = st4(act.h1[1]); tape[3] = st4(act.h2[0]); tape[4] = st4(act.h2[1]);
            return o;
        }
    };
    struct Bwd {
        MlpBwdW2 w;
        CTK_DEV void load(const float* __restrict__ table, float*) { w = mlp_load_bwd2(table); }
        CTK_DEV void begin() {}
        CTK_DEV MlpPair vjp(const float4* tape, float lam0, float lam1) {
            MlpAct act;
            act.h1[0] = ld4(tape + 1); act.h1[1] = ld4(tape + 2); act.h2[0] = ld4(tape + 3); act.h2[1] = ld4(tape + 4);
            return mlp_step_vjp2(w, act, lam0, lam1);
        }
    };
};
using NetMlp = NetMlpT<false>;
using NetMlp3 = NetMlpT<true>;

// ---------------------------------------------------------------------------------------------------------------
// GRU.  Forward table (per lane, entry-major [e][64]), per layer L (KS = 2 for layer 1, 8 for layer 2):
//   Wi [gate r,z,n][tile m][k-step]          6*KS      A[row i][k-slot g] = W_i[gate*32 + 16m + i][input(ks, g)]
//   Wh [gate][m][j]                          48        A[row i][k-slot g] = W_h[gate*32 + 16m + i][hid(j, g)]
//   b_r[m][4] b_z[m][4] b_in[m][4] b_hn[m][4] 32       accumulator initial values (b_r, b_z: input + recurrent bias summed)
// then Wo[8] (rows placed by the io_of_row rule) and b_o[4].  92 + 128 + 12 = 232 entries.
// Reverse table: WoT [m][ks] 4 | layer 2: WiT [input tile 2][24] WhT [hidden tile 2][24] | layer 1: WiT [1][24] WhT [2][24]
//   = 4 + 96 + 72 = 172 entries; k-step (gate G, tile m, register r) of a transposed product has k-slot g = gate neuron
//   G*32 + 16m + 4g + r — exactly where the D layout holds that neuron's adjoint, so no data movement between the element-wise
//   gate adjoints and the products.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GRUG_L1 = 92, GRUG_L2 = 128, GRUG_FWD = GRUG_L1 + GRUG_L2 + 12;      // 232
constexpr int GRUG_BWD = 4 + 96 + 72;                                                // 172
// environments with more than 8 network inputs (K3): layer 1's THIRD k-step, Wi3 [gate][tile] (k-slot g = input 8 + g), kept BEHIND the
// reverse table so that every offset above holds for both; the reverse table needs nothing more (inputs 8 + g are rows 4g + 2 of the one
// input tile it already has).  The table of every GRU handle has the six entries (zeros where S + C <= 8).
constexpr int GRUG_X3 = 6;
constexpr int GRUG_TABLE = GRUG_FWD + GRUG_BWD + GRUG_X3;                            // 410

struct GruLayerTape {       // D layout, per hidden tile
    f32x4 r[2], z[2], n[2], ghn[2], hp[2];
};

CTK_DEV float gru_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

template <bool K3>
struct NetGruT {
    static constexpr int TAPE = 84, LDS_FWD = (GRUG_FWD + (K3 ? GRUG_X3 : 0)) * 64, LDS_BWD = GRUG_BWD * 64, HIDDEN = 64;
    static constexpr bool THREE_KSTEPS = K3;         // K3: a third layer-1 k-step (network inputs 8 + g), as NetMlpT<true>

    // wi3 / x3: the third k-step of layer 1 (X3 only)
    template <int KS, bool X3 = false, class XFn>
    CTK_DEV static void layer_fwd(const float* tab, int lane, XFn&& xb, f32x4 (&h)[2], GruLayerTape* tp, const float* wi3 = nullptr, float x3 = 0.0f) {
        const float* wi = tab;
        const float* wh = tab + 6 * KS * 64;
        const float* bb = wh + 48 * 64;
        f32x4 ar[2], az[2], ani[2], anh[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ar[m][r] = bb[(0 + m * 4 + r) * 64 + lane]; az[m][r] = bb[(8 + m * 4 + r) * 64 + lane];
                ani[m][r] = bb[(16 + m * 4 + r) * 64 + lane]; anh[m][r] = bb[(24 + m * 4 + r) * 64 + lane];
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float x = xb(ks);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ar[m] = CTK_MFMA(wi[((0 * 2 + m) * KS + ks) * 64 + lane], x, ar[m]);
                az[m] = CTK_MFMA(wi[((1 * 2 + m) * KS + ks) * 64 + lane], x, az[m]);
                ani[m] = CTK_MFMA(wi[((2 * 2 + m) * KS + ks) * 64 + lane], x, ani[m]);
            }
        }
        if constexpr (X3) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ar[m] = CTK_MFMA(wi3[(0 * 2 + m) * 64 + lane], x3, ar[m]);
                az[m] = CTK_MFMA(wi3[(1 * 2 + m) * 64 + lane], x3, az[m]);
                ani[m] = CTK_MFMA(wi3[(2 * 2 + m) * 64 + lane], x3, ani[m]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float hv = h[j >> 2][j & 3];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ar[m] = CTK_MFMA(wh[((0 * 2 + m) * 8 + j) * 64 + lane], hv, ar[m]);
                az[m] = CTK_MFMA(wh[((1 * 2 + m) * 8 + j) * 64 + lane], hv, az[m]);
                anh[m] = CTK_MFMA(wh[((2 * 2 + m) * 8 + j) * 64 + lane], hv, anh[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 rr, zz, nn, hn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rr[r] = gru_sigmoid(ar[m][r]);
                zz[r] = gru_sigmoid(az[m][r]);
                nn[r] = ctk_tanhf(ani[m][r] + rr[r] * anh[m][r]);
                hn[r] = (1.0f - zz[r]) * nn[r] + zz[r] * h[m][r];
            }
            if (tp) { tp->r[m] = rr; tp->z[m] = zz; tp->n[m] = nn; tp->ghn[m] = anh[m]; tp->hp[m] = h[m]; }
            h[m] = hn;
        }
    }

    struct Fwd {
        const float* tab;       // LDS
        f32x4 h1[2], h2[2];
        int lane;
        CTK_DEV void load(const float* __restrict__ table, float* lds) {   // all threads of the workgroup stage the table; caller syncs
            for (int i = threadIdx.x; i < GRUG_FWD * 64; i += blockDim.x) lds[i] = table[i];
            if constexpr (K3)
                for (int i = threadIdx.x; i < GRUG_X3 * 64; i += blockDim.x) lds[GRUG_FWD * 64 + i] = table[(GRUG_FWD + GRUG_BWD) * 64 + i];
            tab = lds; lane = threadIdx.x & 63;
        }
        CTK_DEV void begin(const float* __restrict__ hidden) {             // every rollout starts from the carried hidden state
            const int g = lane >> 4;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) { h1[m][r] = hidden[16 * m + 4 * g + r]; h2[m][r] = hidden[32 + 16 * m + 4 * g + r]; }
        }
        CTK_DEV MlpPair step(float x0, float x1, float x2, float4* tape) {
            GruLayerTape t1, t2;
            layer_fwd<2, K3>(tab, lane, [&](int ks) { return ks == 0 ? x0 : x1; }, h1, tape ? &t1 : nullptr, tab + GRUG_FWD * 64, x2);
            const f32x4 a = h1[0], b = h1[1];
            layer_fwd<8>(tab + GRUG_L1 * 64, lane, [&](int j) { return (j >> 2) ? b[j & 3] : a[j & 3]; }, h2, tape ? &t2 : nullptr);
            const float* wo = tab + (GRUG_L1 + GRUG_L2) * 64;
            f32x4 o0, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) o0[r] = wo[(8 + r) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                o0 = CTK_MFMA(wo[j * 64 + lane], h2[j >> 2][j & 3], o0);
                o1 = CTK_MFMA(wo[(j + 1) * 64 + lane], h2[(j + 1) >> 2][(j + 1) & 3], o1);
            }
            if (tape) {
                tape[0] = make_float4(x0, x1, x2, 0.f);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    tape[1 + m] = st4(t1.r[m]); tape[3 + m] = st4(t1.z[m]); tape[5 + m] = st4(t1.n[m]); tape[7 + m] = st4(t1.ghn[m]); tape[9 + m] = st4(t1.hp[m]);
                    tape[11 + m] = st4(t2.r[m]); tape[13 + m] = st4(t2.z[m]); tape[15 + m] = st4(t2.n[m]); tape[17 + m] = st4(t2.ghn[m]); tape[19 + m] = st4(t2.hp[m]);
                }
            }
            return MlpPair{o0[0] + o1[0], o0[1] + o1[1]};
        }
    };

    // adjoint of one GRU cell in place: dh (adjoint of h') -> dgi / dgh gate adjoints; returns the direct part of dh_prev (dh' * z)
    CTK_DEV static void cell_adjoint(const f32x4 (&dh)[2], const float4* tp /* r z n ghn hp, 2 tiles each */, f32x4 (&dar)[2], f32x4 (&daz)[2],
                                     f32x4 (&dan)[2], f32x4 (&dghn)[2], f32x4 (&dhp)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x4 r = ld4(tp + 0 + m), z = ld4(tp + 2 + m), n = ld4(tp + 4 + m), ghn = ld4(tp + 6 + m), hp = ld4(tp + 8 + m);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d = dh[m][q];
                const float dn = d * (1.0f - z[q]);
                const float dz = d * (hp[q] - n[q]);
                dhp[m][q] = d * z[q];
                const float a = dn * (1.0f - n[q] * n[q]);
                dan[m][q] = a;
                dghn[m][q] = a * r[q];
                daz[m][q] = dz * z[q] * (1.0f - z[q]);
                dar[m][q] = (a * ghn[q]) * r[q] * (1.0f - r[q]);
            }
        }
    }

    struct Bwd {
        const float* tab;       // LDS
        f32x4 dh1[2], dh2[2];   // adjoints of the hidden states handed to the EARLIER step
        int lane;
        CTK_DEV void load(const float* __restrict__ table, float* lds) {
            for (int i = threadIdx.x; i < LDS_BWD; i += blockDim.x) lds[i] = table[i];
            tab = lds; lane = threadIdx.x & 63;
        }
        CTK_DEV void begin() {
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            dh1[0] = dh1[1] = dh2[0] = dh2[1] = z;
        }
        // products W^T d for one layer: dx tiles (NI of them) from dgi, dhp tiles += from dgh.  k-step (G, m, r): entry G*8 + m*4 + r
        template <int NI>
        CTK_DEV void layer_products(const float* wiT, const float* whT, const f32x4 (&dar)[2], const f32x4 (&daz)[2], const f32x4 (&dan)[2],
                                    const f32x4 (&dghn)[2], f32x4 (&dx)[2], f32x4 (&dhp)[2]) {
#pragma unroll
            for (int G = 0; G < 3; ++G)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ks = G * 8 + m * 4 + r;
                        const float bi = G == 0 ? dar[m][r] : (G == 1 ? daz[m][r] : dan[m][r]);
                        const float bh = G == 2 ? dghn[m][r] : bi;
#pragma unroll
                        for (int mi = 0; mi < NI; ++mi) dx[mi] = CTK_MFMA(wiT[(mi * 24 + ks) * 64 + lane], bi, dx[mi]);
#pragma unroll
                        for (int mh = 0; mh < 2; ++mh) dhp[mh] = CTK_MFMA(whT[(mh * 24 + ks) * 64 + lane], bh, dhp[mh]);
                    }
        }
        CTK_DEV MlpPair vjp(const float4* tape, float lam0, float lam1) {
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            // out = Wo h2' + bo
            f32x4 d2[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                d2[m] = CTK_MFMA(tab[(m * 2 + 0) * 64 + lane], lam0, dh2[m]);
                d2[m] = CTK_MFMA(tab[(m * 2 + 1) * 64 + lane], lam1, d2[m]);
            }
            f32x4 dar[2], daz[2], dan[2], dghn[2], dhp[2], dx[2];
            // layer 2
            cell_adjoint(d2, tape + 11, dar, daz, dan, dghn, dhp);
            dx[0] = dh1[0]; dx[1] = dh1[1];                     // adjoint of h1' = W_i2^T dgi + what the later step handed back
            layer_products<2>(tab + 4 * 64, tab + (4 + 48) * 64, dar, daz, dan, dghn, dx, dhp);
            dh2[0] = dhp[0]; dh2[1] = dhp[1];
            // layer 1
            const f32x4 d1[2] = {dx[0], dx[1]};
            cell_adjoint(d1, tape + 1, dar, daz, dan, dghn, dhp);
            f32x4 din[2] = {zero, zero};
            layer_products<1>(tab + (4 + 96) * 64, tab + (4 + 96 + 24) * 64, dar, daz, dan, dghn, din, dhp);
            dh1[0] = dhp[0]; dh1[1] = dhp[1];
            return MlpPair{din[0][0], din[0][1], din[0][2]};      // network inputs g, 4 + g, 8 + g (rows 4g + 0 / 1 / 2 of the input tile)
        }
    };
};
using NetGru = NetGruT<false>;

#include "ctk_mlp_wide.h"
