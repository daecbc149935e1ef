// ctk_mlp_wide.h — (S+C)-64-64-S tanh MLP predictor on the fp32 matrix cores, as a policy of the ONE-WAVE template kernels
// (ctk_generic_net.hip).  The reference names a network by its sizes (`Dense-<I>IN-<h1>H1-<h2>H2-<O>OUT-<n>`,
// Control_Toolkit_ASF_Template/config_controllers.yml:8) and hands the name to the predictor (Controllers/controller_mpc.py:67-73): hidden
// layers of up to 32 units run on the 32-unit kernels (narrower ones embedded exactly, ctk_api.hip: ctk_set_predictor_weights_shaped);
// this file is the 64-unit tile count — FOUR 16-row tiles per layer instead of two, the hidden k-loops twice as long — for handles
// created with cfg.predictor_hidden1/2 in 33..64 (narrower layers again embedded exactly).  Same operand conventions as ctk_mlp.h
// (transposed products, accumulator layout of a layer = B operand of the next, hid(j, g) = 16 (j >> 2) + 4 g + (j & 3)); one wave per
// 16-trajectory tile, forward 12 + 64 + 16 MFMAs per step, reverse 8 + 64 + 16.  Built for function first: the two-wave / wide forms
// of ctk_net_split.hip stay 32-unit.
#pragma once
#include "ctk_mlp.h"

constexpr int MLPW_T = 4;                               // 16-row tiles per hidden layer
constexpr int MLPW_HID = 16 * MLPW_T;                   // 64
constexpr int MLPW_KH = 4 * MLPW_T;                     // k-steps over a hidden layer
// per-lane operand table (floats): forward w1[T][3] | w2[T][KH] | w3[KH] | b1[T][4] | b2[T][4] | b3[4]; reverse w3t[T][2] | w2t[T][KH] | w1t[KH]
constexpr int MLPW_FWD_PER_LANE = MLPW_T * 3 + MLPW_T * MLPW_KH + MLPW_KH + 4 * MLPW_T + 4 * MLPW_T + 4;     // 128
constexpr int MLPW_BWD_PER_LANE = MLPW_T * 2 + MLPW_T * MLPW_KH + MLPW_KH;                                    // 88
static_assert(MLPW_FWD_PER_LANE % 4 == 0 && MLPW_BWD_PER_LANE % 4 == 0, "float4 loads");

template <bool K3>
struct NetMlpWideT {
    static constexpr int TAPE = 4 + 8 * MLPW_T, LDS_FWD = 0, LDS_BWD = 0, HIDDEN = 0;     // (x0, x1, x2, -) | h1[T] | h2[T] float4s
    static constexpr bool THREE_KSTEPS = K3;
    struct Fwd {
        float w1[MLPW_T][3], w2[MLPW_T][MLPW_KH], w3[MLPW_KH];
        f32x4 b1[MLPW_T], b2[MLPW_T], b3;
        CTK_DEV void load(const float* __restrict__ table, float*) {
            const float* p = table + (size_t)(threadIdx.x & 63) * MLPW_FWD_PER_LANE;
            int o = 0;
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) w1[m][ks] = p[o++];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m)
#pragma unroll
                for (int j = 0; j < MLPW_KH; ++j) w2[m][j] = p[o++];
#pragma unroll
            for (int j = 0; j < MLPW_KH; ++j) w3[j] = p[o++];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) b1[m][r] = p[o++];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) b2[m][r] = p[o++];
#pragma unroll
            for (int r = 0; r < 4; ++r) b3[r] = p[o++];
        }
        CTK_DEV void begin(const float*) {}
        // x0 / x1 / x2: network inputs g, 4+g, 8+g of the lane's trajectory; returns outputs g and 4+g
        CTK_DEV MlpPair step(float x0, float x1, float x2, float4* tape) {
            f32x4 a[MLPW_T];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) a[m] = CTK_MFMA(w1[m][0], x0, b1[m]);
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) a[m] = CTK_MFMA(w1[m][1], x1, a[m]);
            if constexpr (K3) {
#pragma unroll
                for (int m = 0; m < MLPW_T; ++m) a[m] = CTK_MFMA(w1[m][2], x2, a[m]);
            }
            f32x4 h1[MLPW_T], c[MLPW_T];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) { h1[m] = ctk_tanhf4(a[m]); c[m] = b2[m]; }
#pragma unroll
            for (int j = 0; j < MLPW_KH; ++j) {
                const float b = h1[j >> 2][j & 3];
#pragma unroll
                for (int m = 0; m < MLPW_T; ++m) c[m] = CTK_MFMA(w2[m][j], b, c[m]);
            }
            f32x4 h2[MLPW_T];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) h2[m] = ctk_tanhf4(c[m]);
            f32x4 o0 = b3, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < MLPW_KH; j += 2) {
                o0 = CTK_MFMA(w3[j], h2[j >> 2][j & 3], o0);
                o1 = CTK_MFMA(w3[j + 1], h2[(j + 1) >> 2][(j + 1) & 3], o1);
            }
            if (tape != nullptr) {
                tape[0] = make_float4(x0, x1, x2, 0.f);
#pragma unroll
                for (int m = 0; m < MLPW_T; ++m) { tape[1 + m] = st4(h1[m]); tape[1 + MLPW_T + m] = st4(h2[m]); }
            }
            return MlpPair{o0[0] + o1[0], o0[1] + o1[1]};
        }
    };
    struct Bwd {
        float w3t[MLPW_T][2], w2t[MLPW_T][MLPW_KH], w1t[MLPW_KH];
        CTK_DEV void load(const float* __restrict__ table, float*) {
            const float* p = table + (size_t)64 * MLPW_FWD_PER_LANE + (size_t)(threadIdx.x & 63) * MLPW_BWD_PER_LANE;
            int o = 0;
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) { w3t[m][0] = p[o++]; w3t[m][1] = p[o++]; }
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m)
#pragma unroll
                for (int j = 0; j < MLPW_KH; ++j) w2t[m][j] = p[o++];
#pragma unroll
            for (int j = 0; j < MLPW_KH; ++j) w1t[j] = p[o++];
        }
        CTK_DEV void begin() {}
        // lam0 / lam1: adjoints of the NEXT state's components g / 4+g; returns the adjoints of network inputs g, 4+g, 8+g
        CTK_DEV MlpPair vjp(const float4* tape, float lam0, float lam1) {
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 d2[MLPW_T], d1[MLPW_T], s[MLPW_T];
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) {
                f32x4 t = CTK_MFMA(w3t[m][0], lam0, z);
                t = CTK_MFMA(w3t[m][1], lam1, t);
                const f32x4 h2 = ld4(tape + 1 + MLPW_T + m);
                d2[m] = t * (1.0f - h2 * h2);
                s[m] = z;
            }
#pragma unroll
            for (int j = 0; j < MLPW_KH; ++j) {
                const float b = d2[j >> 2][j & 3];
#pragma unroll
                for (int m = 0; m < MLPW_T; ++m) s[m] = CTK_MFMA(w2t[m][j], b, s[m]);
            }
#pragma unroll
            for (int m = 0; m < MLPW_T; ++m) {
                const f32x4 h1 = ld4(tape + 1 + m);
                d1[m] = s[m] * (1.0f - h1 * h1);
            }
            f32x4 o0 = z, o1 = z;
#pragma unroll
            for (int j = 0; j < MLPW_KH; j += 2) {
                o0 = CTK_MFMA(w1t[j], d1[j >> 2][j & 3], o0);
                o1 = CTK_MFMA(w1t[j + 1], d1[(j + 1) >> 2][(j + 1) & 3], o1);
            }
            return MlpPair{o0[0] + o1[0], o0[1] + o1[1], o0[2] + o1[2]};
        }
    };
};
