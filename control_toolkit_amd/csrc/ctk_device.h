// ctk_device.h — device-side building blocks: cart-pole step, stage/terminal cost, Philox RNG,
// wave reductions.  gfx950 / wave64 only.
#pragma once
#include "ctk_common.h"

#define CTK_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// wave64 reductions (ds_swizzle/DPP via __shfl_xor; all 64 lanes must be active)
// ---------------------------------------------------------------------------------------------
CTK_DEV float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
CTK_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Cart-pole (build-defined predictor, oracle/ctk_oracle.py:Predictor._ode_step).  One explicit
// Euler sub-step; sn/cs = sin/cos of the CURRENT angle (shared with the stage cost).
// ---------------------------------------------------------------------------------------------
struct State4 {
    float x, v, th, om;
};

CTK_DEV void ode_substep(const EnvK& k, State4& s, float F, float sn, float cs) {
    float A = F + k.k_ml * s.om * s.om * sn - k.M_fric * s.v;
    float tmp = A * k.inv_mt;
    float D = k.k43l - k.k_mpl_mt * cs * cs;
    float Nn = k.g * sn - cs * tmp - k.k_jf * s.om;
    float thdd = Nn / D;
    float xdd = tmp - k.k_mpl_mt * thdd * cs;
    float nx = s.x + k.dt * s.v;
    float nv = s.v + k.dt * xdd;
    float nth = s.th + k.dt * s.om;
    float nom = s.om + k.dt * thdd;
    s.x = nx; s.v = nv; s.th = nth; s.om = nom;
}

// full predictor step (intermediate_steps Euler sub-steps); sn/cs of the incoming angle given
CTK_DEV void ode_step(const EnvK& k, State4& s, float q, float sn, float cs) {
    float F = k.u_max * q;
    ode_substep(k, s, F, sn, cs);
    for (int i = 1; i < k.intermediate_steps; ++i) {
        float sn2, cs2;
        sincosf(s.th, &sn2, &cs2);
        ode_substep(k, s, F, sn2, cs2);
    }
}

// Stage cost of (state, u, previous u) — oracle Cost._get_stage_cost; cs = cos(angle).
CTK_DEV float stage_cost(const EnvK& k, const State4& s, float cs, float u, float uprev) {
    float dxn = (s.x - k.target_position) * k.inv_xs;
    float dd = k.dd_weight * dxn * dxn;
    float omc = 1.0f - cs;
    float ep = k.ep_c * omc * omc;
    float ekp = k.ekp_weight * s.om * s.om;
    float cc = k.ccR * u * u;
    float du = u - uprev;
    float ccrc = k.ccrc_weight * du * du;
    return dd + ep + ekp + cc + ccrc;
}

CTK_DEV float terminal_cost(const EnvK& k, const State4& s) {
    float dxn = (s.x - k.target_position) * k.inv_xs;
    float dd = k.dd_weight * dxn * dxn;
    float omc = 1.0f - cosf(s.th);
    float ep = k.ep_c * omc * omc;
    return k.terminal_weight * (dd + ep);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller; mirrors oracle/ctk_oracle.py:device_noise.
// counter = (global_row, block_of_4_columns, call, stream), key = (seed_lo, seed_hi)
// ---------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

CTK_DEV U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

CTK_DEV float u32_unit_open(uint32_t u) { return ((float)(u >> 8) + 1.0f) * (1.0f / 16777216.0f); }     // (0,1]
CTK_DEV float u32_unit_halfopen(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }           // [0,1)

// 4 raw draws for (row, colblock): kind 0 = N(0,1) via Box-Muller, 1 = U[0,1)
CTK_DEV void draw4(const RolloutArgs& a, uint32_t row, uint32_t cb, int kind, float out[4]) {
    U4 r = philox4x32_10(U4{row, cb, a.call, a.stream_id}, a.seed_lo, a.seed_hi);
    if (kind == 1) {
        out[0] = u32_unit_halfopen(r.x); out[1] = u32_unit_halfopen(r.y);
        out[2] = u32_unit_halfopen(r.z); out[3] = u32_unit_halfopen(r.w);
    } else {
        float ra = sqrtf(-2.0f * logf(u32_unit_open(r.x)));
        float rb = sqrtf(-2.0f * logf(u32_unit_open(r.z)));
        float sa, ca, sb, cb2;
        sincosf(6.28318530717958647692f * u32_unit_halfopen(r.y), &sa, &ca);
        sincosf(6.28318530717958647692f * u32_unit_halfopen(r.w), &sb, &cb2);
        out[0] = ra * ca; out[1] = ra * sa; out[2] = rb * cb2; out[3] = rb * sb;
    }
}

// row stride of the per-block sample tile in LDS: odd (conflict-free column walks with
// ds_read_b32: bank = addr/4 mod 32) and >= P+1 so that column P is a readable zero pad.
__host__ __device__ inline int tile_stride(int P) { return (P + 1) | 1; }
