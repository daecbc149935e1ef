// ctk_rollout.h — the fused rollout + cost loop for the analytic (ODE) predictor: one thread per
// trajectory, state in registers, H sequential steps (reference hot loop #1 + #2:
// optimizer_mppi.py:188 predict_core + :159 get_trajectory_cost, fused).
#pragma once
#include "ctk_device.h"

// Loads (or draws) the per-block sample tile into LDS.
//   tile[r * stride + c] = scale * sample[(row0 + r) * P + c]   r < rows_in_block, c < P
// Global reads are fully coalesced (the block's rows are one contiguous span of the
// [N,P,C] buffer); column P.. of every row is a zero pad.
template <int BLOCK>
CTK_DEV void load_tile(float* tile, const float* __restrict__ samples, const RolloutArgs& a, int row0, float scale,
                       int rng_kind) {
    const int P = a.P, stride = tile_stride(P);
    const int rows = min(BLOCK, a.N - row0);
    const int t = threadIdx.x;
    if (rows < BLOCK) {   // last, partial block: rows beyond N read as zeros
        for (int i = t; i < BLOCK * stride; i += BLOCK) tile[i] = 0.0f;
        __syncthreads();
    } else {
        for (int c = P; c < stride; ++c) tile[t * stride + c] = 0.0f;
    }
    if (samples != nullptr) {
        const float* src = samples + (size_t)row0 * P;
        const int total = rows * P;
        const int q = BLOCK / P, rem = BLOCK - q * P;   // advance of (r,c) per BLOCK elements
        int r = t / P, c = t - r * P;
        for (int i = t; i < total; i += BLOCK) {
            tile[r * stride + c] = src[i] * scale;
            r += q; c += rem;
            if (c >= P) { c -= P; ++r; }
        }
    } else {
        // on-device Philox: every thread draws its own row (global row index => shard invariant)
        const bool valid = (row0 + t) < a.N;
        const uint32_t grow = (uint32_t)(a.global_row0 + row0 + t);
        for (int cb = 0; cb * 4 < P; ++cb) {
            float d[4];
            draw4(a, grow, (uint32_t)cb, rng_kind, d);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (cb * 4 + j < P) tile[t * stride + cb * 4 + j] = valid ? d[j] * scale : 0.0f;
        }
    }
}

// Rolls one trajectory; ufn(h) yields the (already clipped) input of step h.
// Returns J = mean over [H stage costs | terminal cost]  (Cost_Functions/__init__.py:90-93).
template <bool WRITE_Q, bool WRITE_TRAJ, class UFn>
CTK_DEV float rollout_ode(const RolloutArgs& a, const EnvK& k, int n, bool valid, UFn&& ufn) {
    State4 s{a.s0[0], a.s0[1], a.s0[2], a.s0[3]};
    float uprev = a.u_prev_dev ? *a.u_prev_dev : a.u_prev;
    float csum = 0.0f;
    const int H = a.H;
    float4* traj = nullptr;
    if constexpr (WRITE_TRAJ) {
        if (a.traj_out) traj = reinterpret_cast<float4*>(a.traj_out) + (size_t)n * (H + 1);
    }
    float u_next = ufn(0);
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = ufn(h + 1);   // issued a whole step ahead of its use: LDS latency hidden
        float sn, cs;
        ctk_sincosf(s.th, &sn, &cs);
        csum += stage_cost(k, s, cs, u, uprev);
        if constexpr (WRITE_TRAJ) {
            if (valid && traj) traj[h] = make_float4(s.x, s.v, s.th, s.om);
        }
        if constexpr (WRITE_Q) {
            if (valid) a.Q_out[(size_t)n * H + h] = u;
        }
        ode_step(k, s, u, sn, cs);
        uprev = u;
    }
    if constexpr (WRITE_TRAJ) {
        if (valid && traj) traj[H] = make_float4(s.x, s.v, s.th, s.om);
    }
    return (csum + terminal_cost(k, s)) * a.inv_Hp1;
}
