// ctk_rollout.h — the fused rollout + cost loop for the analytic (ODE) predictor: one thread per
// trajectory, state in registers, H sequential steps (reference hot loop #1 + #2:
// optimizer_mppi.py:188 predict_core + :159 get_trajectory_cost, fused).
#pragma once
#include "ctk_device.h"

// Loads (or draws) the per-block sample tile into LDS, cooperatively with THREADS threads:
//   tile[r * stride + c] = scale * sample[(row0 + r) * P + c]   r < ROWS, c < P
// Global reads are fully coalesced 16-B-per-lane loads (the block's rows are one contiguous span
// of the [N,P,C] buffer), a batch of four per thread in flight before the first LDS store.
// Rows beyond N and the pad columns P.. of every row read as zeros.  Ends WITHOUT a barrier.
template <int ROWS, int THREADS>
CTK_DEV void load_tile(float* tile, const float* __restrict__ samples, const RolloutArgs& a, int row0, float scale,
                       int rng_kind) {
    const int P = a.P, ts = tile_stride(P);
    const int rows = min(ROWS, a.N - row0);
    const int t = threadIdx.x;
    if (rows < ROWS) {
        for (int i = t; i < ROWS * ts; i += THREADS) tile[i] = 0.0f;
        __syncthreads();
    } else {
        for (int r = t; r < ROWS; r += THREADS)
            for (int c = P; c < ts; ++c) tile[r * ts + c] = 0.0f;
    }
    if (samples != nullptr) {
        const float* src = samples + (size_t)row0 * P;
        const int total = rows * P;
        int done = 0;
        if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const float4* src4 = reinterpret_cast<const float4*>(src);
            const int n4 = total >> 2;
            for (int b0 = 0; b0 < n4; b0 += 4 * THREADS) {
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i4 = b0 + j * THREADS + t;
                    if (i4 < n4) v[j] = src4[i4];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i4 = b0 + j * THREADS + t;
                    if (i4 < n4) {
                        const int flat = i4 << 2;
                        int r = P >= 2 ? (int)__umulhi((uint32_t)flat, a.p_magic) : flat, c = flat - r * P;
                        const float e4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            tile[r * ts + c] = e4[q] * scale;
                            if (++c == P) { c = 0; ++r; }
                        }
                    }
                }
            }
            done = n4 << 2;
        }
        for (int i = done + t; i < total; i += THREADS) {
            const int r = P >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i;
            tile[r * ts + (i - r * P)] = src[i] * scale;
        }
    } else {
        // on-device Philox, addressed by (global row, column block): shard- and launch-shape invariant
        constexpr int TPR = THREADS / ROWS;           // threads per row
        const int r = t % ROWS, cb0 = t / ROWS;
        const bool valid = (row0 + r) < a.N;
        const uint32_t grow = (uint32_t)(a.global_row0 + row0 + r);
        for (int cb = cb0; cb * 4 < P; cb += TPR) {
            float d[4];
            draw4(a, grow, (uint32_t)cb, rng_kind, d);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (cb * 4 + j < P && valid) tile[r * ts + cb * 4 + j] = d[j] * scale;
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
