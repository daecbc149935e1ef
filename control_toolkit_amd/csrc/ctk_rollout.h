// ctk_rollout.h — the fused rollout + cost loop for the analytic (ODE) predictor: one thread per
// trajectory, state in registers, H sequential steps (reference hot loop #1 + #2:
// optimizer_mppi.py:188 predict_core + :159 get_trajectory_cost, fused).
#pragma once
#include "ctk_device.h"
#ifndef STAMP
#define STAMP(i)
#endif

// Loads (or draws) the per-block sample tile into LDS, cooperatively with THREADS threads:
//   tile[r * stride + c] = scale * sample[(row0 + r) * P + c]   r < ROWS, c < P
// Global reads are fully coalesced 16-B-per-lane loads (the block's rows are one contiguous span
// of the [N,P,C] buffer), a batch of four per thread in flight before the first LDS store.
// Rows beyond N and the pad columns P.. of every row read as zeros.  Ends WITHOUT a barrier.
// `early` runs after the first batch of sample loads has been ISSUED and before anything waits on
// them: other global loads placed there share the same memory round trip.
template <int ROWS, int THREADS, class Early>
CTK_DEV void load_tile_early(float* tile, const float* __restrict__ samples, const RolloutArgs& a, int row0, float scale,
                       int rng_kind, Early&& early, int tid = -1) {
    const int P = a.P, ts = tile_stride(P);
    const int rows = min(ROWS, a.N - row0);
    const int t = tid < 0 ? (int)threadIdx.x : tid;       // (tid: the caller's THREADS are a part of the workgroup)
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
        bool early_done = false;
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
                if (b0 == 0) { STAMP(8); early(); early_done = true; STAMP(9); }
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
            STAMP(10);
        }
        if (!early_done) early();
        for (int i = done + t; i < total; i += THREADS) {
            const int r = P >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i;
            tile[r * ts + (i - r * P)] = src[i] * scale;
        }
    } else {
        early();
#ifndef CTK_DIAG_NO_COLD   // diagnostic builds only: measure the kernel without its cold code
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
#endif
    }
}

template <int ROWS, int THREADS>
CTK_DEV void load_tile(float* tile, const float* __restrict__ samples, const RolloutArgs& a, int row0, float scale, int rng_kind) {
    load_tile_early<ROWS, THREADS>(tile, samples, a, row0, scale, rng_kind, [] {});
}

// Rolls one trajectory; ufn(h) yields the (already clipped) input of step h.
// Returns J = mean over [H stage costs | terminal cost]  (Cost_Functions/__init__.py:90-93).
template <bool WRITE_Q, bool WRITE_TRAJ, class UFn>
CTK_DEV float rollout_ode(const RolloutArgs& a, const EnvK& k, int n, bool valid, UFn&& ufn) {
    State4 s{a.s0[0], a.s0[1], a.s0[2], a.s0[3]};
    float uprev = uniform_u_prev0(a);
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

// Recurrence of the MPPI kernel: the inputs arrive as forces F[h] = u_max * u[h] (prepared off the
// critical path together with the input-only cost terms), the per-step work is {sincos, state cost,
// Euler step}.  Returns sum_h (dd + ep + ekp) + terminal; *amax = max |angle| seen (range check of the
// fast sincos, done once after the loop by the caller).
// steps [hb, he) of the recurrence; s / csum / am are carried so that the caller may split the horizon
// (ctk_mppi_rollout runs the first steps while the other waves still prepare the inputs of the later ones)
template <bool WRITE_TRAJ, bool CHECKED, bool SINGLE, class FFn>
CTK_DEV void recur_ode_range(const EnvK& k, float4* traj, bool valid, FFn&& ffn, int hb, int he, State4& s, float& csum, float& am) {
    if (hb >= he) return;
    float F_next = ffn(hb);
#pragma unroll 2
    for (int h = hb; h < he; ++h) {
        const float F = F_next;
        if (h + 1 < he) F_next = ffn(h + 1);
        float sn, cs;
        if constexpr (CHECKED) ctk_sincosf(s.th, &sn, &cs);
        else { ctk_sincosf_fast(s.th, &sn, &cs); am = fmaxf(am, fabsf(s.th)); }
        if constexpr (WRITE_TRAJ) {
            if (valid && traj) traj[h] = make_float4(s.x, s.v, s.th, s.om);
        }
        ode_cost_substep(k, s, F, sn, cs, csum);
        if constexpr (!SINGLE) {
            for (int i = 1; i < k.intermediate_steps; ++i) {
                float sn2, cs2;
                ctk_sincosf(s.th, &sn2, &cs2);
                ode_substep(k, s, F, sn2, cs2);
            }
        }
    }
}

template <bool WRITE_TRAJ, bool CHECKED, bool SINGLE, class FFn>
CTK_DEV float recur_ode_state_cost(const RolloutArgs& a, const EnvK& k, int n, bool valid, FFn&& ffn, float* amax) {
    State4 s{a.s0[0], a.s0[1], a.s0[2], a.s0[3]};
    float csum = 0.0f, am = 0.0f;
    const int H = a.H;
    float4* traj = nullptr;
    if constexpr (WRITE_TRAJ) {
        if (a.traj_out) traj = reinterpret_cast<float4*>(a.traj_out) + (size_t)n * (H + 1);
    }
    recur_ode_range<WRITE_TRAJ, CHECKED, SINGLE>(k, traj, valid, ffn, 0, H, s, csum, am);
    if constexpr (WRITE_TRAJ) {
        if (valid && traj) traj[H] = make_float4(s.x, s.v, s.th, s.om);
    }
    *amax = am;
    return csum + terminal_cost(k, s);
}
