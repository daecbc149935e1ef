// ctk_generic.hip — the environment-agnostic optimizer kernels: every template here is written against the
// Env<ID> interface of ctk_env.h only (S states, C control inputs, step / cost / adjoints), so the same code serves
// any plant the reference would select through predictor_specification / cost_function_specification
// (controller_mpc.py:67-82, cost_function_wrapper.py:59-66).  Shapes follow the reference: samples [N,P,C] / [N,H,C],
// plans Q [N,H,C], u_nom [H,C], trajectories [N,H+1,S]; a row of a [.,H,C] tensor is HC = H*C contiguous floats.
//
//   ctk_g_rollout<ENV, MODE, LOG>   one wave = 64 trajectories, one per lane, state in registers, the block's sample
//        tile staged in LDS by coalesced 16-B loads (or drawn by Philox), inputs formed inline:
//          MODE_MPPI    u = clip(shift(u_nom) + interp(stdev * noise))  + MPPI correction cost + block soft-min record
//                       {rho, a, b[P*C]}                                          (optimizer_mppi.py:154-193)
//          MODE_AFFINE  u = clip(base[h,c] + sample[n,h,c] * scale[h,c])           (CEM optimizer_cem_tf.py:64-66,
//                       random-action optimizer_random_action_tf.py:56-61, plain predict_core rollouts)
//   ctk_g_mppi_update<>             u_nom <- clip(shift(u_nom) + interp(b) / a) per input channel, u = u_nom[0,:]
//   ctk_g_cem_finish / ctk_g_pick_best_first   post-loop bookkeeping with C channels
//   ctk_g_rpgd_descent<ENV>         all Adam iterations of one MPC step in one launch: forward with a state tape,
//        reverse sweep through Env::step_vjp + cost gradients, per-plan clip_by_norm, Adam, clip, final cost pass
//        (optimizer_rpgd.py:306-338, :342)
// The block-record merge (ctk_mppi_merge<false>), the selection (ctk_select_topk), the refit (ctk_cem_refit) and the
// RPGD warm start are shared with the CartPole kernels: they only ever see P*C / H*C columns.
#include <type_traits>
#include "ctk_rollout.h"
#include "ctk_env.h"
#include "ctk_adam.h"
#include "ctk_launch.h"

constexpr int G_TRAJ = 64;

// ---------------------------------------------------------------------------------------------------------------
// rollout + cost.  a.P = number of sample columns per row (P*C for MPPI, H*C for the affine modes), a.p_magic its magic.
// LDS (floats): tile[64][ts] | e[64] | base[HC] | scale[HC] | w0[H] w1[H] i0[H]
// ---------------------------------------------------------------------------------------------------------------
template <int ENV, int MODE, bool LOG>
__global__ __launch_bounds__(G_TRAJ) void ctk_g_rollout(RolloutArgs a, typename Env<ENV>::K k, MppiK m, int cols_per_point,
                                                       const float* __restrict__ samples, const float* __restrict__ base,
                                                       const float* __restrict__ scale, int rng_kind, float* __restrict__ parts) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C, cols = a.P, ts = tile_stride(cols);
    float* tile = lds;
    float* e_s = tile + G_TRAJ * ts;
    float* base_s = e_s + G_TRAJ;             // MPPI: shifted nominal plan; affine: base
    float* scale_s = base_s + HC;             // affine only
    float* w0_s = scale_s + HC;               // MPPI only: interpolation table
    float* w1_s = w0_s + H;
    int* i0_s = reinterpret_cast<int*>(w1_s + H);
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * G_TRAJ;
    const int n = row0 + lane;
    const bool valid = n < a.N;
    (void)cols_per_point;

    load_tile_early<G_TRAJ, G_TRAJ>(tile, samples, a, row0, MODE == CTK_G_MODE_MPPI ? m.stdev : 1.0f, rng_kind, [&] {
        if constexpr (MODE == CTK_G_MODE_MPPI) {
            for (int h = lane; h < H; h += G_TRAJ) {
                const InterpEntry e = a.interp[h];
                i0_s[h] = e.i0; w0_s[h] = e.w0; w1_s[h] = e.w1;
            }
            for (int hc = lane; hc < HC; hc += G_TRAJ) {
                const int h = hc / C, c = hc - h * C;
                base_s[hc] = base[min(h + 1, H - 1) * C + c];                         // optimizer_mppi.py:184 (shift, repeat last)
            }
        } else {
            for (int hc = lane; hc < HC; hc += G_TRAJ) { base_s[hc] = base[hc]; scale_s[hc] = scale[hc]; }
        }
    });
    __syncthreads();

    const float* my = tile + lane * ts;
    float s[S], up[C];
#pragma unroll
    for (int i = 0; i < S; ++i) s[i] = a.s0[i];
#pragma unroll
    for (int c = 0; c < C; ++c) up[c] = a.u_prev_dev ? a.u_prev_dev[c] : a.u_prev[c];
    float csum = 0.0f, corr = 0.0f;
    const int Pm1 = cols / C - 1;             // MPPI: last inducing point
    for (int h = 0; h < H; ++h) {
        float u[C];
        if constexpr (MODE == CTK_G_MODE_MPPI) {
            const int i0 = i0_s[h], i1 = min(i0 + 1, Pm1);
            const float w0 = w0_s[h], w1 = w1_s[h];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float du = my[i0 * C + c] * w0 + my[i1 * C + c] * w1;          // Interpolator.py:97-106, per input channel
                u[c] = fminf(fmaxf(base_s[h * C + c] + du, a.lo[c]), a.hi[c]);        // optimizer_mppi.py:186-187
                corr += m.cc * (m.k_dd * (du * du) + m.R * u[c] * du + m.k_uu * (u[c] * u[c]));   // :154-155, summed over h and c
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c)
                u[c] = fminf(fmaxf(base_s[h * C + c] + my[h * C + c] * scale_s[h * C + c], a.lo[c]), a.hi[c]);
        }
        if constexpr (LOG || MODE == CTK_G_MODE_AFFINE) {
            if (valid && a.Q_out) {
#pragma unroll
                for (int c = 0; c < C; ++c) a.Q_out[(size_t)n * HC + h * C + c] = u[c];
            }
        }
        csum += E::stage_cost(k, s, u, up);
        if constexpr (LOG) {
            if (valid && a.traj_out) {
#pragma unroll
                for (int i = 0; i < S; ++i) a.traj_out[((size_t)n * (H + 1) + h) * S + i] = s[i];
            }
        }
        E::step(k, s, u);
#pragma unroll
        for (int c = 0; c < C; ++c) up[c] = u[c];
    }
    if constexpr (LOG) {
        if (valid && a.traj_out) {
#pragma unroll
            for (int i = 0; i < S; ++i) a.traj_out[((size_t)n * (H + 1) + H) * S + i] = s[i];
        }
    }
    // mean over [H stage costs | terminal] (Cost_Functions/__init__.py:90-93); the MPPI correction is added unscaled
    const float J = (csum + E::terminal_cost(k, s)) * a.inv_Hp1 + corr;
    if (valid) a.J[n] = J;

    if constexpr (MODE == CTK_G_MODE_MPPI) {
        // block-local soft-min record (optimizer_mppi.py:163-168 restricted to this block; merged by ctk_mppi_merge)
        const float rho = wave_min(valid ? J : INFINITY);
        const float e = valid ? expf(m.neg_inv_lbd * (J - rho)) : 0.0f;
        const float asum = wave_sum(e);
        e_s[lane] = e;
        __syncthreads();
        float* rec = parts + (size_t)blockIdx.x * (2 + cols);
        if (lane == 0) { rec[0] = rho; rec[1] = asum; }
        for (int p = lane; p < cols; p += G_TRAJ) {
            float acc = 0.0f;
#pragma unroll 8
            for (int r = 0; r < G_TRAJ; ++r) acc += e_s[r] * tile[r * ts + p];
            rec[2 + p] = acc;
        }
    }
}

// u_nom <- clip(shift(u_nom) + interp(b) / a)  (optimizer_mppi.py:190), u = u_nom[0, :] (:191).
// parts: n_parts <= G_UPD_MAX_PARTS block (or shard) records {rho, a, b[P*C]}, merged here first — rho = min rho_r,
// a = sum a_r e^{-(rho_r - rho)/lambda}, b likewise (optimizer_mppi.py:163-168 re-associated, as ctk_mppi_merge) — so that the
// template path's MPPI step is two launches (rollout, this) instead of three.  LDS: w[n_parts] | b[P*C]
constexpr int G_UPD_MAX_PARTS = 1024;

template <int DUMMY>
__global__ __launch_bounds__(256) void ctk_g_mppi_update(const float* __restrict__ parts, int n_parts, float neg_inv_lbd, int P, int C, int H,
                                                        const InterpEntry* __restrict__ interp, const float* __restrict__ u_nom_in,
                                                        float* __restrict__ u_nom_out, RolloutArgs a, float* __restrict__ u_dev,
                                                        float* __restrict__ u_host, uint32_t seq) {
    extern __shared__ float lds[];
    __shared__ float u_s[CTK_MAX_INPUTS];
    __shared__ float red_s[4];
    const int PC = P * C, rs = 2 + PC, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    float* w_s = lds;                 // [n_parts]
    float* b_s = w_s + n_parts;       // [PC]
    float rho = INFINITY;
    for (int r = t; r < n_parts; r += 256) rho = fminf(rho, parts[(size_t)r * rs]);
    rho = wave_min(rho);
    if (lane == 0) red_s[wave] = rho;
    __syncthreads();
    rho = fminf(fminf(red_s[0], red_s[1]), fminf(red_s[2], red_s[3]));
    __syncthreads();
    float asum = 0.0f;
    for (int r = t; r < n_parts; r += 256) {
        const float w = expf(neg_inv_lbd * (parts[(size_t)r * rs] - rho));
        w_s[r] = w;
        asum += parts[(size_t)r * rs + 1] * w;
    }
    asum = wave_sum(asum);
    if (lane == 0) red_s[wave] = asum;
    __syncthreads();
    const float a_tot = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    for (int pc = t; pc < PC; pc += 256) {
        float acc = 0.0f;
        for (int r = 0; r < n_parts; ++r) acc += parts[(size_t)r * rs + 2 + pc] * w_s[r];
        b_s[pc] = acc;
    }
    __syncthreads();
    for (int hc = t; hc < H * C; hc += 256) {
        const int h = hc / C, c = hc - h * C;
        const InterpEntry e = interp[h];
        const int i1 = min(e.i0 + 1, P - 1);
        const float w = (b_s[e.i0 * C + c] * e.w0 + b_s[i1 * C + c] * e.w1) / a_tot;
        const float o = fminf(fmaxf(u_nom_in[min(h + 1, H - 1) * C + c] + w, a.lo[c]), a.hi[c]);
        u_nom_out[hc] = o;
        if (h == 0) u_s[c] = o;
    }
    __syncthreads();
    if (t == 0) publish_u_vec(u_dev, u_host, u_s, C, seq);
}

// optimizer_cem_tf.py:99-102 with C channels: clip std, shift mean and std by one STEP (C floats), refill the tail with
// the mid-range / initial stdev; u = first input of the best elite (or of the mean: optimizer_cem_naive_grad_tf.py:103)
__global__ __launch_bounds__(256) void ctk_g_cem_finish(const float* __restrict__ Q, int ldq, const int* __restrict__ idx, int H, int C,
                                                       float* __restrict__ mu, float* __restrict__ sd, float std_min, float init_std,
                                                       RolloutArgs a, float* __restrict__ u_dev, float* __restrict__ u_host, uint32_t seq,
                                                       float std_max, int u_from_mu) {
    extern __shared__ float lds[];
    const int HC = H * C, t = threadIdx.x;
    float* m_s = lds;
    float* s_s = lds + HC;
    for (int i = t; i < HC; i += 256) {
        m_s[i] = mu[i];
        s_s[i] = fminf(fmaxf(sd[i], std_min), std_max);
    }
    __syncthreads();
    for (int i = t; i < HC; i += 256) {
        const int c = i % C;
        mu[i] = (i + C < HC) ? m_s[i + C] : (a.lo[c] + a.hi[c]) * 0.5f;
        sd[i] = (i + C < HC) ? s_s[i + C] : init_std;
    }
    if (t == 0) {
        float u[CTK_MAX_INPUTS];
        for (int c = 0; c < C; ++c) u[c] = u_from_mu ? m_s[c] : Q[(size_t)idx[0] * ldq + c];
        publish_u_vec(u_dev, u_host, u, C, seq);
    }
}

__global__ void ctk_g_pick_best_first(const float* __restrict__ Q, int ldq, const int* __restrict__ idx, int C, float* __restrict__ u_dev,
                                      float* __restrict__ u_host, uint32_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float u[CTK_MAX_INPUTS];
        for (int c = 0; c < C; ++c) u[c] = Q[(size_t)idx[0] * ldq + c];
        publish_u_vec(u_dev, u_host, u, C, seq);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// RPGD descent (optimizer_rpgd.py:306-338): a block keeps its 64 plans in LDS as [h*C+c][plan] (stride 65) across the
// iterations; wave 0 runs forward + reverse sweep (one plan per lane), all four waves do Adam with coalesced m/v traffic.
// Tape = the S state components entering every step ([h][i][lane], LDS when it fits, else global scratch); the reverse
// sweep recomputes a step's intermediates from them (Env::step_vjp).
// ---------------------------------------------------------------------------------------------------------------
constexpr int GR_WAVES = 4, GR_BLOCK = G_TRAJ * GR_WAVES, GR_LD = G_TRAJ + 1;

template <int ENV>
__global__ __launch_bounds__(GR_BLOCK) void ctk_g_rpgd_descent(RolloutArgs a, typename Env<ENV>::K k, AdamK ad, float* __restrict__ Q,
                                                              float* __restrict__ m, float* __restrict__ v,
                                                              const float* __restrict__ bc_table, int bc_len, int t0, int iters,
                                                              float* __restrict__ scratch, int tape_in_lds) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C;
    float* q_s = lds;                        // [HC][65]
    float* g_s = q_s + HC * GR_LD;           // [HC][65]
    float* sc_s = g_s + HC * GR_LD;          // [64]
    float* tape_l = sc_s + G_TRAJ;           // [H][NT][64]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * G_TRAJ;
    const int rows = min(G_TRAJ, a.N - row0);
    const int total = rows * HC;
    const size_t gbase = (size_t)row0 * HC;
    constexpr int NT = E::NT;                // taped values per step (Env::fwd_tape / bwd_tape: the sweep recomputes nothing)
    float* tape = tape_in_lds ? tape_l : scratch + (size_t)blockIdx.x * H * NT * 64;

    for (int i = t; i < G_TRAJ * HC; i += GR_BLOCK) {      // a.p_magic = ceil(2^32 / HC)
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        q_s[hc * GR_LD + r] = i < total ? Q[gbase + i] : 0.0f;
    }
    __syncthreads();

    float up0[C];
#pragma unroll
    for (int c = 0; c < C; ++c) up0[c] = a.u_prev_dev ? a.u_prev_dev[c] : a.u_prev[c];
    const float inv = a.inv_Hp1;

    // forward pass of a gradient iteration: no cost, NT taped values per step; the final state in sF
    auto forward_tape = [&](float (&sF)[S]) {
        float s[S];
#pragma unroll
        for (int i = 0; i < S; ++i) s[i] = a.s0[i];
        float un[C];
#pragma unroll
        for (int c = 0; c < C; ++c) un[c] = q_s[c * GR_LD + lane];
        for (int h = 0; h < H; ++h) {
            float u[C], tp[NT];
#pragma unroll
            for (int c = 0; c < C; ++c) u[c] = un[c];
            if (h + 1 < H) {
#pragma unroll
                for (int c = 0; c < C; ++c) un[c] = q_s[((h + 1) * C + c) * GR_LD + lane];
            }
            E::fwd_tape(k, s, u, tp);
#pragma unroll
            for (int i = 0; i < NT; ++i) tape[((size_t)h * NT + i) * 64 + lane] = tp[i];
        }
#pragma unroll
        for (int i = 0; i < S; ++i) sF[i] = s[i];
    };
    // get_action's cost pass (:342): the recurrence of the sampling kernels (Env::cost_step, checked fallback) + the input-only terms
    auto final_cost = [&]() {
        float u[C], up[C], cin = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) up[c] = up0[c];
        for (int h = 0; h < H; ++h) {
#pragma unroll
            for (int c = 0; c < C; ++c) u[c] = q_s[(h * C + c) * GR_LD + lane];
            cin += E::input_cost(k, u, up);
#pragma unroll
            for (int c = 0; c < C; ++c) up[c] = u[c];
        }
        auto run = [&](auto fast) {
            constexpr bool FAST = decltype(fast)::value;
            float s[S], csum = 0.0f, amax = 0.0f;
#pragma unroll
            for (int i = 0; i < S; ++i) s[i] = a.s0[i];
            float un[C];
#pragma unroll
            for (int c = 0; c < C; ++c) un[c] = q_s[c * GR_LD + lane];
            for (int h = 0; h < H; ++h) {
                float f[C];
#pragma unroll
                for (int c = 0; c < C; ++c) f[c] = E::prep_input(k, un[c], c);
                if (h + 1 < H) {
#pragma unroll
                    for (int c = 0; c < C; ++c) un[c] = q_s[((h + 1) * C + c) * GR_LD + lane];
                }
                E::template cost_step<FAST>(k, s, f, csum, amax);
            }
            const float J = csum + E::terminal_cost(k, s);
            return __builtin_amdgcn_ballot_w64(FAST && E::out_of_range(amax)) != 0 ? __builtin_nanf("") : J;
        };
        float J = E::fast_ok(k) ? run(std::true_type{}) : __builtin_nanf("");
        if (__builtin_amdgcn_ballot_w64(J != J) != 0) J = run(std::false_type{});   // wave-uniform: Euler sub-steps or an angle out of range
        return (J + cin) * inv;
    };

    for (int it = 0; it < iters; ++it) {
        if (wave == 0) {
            float sH[S], lam[S];
            forward_tape(sH);
            E::terminal_grad(k, sH, lam);
#pragma unroll
            for (int i = 0; i < S; ++i) lam[i] *= inv;
            float nrm2 = 0.0f;
            float gp_next[C];                 // d stage_{h+1} / d u_h (through u_prev of the next step)
#pragma unroll
            for (int c = 0; c < C; ++c) gp_next[c] = 0.0f;
            for (int h = H - 1; h >= 0; --h) {
                float tp[NT], u[C], upv[C];
#pragma unroll
                for (int i = 0; i < NT; ++i) tp[i] = tape[((size_t)h * NT + i) * 64 + lane];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    u[c] = q_s[(h * C + c) * GR_LD + lane];
                    upv[c] = h > 0 ? q_s[((h - 1) * C + c) * GR_LD + lane] : up0[c];
                }
                float du[C], gu[C], gp[C];
                E::bwd_tape(k, tp, u, lam, du, inv);
                E::input_grad(k, u, upv, gu, gp);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float g = (gu[c] + gp_next[c]) * inv + du[c];
                    g_s[(h * C + c) * GR_LD + lane] = g;
                    nrm2 += g * g;
                    gp_next[c] = gp[c];
                }
            }
            sc_s[lane] = ad.clip / fmaxf(sqrtf(nrm2), ad.clip);       // clip_by_norm over [H,C] (:315,:334)
        }
        __syncthreads();
        const int ti = t0 + it + 1;
        const float bc1 = ti <= bc_len ? bc_table[2 * (ti - 1)] : 1.0f;
        const float bc2 = ti <= bc_len ? bc_table[2 * (ti - 1) + 1] : 1.0f;
        for (int i = t; i < total; i += GR_BLOCK) {
            const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC, c = hc % C;
            float mm = 0.0f, vv = 0.0f;
            if (ad.rule != 2) { mm = m[gbase + i]; vv = v[gbase + i]; }
            const float g = g_s[hc * GR_LD + r] * sc_s[r];
            q_s[hc * GR_LD + r] = adam_update(ad, q_s[hc * GR_LD + r], g, mm, vv, bc1, bc2, a.lo[c], a.hi[c]);
            if (ad.rule != 2) { m[gbase + i] = mm; v[gbase + i] = vv; }
        }
        __syncthreads();
    }
    if (wave == 0) {                          // get_action's forward pass (:342)
        const float J = final_cost();
        if (row0 + lane < a.N) a.J[row0 + lane] = J;
    }
    __syncthreads();
    for (int i = t; i < total; i += GR_BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        Q[gbase + i] = q_s[hc * GR_LD + r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
static uint32_t magic_of(int d) { return d >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; }

size_t ctk_g_rollout_lds(int cols, int H, int C) {
    return (size_t)(G_TRAJ * tile_stride(cols) + G_TRAJ + 2 * H * C + 3 * H) * sizeof(float);
}

int ctk_g_rollout_blocks(int N) { return (N + G_TRAJ - 1) / G_TRAJ; }

const char* ctk_g_rollout_name(int env, int mode, bool log) {
    return ctk_kernel_name("ctk_g_rollout<%d, %d, %4$s>", env, mode == CTK_G_MODE_MPPI ? 0 : 1, 0, log ? "true" : "false");
}

hipError_t ctk_launch_g_rollout(hipStream_t st, int env, int mode, const RolloutArgs& a_in, const float* params, float dt, int isteps,
                                const MppiK& mk, const float* samples, const float* base, const float* scale, int rng_kind,
                                float* parts, bool log, hipEvent_t e0, hipEvent_t e1) {
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        RolloutArgs a = a_in;
        const int cols = (mode == CTK_G_MODE_MPPI ? a_in.P : a_in.H) * E::C;
        a.P = cols; a.p_magic = magic_of(cols); a.C = E::C;
        const typename E::K k = E::derive(params, dt, isteps);
        const dim3 grid(ctk_g_rollout_blocks(a.N)), block(G_TRAJ);
        const size_t lds = ctk_g_rollout_lds(cols, a.H, E::C);
        if (mode == CTK_G_MODE_MPPI) {
            if (log) CTK_LAUNCH((ctk_g_rollout<EV, CTK_G_MODE_MPPI, true>), grid, block, lds, st, e0, e1, a, k, mk, E::C, samples, base, scale, rng_kind, parts);
            else CTK_LAUNCH((ctk_g_rollout<EV, CTK_G_MODE_MPPI, false>), grid, block, lds, st, e0, e1, a, k, mk, E::C, samples, base, scale, rng_kind, parts);
        } else {
            if (log) CTK_LAUNCH((ctk_g_rollout<EV, CTK_G_MODE_AFFINE, true>), grid, block, lds, st, e0, e1, a, k, mk, E::C, samples, base, scale, rng_kind, parts);
            else CTK_LAUNCH((ctk_g_rollout<EV, CTK_G_MODE_AFFINE, false>), grid, block, lds, st, e0, e1, a, k, mk, E::C, samples, base, scale, rng_kind, parts);
        }
    });
    return hipGetLastError();
}

int ctk_g_mppi_update_max_parts() { return G_UPD_MAX_PARTS; }

hipError_t ctk_launch_g_mppi_update(hipStream_t st, const float* parts, int n_parts, float neg_inv_lbd, int P, int C, int H,
                                    const InterpEntry* interp, const float* u_nom_in, float* u_nom_out, const RolloutArgs& a, float* u_dev,
                                    float* u_host, uint32_t seq) {
    hipLaunchKernelGGL(ctk_g_mppi_update<0>, dim3(1), dim3(256), (size_t)(n_parts + P * C) * sizeof(float), st, parts, n_parts, neg_inv_lbd, P, C, H,
                       interp, u_nom_in, u_nom_out, a, u_dev, u_host, seq);
    return hipGetLastError();
}

hipError_t ctk_launch_g_cem_finish(hipStream_t st, const float* Q, const int* idx, int H, int C, float* mu, float* sd, float std_min,
                                   float init_std, const RolloutArgs& a, float* u_dev, float* u_host, uint32_t seq, int ldq, float std_max,
                                   int u_from_mu) {
    hipLaunchKernelGGL(ctk_g_cem_finish, dim3(1), dim3(256), 2 * H * C * sizeof(float), st, Q, ldq, idx, H, C, mu, sd, std_min, init_std, a,
                       u_dev, u_host, seq, std_max, u_from_mu);
    return hipGetLastError();
}

hipError_t ctk_launch_g_pick_best_first(hipStream_t st, const float* Q, const int* idx, int C, float* u_dev, float* u_host, uint32_t seq,
                                        int ldq) {
    hipLaunchKernelGGL(ctk_g_pick_best_first, dim3(1), dim3(64), 0, st, Q, ldq, idx, C, u_dev, u_host, seq);
    return hipGetLastError();
}

static size_t g_rpgd_lds_base(int H, int C) { return (size_t)(2 * H * C * GR_LD + G_TRAJ) * sizeof(float); }

size_t ctk_g_rpgd_descent_lds(int env, int H, bool* tape_in_lds) {
    int NT = 0, C = 0;
    CTK_FOR_ENV(env, EV, { NT = Env<EV>::NT; C = Env<EV>::C; });
    const size_t base = g_rpgd_lds_base(H, C), tape = (size_t)H * NT * 64 * sizeof(float);
    const bool fits = base + tape <= 160 * 1024;
    if (tape_in_lds) *tape_in_lds = fits;
    return fits ? base + tape : base;
}

size_t ctk_g_rpgd_scratch_floats(int env, int N, int H) {
    int NT = 0;
    CTK_FOR_ENV(env, EV, { NT = Env<EV>::NT; });
    return (size_t)((N + G_TRAJ - 1) / G_TRAJ) * H * NT * 64;
}

const char* ctk_g_rpgd_descent_name(int env) { return ctk_kernel_name("ctk_g_rpgd_descent<%d>", env); }

hipError_t ctk_launch_g_rpgd_descent(hipStream_t st, int env, const RolloutArgs& a_in, const float* params, float dt, int isteps, float lr,
                                     float b1, float b2, float eps, float clip, float* Q, float* m, float* v, const float* bc_table,
                                     int bc_len, int t0, int iters, float* scratch, hipEvent_t e0, hipEvent_t e1, int rule) {
    AdamK ad{lr, b1, b2, (float)(1.0 - (double)b1), (float)(1.0 - (double)b2), eps, clip, rule};
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        RolloutArgs a = a_in;
        a.C = E::C; a.p_magic = magic_of(a.H * E::C);
        const typename E::K k = E::derive(params, dt, isteps);
        bool tape_in_lds = false;
        const size_t lds = ctk_g_rpgd_descent_lds(env, a.H, &tape_in_lds);
        const dim3 grid((a.N + G_TRAJ - 1) / G_TRAJ), block(GR_BLOCK);
        CTK_LAUNCH((ctk_g_rpgd_descent<EV>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, scratch,
                   tape_in_lds ? 1 : 0);
    });
    return hipGetLastError();
}
