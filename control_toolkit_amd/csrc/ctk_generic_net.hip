// ctk_generic_net.hip — the environment-agnostic template kernels with a NETWORK predictor in place of the analytic model:
// (S+C)-32-32-S tanh MLP or 2 x 32 GRU + dense (policies NetMlp / NetGru of ctk_net.h, fp32 matrix cores both directions).
// Same structure as ctk_generic.hip (Env<> supplies S, C and the cost terms; the network replaces Env::step), one wave = one
// 16-trajectory MFMA tile, four tiles per workgroup.
// State layout of a tile: lane (c, g) holds components g and 4+g of trajectory c — layer 1's B operands as they stand; the
// cost terms need the whole state, gathered per step with cross-lane reads (every lane group evaluates the same cost).
#include "ctk_rollout.h"
#include "ctk_env.h"
#include "ctk_net.h"
#include "ctk_adam.h"
#include "ctk_launch.h"
#include "ctk_mppi_merge.h"
#include <algorithm>

constexpr int GN_TRAJ = 64, GN_BLOCK = 256, GN_LD = GN_TRAJ + 1;

template <int N>
CTK_DEV float pick(const float (&v)[N], int idx) {      // v[idx] for a lane-dependent idx without scratch memory
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) r = (i == idx) ? v[i] : r;
    return r;
}

// the whole state of the lane's trajectory from the (component g, component 4+g) layout
template <int S>
CTK_DEV void gather_state(float sv0, float sv1, int c, float (&s)[S]) {
#pragma unroll
    for (int j = 0; j < S; ++j) s[j] = __shfl(j < 4 ? sv0 : sv1, c + 16 * (j & 3), 64);
}

// layer-1 B operand of k-step 1 for lane group g: state component 4+g, or control input 4+g-S, or padding
template <int S, int C>
CTK_DEV float second_operand(float sv1, const float (&u)[C], int g) {
    float x1 = (4 + g < S) ? sv1 : 0.0f;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) x1 = (4 + g == S + cc) ? u[cc] : x1;
    return x1;
}
template <int S, int C>
CTK_DEV float third_operand(const float (&u)[C], int g) {              // network input 8+g: a control input where S + C > 8 (S <= 8)
    float x2 = 0.0f;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) x2 = (8 + g == S + cc) ? u[cc] : x2;
    return x2;
}
template <int S, int C>
CTK_DEV float first_operand(float sv0, const float (&u)[C], int g) {   // S < 4 environments: inputs may already start in k-step 0
    float x0 = (g < S) ? sv0 : 0.0f;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) x0 = (g == S + cc) ? u[cc] : x0;
    return x0;
}

// ---------------------------------------------------------------------------------------------------------------
// rollout + cost (MPPI / affine modes of ctk_generic.hip)
// ---------------------------------------------------------------------------------------------------------------
template <int ENV, class NET, int MODE, bool LOG>
__global__ __launch_bounds__(GN_BLOCK) void ctk_g_rollout_net(RolloutArgs a, typename Env<ENV>::K k, MppiK m, const float* __restrict__ samples,
                                                             const float* __restrict__ base, const float* __restrict__ scale, int rng_kind,
                                                             const float* __restrict__ wperm, const float* __restrict__ hidden,
                                                             int net_lds_off, float* __restrict__ parts, NetFuse gz) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    extern __shared__ float lds[];
    __shared__ float red_s[8];
    const int H = a.H, HC = H * C, cols = a.P, ts = tile_stride(cols);
    float* tile = lds;
    float* e_s = tile + GN_TRAJ * ts;
    float* base_s = e_s + GN_TRAJ;
    float* scale_s = base_s + HC;
    float* w0_s = scale_s + HC;
    float* w1_s = w0_s + H;
    int* i0_s = reinterpret_cast<int*>(w1_s + H);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * GN_TRAJ;
    const int tr = wave * 16 + c, n = row0 + tr;
    const bool valid = n < a.N;

    load_tile_early<GN_TRAJ, GN_BLOCK>(tile, samples, a, row0, MODE == CTK_G_MODE_MPPI ? m.stdev : 1.0f, rng_kind, [&] {
        if constexpr (MODE == CTK_G_MODE_MPPI) {
            for (int h = t; h < H; h += GN_BLOCK) {
                const InterpEntry e = a.interp[h];
                i0_s[h] = e.i0; w0_s[h] = e.w0; w1_s[h] = e.w1;
            }
            for (int hc = t; hc < HC; hc += GN_BLOCK) {
                const int h = hc / C, cc = hc - h * C;
                base_s[hc] = base[min(h + 1, H - 1) * C + cc];
            }
        } else {
            for (int hc = t; hc < HC; hc += GN_BLOCK) { base_s[hc] = base[hc]; scale_s[hc] = scale[hc]; }
        }
    });
    __syncthreads();

    typename NET::Fwd net;
    net.load(wperm, lds + net_lds_off);
    if constexpr (NET::LDS_FWD > 0) __syncthreads();
    net.begin(hidden);
    const float* my = tile + tr * ts;
    float sv0 = g < S ? lane_state4(a, g) : 0.0f, sv1 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    float up[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    float csum = 0.0f, corr = 0.0f;
    const int Pm1 = cols / C - 1;
    for (int h = 0; h < H; ++h) {
        float u[C];
        if constexpr (MODE == CTK_G_MODE_MPPI) {
            const int i0 = i0_s[h], i1 = min(i0 + 1, Pm1);
            const float w0 = w0_s[h], w1 = w1_s[h];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const float du = my[i0 * C + cc] * w0 + my[i1 * C + cc] * w1;
                u[cc] = fminf(fmaxf(base_s[h * C + cc] + du, a.lo[cc]), a.hi[cc]);
                corr += m.cc * (m.k_dd * (du * du) + m.R * u[cc] * du + m.k_uu * (u[cc] * u[cc]));
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < C; ++cc)
                u[cc] = fminf(fmaxf(base_s[h * C + cc] + my[h * C + cc] * scale_s[h * C + cc], a.lo[cc]), a.hi[cc]);
        }
        if constexpr (LOG || MODE == CTK_G_MODE_AFFINE) {
            if (valid && g == 0 && a.Q_out) {
#pragma unroll
                for (int cc = 0; cc < C; ++cc) a.Q_out[(size_t)n * HC + h * C + cc] = u[cc];
            }
        }
        float s[S];
        gather_state<S>(sv0, sv1, c, s);
        csum += E::stage_cost(k, s, u, up);
        if constexpr (LOG) {
            if (valid && g == 0 && a.traj_out) {
#pragma unroll
                for (int i = 0; i < S; ++i) a.traj_out[((size_t)n * (H + 1) + h) * S + i] = s[i];
            }
        }
        const MlpPair o = net.step(first_operand<S, C>(sv0, u, g), second_operand<S, C>(sv1, u, g), third_operand<S, C>(u, g), nullptr);
        sv0 = o.lo; sv1 = o.hi;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) up[cc] = u[cc];
    }
    float sT[S];
    gather_state<S>(sv0, sv1, c, sT);
    if constexpr (LOG) {
        if (valid && g == 0 && a.traj_out) {
#pragma unroll
            for (int i = 0; i < S; ++i) a.traj_out[((size_t)n * (H + 1) + H) * S + i] = sT[i];
        }
    }
    const float J = (csum + E::terminal_cost(k, sT)) * a.inv_Hp1 + corr;
    if (valid && g == 0) a.J[n] = J;

    if constexpr (MODE == CTK_G_MODE_MPPI) {
        const float rw = wave_min(valid ? J : INFINITY);           // every trajectory appears in all four lane groups: min unaffected
        if (lane == 0) red_s[wave] = rw;
        __syncthreads();
        const float rho = fminf(fminf(red_s[0], red_s[1]), fminf(red_s[2], red_s[3]));
        const float e = valid ? expf(m.neg_inv_lbd * (J - rho)) : 0.0f;
        const float aw = wave_sum(g == 0 ? e : 0.0f);
        if (g == 0) e_s[tr] = e;
        if (lane == 0) red_s[4 + wave] = aw;
        __syncthreads();
        float* rec = parts + (size_t)blockIdx.x * (2 + cols);
        const bool use_ll = gz.mode != 0;            // kernel-argument uniform: the records are handed over inside this launch
        unsigned long long* llr = gz.ll + (size_t)blockIdx.x * (2 + cols);
        if (t == 0) {
            const float aw_b = (red_s[4] + red_s[5]) + (red_s[6] + red_s[7]);
            if (use_ll) { ll_store(llr, rho, gz.up.seq); ll_store(llr + 1, aw_b, gz.up.seq); }
            else { rec[0] = rho; rec[1] = aw_b; }
        }
        for (int p = t; p < cols; p += GN_BLOCK) {
            float acc = 0.0f;
#pragma unroll 8
            for (int r = 0; r < GN_TRAJ; ++r) acc += e_s[r] * tile[r * ts + p];
            if (use_ll) ll_store(llr + 2 + p, acc, gz.up.seq);
            else rec[2 + p] = acc;
        }
        if (use_ll && blockIdx.x == 0) {             // block 0 gathers every block's words, merges, updates / emits the shard record
            __syncthreads();
            mppi_ll_tail<C>(lds, gz.ll, (int)gridDim.x, cols, m.neg_inv_lbd, gz.mode, gz.out_rec, gz.up);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// RPGD descent with a network predictor: forward with what the adjoint needs taped to an L2-resident global scratch
// ([h][lane][NET::TAPE] floats per wave: the step's inputs and the activations / gates), reverse sweep on MFMA (NET::Bwd::vjp;
// for the GRU back-propagation through time with the hidden-state adjoints carried), cost gradients from Env<>, per-plan
// clip_by_norm, Adam, clip, final cost pass.  64 plans per workgroup, 16 per wave.
// ---------------------------------------------------------------------------------------------------------------
template <int ENV, class NET>
__global__ __launch_bounds__(GN_BLOCK) void ctk_g_rpgd_descent_net(RolloutArgs a, typename Env<ENV>::K k, AdamK ad, float* __restrict__ Q,
                                                                  float* __restrict__ m, float* __restrict__ v,
                                                                  const float* __restrict__ bc_table, int bc_len, int t0, int iters,
                                                                  const float* __restrict__ wperm, const float* __restrict__ wperm_bwd,
                                                                  const float* __restrict__ hidden, float* __restrict__ scratch) {
    constexpr int GN_TAPE = NET::TAPE;
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    extern __shared__ float lds[];
    const int H = a.H, HC = H * C;
    float* q_s = lds;
    float* g_s = q_s + HC * GN_LD;
    float* sc_s = g_s + HC * GN_LD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g = lane >> 4;
    const int col = wave * 16 + c;
    const int row0 = blockIdx.x * GN_TRAJ;
    const int rows = min(GN_TRAJ, a.N - row0);
    const int total = rows * HC;
    const size_t gbase = (size_t)row0 * HC;
    float* tape = scratch + (size_t)(blockIdx.x * 4 + wave) * H * 64 * GN_TAPE;

    for (int i = t; i < GN_TRAJ * HC; i += GN_BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        q_s[hc * GN_LD + r] = i < total ? Q[gbase + i] : 0.0f;
    }
    __syncthreads();

    typename NET::Fwd nf;
    typename NET::Bwd nb;
    nf.load(wperm, sc_s + GN_TRAJ);
    nb.load(wperm_bwd, sc_s + GN_TRAJ + NET::LDS_FWD);
    if constexpr (NET::LDS_FWD + NET::LDS_BWD > 0) __syncthreads();
    float up0[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) up0[cc] = a.u_prev_dev ? a.u_prev_dev[cc] : a.u_prev[cc];
    const float inv = a.inv_Hp1;
    const float s00 = g < S ? lane_state4(a, g) : 0.0f, s01 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;

    for (int it = 0; it < iters; ++it) {
        // ---- forward, taping the step inputs and activations
        float sv0 = s00, sv1 = s01;
        nf.begin(hidden);
        for (int h = 0; h < H; ++h) {
            float u[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = q_s[(h * C + cc) * GN_LD + col];
            const MlpPair o = nf.step(first_operand<S, C>(sv0, u, g), second_operand<S, C>(sv1, u, g), third_operand<S, C>(u, g),
                                      reinterpret_cast<float4*>(tape + ((size_t)h * 64 + lane) * GN_TAPE));   // tape[0] = the step's (x0, x1, x2)
            sv0 = o.lo; sv1 = o.hi;
        }
        // ---- reverse sweep
        float sH[S], gT[S];
        gather_state<S>(sv0, sv1, c, sH);
        E::terminal_grad(k, sH, gT);
        float lam0 = g < S ? pick<S>(gT, g) * inv : 0.0f, lam1 = 4 + g < S ? pick<S>(gT, 4 + g) * inv : 0.0f;
        float nrm2 = 0.0f;
        float gp_next[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) gp_next[cc] = 0.0f;
        nb.begin();
        for (int h = H - 1; h >= 0; --h) {
            const float4* tq = reinterpret_cast<const float4*>(tape + ((size_t)h * 64 + lane) * GN_TAPE);
            const float4 in01 = tq[0];
            const float p0 = in01.x, p1 = in01.y;              // network inputs g, 4+g of step h: the state components where < S
            const MlpPair d = nb.vjp(tq, lam0, lam1);
            float s[S], gs[S], u[C], upv[C], gu[C], gp[C];
            gather_state<S>(p0, p1, c, s);
            E::stage_grad_state(k, s, gs);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                u[cc] = q_s[(h * C + cc) * GN_LD + col];
                upv[cc] = h > 0 ? q_s[((h - 1) * C + cc) * GN_LD + col] : up0[cc];
            }
            E::input_grad(k, u, upv, gu, gp);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int kk = S + cc;                         // network input index of control input cc: lane group kk % 4, half kk / 4
                if (g == (kk & 3)) {
                    const float gq = (gu[cc] + gp_next[cc]) * inv + (kk >= 8 ? d.ex : (kk >= 4 ? d.hi : d.lo));
                    g_s[(h * C + cc) * GN_LD + col] = gq;
                    nrm2 += gq * gq;
                }
                gp_next[cc] = gp[cc];
            }
            lam0 = g < S ? pick<S>(gs, g) * inv + d.lo : 0.0f;
            lam1 = 4 + g < S ? pick<S>(gs, 4 + g) * inv + d.hi : 0.0f;
        }
        nrm2 = sum_over_groups(nrm2);
        if (g == 0) sc_s[col] = ad.clip / fmaxf(sqrtf(nrm2), ad.clip);
        __syncthreads();
        const int ti = t0 + it + 1;
        const float bc1 = ti <= bc_len ? bc_table[2 * (ti - 1)] : 1.0f;
        const float bc2 = ti <= bc_len ? bc_table[2 * (ti - 1) + 1] : 1.0f;
        for (int i = t; i < total; i += GN_BLOCK) {
            const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC, cc = hc % C;
            float mm = 0.0f, vv = 0.0f;
            if (ad.rule != 2) { mm = m[gbase + i]; vv = v[gbase + i]; }
            const float gg = g_s[hc * GN_LD + r] * sc_s[r];
            q_s[hc * GN_LD + r] = adam_update(ad, q_s[hc * GN_LD + r], gg, mm, vv, bc1, bc2, a.lo[cc], a.hi[cc]);
            if (ad.rule != 2) { m[gbase + i] = mm; v[gbase + i] = vv; }
        }
        __syncthreads();
    }
    {   // get_action's forward pass (optimizer_rpgd.py:342): costs of the refined plans
        nf.begin(hidden);
        float sv0 = s00, sv1 = s01, csum = 0.0f, up[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) up[cc] = up0[cc];
        for (int h = 0; h < H; ++h) {
            float u[C], s[S];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) u[cc] = q_s[(h * C + cc) * GN_LD + col];
            gather_state<S>(sv0, sv1, c, s);
            csum += E::stage_cost(k, s, u, up);
            const MlpPair o = nf.step(first_operand<S, C>(sv0, u, g), second_operand<S, C>(sv1, u, g), third_operand<S, C>(u, g), nullptr);
            sv0 = o.lo; sv1 = o.hi;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) up[cc] = u[cc];
        }
        float sT[S];
        gather_state<S>(sv0, sv1, c, sT);
        const float J = (csum + E::terminal_cost(k, sT)) * inv;
        if (g == 0 && row0 + col < a.N) a.J[row0 + col] = J;
    }
    __syncthreads();
    for (int i = t; i < total; i += GN_BLOCK) {
        const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, hc = i - r * HC;
        Q[gbase + i] = q_s[hc * GN_LD + r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// predictor.update(s, Q0) for the GRU (optimizer_mppi.py:195-197): one wave advances the carried hidden state by the measured
// state and the applied input (all 16 MFMA columns carry the same values; column 0 writes back).  hidden = [h1[32] | h2[32]].
template <int ENV>
__global__ __launch_bounds__(64) void ctk_g_gru_advance(RolloutArgs a, const float* __restrict__ u_dev, const float* __restrict__ wperm,
                                                       float* __restrict__ hidden) {
    using E = Env<ENV>;
    constexpr int S = E::S, C = E::C;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    typename NetGruT<(S + C > 8)>::Fwd net;
    net.load(wperm, lds);
    __syncthreads();
    net.begin(hidden);
    float u[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) u[cc] = u_dev ? u_dev[cc] : a.u_prev[cc];
    const float sv0 = g < S ? lane_state4(a, g) : 0.0f, sv1 = 4 + g < S ? lane_state4(a, g, 4) : 0.0f;
    (void)net.step(first_operand<S, C>(sv0, u, g), second_operand<S, C>(sv1, u, g), third_operand<S, C>(u, g), nullptr);
    __syncthreads();                       // every lane has read `hidden` (begin) before column 0 overwrites it
    if (c == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { hidden[16 * m + 4 * g + r] = net.h1[m][r]; hidden[32 + 16 * m + 4 * g + r] = net.h2[m][r]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
static uint32_t magic_of(int d) { return d >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; }

size_t ctk_g_net_table_floats(int net) {
    if (net == NET_MLP64) return (size_t)64 * (MLPW_FWD_PER_LANE + MLPW_BWD_PER_LANE);
    return net == NET_GRU ? (size_t)GRUG_TABLE * 64 : (size_t)64 * (MLP_FWD_PER_LANE + MLP_BWD_PER_LANE);
}
size_t ctk_g_net_hidden_floats(int net) { return net == NET_GRU ? NetGru::HIDDEN : 0; }
static const float* bwd_table(int net, const float* wperm) { return net == NET_GRU ? wperm + (size_t)GRUG_FWD * 64 : wperm; }
static size_t net_lds_fwd(int net) { return net == NET_GRU ? NetGruT<true>::LDS_FWD : 0; }     // (the larger of the two forms: six entries)
static size_t net_lds_bwd(int net) { return net == NET_GRU ? NetGru::LDS_BWD : 0; }

int ctk_g_rollout_net_cols(int env, int mode, int P, int H) {
    int C = 0;
    CTK_FOR_ENV(env, EV, { C = Env<EV>::C; });
    return (mode == CTK_G_MODE_MPPI ? P : H) * C;
}

// workgroups = block records of one MPPI launch
int ctk_g_rollout_net_blocks(int env, int net, int mode, int N, int P, int H) {
    return ctk_g_rollout_split_ok(env, net, N, H, ctk_g_rollout_net_cols(env, mode, P, H)) ? ctk_g_rollout_split_blocks(N) : ctk_g_rollout_blocks(N);
}

const char* ctk_g_rollout_net_name(int env, int net, int mode, bool log, int N, int P, int H) {
    if (ctk_g_rollout_split_ok(env, net, N, H, ctk_g_rollout_net_cols(env, mode, P, H))) return ctk_g_rollout_split_name(env, net, mode, log, N, H, ctk_g_rollout_net_cols(env, mode, P, H));
    int io = 0;
    CTK_FOR_ENV(env, EV, { io = Env<EV>::S + Env<EV>::C; });
    return ctk_kernel_name("ctk_g_rollout_net<%d, %4$s, %d, %5$s>", env, mode, 0,
                           net == NET_MLP64 ? (io > 8 ? "NetMlpWideT<true>" : "NetMlpWideT<false>") : net == NET_GRU ? (io > 8 ? "NetGruT<true>" : "NetGru") : (io > 8 ? "NetMlpT<true>" : "NetMlp"),
                           log ? "true" : "false");
}

size_t ctk_g_rollout_net_lds(int env, int net, int N, int cols, int H, int C) {
    if (ctk_g_rollout_split_ok(env, net, N, H, cols)) return ctk_g_rollout_split_lds(net, cols, H, C);
    return ctk_g_rollout_lds(cols, H, C) + net_lds_fwd(net) * sizeof(float);
}

// the kernel-side fuse argument of an MPPI launch (mode 0 unless the caller asked for the in-launch hand-off and it fits)
NetFuse ctk_net_fuse(const MppiFuse* fuse, int mode, const RolloutArgs& a, int C, const float* u_nom, int blocks, int cols) {
    NetFuse gz{};
    if (fuse == nullptr || mode != CTK_G_MODE_MPPI || fuse->mode == 0 || fuse->ll == nullptr) return gz;
    if (!ctk_ll_records_ok(blocks, cols) || !merge_can_stage(cols, blocks)) return gz;
    gz.mode = fuse->mode; gz.ll = fuse->ll; gz.out_rec = fuse->out_rec;
    gz.up = MppiUpdateArgs{nullptr, nullptr, nullptr, nullptr, a.H, a.interp, u_nom, fuse->u_nom_out, a.lo[0], a.hi[0], fuse->u_dev, fuse->u_host, fuse->seq};
    gz.up.C = C;
    for (int c = 0; c < C; ++c) { gz.up.lo_c[c] = a.lo[c]; gz.up.hi_c[c] = a.hi[c]; }
    return gz;
}

// may an MPPI step of this handle run as ONE launch?  (the API asks before it chooses the fuse mode)
bool ctk_g_rollout_net_fusable(int env, int net, int N, int P, int H) {
    const int blocks = ctk_g_rollout_net_blocks(env, net, CTK_G_MODE_MPPI, N, P, H);
    // the merge tail is written for 256-thread workgroups: the two-wave MLP form (128 threads) keeps the separate update launch (~1 us)
    if (net != NET_GRU && ctk_g_rollout_split_ok(env, net, N, H, ctk_g_rollout_net_cols(env, CTK_G_MODE_MPPI, P, H))) return false;
    return ctk_ll_records_ok(blocks, ctk_g_rollout_net_cols(env, CTK_G_MODE_MPPI, P, H)) && merge_can_stage(ctk_g_rollout_net_cols(env, CTK_G_MODE_MPPI, P, H), blocks);
}

template <int EV, class NETT>
static void launch_rollout_net(hipStream_t st, int mode, const RolloutArgs& a_in, const float* params, float dt, int isteps, const MppiK& mk,
                               const float* samples, const float* base, const float* scale, int rng_kind, const float* wperm,
                               const float* hidden, float* parts, bool log, hipEvent_t e0, hipEvent_t e1, const MppiFuse* fuse) {
    using E = Env<EV>;
    RolloutArgs a = a_in;
    const int cols = (mode == CTK_G_MODE_MPPI ? a_in.P : a_in.H) * E::C;
    a.P = cols; a.p_magic = magic_of(cols); a.C = E::C;
    const typename E::K k = E::derive(params, dt, isteps);
    const dim3 grid(ctk_g_rollout_blocks(a.N)), block(GN_BLOCK);
    const size_t lds0 = ctk_g_rollout_lds(cols, a.H, E::C);
    const NetFuse gz = ctk_net_fuse(fuse, mode, a, E::C, base, (int)grid.x, cols);
    const size_t lds = std::max(lds0 + NETT::LDS_FWD * sizeof(float), gz.mode ? merge_lds_staged(cols, (int)grid.x) : 0);
    const int off = (int)(lds0 / sizeof(float));
    if (mode == CTK_G_MODE_MPPI) {
        if (log) CTK_LAUNCH((ctk_g_rollout_net<EV, NETT, CTK_G_MODE_MPPI, true>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, off, parts, gz);
        else CTK_LAUNCH((ctk_g_rollout_net<EV, NETT, CTK_G_MODE_MPPI, false>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, off, parts, gz);
    } else {
        if (log) CTK_LAUNCH((ctk_g_rollout_net<EV, NETT, CTK_G_MODE_AFFINE, true>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, off, parts, gz);
        else CTK_LAUNCH((ctk_g_rollout_net<EV, NETT, CTK_G_MODE_AFFINE, false>), grid, block, lds, st, e0, e1, a, k, mk, samples, base, scale, rng_kind, wperm, hidden, off, parts, gz);
    }
}

hipError_t ctk_launch_g_rollout_net(hipStream_t st, int env, int net, int mode, const RolloutArgs& a, const float* params, float dt, int isteps,
                                    const MppiK& mk, const float* samples, const float* base, const float* scale, int rng_kind,
                                    const float* wperm, float* parts, bool log, hipEvent_t e0, hipEvent_t e1, const MppiFuse* fuse) {
    const float* hidden = wperm + ctk_g_net_table_floats(net);
    if (ctk_g_rollout_split_ok(env, net, a.N, a.H, ctk_g_rollout_net_cols(env, mode, a.P, a.H)))   // one tile over several waves (ctk_net_split.hip)
        return ctk_launch_g_rollout_split(st, env, net, mode, a, params, dt, isteps, mk, samples, base, scale, rng_kind, wperm, hidden, parts, log, e0, e1, fuse);
    CTK_FOR_ENV(env, EV, {
        using MLP = NetMlpT<(Env<EV>::S + Env<EV>::C > 8)>;      // a third layer-1 k-step where the environment has more than 8 network inputs
        if (net == NET_GRU) launch_rollout_net<EV, NetGruT<(Env<EV>::S + Env<EV>::C > 8)>>(st, mode, a, params, dt, isteps, mk, samples, base, scale, rng_kind, wperm, hidden, parts, log, e0, e1, fuse);
        else if (net == NET_MLP64)
            launch_rollout_net<EV, NetMlpWideT<(Env<EV>::S + Env<EV>::C > 8)>>(st, mode, a, params, dt, isteps, mk, samples, base, scale, rng_kind, wperm, hidden, parts, log, e0, e1, fuse);
        else launch_rollout_net<EV, MLP>(st, mode, a, params, dt, isteps, mk, samples, base, scale, rng_kind, wperm, hidden, parts, log, e0, e1, fuse);
    });
    return hipGetLastError();
}

size_t ctk_g_rpgd_descent_net_lds(int env, int net, int N, int H) {
    int C = 0;
    CTK_FOR_ENV(env, EV, { C = Env<EV>::C; });
    if (ctk_g_rpgd_split_ok(env, net, N, H)) return ctk_g_rpgd_descent_split_lds(net, H, C);
    return (size_t)(2 * H * C * GN_LD + GN_TRAJ + net_lds_fwd(net) + net_lds_bwd(net)) * sizeof(float);
}

size_t ctk_g_rpgd_scratch_floats_net(int net, int N, int H) {
    if (net == NET_MLP64) return std::max((size_t)((N + GN_TRAJ - 1) / GN_TRAJ) * 4 * H * 64 * NetMlpWideT<false>::TAPE, ctk_g_rpgd_scratch_floats_wide(N, H));
    const size_t one_wave = (size_t)((N + GN_TRAJ - 1) / GN_TRAJ) * 4 * H * 64 * (net == NET_GRU ? NetGru::TAPE : NetMlp::TAPE);
    return std::max(std::max(one_wave, ctk_g_rpgd_scratch_floats_split(net, N, H)), net == NET_MLP ? ctk_g_rpgd_scratch_floats_wide(N, H) : (size_t)0);
}

const char* ctk_g_rpgd_descent_net_name(int env, int net, int N, int H) {
    if (net == NET_MLP64 && ctk_g_rpgd_persist64_ok(env, N, H)) return ctk_g_rpgd_persist64_name(env);
    if (ctk_g_rpgd_wide_ok(env, net, N, H)) return ctk_g_rpgd_wide_name(env, N, H);
    if (ctk_g_rpgd_split_ok(env, net, N, H)) return ctk_g_rpgd_descent_split_name(env, net);
    int io = 0;
    CTK_FOR_ENV(env, EV, { io = Env<EV>::S + Env<EV>::C; });
    return ctk_kernel_name("ctk_g_rpgd_descent_net<%d, %4$s>", env, 0, 0,
                           net == NET_MLP64 ? (io > 8 ? "NetMlpWideT<true>" : "NetMlpWideT<false>") : net == NET_GRU ? (io > 8 ? "NetGruT<true>" : "NetGru") : (io > 8 ? "NetMlpT<true>" : "NetMlp"));
}

hipError_t ctk_launch_g_rpgd_descent_net(hipStream_t st, int env, int net, const RolloutArgs& a_in, const float* params, float dt, int isteps,
                                         float lr, float b1, float b2, float eps, float clip, float* Q, float* m, float* v,
                                         const float* bc_table, int bc_len, int t0, int iters, const float* wperm, float* scratch,
                                         hipEvent_t e0, hipEvent_t e1, int rule, uint32_t* err_word, RpgdPersist* pers) {
    AdamK ad{lr, b1, b2, (float)(1.0 - (double)b1), (float)(1.0 - (double)b2), eps, clip, rule};
    const float* hidden = wperm + ctk_g_net_table_floats(net);
    const float* wb = bwd_table(net, wperm);
    if (net == NET_MLP64 && pers != nullptr && iters >= 1 && iters <= 63 && ctk_g_rpgd_persist64_ok(env, a_in.N, a_in.H))   // the 64-unit network: one launch (ctk_net_split.hip)
        return ctk_launch_g_rpgd_persist64(st, env, a_in, params, dt, isteps, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, scratch, e0, e1, err_word, pers);
    if (ctk_g_rpgd_wide_ok(env, net, a_in.N, a_in.H))       // MLP, N <= 4 096: phase + grid-wide Jacobian launches (ctk_net_split.hip)
        return ctk_launch_g_rpgd_wide_split(st, env, a_in, params, dt, isteps, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, scratch, e0, e1, err_word, pers);
    if (ctk_g_rpgd_split_ok(env, net, a_in.N, a_in.H))      // one tile over several waves while the population leaves SIMDs idle (ctk_net_split.hip)
        return ctk_launch_g_rpgd_descent_split(st, env, net, a_in, params, dt, isteps, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wb, hidden, scratch, e0, e1);
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        RolloutArgs a = a_in;
        a.C = E::C; a.p_magic = magic_of(a.H * E::C);
        const typename E::K k = E::derive(params, dt, isteps);
        const dim3 grid((a.N + GN_TRAJ - 1) / GN_TRAJ), block(GN_BLOCK);
        const size_t lds = ctk_g_rpgd_descent_net_lds(env, net, 1 << 30, a.H);   // this (one-wave) form
        if (net == NET_GRU)
            CTK_LAUNCH((ctk_g_rpgd_descent_net<EV, NetGruT<(E::S + E::C > 8)>>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wb, hidden, scratch);
        else if (net == NET_MLP64)
            CTK_LAUNCH((ctk_g_rpgd_descent_net<EV, NetMlpWideT<(E::S + E::C > 8)>>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wb, hidden, scratch);
        else
            CTK_LAUNCH((ctk_g_rpgd_descent_net<EV, NetMlpT<(E::S + E::C > 8)>>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm, wb, hidden, scratch);
    });
    return hipGetLastError();
}

hipError_t ctk_launch_g_gru_advance(hipStream_t st, int env, const RolloutArgs& a, const float* u_dev, float* wperm) {
    float* hidden = wperm + ctk_g_net_table_floats(NET_GRU);
    static const bool one_wave = getenv("CTK_NET_ONE_WAVE") != nullptr;
    if (!one_wave) return ctk_launch_g_gru_advance4(st, env, a, u_dev, wperm, hidden);      // ctk_net_split.hip: the four-wave step
    CTK_FOR_ENV(env, EV, {
        hipLaunchKernelGGL((ctk_g_gru_advance<EV>), dim3(1), dim3(64), NetGruT<true>::LDS_FWD * sizeof(float), st, a, u_dev, wperm, hidden);
    });
    return hipGetLastError();
}
