// ctk_sampled.hip — rollouts whose inputs are an affine map of per-step samples, and the
// selection / refit kernels around them:
//   CEM            Q = clip(mu[h] + eps[n,h]*std[h])          optimizer_cem_tf.py:64-66
//   random-action  Q = lo + u01[n,h]*(hi-lo)                  optimizer_random_action_tf.py:56-61
//   plain rollout  Q given (base 0, scale 1, no clip)         predictor.predict_core(s, Q)
// plus smallest-K selection under the total order (J, index) (tf.argsort at
// optimizer_cem_tf.py:73 / optimizer_rpgd.py:345 with ties fixed by index), the CEM elite
// refit (:77-78) and the post-loop shift (:99-102).
#include "ctk_rollout.h"
#include "ctk_env.h"
#include "ctk_mlp.h"
#include "ctk_gru.h"
#include "ctk_launch.h"

constexpr int SAMP_TRAJ = 64;                 // trajectories per block
constexpr int SAMP_WAVES = 4;
constexpr int SAMP_BLOCK = SAMP_TRAJ * SAMP_WAVES;

// Same anatomy as ctk_mppi_rollout (ctk_mppi.hip): 4-wave prologues (coalesced sample tile -> LDS; inputs
// of all H steps, input-only cost terms, coalesced write of the plans Q), then the recurrence on one
// wave (ODE, one trajectory per lane) or on all four (MLP, 16 trajectories per wave on MFMA).
// GRU: the four waves share ONE 16-trajectory column block (ctk_gru.h), so a workgroup owns 16 trajectories.
// LDS carve (floats): tile[TRAJ][ts] | ubuf[TRAJ][us] | cin[256/TRAJ][TRAJ] | base[H] | scale[H] | (GRU) exchange slots
// H here = H*C flat (step, input) columns of a plan (C control inputs; CartPole: C = 1)
__host__ __device__ inline int affine_carve_floats(int H, int traj) {
    const int f = traj * tile_stride(H) + traj * ((H + 1) | 1) + SAMP_BLOCK + 2 * H;
    return (f + 3) & ~3;
}
__host__ __device__ inline int affine_traj(int pred) { return pred == CTK_PRED_GRU ? GRU_TRAJ : SAMP_TRAJ; }

// Optional in-launch arg-min tail (random-action, optimizer_random_action_tf.py:62-68: u = first input of the cheapest
// plan): every block hands {key(J), index, first input} of its cheapest rollout to block 0 as {payload, sequence number}
// words (the hand-off form of ctk_mppi.hip), block 0 picks the global minimum under the total order (J, index) — what
// ctk_select_topk(K = 1) + ctk_pick_best_first do in two more launches — and publishes u.
struct BestArgs {
    unsigned long long* ll;   // [blocks][2 + C] words {key, index, first input[C]}; nullptr: no tail
    uint32_t seq;
    float* u_dev;
    float* u_host;
    int* idx_out;
};
CTK_DEV void ll_put(unsigned long long* p, uint32_t payload, uint32_t seq) {
    __hip_atomic_store(p, ((unsigned long long)seq << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// *expired is raised when the bounded poll runs out (the payload is then stale: the caller reports it, see below)
CTK_DEV uint32_t ll_get(const unsigned long long* p, uint32_t seq, bool* expired) {
    unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; (uint32_t)(w >> 32) != seq && spin < (1 << 22); ++spin) {
        __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((uint32_t)(w >> 32) != seq) *expired = true;
    return (uint32_t)w;
}

// ENV: the environment (ctk_env.h); the analytic-predictor instantiation is written against Env<ENV> only (C control inputs:
// H*C sample columns and inputs per trajectory, per-channel base / scale / clip; recurrence through Env::cost_step); the network
// predictors' instantiations are CartPole's.  P_ = H*C sample columns of a row, pmagic_ its magic.
template <int ENV, int PRED, bool WTRAJ>
// (argument order: see ctk_mppi_rollout — the leading 14 dwords are preloaded into SGPRs at wave launch)
__global__ __launch_bounds__(SAMP_BLOCK) void ctk_affine_rollout(const float* __restrict__ samples, const float* __restrict__ base,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ wperm, int rng_kind, int N_, int H_,
                                                                 int P_, uint32_t pmagic_, RolloutArgs a_in, typename Env<ENV>::K k, BestArgs best) {
    using E = Env<ENV>;
    constexpr int C = E::C, S = E::S;
    static_assert(PRED == CTK_PRED_ODE || ENV == CTK_ENV_CARTPOLE, "network predictors: CartPole instantiations only");
    extern __shared__ float lds[];
    RolloutArgs a = a_in;
    a.N = N_; a.H = H_; a.P = P_; a.p_magic = pmagic_;
    constexpr int TRAJ = (PRED == CTK_PRED_GRU) ? GRU_TRAJ : SAMP_TRAJ;
    constexpr int CHUNKS = SAMP_BLOCK / TRAJ;
    const int H = a.H, HC = H * C, ts = tile_stride(a.P), us = (HC + 1) | 1;   // a.P == H*C here: one sample per step and input
    float* tile = lds;
    float* ubuf = tile + TRAJ * ts;
    float* cin_s = ubuf + TRAJ * us;
    float* base_s = cin_s + SAMP_BLOCK;
    float* scale_s = base_s + HC;
    float* gru_ex = lds + affine_carve_floats(HC, TRAJ);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * TRAJ;
    const int n = row0 + lane;
    const bool valid = lane < TRAJ && n < a.N;

    load_tile_early<TRAJ, SAMP_BLOCK>(tile, samples, a, row0, 1.0f, rng_kind, [&] {
        for (int h = t; h < HC; h += SAMP_BLOCK) { base_s[h] = base[h]; scale_s[h] = scale[h]; }
    });
    __syncthreads();

    // inputs of the steps [hbeg, hend) of trajectory ptraj into ubuf + their input-only stage-cost terms (returned)
    auto prepare = [&](int ptraj, int hbeg, int hend) {
        const float* my = tile + ptraj * ts;
        auto input_at = [&](int h, int c) { return fminf(fmaxf(base_s[h * C + c] + my[h * C + c] * scale_s[h * C + c], a.lo[c]), a.hi[c]); };
        float cin = 0.0f;
        float uprev[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
            uprev[c] = (hbeg == 0 || hbeg >= H) ? (a.u_prev_dev ? a.u_prev_dev[c] : a.u_prev[c]) : input_at(hbeg - 1, c);
#pragma unroll 2
        for (int h = hbeg; h < hend; ++h) {
            float u[C];
#pragma unroll
            for (int c = 0; c < C; ++c) u[c] = input_at(h, c);
            cin += E::input_cost(k, u, uprev);
#pragma unroll
            for (int c = 0; c < C; ++c) { uprev[c] = u[c]; ubuf[ptraj * us + h * C + c] = u[c]; }
        }
        return cin;
    };
    // the plans, coalesced: Q[row0*H + i] for the block's contiguous span (elite refit / logging read it)
    auto write_plans = [&](int first, int stride) {
        const int total = min(TRAJ, a.N - row0) * HC;
        float* dst = a.Q_out + (size_t)row0 * HC;
        for (int i = first; i < total; i += stride) {
            const int r = HC >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i;
            dst[i] = ubuf[r * us + (i - r * HC)];
        }
    };

    if constexpr (PRED == CTK_PRED_ODE) {
        // Two phases, as in ctk_mppi_rollout: all four waves prepare the first S1 steps; then wave 0 runs the recurrence
        // over them while waves 1..3 prepare the rest; they meet when wave 0 reaches step S1, after which waves 1..3
        // write the plans out while wave 0 finishes the horizon.  Per-wave sums of the input-only terms: a wave's
        // phase-A part is carried into its phase-B sum (slot = wave).
        const int S1 = min(H, 16), Ha = (S1 + SAMP_WAVES - 1) / SAMP_WAVES;
        const float cin_a = prepare(lane, min(S1, wave * Ha), min(S1, wave * Ha + Ha));
        if (wave == 0) cin_s[lane] = cin_a;
        __syncthreads();
        const float* myu = ubuf + lane * us;
        float sx[S];
#pragma unroll
        for (int i = 0; i < S; ++i) sx[i] = a.s0[i];
        float csum = 0.0f, amax = 0.0f;
        float* traj = nullptr;
        if constexpr (WTRAJ) {
            if (a.traj_out) traj = a.traj_out + (size_t)n * (H + 1) * S;
        }
        const bool single = E::fast_ok(k);
        if (wave == 0) {
            if (single) recur_env_range<ENV, WTRAJ, true, true>(k, traj, valid, myu, 0, S1, sx, csum, amax);
        } else {
            const int Hb = (H - S1 + SAMP_WAVES - 2) / (SAMP_WAVES - 1);
            cin_s[wave * SAMP_TRAJ + lane] = cin_a + prepare(lane, min(H, S1 + (wave - 1) * Hb), min(H, S1 + (wave - 1) * Hb + Hb));
        }
        __syncthreads();
        if (wave == 0) {
            float J = 0.0f;
            if (single) {
                recur_env_range<ENV, WTRAJ, true, true>(k, traj, valid, myu, S1, H, sx, csum, amax);
                if constexpr (WTRAJ) {
                    if (valid && traj) store_state<S>(traj + (size_t)H * S, sx);
                }
                J = csum + E::terminal_cost(k, sx);
            }
            if (!single || __builtin_expect(__builtin_amdgcn_ballot_w64(E::out_of_range(amax)) != 0, 0)) {
#pragma unroll
                for (int i = 0; i < S; ++i) sx[i] = a.s0[i];
                csum = 0.0f;
                recur_env_range<ENV, WTRAJ, false, true>(k, traj, valid, myu, 0, H, sx, csum, amax);
                if constexpr (WTRAJ) {
                    if (valid && traj) store_state<S>(traj + (size_t)H * S, sx);
                }
                J = csum + E::terminal_cost(k, sx);
            }
            J += (cin_s[lane] + cin_s[SAMP_TRAJ + lane]) + (cin_s[2 * SAMP_TRAJ + lane] + cin_s[3 * SAMP_TRAJ + lane]);
            J *= a.inv_Hp1;
            if (valid) a.J[n] = J;
            if (best.ll) {
                const uint32_t key = valid ? f32_sortable(J) : 0xFFFFFFFFu;
                const uint32_t kmin = wave_min_u32(key);
                const uint32_t imin = wave_min_u32(key == kmin && valid ? (uint32_t)n : 0x7FFFFFFFu);   // ties: smallest index
                constexpr int BW = 2 + C;           // words per workgroup: key, index, the plan's first input [C]
                if (lane == 0) {
                    unsigned long long* r = best.ll + (size_t)blockIdx.x * BW;
                    ll_put(r, kmin, best.seq);
                    ll_put(r + 1, imin, best.seq);
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        ll_put(r + 2 + c, imin < (uint32_t)a.N ? __builtin_bit_cast(uint32_t, ubuf[(imin - (uint32_t)row0) * us + c]) : 0u, best.seq);
                }
                if (blockIdx.x == 0) {              // every block finishes unconditionally: the polls terminate (and are bounded)
                    uint32_t bk = 0xFFFFFFFFu, bi = 0x7FFFFFFFu, bu[C] = {};
                    bool expired = false;
                    for (int b = lane; b < (int)gridDim.x; b += 64) {
                        const uint32_t kb = ll_get(best.ll + (size_t)b * BW, best.seq, &expired), ib = ll_get(best.ll + (size_t)b * BW + 1, best.seq, &expired);
                        uint32_t ub[C];
#pragma unroll
                        for (int c = 0; c < C; ++c) ub[c] = ll_get(best.ll + (size_t)b * BW + 2 + c, best.seq, &expired);
                        if (kb < bk || (kb == bk && ib < bi)) {
                            bk = kb; bi = ib;
#pragma unroll
                            for (int c = 0; c < C; ++c) bu[c] = ub[c];
                        }
                    }
                    // a stale record may have won or lost wrongly: tell the host (ctk_api.hip:finish_step -> CTK_ERR_STATE)
                    if (expired) __hip_atomic_store(reinterpret_cast<uint32_t*>(best.u_host) + 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const uint32_t gk = wave_min_u32(bk);
                    const uint32_t gi = wave_min_u32(bk == gk ? bi : 0x7FFFFFFFu);
                    if (bk == gk && bi == gi) {     // exactly one lane holds the winner
                        best.idx_out[0] = (int)gi;
                        if constexpr (C == 1) publish_u(best.u_dev, best.u_host, __builtin_bit_cast(float, bu[0]), best.seq);
                        else {
                            float uo[C];
#pragma unroll
                            for (int c = 0; c < C; ++c) uo[c] = __builtin_bit_cast(float, bu[c]);
                            publish_u_vec(best.u_dev, best.u_host, uo, C, best.seq);
                        }
                    }
                }
            }
        } else {
            write_plans(t - 64, SAMP_BLOCK - 64);
        }
        return;
    }

    {   // network predictors: inputs of all H steps first; thread (trajectory t % TRAJ, chunk t / TRAJ)
        const int ptraj = t % TRAJ, chunk = t / TRAJ;
        const int Hc = (H + CHUNKS - 1) / CHUNKS;
        cin_s[chunk * TRAJ + ptraj] = prepare(ptraj, min(H, chunk * Hc), min(H, chunk * Hc + Hc));
    }
    __syncthreads();
    write_plans(t, SAMP_BLOCK);

    if constexpr (PRED == CTK_PRED_MLP) {
        const int tr = wave * CTK_MLP_TRAJ_PER_WAVE + (lane & 15);
        const float* myu = ubuf + tr * us;
        const MlpFwdT w = mlp_load_fwd_thin(wperm);
        float J = rollout_mlp<false, WTRAJ, false>(a, k, w, row0 + wave * CTK_MLP_TRAJ_PER_WAVE, [&](int h) { return myu[h]; });
        J += ((cin_s[tr] + cin_s[SAMP_TRAJ + tr]) + (cin_s[2 * SAMP_TRAJ + tr] + cin_s[3 * SAMP_TRAJ + tr])) * a.inv_Hp1;
        if (lane < 16 && row0 + tr < a.N) a.J[row0 + tr] = J;
    } else if constexpr (PRED == CTK_PRED_GRU) {
        // the four waves share the workgroup's 16 trajectories; wave 0 ends with J (lane = trajectory for lanes 0..15)
        const float* myu = ubuf + (lane & 15) * us;
        float J = rollout_gru<WTRAJ, false>(a, k, wperm, wperm + GRU_TABLE_FLOATS, gru_ex, row0, [&](int h) { return myu[h]; });
        if (wave == 0 && valid) {
            float cs = 0.0f;
#pragma unroll
            for (int cnk = 0; cnk < CHUNKS; ++cnk) cs += cin_s[cnk * TRAJ + lane];
            a.J[n] = J + cs * a.inv_Hp1;
        }
    }
}

// predictor.update(s, Q0) for the GRU (optimizer_mppi.py:195-197): one workgroup of four waves (the split
// step of ctk_gru.h); all 16 MFMA columns carry the same (s, u); wave 0's lanes of column 0 write the new
// hidden state back in place (table | hidden, see ctk_api.hip:permute_gru_weights).
__global__ __launch_bounds__(GRU_BLOCK) void ctk_gru_advance(float s0, float s1, float s2, float s3, const float* __restrict__ u_dev,
                                                             float u_val, float* __restrict__ wperm) {
    __shared__ __attribute__((aligned(16))) float ex[GRU_EX_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    float* hidden = wperm + GRU_TABLE_FLOATS;
    const GruW w = gru_load_weights(wperm, wave, lane);
    GruState st = gru_load_state(hidden, g);
    const float sv = g == 0 ? s0 : (g == 1 ? s1 : (g == 2 ? s2 : s3));
    const float u = u_dev ? *u_dev : u_val;
    (void)gru_step(w, st, sv, u, g, ex, wave, lane);   // its barriers order every wave's loads of `hidden` before the stores below
    if (wave == 0 && c == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hidden[16 * m + 4 * g + r] = st.h1[m][r];
                hidden[32 + 16 * m + 4 * g + r] = st.h2[m][r];
            }
    }
}

hipError_t ctk_launch_gru_advance(hipStream_t st, const float* s, const float* u_dev, float u_val, float* wperm) {
    hipLaunchKernelGGL(ctk_gru_advance, dim3(1), dim3(GRU_BLOCK), 0, st, s[0], s[1], s[2], s[3], u_dev, u_val, wperm);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// rank-by-counting selection: rank(i) = #{ j : (J_j, j) < (J_i, i) }, a permutation under the total
// order (cost, index), so idx_out[rank] = i for rank < K is a race-free scatter and idx_out comes
// out sorted ascending — what tf.argsort gives, ties fixed by index.
// Work N^2 comparisons, spread over N/16 blocks x 16 waves (256 blocks at N = 4096: the whole
// chip): a block owns 16 rows; wave w scans the j-slice [w*L, (w+1)*L); lane (r = lane & 15,
// q = lane >> 4) compares row r against the j = 4t + q of the slice, the slice staged in LDS as
// sortable 32-bit keys (coalesced load; 4 distinct addresses per LDS read, broadcast within 16 lanes).
// ---------------------------------------------------------------------------------------------
constexpr int SEL_WAVES = 16;
constexpr int SEL_ROWS = 16;
constexpr int SEL_CHUNK = 1024;   // keys per wave and pass (LDS: 16 waves x 1024 x 4 B = 64 KiB)


// J_i = J[i * ldj] (ldj = 1 for a plain cost vector; candidate records of the sharded path are strided)
__global__ __launch_bounds__(64 * SEL_WAVES) void ctk_select_topk(const float* __restrict__ J, int ldj, int N, int K,
                                                                  int* __restrict__ idx_out) {
    __shared__ uint32_t key_s[SEL_WAVES][SEL_CHUNK];
    __shared__ int cnt_s[SEL_WAVES][SEL_ROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int i = blockIdx.x * SEL_ROWS + r;
    const uint32_t ki = f32_sortable((i < N) ? J[(size_t)i * ldj] : INFINITY);
    const int L = (N + SEL_WAVES - 1) / SEL_WAVES;
    const int j0 = wave * L, j1 = min(N, j0 + L);
    int cnt = 0;
    for (int jb = j0; jb < j1; jb += SEL_CHUNK) {
        const int n = min(SEL_CHUNK, j1 - jb);
        for (int t = lane; t < n; t += 64) key_s[wave][t] = f32_sortable(J[(size_t)(jb + t) * ldj]);
        // wave-private slice: LDS ops of one wave complete in order, no barrier needed
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        for (int t = q; t < n; t += 4) {
            const uint32_t kj = key_s[wave][t];
            const int j = jb + t;
            cnt += (kj < ki) | ((kj == ki) & (j < i));
        }
    }
    cnt += __shfl_xor(cnt, 16, 64);
    cnt += __shfl_xor(cnt, 32, 64);
    if (lane < SEL_ROWS) cnt_s[wave][lane] = cnt;
    __syncthreads();
    if (wave == 0 && lane < SEL_ROWS && i < N) {
        int rk = 0;
#pragma unroll
        for (int w = 0; w < SEL_WAVES; ++w) rk += cnt_s[w][lane];
        if (rk < K) idx_out[rk] = i;
    }
}

// elite refit: one block per horizon step h; mu = mean_k Q[idx[k],h]; sd = population std.
// Q row r = Q[r * ldq .. + H) (ldq = H for the plan matrix; 2+H with Q pointing at column 2 of candidate records)
__global__ __launch_bounds__(256) void ctk_cem_refit(const float* __restrict__ Q, int ldq, const int* __restrict__ idx, int K, int H,
                                                     float* __restrict__ mu, float* __restrict__ sd) {
    __shared__ float red[4];
    const int h = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    float s = 0.0f;
    for (int kk = t; kk < K; kk += 256) s += Q[(size_t)idx[kk] * ldq + h];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)K;
    __syncthreads();
    float v = 0.0f;
    for (int kk = t; kk < K; kk += 256) {
        const float d = Q[(size_t)idx[kk] * ldq + h] - mean;
        v += d * d;
    }
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (t == 0) {
        mu[h] = mean;
        sd[h] = sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K);   // tf.math.reduce_std: ddof = 0
    }
}

// optimizer_cem_tf.py:99-102: clip std, shift both by one step, refill the tail; u = elite[0,0]
// std_max / u_from_mu: optimizer_cem_naive_grad_tf.py:101-104 clips the stdev to [min, 10] and applies the
// MEAN's first input; optimizer_cem_tf.py:99-101 clips to [min, 1e8] and applies the best elite's
__global__ __launch_bounds__(256) void ctk_cem_finish(const float* __restrict__ Q, int ldq, const int* __restrict__ idx, int H,
                                                      float* __restrict__ mu, float* __restrict__ sd, float std_min,
                                                      float init_std, float mid, float* __restrict__ u_dev,
                                                      float* __restrict__ u_host, uint32_t seq, float std_max, int u_from_mu) {
    extern __shared__ float lds[];
    float* m_s = lds;
    float* s_s = lds + H;
    const int t = threadIdx.x;
    for (int h = t; h < H; h += 256) {
        m_s[h] = mu[h];
        s_s[h] = fminf(fmaxf(sd[h], std_min), std_max);
    }
    __syncthreads();
    for (int h = t; h < H; h += 256) {
        mu[h] = (h + 1 < H) ? m_s[h + 1] : mid;
        sd[h] = (h + 1 < H) ? s_s[h + 1] : init_std;
    }
    __syncthreads();
    if (t == 0) publish_u(u_dev, u_host, u_from_mu ? m_s[0] : Q[(size_t)idx[0] * ldq], seq);
}

__global__ void ctk_pick_best_first(const float* __restrict__ Q, int ldq, const int* __restrict__ idx, int H,
                                    float* __restrict__ u_dev, float* __restrict__ u_host, uint32_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) publish_u(u_dev, u_host, Q[(size_t)idx[0] * ldq], seq);
}

// plans of a CEM-with-gradient iteration, without rolling them out (the descent kernel does that)
__global__ __launch_bounds__(256) void ctk_sample_plans(RolloutArgs a, const float* __restrict__ samples,
                                                        const float* __restrict__ mu, const float* __restrict__ sd,
                                                        float* __restrict__ Q) {
    const int H = a.H * a.C;                  // a row of [N,H,C] is H*C contiguous floats; h below is the flat (step, input) column
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.N * H) return;
    const int n = gid / H, h = gid - n * H;   // gid ranges over N*H: beyond the exactness range of the p_magic trick
    const int c = h % a.C;
    float e;
    if (samples != nullptr) {
        e = samples[gid];
    } else {
        float d4[4];
        draw4(a, (uint32_t)(a.global_row0 + n), (uint32_t)(h >> 2), 0, d4);
        e = d4[h & 3];
    }
    Q[gid] = fminf(fmaxf(mu[h] + e * sd[h], a.lo[c]), a.hi[c]);
}

hipError_t ctk_launch_sample_plans(hipStream_t st, const RolloutArgs& a, const float* samples, const float* mu, const float* sd, float* Q) {
    const int total = a.N * a.H * a.C;
    hipLaunchKernelGGL(ctk_sample_plans, dim3((total + 255) / 256), dim3(256), 0, st, a, samples, mu, sd, Q);
    return hipGetLastError();
}

// population of one cem-grad-bharadhwaj iteration (optimizer_cem_grad_bharadhwaj_tf.py:94-97):
// rows [0,K) = the elites (first iteration: fresh samples, :158; afterwards the previous iteration's best K of
// the refined population, :118-119), rows [K,N) = fresh samples mu + std*eps; everything clipped.
__global__ __launch_bounds__(256) void ctk_cem_build_population(RolloutArgs a, int K, int first, const float* __restrict__ Q_prev,
                                                                const int* __restrict__ idx, const float* __restrict__ eps_elite,
                                                                const float* __restrict__ eps_rest, const float* __restrict__ mu,
                                                                const float* __restrict__ sd, float* __restrict__ Q) {
    const int H = a.H * a.C;                  // flat (step, input) columns of a row, as in ctk_sample_plans
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.N * H) return;
    const int n = gid / H, h = gid - n * H;   // gid ranges over N*H: beyond the exactness range of the p_magic trick
    const int c = h % a.C;
    float q;
    if (n < K && !first) {
        q = Q_prev[(size_t)idx[n] * H + h];
    } else {
        const float* src = n < K ? eps_elite : eps_rest;
        float e;
        if (src != nullptr) {
            e = src[(size_t)(n < K ? n : n - K) * H + h];
        } else {   // Philox: stream_id separates the iterations, elites of the first iteration use the same row space
            float d4[4];
            draw4(a, (uint32_t)(a.global_row0 + n), (uint32_t)(h >> 2), 0, d4);
            e = d4[h & 3];
        }
        q = mu[h] + sd[h] * e;                                   // :122-128
    }
    Q[gid] = fminf(fmaxf(q, a.lo[c]), a.hi[c]);                        // :96
}

hipError_t ctk_launch_cem_build_population(hipStream_t st, const RolloutArgs& a, int K, int first, const float* Q_prev, const int* idx,
                                           const float* eps_elite, const float* eps_rest, const float* mu, const float* sd, float* Q) {
    const int total = a.N * a.H * a.C;
    hipLaunchKernelGGL(ctk_cem_build_population, dim3((total + 255) / 256), dim3(256), 0, st, a, K, first, Q_prev, idx, eps_elite, eps_rest,
                       mu, sd, Q);
    return hipGetLastError();
}

// sharded selection: this shard's best K plans as records {J, global index (int bits), Q[H]},
// sorted ascending by (J, index) — what is all-gathered; the global top-K is the top-K of their union.
__global__ __launch_bounds__(256) void ctk_pack_candidates(const float* __restrict__ J, const float* __restrict__ Q,
                                                           const int* __restrict__ idx, int K, int H, int global_offset,
                                                           float* __restrict__ cand) {
    const int rs = 2 + H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K * rs; i += gridDim.x * blockDim.x) {
        const int kk = i / rs, f = i - kk * rs, src = idx[kk];
        float v;
        if (f == 0) v = J[src];
        else if (f == 1) v = __builtin_bit_cast(float, global_offset + src);
        else v = Q[(size_t)src * H + (f - 2)];
        cand[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// step log: four contiguous copies in one launch; blockIdx.y selects the job, float4 body + scalar tail
struct CopyJobs { CopyJob j[4]; };
__global__ __launch_bounds__(256) void ctk_log_copy(CopyJobs jobs) {
    const CopyJob job = jobs.j[blockIdx.y];
    const unsigned n = job.n;
    const unsigned tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(job.src) | reinterpret_cast<uintptr_t>(job.dst)) & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(job.src);
        float4* d4 = reinterpret_cast<float4*>(job.dst);
        const unsigned n4 = n >> 2;
        for (unsigned i = tid; i < n4; i += stride) d4[i] = s4[i];
        for (unsigned i = (n4 << 2) + tid; i < n; i += stride) job.dst[i] = job.src[i];
    } else {
        for (unsigned i = tid; i < n; i += stride) job.dst[i] = job.src[i];
    }
}

hipError_t ctk_launch_copy4(hipStream_t st, const CopyJob (&jobs)[4]) {
    CopyJobs cj;
    unsigned nmax = 0;
    for (int i = 0; i < 4; ++i) { cj.j[i] = jobs[i]; nmax = jobs[i].n > nmax ? jobs[i].n : nmax; }
    if (nmax == 0) return hipSuccess;
    unsigned blocks = (nmax / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(ctk_log_copy, dim3(blocks, 4), dim3(256), 0, st, cj);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// template arguments <environment, predictor, materialise>
const char* ctk_affine_rollout_name(int pred, bool log) {
    if (pred == CTK_PRED_ODE) return log ? "ctk_affine_rollout<0, 0, true>" : "ctk_affine_rollout<0, 0, false>";
    if (pred == CTK_PRED_GRU) return log ? "ctk_affine_rollout<0, 2, true>" : "ctk_affine_rollout<0, 2, false>";
    return log ? "ctk_affine_rollout<0, 1, true>" : "ctk_affine_rollout<0, 1, false>";
}
const char* ctk_affine_rollout_env_name(int env, bool log) {
    return ctk_kernel_name("ctk_affine_rollout<%d, 0, %4$s>", env, 0, 0, log ? "true" : "false");
}

hipError_t ctk_launch_affine_rollout(hipStream_t st, int pred, const RolloutArgs& a, const EnvK& k, const float* samples,
                                     int rng_kind, const float* base, const float* scale, const float* wperm, bool log,
                                     hipEvent_t e0, hipEvent_t e1, const AffineBest* bst) {
    BestArgs bargs{nullptr, 0u, nullptr, nullptr, nullptr};
    if (bst && bst->ll && pred == CTK_PRED_ODE) bargs = BestArgs{bst->ll, bst->seq, bst->u_dev, bst->u_host, bst->idx_out};
    const int tr = affine_traj(pred);
    const dim3 grid((a.N + tr - 1) / tr), block(SAMP_BLOCK);
    const size_t lds = ctk_affine_rollout_lds(a.H, pred);
    if (pred == CTK_PRED_ODE) {
        if (log) CTK_LAUNCH((ctk_affine_rollout<CTK_ENV_CARTPOLE, CTK_PRED_ODE, true>), grid, block, lds, st, e0, e1, samples, base, scale, wperm, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
        else CTK_LAUNCH((ctk_affine_rollout<CTK_ENV_CARTPOLE, CTK_PRED_ODE, false>), grid, block, lds, st, e0, e1, samples, base, scale, wperm, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
    } else if (pred == CTK_PRED_MLP) {
        if (log) CTK_LAUNCH((ctk_affine_rollout<CTK_ENV_CARTPOLE, CTK_PRED_MLP, true>), grid, block, lds, st, e0, e1, samples, base, scale, wperm, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
        else CTK_LAUNCH((ctk_affine_rollout<CTK_ENV_CARTPOLE, CTK_PRED_MLP, false>), grid, block, lds, st, e0, e1, samples, base, scale, wperm, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
    } else {
        if (log) CTK_LAUNCH((ctk_affine_rollout<CTK_ENV_CARTPOLE, CTK_PRED_GRU, true>), grid, block, lds, st, e0, e1, samples, base, scale, wperm, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
        else CTK_LAUNCH((ctk_affine_rollout<CTK_ENV_CARTPOLE, CTK_PRED_GRU, false>), grid, block, lds, st, e0, e1, samples, base, scale, wperm, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
    }
    return hipGetLastError();
}

// The same 4-wave kernel for any environment's analytic predictor: a.H steps, a.C inputs (a.lo / a.hi per input); samples
// [N, H, C] or nullptr (Philox over the flat columns); base / scale [H*C]; bst as above with 2 + C words per workgroup.
hipError_t ctk_launch_affine_rollout_env(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a_in,
                                         const float* samples, int rng_kind, const float* base, const float* scale, bool log,
                                         hipEvent_t e0, hipEvent_t e1, const AffineBest* bst) {
    BestArgs bargs{nullptr, 0u, nullptr, nullptr, nullptr};
    if (bst && bst->ll) bargs = BestArgs{bst->ll, bst->seq, bst->u_dev, bst->u_host, bst->idx_out};
    const dim3 grid((a_in.N + SAMP_TRAJ - 1) / SAMP_TRAJ), block(SAMP_BLOCK);
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        RolloutArgs a = a_in;
        const int HC = a.H * E::C;
        a.C = E::C; a.P = HC;
        a.p_magic = HC >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)HC - 1) / (uint64_t)HC) : 0u;
        const typename E::K k = E::derive(params, dt, isteps);
        const size_t lds = (size_t)affine_carve_floats(HC, SAMP_TRAJ) * sizeof(float);
        if (log) CTK_LAUNCH((ctk_affine_rollout<EV, CTK_PRED_ODE, true>), grid, block, lds, st, e0, e1, samples, base, scale, (const float*)nullptr, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
        else CTK_LAUNCH((ctk_affine_rollout<EV, CTK_PRED_ODE, false>), grid, block, lds, st, e0, e1, samples, base, scale, (const float*)nullptr, rng_kind, a.N, a.H, a.P, a.p_magic, a, k, bargs);
    });
    return hipGetLastError();
}
size_t ctk_affine_rollout_env_lds(int env, int H) {
    int C = 1;
    CTK_FOR_ENV(env, EV, { C = Env<EV>::C; });
    return (size_t)affine_carve_floats(H * C, SAMP_TRAJ) * sizeof(float);
}

int ctk_affine_rollout_blocks(int pred, int N) { const int tr = affine_traj(pred); return (N + tr - 1) / tr; }

size_t ctk_affine_rollout_lds(int H, int pred) {
    return (size_t)(affine_carve_floats(H, affine_traj(pred)) + (pred == CTK_PRED_GRU ? GRU_EX_FLOATS : 0)) * sizeof(float);
}

hipError_t ctk_launch_select_topk(hipStream_t st, const float* J, int N, int K, int* idx_out, int ldj) {
    hipLaunchKernelGGL(ctk_select_topk, dim3((N + SEL_ROWS - 1) / SEL_ROWS), dim3(64 * SEL_WAVES), 0, st, J, ldj, N, K, idx_out);
    return hipGetLastError();
}

hipError_t ctk_launch_cem_refit(hipStream_t st, const float* Q, const int* idx, int K, int H, float* mu, float* sd, int ldq) {
    hipLaunchKernelGGL(ctk_cem_refit, dim3(H), dim3(256), 0, st, Q, ldq, idx, K, H, mu, sd);
    return hipGetLastError();
}

hipError_t ctk_launch_pack_candidates(hipStream_t st, const float* J, const float* Q, const int* idx, int K, int H, int global_offset,
                                      float* cand) {
    const int total = K * (2 + H);
    hipLaunchKernelGGL(ctk_pack_candidates, dim3((total + 255) / 256 > 64 ? 64 : (total + 255) / 256), dim3(256), 0, st, J, Q, idx, K, H,
                       global_offset, cand);
    return hipGetLastError();
}

hipError_t ctk_launch_cem_finish(hipStream_t st, const float* Q, const int* idx, int H, float* mu, float* sd, float std_min,
                                 float init_std, float mid, float* u_dev, float* u_host, uint32_t seq, int ldq, float std_max,
                                 int u_from_mu) {
    hipLaunchKernelGGL(ctk_cem_finish, dim3(1), dim3(256), 2 * H * sizeof(float), st, Q, ldq, idx, H, mu, sd, std_min, init_std, mid,
                       u_dev, u_host, seq, std_max, u_from_mu);
    return hipGetLastError();
}

hipError_t ctk_launch_pick_best_first(hipStream_t st, const float* Q, const int* idx, int H, float* u_dev, float* u_host, uint32_t seq,
                                      int ldq) {
    hipLaunchKernelGGL(ctk_pick_best_first, dim3(1), dim3(64), 0, st, Q, ldq, idx, H, u_dev, u_host, seq);
    return hipGetLastError();
}
