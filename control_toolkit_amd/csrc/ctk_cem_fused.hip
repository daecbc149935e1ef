// ctk_cem_fused.hip — one CEM step (all outer iterations) in ONE launch, analytic predictor of any environment (ctk_env.h: C control
// inputs -> H*C columns per plan, below "H" where a column count is meant), <= CTK_CEM_FUSED_MAX_BLOCKS
// workgroups (all co-resident: one per CU).  Replaces, per outer iteration, the three launches rollout -> ctk_select_topk ->
// ctk_cem_refit and, after the loop, ctk_cem_finish (optimizer_cem_tf.py:61-80,83-111): SURVEY 8e's all-reduce form of the
// elite refit applied INSIDE one GPU, with the {value, tag} word hand-off of ctk_mppi.hip between workgroups.
//
// Per outer iteration, every workgroup (64 rollouts, 256 threads):
//   1. rollout    Q = clip(mu + eps * std) (:64-66), costs J of its 64 rows (anatomy of ctk_affine_rollout<ODE>; the NEXT
//                 iteration's sample tile is fetched by waves 1..3 while wave 0 runs the recurrence — it does not depend on mu/std);
//   2. hop 1      publishes its 64 costs as words {sortable key, tag}; polls all N words into LDS;
//   3. selection  finds the K-th smallest (key, index) of all N REDUNDANTLY (4-pass radix select on an LDS histogram; ties
//                 broken by index, the total order of ctk_select_topk / tf.argsort) -> which of ITS rows are elite (:73-75);
//   4. hop 2      publishes {n_b, m_b[H], M2_b[H]} of its elite rows: m_b = mean of d = q - mu (mu: the mean the samples were drawn around),
//                 M2_b = centred sum of squares — formed in one pass in double, published as floats; polls every workgroup's record;
//   5. refit      ONE pass over the records in a fixed order, in double: A = sum n_b m_b, B = sum (M2_b + n_b m_b^2) -> mean = mu + A/K,
//                 M2 = B - A^2/K (the 53 bits absorb the cancellation; the shift by mu keeps it small), identical bits in every
//                 workgroup -> mu, population std (:77-78) in LDS for the next iteration.  (First form: Chan's pairwise update in
//                 float — two passes, three barriers: 2.0 us per iteration against 1.0.  Raw sums as doubles, two words each: the
//                 records double and their gather eats the gain, 3.3 against 1.9 us.)
// After the loop: the workgroup that owns the cheapest row publishes u = elite[0,0] (:101); workgroup 0 clips the std, shifts
// both by one step and refills the tail (:99-102) into the handle's mu / std.
// Only the last iteration's plans, costs (and trajectories, WTRAJ) reach memory; BEST_IDX is materialised on demand by
// ctk_select_topk from those costs (ctk_api.hip: locate_buffer).
// Every wait is bounded by a wall clock; on expiry the error word behind {u, seq} is raised (ctk_api.hip:finish_step ->
// CTK_ERR_STATE) — never a silently wrong result.
#include "ctk_rollout.h"
#include "ctk_env.h"
#include "ctk_launch.h"

#ifdef CTK_CEM_STAMPS   // diagnostic build (tools/diag_cem_fused.hip); never compiled into libctk_hip.so
#define CSTAMP(i)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0 && a.stamps) a.stamps[(blockIdx.x * 8 + it) * 16 + (i)] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define CSTAMP(i)
#endif

// 64 rollouts per workgroup: wave 0 runs their recurrence (one per lane).  EIGHT waves per workgroup (CF_WAVES; launch_bounds(512)):
// everything between two recurrences (input preparation, hand-off polls, selection, moments, merge) is spread over 512 threads, two
// waves per SIMD — with four waves a SIMD holds ONE wave, every instruction of these phases issues at 4+ cycles and every LDS latency is exposed (measured: selection
// 4.0 us, merge 1.7 us per outer iteration at cfg3).
constexpr int CF_TRAJ = 64, CF_WAVES = 8, CF_BLOCK = CF_TRAJ * CF_WAVES;
constexpr int CF_CHUNK = 8;   // keys per thread and chunk of the counting loop
constexpr int CF_LLW = 8;   // hand-off words in flight per thread

struct CemFusedK {
    int its, K, nblk;
    unsigned long long per_it;      // samples per outer iteration (N * H)
    unsigned long long* llJ;        // [N]              {sortable key of J_n, tag}
    unsigned long long* llS;        // [nblk][1 + 2H]   {n_b | m_b[H] | M2_b[H], tag}
    uint32_t tag0;                  // tag of iteration it = tag0 + it (host: consecutive across launches, never 0)
    float std_min, std_max, init_std;
    float mid[CTK_MAX_INPUTS];      // per input: the tail refill of the mean (:99-102)
    float* mu; float* sd;           // [H*C] device, in / out
    float* u_dev; float* u_host; int* idx_out; uint32_t seq;
    unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz) per hop
};

// LDS carve (4-byte words)
struct CemCarve {
    int tile0, tile1, ubuf, cin, mu, sd, keys, recs, part, hist, misc, total;
};
__host__ __device__ inline CemCarve cem_carve(int N, int H, int nblk) {
    const int ts = tile_stride(H), us = (H + 1) | 1, rs = 1 + 2 * H;
    CemCarve c;
    int o = 0;
    c.tile0 = o; o += CF_TRAJ * ts;
    c.tile1 = o; o += CF_TRAJ * ts;
    c.ubuf = o; o += CF_TRAJ * us;
    c.cin = o; o += CF_BLOCK;
    c.mu = o; o += H;
    c.sd = o; o += H;
    c.keys = o; o += (N + CF_BLOCK * CF_CHUNK - 1) / (CF_BLOCK * CF_CHUNK) * (CF_BLOCK * CF_CHUNK);   // padded: the counting loop reads whole chunks
    c.recs = o; o += nblk * rs;
    o = (o + 1) & ~1;
    c.part = o; o += 4 * CF_BLOCK;          // [SEG][H] {S1, S2} doubles of the segmented refit (SEG * H <= CF_BLOCK)
    c.hist = o; o += 256;
    c.misc = o; o += 80 + 2 * CF_WAVES;   // [0..7] selection scalars | [8] n_b | [16..79] elite rows | [80..) two per-wave reduction rows
    c.total = (o + 3) & ~3;
    return c;
}

CTK_DEV unsigned long long ll_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CTK_DEV void ll_st(unsigned long long* p, uint32_t payload, uint32_t tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Polls words [0, n) of `src` (thread t takes t, t + nthreads, ...) until each carries `tag`; sink(i, payload).
// All of a thread's pending words are re-polled TOGETHER (one memory round trip per round, not one per word: the words of a
// batch come from CF_LLW different workgroups, and a stale first read of each would otherwise cost its own round trip).
template <class Sink>
CTK_DEV bool ll_gather(const unsigned long long* src, int n, uint32_t tag, int t, int nthreads, unsigned long long ticks, Sink&& sink) {
    bool expired = false;
    const unsigned long long t0 = wall_clock64();
    for (int i0 = t; i0 < n; i0 += nthreads * CF_LLW) {
        unsigned long long w[CF_LLW];
#pragma unroll
        for (int j = 0; j < CF_LLW; ++j) {
            const int i = i0 + j * nthreads;
            w[j] = i < n ? ll_ld(src + i) : ((unsigned long long)tag << 32);
        }
        for (;;) {
            bool pending = false;
#pragma unroll
            for (int j = 0; j < CF_LLW; ++j) pending |= (uint32_t)(w[j] >> 32) != tag;
            if (!pending) break;
            if (wall_clock64() - t0 > ticks) { expired = true; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int j = 0; j < CF_LLW; ++j) {
                const int i = i0 + j * nthreads;
                if ((uint32_t)(w[j] >> 32) != tag) w[j] = ll_ld(src + i);
            }
        }
#pragma unroll
        for (int j = 0; j < CF_LLW; ++j) {
            const int i = i0 + j * nthreads;
            if (i < n) sink(i, (uint32_t)w[j]);
        }
    }
    return expired;
}

// sample tile of one iteration into LDS by the threads tsub in [0, nsub) (a subset of the workgroup): tile[r*ts + c] = eps[row0+r][c];
// rows beyond N and the pad columns read as zeros.  No barrier inside.
CTK_DEV void cem_fetch_tile(float* tile, const float* __restrict__ samples, const RolloutArgs& a, int row0, int tsub, int nsub) {
    const int P = a.P, ts = tile_stride(P);
    const int rows = max(0, min(CF_TRAJ, a.N - row0));
    if (rows < CF_TRAJ) {
        for (int i = tsub + rows * ts; i < CF_TRAJ * ts; i += nsub) tile[i] = 0.0f;
    }
    for (int r = tsub; r < rows; r += nsub)
        for (int c = P; c < ts; ++c) tile[r * ts + c] = 0.0f;
    if (samples != nullptr) {
        const float* src = samples + (size_t)row0 * P;
        const int total = rows * P;
        int done = 0;
        if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const float4* src4 = reinterpret_cast<const float4*>(src);
            const int n4 = total >> 2;
            for (int b0 = 0; b0 < n4; b0 += 4 * nsub) {
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i4 = b0 + j * nsub + tsub;
                    if (i4 < n4) v[j] = src4[i4];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i4 = b0 + j * nsub + tsub;
                    if (i4 < n4) {
                        const int flat = i4 << 2;
                        int r = P >= 2 ? (int)__umulhi((uint32_t)flat, a.p_magic) : flat, c = flat - r * P;
                        const float e4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            tile[r * ts + c] = e4[q];
                            if (++c == P) { c = 0; ++r; }
                        }
                    }
                }
            }
            done = n4 << 2;
        }
        for (int i = done + tsub; i < total; i += nsub) {
            const int r = P >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i;
            tile[r * ts + (i - r * P)] = src[i];
        }
    } else {
        // on-device Philox, addressed by (global row, column block, call, stream = iteration): the draws of ctk_affine_rollout
        const int tpr = nsub / CF_TRAJ;              // threads per row (nsub is a multiple of 64)
        const int r = tsub % CF_TRAJ, cb0 = tsub / CF_TRAJ;
        if (r < rows) {
            const uint32_t grow = (uint32_t)(a.global_row0 + row0 + r);
            for (int cb = cb0; cb * 4 < P; cb += tpr) {
                float d[4];
                draw4(a, grow, (uint32_t)cb, 0, d);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cb * 4 + j < P) tile[r * ts + cb * 4 + j] = d[j];
            }
        }
    }
}

template <int ENV, bool WTRAJ>
__global__ __launch_bounds__(CF_BLOCK) void ctk_cem_fused(const float* __restrict__ samples, RolloutArgs a_in, typename Env<ENV>::K k, CemFusedK cf) {
    using E = Env<ENV>;
    constexpr int C = E::C, S = E::S;
    extern __shared__ float lds[];
    RolloutArgs a = a_in;
    // Hs steps; H = Hs*C flat (step, input) columns of a plan = sample columns of a row (a.P): one sample per step and input
    const int N = a.N, Hs = a.H, H = Hs * C, ts = tile_stride(a.P), us = (H + 1) | 1, rs = 1 + 2 * H;
    const CemCarve cv = cem_carve(N, H, cf.nblk);
    float* tiles[2] = {lds + cv.tile0, lds + cv.tile1};
    float* ubuf = lds + cv.ubuf;
    float* cin_s = lds + cv.cin;
    float* mu_s = lds + cv.mu;
    float* sd_s = lds + cv.sd;
    uint32_t* keys = reinterpret_cast<uint32_t*>(lds + cv.keys);
    float* recs = lds + cv.recs;
    double* part = reinterpret_cast<double*>(lds + cv.part);
    int* hist = reinterpret_cast<int*>(lds + cv.hist);
    int* sel = reinterpret_cast<int*>(lds + cv.misc);            // [0] prefix (as bits) [1] want
    int* nb_s = reinterpret_cast<int*>(lds + cv.misc) + 8;
    int* erow = reinterpret_cast<int*>(lds + cv.misc) + 16;       // [64] this workgroup's elite rows, ascending
    uint32_t* red = reinterpret_cast<uint32_t*>(lds + cv.misc) + 80;   // [2][CF_WAVES]
    auto red_min = [&](int row) { uint32_t v = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < CF_WAVES; ++w) v = min(v, red[row * CF_WAVES + w]);
        return v; };
    auto red_sum = [&](int row) { uint32_t v = 0u;
#pragma unroll
        for (int w = 0; w < CF_WAVES; ++w) v += red[row * CF_WAVES + w];
        return v; };
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * CF_TRAJ;
    const int n = row0 + lane;
    const bool valid = n < N;                                     // wave 0: lane = row of the workgroup
    float up0[C];
#pragma unroll
    for (int c = 0; c < C; ++c) up0[c] = a.u_prev_dev ? a.u_prev_dev[c] : a.u_prev[c];
    bool expired = false;

    for (int h = t; h < H; h += CF_BLOCK) { mu_s[h] = cf.mu[h]; sd_s[h] = cf.sd[h]; }
    a.stream_id = 0;
    cem_fetch_tile(tiles[0], samples, a, row0, t, CF_BLOCK);
    __syncthreads();

    for (int it = 0; it < cf.its; ++it) {
        const bool last_it = it + 1 == cf.its;
        const uint32_t tag = cf.tag0 + (uint32_t)it;
        float* tile = tiles[it & 1];
        CSTAMP(0);
        // ---- 1. rollout ---------------------------------------------------------------------------------------------
        auto prepare = [&](int ptraj, int hbeg, int hend) {
            const float* my = tile + ptraj * ts;
            auto input_at = [&](int h, int c) { return fminf(fmaxf(mu_s[h * C + c] + my[h * C + c] * sd_s[h * C + c], a.lo[c]), a.hi[c]); };   // :64-66
            float cin = 0.0f;
            float uprev[C];
#pragma unroll
            for (int c = 0; c < C; ++c) uprev[c] = (hbeg == 0 || hbeg >= Hs) ? up0[c] : input_at(hbeg - 1, c);
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
        const int S1 = min(Hs, 16), Ha = (S1 + CF_WAVES - 1) / CF_WAVES;
        const float cin_a = prepare(lane, min(S1, wave * Ha), min(S1, wave * Ha + Ha));
        if (wave == 0) cin_s[lane] = cin_a;
        __syncthreads();
        CSTAMP(1);
        const float* myu = ubuf + lane * us;
        float sx[S];
#pragma unroll
        for (int i = 0; i < S; ++i) sx[i] = a.s0[i];
        float csum = 0.0f, amax = 0.0f;
        float* traj = nullptr;
        if constexpr (WTRAJ) {
            if (a.traj_out && last_it) traj = a.traj_out + (size_t)n * (Hs + 1) * S;
        }
        const bool single = E::fast_ok(k);
        if (wave == 0) {
            if (single) recur_env_range<ENV, WTRAJ, true, true>(k, traj, valid, myu, 0, S1, sx, csum, amax);
        } else {
            const int Hb = (Hs - S1 + CF_WAVES - 2) / (CF_WAVES - 1);
            cin_s[wave * CF_TRAJ + lane] = cin_a + prepare(lane, min(Hs, S1 + (wave - 1) * Hb), min(Hs, S1 + (wave - 1) * Hb + Hb));
        }
        __syncthreads();
        if (wave == 0) {
            float J = 0.0f;
            if (single) {
                recur_env_range<ENV, WTRAJ, true, true>(k, traj, valid, myu, S1, Hs, sx, csum, amax);
                if constexpr (WTRAJ) {
                    if (valid && traj) store_state<S>(traj + (size_t)Hs * S, sx);
                }
                J = csum + E::terminal_cost(k, sx);
            }
            if (!single || __builtin_expect(__builtin_amdgcn_ballot_w64(E::out_of_range(amax)) != 0, 0)) {
#pragma unroll
                for (int i = 0; i < S; ++i) sx[i] = a.s0[i];
                csum = 0.0f;
                recur_env_range<ENV, WTRAJ, false, true>(k, traj, valid, myu, 0, Hs, sx, csum, amax);
                if constexpr (WTRAJ) {
                    if (valid && traj) store_state<S>(traj + (size_t)Hs * S, sx);
                }
                J = csum + E::terminal_cost(k, sx);
            }
            float cin = 0.0f;
#pragma unroll
            for (int w = 0; w < CF_WAVES; ++w) cin += cin_s[w * CF_TRAJ + lane];
            J += cin;
            J *= a.inv_Hp1;
            CSTAMP(2);
            if (valid) {
                ll_st(cf.llJ + n, f32_sortable(J), tag);          // ---- 2. hop 1: publish
                if (last_it) a.J[n] = J;
            }
        } else {
            const int tsub = t - 64, nsub = CF_BLOCK - 64;
            if (last_it && a.Q_out) {                             // the plans, coalesced (ctk_read / logging)
                const int total = max(0, min(CF_TRAJ, N - row0)) * H;
                float* dst = a.Q_out + (size_t)row0 * H;
                for (int i = tsub; i < total; i += nsub) {
                    const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i;
                    dst[i] = ubuf[r * us + (i - r * H)];
                }
            }
            if (!last_it) {                                       // next iteration's samples: independent of mu / std
                RolloutArgs an = a;
                an.stream_id = (uint32_t)(it + 1);
                cem_fetch_tile(tiles[(it + 1) & 1], samples ? samples + (size_t)cf.per_it * (it + 1) : nullptr, an, row0, tsub, nsub);
            }
        }
        // all N costs into LDS (waves 1..3 start polling while wave 0 still runs the recurrence); their range on the way
        uint32_t kmin_t = 0xFFFFFFFFu, kmax_t = 0u;
        expired |= ll_gather(cf.llJ, N, tag, t, CF_BLOCK, cf.timeout_ticks, [&](int i, uint32_t v) {
            keys[i] = v; kmin_t = min(kmin_t, v); kmax_t = max(kmax_t, v);
        });
        CSTAMP(3);
        kmin_t = wave_min_u32(kmin_t);
        kmax_t = ~wave_min_u32(~kmax_t);
        if (lane == 0) { red[wave] = kmin_t; red[CF_WAVES + wave] = ~kmax_t; }
        __syncthreads();
        const uint32_t kbase = red_min(0);
        const uint32_t krange = ~red_min(1) - kbase;
        if (t == 0) { sel[0] = 0; sel[1] = cf.K; }

        // ---- 3. K-th smallest key: MSB-first radix select over d = key - kbase, 8 bits per pass.  Only the bits the range
        //      needs are walked, and the first digit buckets the costs LINEARLY over [min, max] (the raw top bits of a float are
        //      nearly constant over a population's costs: every key in one bin serialises the LDS atomics).  (Compacting the first
        //      pass's bucket and ranking its keys by brute force instead of the later passes was measured: slower, 4.0 vs 3.0 us.)
        const int nbits = 32 - __builtin_clz(krange | 1u);
        const int passes = (nbits + 7) >> 3;
        for (int pass = 0; pass < passes; ++pass) {
            const int hi = nbits - 8 * pass, lo = max(hi - 8, 0);   // this pass's digit = bits [lo, hi) of d: the first one is full
            const uint32_t dmask = (1u << (hi - lo)) - 1u;
            if (t < 256) hist[t] = 0;
            __syncthreads();
            const uint32_t prefix = (uint32_t)sel[0];
            const int want = sel[1];
            for (int j0 = t; j0 < N; j0 += CF_BLOCK * CF_CHUNK) { // unconditional LDS reads in flight (keys[] is padded), then the counting
                uint32_t dj[CF_CHUNK];
#pragma unroll
                for (int u = 0; u < CF_CHUNK; ++u) dj[u] = keys[j0 + u * CF_BLOCK] - kbase;
#pragma unroll
                for (int u = 0; u < CF_CHUNK; ++u) {
                    const bool act = (j0 + u * CF_BLOCK < N) & (pass == 0 || (dj[u] >> hi) == prefix);
                    if (act) atomicAdd(&hist[(dj[u] >> lo) & dmask], 1);
                }
            }
            __syncthreads();
            if (pass == 0) CSTAMP(9);
            if (wave == 0) {
                const int b0 = hist[4 * lane], b1 = hist[4 * lane + 1], b2 = hist[4 * lane + 2], b3 = hist[4 * lane + 3];
                const int c = b0 + b1 + b2 + b3;
                // inclusive prefix over the 64 lanes: DPP row shifts inside each row of 16, then the three row totals
                int inc = c;
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);   // row_shr:1, zero fill
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);   // row_shr:2
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);   // row_shr:4
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);   // row_shr:8
                const int r0 = __builtin_amdgcn_readlane(inc, 15), r1 = __builtin_amdgcn_readlane(inc, 31), r2 = __builtin_amdgcn_readlane(inc, 47);
                inc += lane >= 48 ? r0 + r1 + r2 : (lane >= 32 ? r0 + r1 : (lane >= 16 ? r0 : 0));
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(inc >= want);
                const int first = hit ? __builtin_ctzll(hit) : 64;   // hit != 0: the histogram holds >= want keys
                if (lane == first) {
                    int below = inc - c, dgt = 4 * lane, bsel = b0;
                    if (below + b0 >= want) { dgt += 0; }
                    else if (below + b0 + b1 >= want) { below += b0; dgt += 1; bsel = b1; }
                    else if (below + b0 + b1 + b2 >= want) { below += b0 + b1; dgt += 2; bsel = b2; }
                    else { below += b0 + b1 + b2; dgt += 3; bsel = b3; }
                    sel[0] = (int)((prefix << (hi - lo)) | (uint32_t)dgt);
                    sel[1] = want - below;
                    sel[2] = bsel;                                // keys in the chosen bin (last pass: keys == the K-th smallest)
                }
            }
            __syncthreads();
            if (pass == 0) CSTAMP(10);
            if (pass == 1) CSTAMP(11);
        }
        CSTAMP(4);
        const uint32_t T32 = kbase + (uint32_t)sel[0];            // the K-th smallest key
        const int r_ties = sel[1];                                // of the keys == T32, the first r_ties in index order are elite
        // ties in front of this workgroup's rows — only when the cut falls INSIDE a group of equal keys (workgroup-uniform)
        const bool cut_in_tie = sel[2] != r_ties;
        __syncthreads();                                          // red[] (the range) and sel[] have been read by everyone
        if (cut_in_tie) {
            int tb = 0;
            for (int j = t; j < min(row0, N); j += CF_BLOCK) tb += keys[j] == T32;
            tb = (int)wave_sum((float)tb);                        // exact: < 2^24
            if (lane == 0) red[wave] = (uint32_t)tb;
        } else if (lane == 0) red[wave] = 0u;
        __syncthreads();
        const int ties_before = (int)red_sum(0);
        if (wave == 0) {
            const uint32_t ki = valid ? keys[n] : 0xFFFFFFFFu;
            const bool tie = valid && ki == T32;
            const unsigned long long tm = __builtin_amdgcn_ballot_w64(tie);
            const int my_tie_rank = ties_before + __builtin_popcountll(tm & ((1ull << lane) - 1ull));
            const bool elite = valid && (ki < T32 || (tie && my_tie_rank < r_ties));
            const unsigned long long em = __builtin_amdgcn_ballot_w64(elite);
            if (elite) erow[__builtin_popcountll(em & ((1ull << lane) - 1ull))] = lane;
            if (lane == 0) nb_s[0] = __builtin_popcountll(em);
        }
        __syncthreads();

        CSTAMP(5);
        // ---- 4. local moments of the elite rows, hop 2 --------------------------------------------------------------
        const int nb = nb_s[0];
        unsigned long long* myrec = cf.llS + (size_t)blockIdx.x * rs;
        if (t == 0) ll_st(myrec, (uint32_t)nb, tag);
        for (int h = t; h < H; h += CF_BLOCK) {
            const double mu0 = (double)mu_s[h];
            double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
            for (int e = 0; e < nb; ++e) { const double d = (double)ubuf[erow[e] * us + h] - mu0; s1 += d; s2 = fma(d, d, s2); }
            const double mb = nb > 0 ? s1 / (double)nb : 0.0;
            const double m2 = fma(-mb, s1, s2);                   // sum (d - m_b)^2 = s2 - s1^2 / n_b
            ll_st(myrec + 1 + h, __builtin_bit_cast(uint32_t, (float)mb), tag);
            ll_st(myrec + 1 + H + h, __builtin_bit_cast(uint32_t, (float)(m2 > 0.0 ? m2 : 0.0)), tag);
        }
        CSTAMP(6);
        expired |= ll_gather(cf.llS, cf.nblk * rs, tag, t, CF_BLOCK, cf.timeout_ticks,
                             [&](int i, uint32_t v) { reinterpret_cast<uint32_t*>(recs)[i] = v; });
        __syncthreads();

        CSTAMP(7);
        // ---- 5. refit (:77-78; population std): A = sum n_b m_b, B = sum (M2_b + n_b m_b^2) in double, in a fixed order (segments of the
        //      workgroup range, then the segments): every workgroup arrives at the same bits.  One pass, one barrier.
        //      (One thread per column walking all workgroups: 4.4 us at cfg3.)
        {
            constexpr int SEGMAX = 16;
            const double invK = 1.0 / (double)cf.K;               // == 1 / sum_b n_b: the elite set has exactly K rows
            const int* nrec = reinterpret_cast<const int*>(recs);
            auto add_rec = [&](int bq, int h, double& A, double& B) {
                const double nbq = (double)nrec[bq * rs], mb = (double)recs[bq * rs + 1 + h];
                A = fma(nbq, mb, A);
                B += (double)recs[bq * rs + 1 + H + h] + nbq * mb * mb;
            };
            auto finish = [&](int h, double A, double B) {
                const double mshift = A * invK;
                const double var = fma(-mshift, mshift, B * invK);
                mu_s[h] = (float)((double)mu_s[h] + mshift);
                sd_s[h] = (float)sqrt(var > 0.0 ? var : 0.0);     // tf.math.reduce_std: ddof = 0
            };
            const bool multi = H <= CF_BLOCK && cf.nblk > 8;      // few workgroups: the plain walk is shorter than the barrier
            const int SEG = multi ? min(SEGMAX, CF_BLOCK / H) : 1, per = (cf.nblk + SEG - 1) / SEG;
            if (multi) {
                const int hcol = t % H, sg = t / H;
                if (sg < SEG) {
                    double A = 0.0, B = 0.0;
                    const int bb = sg * per, be = min(cf.nblk, bb + per);
                    for (int bq = bb; bq < be; ++bq) add_rec(bq, hcol, A, B);
                    part[(sg * H + hcol) * 2] = A; part[(sg * H + hcol) * 2 + 1] = B;
                }
                __syncthreads();
                if (t < H) {
                    double A = 0.0, B = 0.0;
                    for (int q = 0; q < SEG; ++q) { A += part[(q * H + t) * 2]; B += part[(q * H + t) * 2 + 1]; }
                    finish(t, A, B);
                }
            } else {
                for (int h = t; h < H; h += CF_BLOCK) {
                    double A = 0.0, B = 0.0;
                    for (int bq = 0; bq < cf.nblk; ++bq) add_rec(bq, h, A, B);
                    finish(h, A, B);
                }
            }
        }
        __syncthreads();
        CSTAMP(8);

        if (last_it) {
            // u = elite[0,0,:] (:101): first input of the cheapest row under (J, index), published by its owner
            const uint32_t gk = kbase;                            // the cheapest cost's key (this iteration's range, above)
            int best = 0x7FFFFFFF;
            for (int j = t; j < N; j += CF_BLOCK)
                if (keys[j] == gk) { best = j; break; }           // j ascending per thread: its smallest match
            best = (int)wave_min_u32((uint32_t)best);
            __syncthreads();                                      // red[] is read above by everyone
            if (lane == 0) red[wave] = (uint32_t)best;
            __syncthreads();
            const int gbest = (int)red_min(0);
            if (t == 0 && gbest >= row0 && gbest < row0 + CF_TRAJ) {
                cf.idx_out[0] = gbest;
                if (expired) __hip_atomic_store(reinterpret_cast<uint32_t*>(cf.u_host) + 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if constexpr (C == 1) publish_u(cf.u_dev, cf.u_host, ubuf[(gbest - row0) * us], cf.seq);
                else publish_u_vec(cf.u_dev, cf.u_host, ubuf + (gbest - row0) * us, C, cf.seq);
            }
            // :99-102 clip the std, shift both by one step, refill the tail — the handle's distribution for the next MPC step
            if (blockIdx.x == 0) {
                for (int h = t; h < H; h += CF_BLOCK) {
                    cf.mu[h] = (h + C < H) ? mu_s[h + C] : cf.mid[h - (H - C)];      // shift by one STEP = C columns
                    cf.sd[h] = (h + C < H) ? fminf(fmaxf(sd_s[h + C], cf.std_min), cf.std_max) : cf.init_std;
                }
            }
        }
    }
    // a wait that ran out in a workgroup that does not own the best row still has to reach the host
    if (expired && t == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(cf.u_host) + 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// H below: flat columns of a plan (mpc_horizon * control inputs)
int ctk_cem_fused_blocks(int N) { return (N + CF_TRAJ - 1) / CF_TRAJ; }
size_t ctk_cem_fused_ll_words(int N, int H) { return (size_t)N + (size_t)ctk_cem_fused_blocks(N) * (1 + 2 * H); }
size_t ctk_cem_fused_lds(int N, int H) { return (size_t)cem_carve(N, H, ctk_cem_fused_blocks(N)).total * sizeof(float); }
bool ctk_cem_fusable(int pred, int N, int H) {
    return pred == CTK_PRED_ODE && ctk_cem_fused_blocks(N) <= CTK_CEM_FUSED_MAX_BLOCKS && ctk_cem_fused_lds(N, H) <= 128 * 1024;
}
const char* ctk_cem_fused_name(int env, bool log) {
    return ctk_kernel_name("ctk_cem_fused<%d, %4$s>", env, 0, 0, log ? "true" : "false");
}

// a_in.H steps, a_in.C inputs (limits per input); the kernel constants are derived from the environment's parameter table
hipError_t ctk_launch_cem_fused(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a_in, const float* samples,
                                const CemFusedLaunch& c, bool log, hipEvent_t e0, hipEvent_t e1) {
    const int nblk = ctk_cem_fused_blocks(a_in.N);
    const dim3 grid(nblk), block(CF_BLOCK);
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        RolloutArgs a = a_in;
        const int HC = a.H * E::C;
        a.C = E::C; a.P = HC;
        a.p_magic = HC >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)HC - 1) / (uint64_t)HC) : 0u;
        const typename E::K k = E::derive(params, dt, isteps);
        CemFusedK cf{};
        cf.its = c.its; cf.K = c.K; cf.nblk = nblk; cf.per_it = (unsigned long long)a.N * HC;
        cf.llJ = c.ll; cf.llS = c.ll + a.N; cf.tag0 = c.tag0;
        cf.std_min = c.std_min; cf.std_max = c.std_max; cf.init_std = c.init_std;
        for (int i = 0; i < E::C; ++i) cf.mid[i] = 0.5f * (a.lo[i] + a.hi[i]);
        cf.mu = c.mu; cf.sd = c.sd; cf.u_dev = c.u_dev; cf.u_host = c.u_host; cf.idx_out = c.idx_out; cf.seq = c.seq;
        cf.timeout_ticks = (unsigned long long)(c.timeout_s * 1.0e8);
        const size_t lds = ctk_cem_fused_lds(a.N, HC);
        if (log) CTK_LAUNCH((ctk_cem_fused<EV, true>), grid, block, lds, st, e0, e1, samples, a, k, cf);
        else CTK_LAUNCH((ctk_cem_fused<EV, false>), grid, block, lds, st, e0, e1, samples, a, k, cf);
    });
    return hipGetLastError();
}
