// ctk_mppi_merge.h — the soft-min merge of MPPI partial records {rho, a, b[P]} (optimizer_mppi.py:163-168 re-associated) and the
// MPPI update, as device functions of ONE 256-thread workgroup: shared by ctk_mppi.hip (the merge kernels, the fused tail of
// ctk_mppi_rollout) and by the network template kernels (ctk_generic_net.hip, ctk_gru4.hip), whose MPPI launches end in the same
// low-latency hand-off (mppi_ll_tail below).
#pragma once
#include "ctk_device.h"
#include "ctk_common.h"

// ---------------------------------------------------------------------------------------------
// merge of partial records {rho, a, b[P]} by one 256-thread block.
// FINAL=false: writes one merged record to `out_rec`.
// FINAL=true : applies the MPPI update and publishes u.
// SC1: the records were handed over inside ONE launch (fused tail below): every load of them is an
//      agent-scope relaxed atomic load (global_load ... sc1), cdna_hip_programming.md G16.
// ---------------------------------------------------------------------------------------------
constexpr int MERGE_BLOCK = 256;
constexpr int MERGE_CHUNK = 1024;

// LD: 0 plain loads (records written by an earlier launch); 1 agent-scope (handed over inside ONE launch);
//     2 system-scope (records stored by peer GPUs into this GPU's uncached exchange buffer, ctk_mppi_p2p_exchange)
template <int LD>
CTK_DEV float ld_rec(const float* p) {
    if constexpr (LD == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (LD == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else return *p;
}
CTK_DEV void st_rec(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct MppiUpdateArgs {
    // per-step tables already resident in this block's LDS (fused tail) — nullptr: read them from memory
    // (un_l: the shifted nominal plan [H*C])
    const float* w0_l = nullptr; const float* w1_l = nullptr; const float* un_l = nullptr; const int* i0_l = nullptr;
    int H;
    const InterpEntry* interp;
    const float* u_nom_in;
    float* u_nom_out;
    float lo, hi;          // C == 1
    float* u_dev;
    float* u_host;
    uint32_t seq;
    int C = 1;             // control inputs: records carry b[P*C], the update runs per channel
    float lo_c[CTK_MAX_INPUTS] = {}, hi_c[CTK_MAX_INPUTS] = {};   // C > 1
};

// scratch: >= 8 + (P + 1) + min(cnt, MERGE_CHUNK) floats of LDS, plus cnt*(2+P) more when `stage`
// (all records fetched into LDS by ONE wide pass: one memory round trip instead of one per record).
// CH: control inputs of the FINAL update (compile time: the C == 1 instantiations are CartPole's statement sequence, unchanged)
template <bool FINAL, int SC1, int CH = 1>
CTK_DEV void mppi_merge_block(float* scratch, const float* base, int cnt, int P, float neg_inv_lbd, float* out_rec,
                              const MppiUpdateArgs& up, int stage) {
    float* red = scratch;             // [4] cross-wave scratch
    float* b_s = scratch + 8;         // [P + 1] merged numerator
    float* sc_s = b_s + P + 1;        // [chunk] per-record rescale factors
    float* st_s = sc_s + (cnt < MERGE_CHUNK ? cnt : MERGE_CHUNK);   // [cnt][2+P] staged records
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int rs = 2 + P;
    if (stage == 1) {
        const int tot = cnt * rs;
        for (int i0 = 0; i0 < tot; i0 += 4 * MERGE_BLOCK) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * MERGE_BLOCK + t;
                if (i < tot) v[j] = ld_rec<SC1>(base + i);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * MERGE_BLOCK + t;
                if (i < tot) st_s[i] = v[j];
            }
        }
        __syncthreads();
    }
    auto rec_at = [&](int i, int f) -> float { return stage != 0 ? st_s[i * rs + f] : ld_rec<SC1>(base + (size_t)i * rs + f); };

    float r = INFINITY;
    for (int i = t; i < cnt; i += MERGE_BLOCK) r = fminf(r, rec_at(i, 0));
    r = wave_min(r);
    if (lane == 0) red[wave] = r;
    __syncthreads();
    const float rho = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    __syncthreads();

    float a_acc = 0.0f;
    float b_acc[4] = {0.f, 0.f, 0.f, 0.f};   // thread t owns columns t, t+256, ... (P <= 1024)
    for (int c0 = 0; c0 < cnt; c0 += MERGE_CHUNK) {
        const int cn = min(MERGE_CHUNK, cnt - c0);
        for (int i = t; i < cn; i += MERGE_BLOCK) {
            const float sc = expf(neg_inv_lbd * (rec_at(c0 + i, 0) - rho));   // e^{-(rho_r - rho)/lambda}
            sc_s[i] = sc;
            a_acc += rec_at(c0 + i, 1) * sc;
        }
        __syncthreads();
        if (stage != 0 && cnt > 128 && cnt <= MERGE_CHUNK && 2 * P <= MERGE_BLOCK) {
            // many narrow records (a shard of configs[4]: 256 records of 11 columns): one thread per column would walk all of them with
            // 245 threads idle.  Thread (slice, column) sums every slices-th record; the slices meet through LDS in slice order.
            // (cnt > 128 with staged records did not exist before round 4: no earlier result changes its association)
            const int slices = MERGE_BLOCK / P, sl = t / P, p = t - sl * P;
            float* part_s = st_s + (size_t)cnt * rs;           // [slices][P], behind the staged records (merge_lds_staged reserves it)
            if (sl < slices) {
                float acc = 0.0f;
                for (int i = sl; i < cn; i += slices) acc += rec_at(i, 2 + p) * sc_s[i];
                part_s[sl * P + p] = acc;
            }
            __syncthreads();
            if (t < P) {
                float acc = 0.0f;
                for (int q = 0; q < slices; ++q) acc += part_s[q * P + t];
                b_acc[0] = acc;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = t + j * MERGE_BLOCK;
                if (p < P) {
                    float acc = b_acc[j];
                    for (int i = 0; i < cn; ++i) acc += rec_at(c0 + i, 2 + p) * sc_s[i];
                    b_acc[j] = acc;
                }
            }
        }
        __syncthreads();
    }
    a_acc = wave_sum(a_acc);
    if (lane == 0) red[wave] = a_acc;
    __syncthreads();
    const float a_tot = red[0] + red[1] + red[2] + red[3];

    if constexpr (!FINAL) {
        if (t == 0) { out_rec[0] = rho; out_rec[1] = a_tot; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = t + j * MERGE_BLOCK;
            if (p < P) out_rec[2 + p] = b_acc[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = t + j * MERGE_BLOCK;
            if (p < P) b_s[p] = b_acc[j];
        }
        if (t == 0) b_s[P] = 0.0f;   // pad read by i0+1 when P == 1
        __syncthreads();
        if constexpr (CH == 1) {
            for (int h = t; h < up.H; h += MERGE_BLOCK) {
                InterpEntry e; float un;
                if (up.w0_l) { e = InterpEntry{up.i0_l[h], up.w0_l[h], up.w1_l[h]}; un = up.un_l[h]; }
                else { e = up.interp[h]; un = up.u_nom_in[min(h + 1, up.H - 1)]; }
                const float w = (b_s[e.i0] * e.w0 + b_s[e.i0 + 1] * e.w1) / a_tot;
                const float o = fminf(fmaxf(un + w, up.lo), up.hi);   // optimizer_mppi.py:190
                up.u_nom_out[h] = o;
                if (h == 0) publish_u(up.u_dev, up.u_host, o, up.seq);   // :191 u = u_nom[0,0,:]
            }
        } else {
            // P here = P*C record columns; inducing point i of channel c is column i*C + c
            constexpr int C = CH;
            const int Pp = P / C;
            float* u_s = scratch;             // red[] is dead: the C outputs of step 0
            for (int hc = t; hc < up.H * C; hc += MERGE_BLOCK) {
                const int h = hc / C, c = hc - h * C;
                InterpEntry e; float un;
                if (up.w0_l) { e = InterpEntry{up.i0_l[h], up.w0_l[h], up.w1_l[h]}; un = up.un_l[hc]; }
                else { e = up.interp[h]; un = up.u_nom_in[min(h + 1, up.H - 1) * C + c]; }
                const int i1 = min(e.i0 + 1, Pp - 1);
                const float w = (b_s[e.i0 * C + c] * e.w0 + b_s[i1 * C + c] * e.w1) / a_tot;
                const float o = fminf(fmaxf(un + w, up.lo_c[c]), up.hi_c[c]);   // optimizer_mppi.py:190
                up.u_nom_out[hc] = o;
                if (h == 0) u_s[c] = o;
            }
            __syncthreads();
            if (t == 0) publish_u_vec(up.u_dev, up.u_host, u_s, C, up.seq);   // :191 u = u_nom[0,0,:]
        }
    }
}


// start of the staged records inside the merge scratch (see mppi_merge_block)
CTK_DEV float* merge_stage_ptr(float* scratch, int cnt, int P) { return scratch + 8 + (P + 1) + (cnt < MERGE_CHUNK ? cnt : MERGE_CHUNK); }

// Low-latency hand-off of a record word: value and the launch's sequence number travel in ONE 8-byte store, so
// the reader polls the data itself — no "drain my stores, then signal" step and no ticket (cf. RCCL's LL protocol).
CTK_DEV void ll_store(unsigned long long* p, float v, uint32_t seq) {
    __hip_atomic_store(p, ((unsigned long long)seq << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// The fused tail of an MPPI rollout launch of 256-thread workgroups whose blocks have published their records as {value, seq} words
// (ll_store): called by block 0 (all threads, after a barrier that retires its own use of `lds`), which polls every word until it
// carries this launch's sequence number — the words ARE the data — stages them in LDS, merges, and either applies the update and
// publishes u (mode 1) or writes the shard's ONE record (mode 2).  lds: merge_lds_staged(P, nb) bytes.  A bounded poll that runs out
// raises the error word behind {u, seq}: ctk_api.hip:finish_step returns CTK_ERR_STATE.
template <int CH>
CTK_DEV void mppi_ll_tail(float* lds, const unsigned long long* ll, int nb, int P, float neg_inv_lbd, int mode, float* out_rec,
                          const MppiUpdateArgs& up) {
    const int t = threadIdx.x, tot = nb * (2 + P);
    float* st = merge_stage_ptr(lds, nb, P);
    bool expired = false;
    constexpr int LLW = 8;                // words in flight per thread
    for (int i0 = t; i0 < tot; i0 += MERGE_BLOCK * LLW) {
        unsigned long long w[LLW];
#pragma unroll
        for (int j = 0; j < LLW; ++j) {
            const int i = i0 + j * MERGE_BLOCK;
            if (i < tot) w[j] = __hip_atomic_load(ll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < LLW; ++j) {
            const int i = i0 + j * MERGE_BLOCK;
            if (i < tot) {
                for (int spin = 0; (uint32_t)(w[j] >> 32) != up.seq && spin < (1 << 22); ++spin) {
                    __builtin_amdgcn_s_sleep(1);
                    w[j] = __hip_atomic_load(ll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const bool got = (uint32_t)(w[j] >> 32) == up.seq;
                expired |= !got;
                st[i] = got ? __builtin_bit_cast(float, (uint32_t)w[j]) : __builtin_nanf("");
            }
        }
    }
    if (expired && up.u_host)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(up.u_host) + 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (mode == 1) mppi_merge_block<true, 0, CH>(lds, nullptr, nb, P, neg_inv_lbd, nullptr, up, 2);
    else mppi_merge_block<false, 0, CH>(lds, nullptr, nb, P, neg_inv_lbd, out_rec, up, 2);
}

// kernel argument of the network template kernels' MPPI launches (ctk_generic_net.hip, ctk_gru4.hip): mode 0 = block records only
struct NetFuse {
    int mode = 0;                       // 1 merge + update + publish u; 2 merge into ONE record (sharded step_begin)
    unsigned long long* ll = nullptr;   // [blocks][2 + P*C] {value, seq} words
    float* out_rec = nullptr;           // mode 2
    MppiUpdateArgs up{};
};

// LDS of a merge by one workgroup (bytes); with all records staged in LDS (used when it stays <= 64 KiB)
inline size_t merge_lds(int P, int cnt) { return (size_t)(8 + P + 1 + (cnt < MERGE_CHUNK ? cnt : MERGE_CHUNK)) * sizeof(float); }
inline size_t merge_lds_staged(int P, int cnt) { return merge_lds(P, cnt) + ((size_t)cnt * (2 + P) + MERGE_BLOCK) * sizeof(float); }   // (+ the column slices' partial sums)
inline bool merge_can_stage(int P, int cnt) { return merge_lds_staged(P, cnt) <= 64 * 1024; }
