// ctk_mppi.hip — MPPI on gfx950 (replaces reference Optimizers/optimizer_mppi.py:181-193).
//
//   ctk_mppi_rollout_ode   64 trajectories per 256-thread block.  All four waves do the parts
//        that do not depend on the state (coalesced HBM read of the block's [64,P] perturbation
//        tile into LDS, interpolation + nominal + clip + MPPI correction for all H steps into an
//        LDS input buffer); then ONE wave runs the H-step recurrence, one trajectory per lane,
//        state in registers, inputs prefetched from LDS {stage cost, Euler step}; then all four
//        waves form the block-local soft-min partial (rho_b, a_b, b_b[P]) by wave shuffles + LDS
//        column sums (sum_n e_n * delta_u_n is linear in the inducing points, so the reduction
//        runs over P values per trajectory instead of H).
//   ctk_mppi_merge         merges partial records {rho, a, b[P]} (blocks of one GPU, or the
//        all-gathered records of several GPUs — SURVEY.md 8e) and either emits one record or
//        applies the update u_nom <- clip(shift(u_nom) + interp(b)/a)   (:163-168,:184,:190).
#ifdef CTK_STAMPS   // diagnostic build (tools/diag_mppi_stamps.hip); never compiled into libctk_hip.so
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long _t;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0 && a.stamps) a.stamps[blockIdx.x * 16 + (i)] = _t;                     \
    } while (0)
#else
#define STAMP(i)
#endif

#include "ctk_rollout.h"
#include "ctk_env.h"
#include "ctk_mlp.h"
#include "ctk_gru.h"
#include "ctk_launch.h"
#include "ctk_mppi_merge.h"
#include <cstring>

constexpr int MPPI_TRAJ = 64;     // trajectories per block: one wave runs the recurrence
constexpr int MPPI_WAVES = 4;     // waves per block: the prologue / epilogue are spread over all four
constexpr int MPPI_BLOCK = MPPI_TRAJ * MPPI_WAVES;

// LDS carve (floats): tile[64][ts] | ubuf[64][us] | corr[4][64] | e[64] | colsum[4][P] | w0,w1,un,i0 [H] each
__host__ __device__ inline int ubuf_stride(int H) { return (H + 1) | 1; }

// ---------------------------------------------------------------------------------------------
// Multi-GPU exchange without a collective library call (xGMI is point-to-point): every rank owns an
// uncached, IPC-exported exchange buffer  recs[2 parities][W][2+P] | flags[2][W]  that all peers have mapped.
// One 256-thread block per rank and step: store my shard record into slot [parity][rank] of EVERY rank's
// buffer (system-scope stores over xGMI), fence, raise flag [parity][rank] = seq there; wait until all W
// flags in MY buffer carry seq; merge the W records and apply the MPPI update — every rank arrives at the
// identical u_nom.  Parity double-buffering: a peer can run at most one step ahead (it needs my flag of step
// seq+1 to finish that step), so it never overwrites a slot I am still reading.  The wait is bounded by a
// wall-clock timeout; on expiry the step publishes NaN and an error word the host turns into CTK_ERR_STATE.
// ---------------------------------------------------------------------------------------------
struct P2PArgs {
    float* bufs[CTK_P2P_MAX_WORLD];   // exchange buffer of every rank (bufs[rank] is local memory)
    int rank, world, rs;              // rs = 2 + P floats per record
    uint32_t seq;
    uint32_t* err_host;               // pinned, device-visible error word
    unsigned long long timeout_ticks; // wall_clock64 ticks (100 MHz)
};

// the exchange, by one 256-thread block (its own launch, or the tail of the rollout launch's block 0); `bad` is a
// workgroup-shared word
CTK_DEV void p2p_exchange_and_update(float* lds, int* bad, const P2PArgs& x, int P, float neg_inv_lbd, const MppiUpdateArgs& up,
                                     int stage_ok) {
    const int t = threadIdx.x, W = x.world, rs = x.rs, par = (int)(x.seq & 1u);
    float* mine = x.bufs[x.rank];
    const size_t slot = (size_t)(par * W + x.rank) * rs, flags0 = (size_t)2 * W * rs;
    if (t == 0) *bad = 0;
    for (int w = 0; w < W; ++w) {
        if (w == x.rank) continue;
        float* dst = x.bufs[w] + slot;
        for (int i = t; i < rs; i += MERGE_BLOCK)
            __hip_atomic_store(dst + i, ld_rec<2>(mine + slot + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (t < W) {
        uint32_t* f = reinterpret_cast<uint32_t*>(x.bufs[t] + flags0) + par * W + x.rank;
        __hip_atomic_store(f, x.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (t < W) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(mine + flags0) + par * W + t;
        const unsigned long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - x.seq) < 0) {
            if (wall_clock64() - t0 > x.timeout_ticks) { atomicExch(bad, 1); break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    if (*bad) {
        if (t == 0) {
            __hip_atomic_store(x.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            publish_u(up.u_dev, up.u_host, __builtin_nanf(""), up.seq);
        }
        return;
    }
    mppi_merge_block<true, 2>(lds, mine + (size_t)par * W * rs, W, P, neg_inv_lbd, nullptr, up, stage_ok != 0 ? 1 : 0);
}

__global__ __launch_bounds__(MERGE_BLOCK) void ctk_mppi_p2p_exchange(P2PArgs x, int P, float neg_inv_lbd, MppiUpdateArgs up, int stage_ok) {
    extern __shared__ float lds[];
    __shared__ int bad;
    p2p_exchange_and_update(lds, &bad, x, P, neg_inv_lbd, up, stage_ok);
}

// grid.x blocks; block b merges records [b*per_block, ...)
template <bool FINAL>
__global__ __launch_bounds__(MERGE_BLOCK) void ctk_mppi_merge(const float* __restrict__ parts, int n_parts, int per_block,
                                                             int P, float neg_inv_lbd, float* __restrict__ out_rec,
                                                             MppiUpdateArgs up, int stage_ok) {
    extern __shared__ float lds[];
    const int first = blockIdx.x * per_block;
    const int cnt = min(per_block, n_parts - first);
    mppi_merge_block<FINAL, false>(lds, parts + (size_t)first * (2 + P), cnt, P, neg_inv_lbd,
                                   out_rec ? out_rec + (size_t)blockIdx.x * (2 + P) : nullptr, up, stage_ok != 0 ? 1 : 0);
}

// floats of the rollout carve (everything but the GRU exchange slots), rounded up so that what follows is
// 16-byte aligned.
// traj = trajectories per workgroup: 64 (ODE: one wave runs them, one per lane; MLP: 16 per wave) or
// GRU_TRAJ = 16 (GRU: the four waves share one 16-trajectory MFMA column block, ctk_gru.h)
// PC = P*C sample columns, HC = H*C inputs per trajectory (C control inputs; CartPole: C = 1)
__host__ __device__ inline int mppi_carve_floats(int PC, int HC, int traj, int H) {
    const int f = traj * tile_stride(PC) + traj * ubuf_stride(HC) + MPPI_BLOCK + traj + MPPI_WAVES * PC + 3 * H + HC;
    return (f + 3) & ~3;
}
__host__ __device__ inline int mppi_carve_floats(int P, int H, int traj) { return mppi_carve_floats(P, H, traj, H); }
constexpr int MPPI_PAIR_TRAJ = 32;   // MLP pair form: two tiles per workgroup, two waves per tile
__host__ __device__ inline int mppi_traj(int pred) { return pred == CTK_PRED_GRU ? GRU_TRAJ : pred == CTK_PRED_MLP_PAIR ? MPPI_PAIR_TRAJ : MPPI_TRAJ; }

// In-launch tail of the rollout kernel (single-GPU, <= 256 blocks): the block whose ticket is last
// merges all block records and applies the update, saving the second launch and its boundary.
struct FuseArgs {
    int mode;             // 0: records only; 1: last block merges + updates u_nom/u; 2: last block emits ONE merged record
    int stage_ok;         // the launch's LDS holds all records staged (merge_lds with staging)
    const P2PArgs* p2p;   // mode 3: device-resident exchange description (ctk_p2p_connect); block 0 goes on to exchange + update
    uint32_t p2p_seq;     // mode 3: this step's exchange sequence number (parity = buffer half)
    unsigned long long* ll;   // {value, seq} words [blocks][2+P] for the low-latency hand-off (needs stage_ok); nullptr: ticket path
    unsigned* counter;    // zero before the launch; the last block resets it
    float* out_rec;       // mode 2
    MppiUpdateArgs up;    // mode 1
};

// Argument order: the 14 dwords the first global loads depend on come first — with
// -amdgpu-kernarg-preload-count=14 (Makefile) the command processor delivers them in SGPRs at wave launch, so the
// sample loads do not wait for a cold scalar-cache miss on the kernel-argument segment (MI355X: gfx940+ feature).
// N_/H_/P_/pmagic_ duplicate fields of `a` for that reason.
// P2P: the instantiations whose merging block goes on to exchange records with peer GPUs (fuse mode 3, ctk_p2p_step);
// kept apart because that tail costs the network-predictor kernels registers they need in the recurrence (measured:
// GRU 83 -> 109 us per launch when every instantiation carried it).
// ENV: the environment (ctk_env.h).  The analytic-predictor instantiation (PRED = ODE) is written against Env<ENV> only — C
// control inputs (sample columns P*C, inputs H*C per trajectory, per-channel interpolation / clip / correction / update), the
// recurrence through Env::cost_step, the input-only cost terms through Env::input_cost; CartPole is Env<0>, C = 1.  The network
// predictors' instantiations are CartPole's (other environments: ctk_generic_net.hip).
// P_ = inducing points; pmagic_ = magic of the P_*C sample columns of a row.
template <int ENV, int PRED, bool LOG, bool P2P = false>
__global__ __launch_bounds__(MPPI_BLOCK) void ctk_mppi_rollout(const float* __restrict__ samples,
                                                               const float* __restrict__ u_nom,
                                                               const InterpEntry* __restrict__ interp,
                                                               const float* __restrict__ wperm,
                                                               float* __restrict__ parts, int N_, int H_, int P_,
                                                               uint32_t pmagic_, RolloutArgs a_in, typename Env<ENV>::K k, MppiK m, FuseArgs fz) {
    extern __shared__ float lds[];
#include "ctk_mppi_body.inc"
}

// ---------------------------------------------------------------------------------------------
// The RESIDENT form (opt-in: ctk_resident_enable): the same step — the same statements, ctk_mppi_body.inc — served
// by a kernel that stays on the device and takes its per-step inputs (state, previous input, sample pointer, sequence number, Philox
// position, which u_nom buffer is current) from a MAILBOX instead of a launch: what a launch-per-step loop pays per step in the runtime
// and the command processor is paid once.
//   * the mailbox lives in fine-grained DEVICE memory that the host stores into through the PCIe BAR (posted writes): every workgroup
//     polls local memory, and no PCIe READ is left on the request path (a read round trip is 1.4 - 5 us depending on the box,
//     tools/diag_mailbox_vram.hip).  Where the host cannot reach device memory (ctk_api.hip probes it) the mailbox is pinned host memory:
//     block 0 polls it — ONE poller: sixteen queue on the read path — fetches the request in one coalesced access and relays it through
//     device memory.  Everything a step leaves for the next one (u_nom, u) is published by the step's closing fence and acquired with
//     the next request;
//   * every wait is bounded by the wall clock: no request for `idle_ticks` -> the kernel leaves (the next ctk_step launches it again);
//     the other workgroups follow block 0 (its "left" flag) or give up after 4 x that.  It therefore never holds the device longer than
//     idle_ticks beyond its last step — a device-wide synchronize of anybody else waits at most that long; ctk_resident_stop / every
//     other API call on the handle ends it at once (cmd = EXIT);
//   * leaving is announced (state = LEAVING, fence, re-read the request number; then LEFT as the kernel's last store).  A request that
//     crosses the announcement in flight is either seen by the re-read and served, or lost with the kernel — the host sees LEFT while it
//     waits for the result and launches the kernel again with that request as its first.
// ---------------------------------------------------------------------------------------------
CTK_DEV uint32_t box_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
CTK_DEV void box_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// what a launch passes as kernel arguments, in DEVICE memory here: as by-value kernel arguments these ~120 dwords stay live in SGPRs across
// the request loop (218 spilled, read back lane by lane inside the recurrence: block 0's step 14.8 us); behind a pointer they are
// scalar-loaded where a step needs them, as a launched kernel loads its kernarg segment
template <class K>
struct ResidentArgs {
    const InterpEntry* interp;
    float* parts;
    int N, H, P;
    uint32_t pmagic;
    float* unom0;
    float* unom1;
    RolloutArgs base;
    K k;
    MppiK m;
    FuseArgs fz0;
};

template <int ENV>
__global__ __launch_bounds__(MPPI_BLOCK) void ctk_mppi_resident(const ResidentArgs<typename Env<ENV>::K>* __restrict__ ra,
                                                                const CtkResidentBox* box, int box_local, CtkResidentStat* stat, CtkResidentBox* relay,
                                                                unsigned long long idle_ticks, uint32_t first_req) {
    extern __shared__ float lds[];
    __shared__ CtkResidentBox req_s;
    __shared__ float uown_s[CTK_MAX_INPUTS];          // the optimizer's own last output, as the prepared inputs assumed it
    __shared__ int upd_ok_s;
    // ---- the names the body's parts expect; what changes per step is assigned before the part that reads it
    constexpr int PRED = CTK_PRED_ODE;
    constexpr bool LOG = false, P2P = false;
    const InterpEntry* interp = ra->interp;
    float* parts = ra->parts;
    const int N_ = ra->N, H_ = ra->H, P_ = ra->P;
    const uint32_t pmagic_ = ra->pmagic;
    float* const unom0 = ra->unom0; float* const unom1 = ra->unom1;
    const typename Env<ENV>::K& k = ra->k;
    const MppiK& m = ra->m;
    const float* samples = nullptr;
    const float* u_nom = unom0;
    const float* wperm = nullptr;
    const RolloutArgs& a_in = ra->base;
    FuseArgs fz = ra->fz0;
    const float* const d_u = ra->base.u_prev_dev;     // where the step publishes its own output
#ifdef CTK_RES_STAMPS                                   // diagnostic build only: where block 0's step goes (shader cycles since the request was fetched)
    unsigned long long c_req = 0;
#pragma push_macro("STAMP")
#undef STAMP
#define STAMP(i) do { if ((blockIdx.x == 0 || blockIdx.x == 5) && threadIdx.x == 0 && (i) >= 3) __hip_atomic_store(&stat->stamps[(blockIdx.x ? 6 : 0) + (i) - 2], (uint32_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#endif
#define CTK_BODY_S1 16                                // as the launched kernel: the same split of the steps over the waves = the same sums, bit for bit
#include "ctk_mppi_body_1_decl.inc"
#include "ctk_mppi_body_3_defs.inc"
#undef CTK_BODY_S1
    // inputs of a whole step (sample tile, tables, clipped inputs, input-only costs) from `samples` / `a.call` / `u_nom` / `a.u_prev*`
    auto do_pre = [&]() {
#include "ctk_mppi_body_2_pro1.inc"
#include "ctk_mppi_body_4_phase_a.inc"
        // phase B too (part 5 runs it under the first S1 steps of the recurrence; it is idempotent, so running it here as well only means
        // that the recurrence finds every input ready): waves 1..3, their thirds of [S1, H)
        if (wave != 0) {
            const int Hb = (H - S1 + MPPI_WAVES - 2) / (MPPI_WAVES - 1);
            prologue2(lane, min(H, S1 + (wave - 1) * Hb), min(H, S1 + (wave - 1) * Hb + Hb), wave, corr_keep);
        }
        __syncthreads();
    };
    // recurrence from `a.s0`, costs, block record, hand-off; block 0: merge, update, publish {u, seq}
    auto do_post = [&]() {
#include "ctk_mppi_body_5_post.inc"
    };
#ifdef CTK_RES_STAMPS
#pragma pop_macro("STAMP")
#endif
    uint32_t served = first_req - 1u;                 // request number of the last step this workgroup has taken
    unsigned long long t_seen = 0, t_relayed = 0;
    bool pre_ok = false;                              // the inputs of the NEXT step are already in LDS, formed for ...
    const float* pre_samples = nullptr;               // ... these draws (nullptr: the in-kernel sampler at position pre_call)
    uint32_t pre_call = 0, pre_cur = 0;
    for (;;) {
        // ---- the next request.  Wave 0: lane 0 polls ONE word; then the request is moved as a whole — its dwords by as many lanes in one
        //      coalesced access (a dword at a time over PCIe is a round trip each: 24 us measured).  box_local: every workgroup reads the
        //      box itself (device memory); else block 0 reads it (host memory) and relays it.  The relay's cmd word is also the "block 0 has
        //      left" flag the other workgroups watch.
        if (threadIdx.x < 64) {
            constexpr int BOXW = (int)(sizeof(CtkResidentBox) / sizeof(uint32_t));
            static_assert(BOXW <= 64 && offsetof(CtkResidentBox, req) == 0 && offsetof(CtkResidentBox, cmd) == 4 &&
                          offsetof(CtkResidentBox, upd) == sizeof(CtkResidentBox) - 4, "request layout");
            const int lane = threadIdx.x;
            const unsigned long long t0 = wall_clock64();
            int leave = 0, from_relay = 0;
            uint32_t w = 0;
            constexpr int TAILW = (int)(offsetof(CtkResidentBox, tail) / sizeof(uint32_t));
            // box_local: the whole box in ONE pass per poll (its dwords by as many lanes); req == tail != served: a complete new request
            auto poll_box = [&]() {
                const uint32_t x = lane < BOXW ? __hip_atomic_load(reinterpret_cast<const uint32_t*>(box) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                const uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), tl = (uint32_t)__builtin_amdgcn_readlane((int)x, TAILW);
                w = x;
                return head != served && tl == head;
            };
            if (blockIdx.x == 0) {
                bool have = false;                                       // w already holds the request (box_local)
                for (;;) {
                    if (box_local) { if (poll_box()) { have = true; break; } }
                    else if (__builtin_amdgcn_readfirstlane((int)(lane == 0 ? __hip_atomic_load(&box->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u)) != (int)served) break;
                    if (wall_clock64() - t0 > idle_ticks) {              // nothing to do: leave, unless a request slips in right now
                        if (lane == 0) { box_store(&stat->state, CTK_RES_LEAVING); __threadfence_system(); }
                        const int slipped = __builtin_amdgcn_readfirstlane((int)(lane == 0 ? (box_load(&box->req) != served ? 1u : 0u) : 0u));
                        if (slipped) { if (lane == 0) box_store(&stat->state, CTK_RES_RUNNING); continue; }   // (its tail may still be in flight: poll on)
                        leave = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                t_seen = wall_clock64();
                if (!leave) {
                    if (!have && lane < BOXW) w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(box) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    w = lane == 0 ? served + 1u : (lane == 1 ? (uint32_t)CTK_RES_CMD_EXIT : 0u);
                }
                if (lane < BOXW) reinterpret_cast<uint32_t*>(&req_s)[lane] = w;
                if (!box_local || leave) {
                    // hand it to the other workgroups; the release also publishes what the last step left (u_nom, u)
                    if (lane != 0 && lane < BOXW - 1)     // (not the last word: relay->upd is block 0's "update published" flag)
                        __hip_atomic_store(reinterpret_cast<uint32_t*>(relay) + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane == 0) __hip_atomic_store(&relay->req, w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                t_relayed = wall_clock64();
            } else {
                leave = 1;
                for (;;) {                                               // (wave-uniform: every lane takes every turn)
                    if (box_local && poll_box()) { leave = 0; break; }
                    const uint32_t rc = lane == 0 ? __hip_atomic_load(&relay->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    if (__builtin_amdgcn_readfirstlane((int)rc) == CTK_RES_CMD_EXIT) break;                      // block 0 has left
                    if (!box_local) {
                        const uint32_t rr = lane == 0 ? __hip_atomic_load(&relay->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : served;
                        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)rr) != served) { leave = 0; from_relay = 1; break; }
                    }
                    if (wall_clock64() - t0 > 4 * idle_ticks + 100000ull) break;       // never wait for ever
                    __builtin_amdgcn_s_sleep(1);
                }
                if (leave) w = lane == 1 ? (uint32_t)CTK_RES_CMD_EXIT : 0u;
                else if (from_relay && lane < BOXW) w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(relay) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane < BOXW) reinterpret_cast<uint32_t*>(&req_s)[lane] = w;
            }
        }
        __syncthreads();
        if (req_s.cmd != CTK_RES_CMD_STEP) break;        // workgroup-uniform
        served = req_s.req;
        // ---- are the prepared inputs those of THIS request?  (workgroup-uniform: every thread evaluates the same LDS words)
        bool hit = pre_ok && req_s.samples == pre_samples && (req_s.samples != nullptr || req_s.call == pre_call) && req_s.cur == pre_cur;
        if (hit && !req_s.dev_uprev) {
#pragma unroll
            for (int c = 0; c < C; ++c) hit = hit && __builtin_bit_cast(uint32_t, req_s.u_prev[c]) == __builtin_bit_cast(uint32_t, uown_s[c]);
        }
#pragma unroll
        for (int i = 0; i < S; ++i) a.s0[i] = req_s.s[i];
        fz.up.seq = req_s.seq;
        fz.up.u_nom_in = req_s.cur ? unom1 : unom0; fz.up.u_nom_out = req_s.cur ? unom0 : unom1;
        fz.up.w0_l = nullptr; fz.up.w1_l = nullptr; fz.up.un_l = nullptr; fz.up.i0_l = nullptr;
        // (no acquire fence on the way here: the request was read with uncached atomics, in program order behind the poll that saw its
        //  number; what the prepared inputs were formed from was acquired when they were formed.  An acquire HERE would empty the L2 that
        //  the inputs' preparation has just warmed for the merge tail: + 2 us measured.)
        if (!hit) {                                       // first step, an unexpected sample buffer / previous input: form them now
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // draws written by the host or by another kernel; u_nom / u of the last step
            samples = req_s.samples; u_nom = fz.up.u_nom_in; a.call = req_s.call;
#pragma unroll
            for (int c = 0; c < C; ++c) a.u_prev[c] = req_s.u_prev[c];
            a.u_prev_dev = req_s.dev_uprev ? d_u : nullptr;
            do_pre();
        }
        const unsigned long long c_post0 = clock64();
#ifdef CTK_RES_STAMPS
        c_req = c_post0;
        if ((blockIdx.x == 0 || blockIdx.x == 5) && threadIdx.x == 0) __hip_atomic_store(&stat->stamps[blockIdx.x ? 6 : 0], (uint32_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        do_post();
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            __hip_atomic_store(&stat->c_body, (uint32_t)(clock64() - c_post0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&stat->t_relay, (uint32_t)(t_relayed - t_seen) | (hit ? 0x80000000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&stat->t_body, (uint32_t)(wall_clock64() - t_relayed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // ---- while the host works on the result: the NEXT step's inputs.  They need this step's update (u_nom, u), which block 0
        //      publishes behind its merge: fence, then the request number in relay->upd (agent scope); the others wait for it — bounded
        __threadfence();
        __syncthreads();                                  // req_s is stable until the next wait; the LDS carve is free again
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) { __hip_atomic_store(&relay->upd, req_s.req, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); upd_ok_s = 1; }
        } else if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            int ok = 1;
            while (__hip_atomic_load(&relay->upd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != req_s.req) {
                if (wall_clock64() - t0 > 20000ull) { ok = 0; break; }          // 200 us: then this workgroup forms its inputs at the next request
                __builtin_amdgcn_s_sleep(1);
            }
            upd_ok_s = ok;
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        pre_ok = false;
        if (upd_ok_s && req_s.next_known) {
            if (threadIdx.x < C) uown_s[threadIdx.x] = __hip_atomic_load(d_u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pre_samples = req_s.next_samples; pre_call = req_s.call + 1u; pre_cur = req_s.cur ^ 1u;
            samples = pre_samples; u_nom = pre_cur ? unom1 : unom0; a.call = pre_call; a.u_prev_dev = d_u;
            do_pre();                                     // ends with a workgroup barrier: uown_s is visible too
            pre_ok = true;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        box_store(&stat->served, served);
        box_store(&stat->state, CTK_RES_LEFT);
    }
}

// ---------------------------------------------------------------------------------------------
// Throughput variant (ODE, N >= CTK_MPPI_THROUGHPUT_MIN_N): one wave per block, 64 trajectories, the
// inputs formed inline in the recurrence instead of through an LDS input buffer.  LDS per block is the
// sample tile only (13.5 KiB at P = 50 instead of 27 KiB), so ~11 recurrence waves are resident per CU
// instead of 5 and the VALU issue slots of every SIMD are covered by several waves.  Same arithmetic in
// the same order as the 4-wave kernel's prologue 2 + recurrence; block records merged by separate launches.
// ---------------------------------------------------------------------------------------------
template <bool LOG>
__global__ __launch_bounds__(64) void ctk_mppi_rollout_tp(RolloutArgs a, EnvK k, MppiK m, const float* __restrict__ samples,
                                                          const float* __restrict__ u_nom,
                                                          const InterpEntry* __restrict__ interp, float* __restrict__ parts) {
    extern __shared__ float lds[];
    const int P = a.P, H = a.H, ts = tile_stride(P);
    float* tile = lds;                 // [64][ts]
    float* e_s = tile + 64 * ts;       // [64]
    float* w0_s = e_s + 64;            // per-step tables [H] x 4
    float* w1_s = w0_s + H;
    float* un_s = w1_s + H;
    int* i0_s = reinterpret_cast<int*>(un_s + H);
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    const int n = row0 + lane;
    const bool valid = n < a.N;

    load_tile_early<64, 64>(tile, samples, a, row0, m.stdev, 0, [&] {
        for (int h = lane; h < H; h += 64) {
            const InterpEntry e = interp[h];
            i0_s[h] = e.i0; w0_s[h] = e.w0; w1_s[h] = e.w1;
            un_s[h] = u_nom[min(h + 1, H - 1)];
        }
    });
    __syncthreads();

    const float* my = tile + lane * ts;
    const bool ident = a.identity_interp != 0;
    // The input-only cost terms are quadratic forms of (du, u, u - u_prev): four running sums in the recurrence,
    //   correction (:154-155)  cc * (k_dd S(du^2) + R S(u du) + k_uu S(u^2)),   stage cost  ccR S(u^2) + ccrc S((u - u_prev)^2),
    // scaled once after the loop — 5 instructions per step instead of 13 (this kernel is VALU-bound: profiles/r01_mppi_largeN_pmc.txt).
    float s_dd = 0.0f, s_ud = 0.0f, s_uu = 0.0f, s_rc = 0.0f;
    float uprev = uniform_u_prev0(a);
    float amax = 0.0f;
    auto F_at = [&](int h) {
        float du;
        if (ident) du = my[h];
        else { const int i0 = i0_s[h]; du = my[i0] * w0_s[h] + my[i0 + 1] * w1_s[h]; }
        const float u = fminf(fmaxf(un_s[h] + du, a.lo[0]), a.hi[0]);
        const float dr = u - uprev;
        s_dd = fmaf(du, du, s_dd); s_ud = fmaf(u, du, s_ud); s_uu = fmaf(u, u, s_uu); s_rc = fmaf(dr, dr, s_rc);
        uprev = u;
        if constexpr (LOG) {
            if (valid) a.Q_out[(size_t)n * H + h] = u;
        }
        return k.u_max * u;
    };
    float J;
    if (k.intermediate_steps == 1) J = recur_ode_state_cost<LOG, false, true>(a, k, n, valid, F_at, &amax);
    else J = recur_ode_state_cost<LOG, true, false>(a, k, n, valid, F_at, &amax);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(amax <= CTK_SINCOS_FAST_LIMIT)) != 0, 0)) {
        s_dd = 0.0f; s_ud = 0.0f; s_uu = 0.0f; s_rc = 0.0f; uprev = uniform_u_prev0(a);
        J = recur_ode_state_cost<LOG, true, false>(a, k, n, valid, F_at, &amax);
    }
    const float corr = m.cc * (m.k_dd * s_dd + m.R * s_ud + m.k_uu * s_uu);
    const float cin = k.ccR * s_uu + k.ccrc_weight * s_rc;
    J = (J + cin) * a.inv_Hp1 + corr;
    if (valid) a.J[n] = J;

    const float rho = wave_min(valid ? J : INFINITY);
    const float e = valid ? expf(m.neg_inv_lbd * (J - rho)) : 0.0f;
    const float asum = wave_sum(e);
    e_s[lane] = e;
    __syncthreads();
    float* rec = parts + (size_t)blockIdx.x * (2 + P);
    if (lane == 0) { rec[0] = rho; rec[1] = asum; }
    for (int p = lane; p < P; p += 64) {
        float acc = 0.0f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) acc += e_s[r] * tile[r * ts + p];
        rec[2 + p] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Throughput variant that STREAMS the samples (identity interpolation, samples in a buffer).  With P = H the whole-horizon LDS
// tile is what limits residency (13.5 KiB per wave -> ~8 waves per CU, profiles/r01_mppi_largeN_pmc.txt), and at ~2 waves per
// SIMD the recurrence is bound by its own dependent-issue latency, not by VALU throughput (a wave64 VALU op issues in 2 cycles
// when enough waves share the SIMD; measured: tools/diag_pk_rate.hip).  Here the wave moves TPS_CK = 16 steps at a time:
// 64 rows x 16 steps are read as 64-byte row segments (16 lanes per row: every fetched line is used once), transposed through
// a 4.3 KiB wave-private LDS patch into 16 registers per lane, and the next chunk's loads are in flight while the current
// 16 steps run.  The epilogue's weighted column sums re-read the block's rows (row-coalesced; they are still in L2 / MALL).
// (Reading each lane's row directly, a dword per step, was tried first: 16x amplification of the L2->L1 traffic, no gain.)
// ---------------------------------------------------------------------------------------------
constexpr int TPS_CK = 16, TPS_LD = TPS_CK + 1;

// RES (round 3, opt-in — measured slower, see tps_resident): the patch becomes the wave's WHOLE [64][H+1] tile, filled chunk by chunk as
// the stream arrives, so the epilogue's weighted column sums read LDS instead of re-reading the block's rows.
template <bool LOG, bool RES = false>
__global__ __launch_bounds__(64) void ctk_mppi_rollout_tps(RolloutArgs a, EnvK k, MppiK m, const float* __restrict__ samples,
                                                           const float* __restrict__ u_nom, float* __restrict__ parts) {
    extern __shared__ float lds[];
    const int P = a.P, H = a.H;               // P == H (identity interpolation)
    const int LD = RES ? (H | 1) : TPS_LD;    // RES: an odd row stride >= H (conflict-free row walk)
    float* xs = lds;                          // [64][LD]: transposition patch, or (RES) the whole tile
    float* e_s = xs + 64 * LD;                // [64]
    float* un_s = e_s + 64;                   // [H] shifted nominal input
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    const int n = row0 + lane;
    const bool valid = n < a.N;
    const int last = a.N - 1 - row0;          // last valid row of the block (rows beyond N re-read it; their weight is 0)
    // element e = lane + 64 i of a chunk: row e / 16, column e % 16 -> lanes 16j .. 16j+15 read one 64-byte row segment
    const int crow = lane >> 4, ccol = lane & 15;
    const float* cbase = samples + (size_t)row0 * P;
    float nxt[TPS_CK], cur[TPS_CK];
    auto fetch = [&](int h0) {
        const int c = min(h0 + ccol, H - 1);
#pragma unroll
        for (int i = 0; i < TPS_CK; ++i) nxt[i] = cbase[(size_t)min(crow + 4 * i, last) * P + c];
    };
    auto transpose = [&](int h0) {            // nxt (chunk layout) -> cur (this lane's 16 steps); the patch / tile is the wave's own
        if constexpr (RES) {
            if (h0 + ccol < H) {
#pragma unroll
                for (int i = 0; i < TPS_CK; ++i) xs[(crow + 4 * i) * LD + h0 + ccol] = nxt[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int i = 0; i < TPS_CK; ++i) cur[i] = xs[lane * LD + min(h0 + i, H - 1)];
        } else {
#pragma unroll
            for (int i = 0; i < TPS_CK; ++i) xs[(crow + 4 * i) * TPS_LD + ccol] = nxt[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int i = 0; i < TPS_CK; ++i) cur[i] = xs[lane * TPS_LD + i];
        }
    };
    fetch(0);
    for (int h = lane; h < H; h += 64) un_s[h] = u_nom[min(h + 1, H - 1)];   // optimizer_mppi.py:184 (shift)
    __syncthreads();

    const float up0 = uniform_u_prev0(a);
    float s_dd = 0.0f, s_ud = 0.0f, s_uu = 0.0f, s_rc = 0.0f, uprev = up0;   // quadratic-form sums, as in ctk_mppi_rollout_tp
    float J;
    bool redo = k.intermediate_steps != 1;
    if (!redo) {
        State4 s{a.s0[0], a.s0[1], a.s0[2], a.s0[3]};
        float csum = 0.0f, am = 0.0f, e_dd = 0.0f, e_ud = 0.0f;
        float4* traj = nullptr;
        if constexpr (LOG) {
            if (a.traj_out) traj = reinterpret_cast<float4*>(a.traj_out) + (size_t)n * (H + 1);
        }
        for (int h0 = 0; h0 < H; h0 += TPS_CK) {
            transpose(h0);
            if (h0 + TPS_CK < H) fetch(h0 + TPS_CK);
#pragma unroll
            for (int i = 0; i < TPS_CK; ++i) {
                const int h = h0 + i;
                if (h < H) {
                    // u = clip(u_nom + stdev * eps) as one fma + one median; the sums over the RAW draw (scaled once after the loop)
                    const float eps = cur[i];
                    const float u = __builtin_amdgcn_fmed3f(fmaf(eps, m.stdev, un_s[h]), a.lo[0], a.hi[0]);
                    const float dr = u - uprev;
                    e_dd = fmaf(eps, eps, e_dd); e_ud = fmaf(u, eps, e_ud); s_uu = fmaf(u, u, s_uu); s_rc = fmaf(dr, dr, s_rc);
                    uprev = u;
                    float sn, cs;
                    ctk_sincosf_fast(s.th, &sn, &cs);
                    am = fmaxf(am, fabsf(s.th));
                    if constexpr (LOG) {
                        if (valid) a.Q_out[(size_t)n * H + h] = u;
                        if (valid && traj) traj[h] = make_float4(s.x, s.v, s.th, s.om);
                    }
                    ode_cost_substep(k, s, k.u_max * u, sn, cs, csum);
                }
            }
        }
        if constexpr (LOG) {
            if (valid && traj) traj[H] = make_float4(s.x, s.v, s.th, s.om);
        }
        s_dd = m.stdev * m.stdev * e_dd; s_ud = m.stdev * e_ud;
        J = csum + terminal_cost(k, s);
        redo = __builtin_amdgcn_ballot_w64(!(am <= CTK_SINCOS_FAST_LIMIT)) != 0;
    }
    if (__builtin_expect(redo, 0)) {   // Euler sub-steps, or an angle beyond the fast sincos range somewhere in the wave: checked recurrence
        s_dd = 0.0f; s_ud = 0.0f; s_uu = 0.0f; s_rc = 0.0f; uprev = up0;
        const float* my = samples + (size_t)min(n, a.N - 1) * P;
        float amax;
        auto F_at = [&](int h) {
            const float du = my[h] * m.stdev;
            const float u = fminf(fmaxf(un_s[h] + du, a.lo[0]), a.hi[0]);
            const float dr = u - uprev;
            s_dd = fmaf(du, du, s_dd); s_ud = fmaf(u, du, s_ud); s_uu = fmaf(u, u, s_uu); s_rc = fmaf(dr, dr, s_rc);
            uprev = u;
            if constexpr (LOG) {
                if (valid) a.Q_out[(size_t)n * H + h] = u;
            }
            return k.u_max * u;
        };
        J = recur_ode_state_cost<LOG, true, false>(a, k, n, valid, F_at, &amax);
    }
    J = (J + k.ccR * s_uu + k.ccrc_weight * s_rc) * a.inv_Hp1 + m.cc * (m.k_dd * s_dd + m.R * s_ud + m.k_uu * s_uu);
    if (valid) a.J[n] = J;

    const float rho = wave_min(valid ? J : INFINITY);
    const float e = valid ? expf(m.neg_inv_lbd * (J - rho)) : 0.0f;
    const float asum = wave_sum(e);
    e_s[lane] = e;
    __syncthreads();
    float* rec = parts + (size_t)blockIdx.x * (2 + P);
    if (lane == 0) { rec[0] = rho; rec[1] = asum; }
    // b[p] = sum_r e_r eps[r][p] over the block's 64 rows, in the CHUNK layout of the main loop: lane (row group crow, column ccol)
    // holds rows crow, crow+4, ... of column h0 + ccol — 16 row-segment loads all in flight, 16 FMAs against the rows' weights (LDS
    // broadcast), two cross-group adds; every lane busy.  (Each lane walking one column over 64 rows: 16 of the kernel's 113 us at
    // N = 2^20 — 50 of 64 lanes active, 8 loads in flight.)  The rows are still in L2 / MALL.  Rows beyond N carry weight 0.
    auto fetch_into = [&](float (&dst)[TPS_CK], int h0) {
        const int c = min(h0 + ccol, H - 1);
#pragma unroll
        for (int i = 0; i < TPS_CK; ++i) dst[i] = cbase[(size_t)min(crow + 4 * i, last) * P + c];
    };
    auto column_sums = [&](const float (&v)[TPS_CK], int h0) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < TPS_CK; ++i) acc = fmaf(e_s[crow + 4 * i], v[i], acc);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (lane < TPS_CK && h0 + lane < H) rec[2 + h0 + lane] = acc * m.stdev;
    };
    if constexpr (RES) {                                  // the tile is resident: no second pass through memory
        for (int h0 = 0; h0 < H; h0 += TPS_CK) {
            const int c = min(h0 + ccol, H - 1);
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < TPS_CK; ++i) acc = fmaf(e_s[crow + 4 * i], xs[(crow + 4 * i) * LD + c], acc);
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            if (lane < TPS_CK && h0 + lane < H) rec[2 + h0 + lane] = acc * m.stdev;
        }
        return;
    }
    fetch_into(nxt, 0);                                   // two chunks in flight: one being summed, the next one landing
    for (int h0 = 0; h0 < H; h0 += 2 * TPS_CK) {
        if (h0 + TPS_CK < H) fetch_into(cur, h0 + TPS_CK);
        column_sums(nxt, h0);
        if (h0 + 2 * TPS_CK < H) fetch_into(nxt, h0 + 2 * TPS_CK);
        if (h0 + TPS_CK < H) column_sums(cur, h0 + TPS_CK);
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
bool ctk_mppi_uses_throughput_kernel(int pred, int N) { return pred == CTK_PRED_ODE && N >= CTK_MPPI_THROUGHPUT_MIN_N; }

// MLP at sizes that leave SIMDs idle -> the pair form
static int kernel_pred(int pred, int N) {
    static const bool off = getenv("CTK_MPPI_NO_PAIR") != nullptr;   // diagnostic switch: A/B the two forms
    return (pred == CTK_PRED_MLP && N <= CTK_MPPI_PAIR_MAX_N && !off) ? CTK_PRED_MLP_PAIR : pred;
}

// which throughput form a launch takes: 2 = streaming (ctk_mppi_rollout_tps: sample buffer + identity interpolation), 1 = whole-horizon
// tile (ctk_mppi_rollout_tp: in-kernel sampler, or P < H), 0 = not a throughput launch.  ONE predicate for the launcher and the name.
static int throughput_form(int pred, int N, bool have_samples, bool identity_interp) {
    if (!ctk_mppi_uses_throughput_kernel(pred, N)) return 0;
    static const bool no_direct = getenv("CTK_MPPI_TP_TILE") != nullptr;   // diagnostic switch: A/B the two throughput forms
    return (have_samples && identity_interp && !no_direct) ? 2 : 1;
}

// the streaming kernel's resident-tile form (RES) is opt-in: measured SLOWER than the patch + re-read form (N = 2^20: 112.4 vs 105.4 us) —
// the re-read hits L2 / MALL, and 13 KiB of LDS per wave costs a twelfth of the residency.  CTK_MPPI_TPS_RESIDENT: diagnostic switch (A/B)
static bool tps_resident(int H) {
    static const bool on = getenv("CTK_MPPI_TPS_RESIDENT") != nullptr;
    return on && (size_t)64 * (H | 1) * sizeof(float) <= 16 * 1024;
}

// name of the kernel ctk_launch_mppi_rollout runs for these arguments (identity_interp as RolloutArgs carries it: period 1 AND P == H)
const char* ctk_mppi_rollout_name(int pred, bool log, int N, bool identity_interp, bool have_samples, bool p2p, int H) {
    const int tf = throughput_form(pred, N, have_samples, identity_interp);
    if (tf == 2) return ctk_kernel_name("ctk_mppi_rollout_tps<%4$s, %5$s>", 0, 0, 0, log ? "true" : "false", (H > 0 && tps_resident(H)) ? "true" : "false");
    if (tf == 1) return log ? "ctk_mppi_rollout_tp<true>" : "ctk_mppi_rollout_tp<false>";
    // template arguments <environment, predictor form, materialise, peer-to-peer tail>: what rocprofv3's kernel trace shows
    const int kp = pred == CTK_PRED_ODE ? CTK_PRED_ODE : pred == CTK_PRED_GRU ? CTK_PRED_GRU : kernel_pred(pred, N);
    return ctk_kernel_name("ctk_mppi_rollout<0, %d, %4$s, %5$s>", kp, 0, 0, log ? "true" : "false", p2p ? "true" : "false");
}

int ctk_mppi_num_blocks(int N, int pred) {
    const int tr = mppi_traj(kernel_pred(pred, N));
    return (N + tr - 1) / tr;
}


// C control inputs (CartPole kernels: 1): P*C sample columns, H*C inputs per trajectory
size_t ctk_mppi_rollout_lds(int P, int H, int pred, int N, int C) {
    const int kp = kernel_pred(pred, N);
    const size_t roll = (size_t)(mppi_carve_floats(P * C, H * C, mppi_traj(kp), H) + (kp == CTK_PRED_GRU ? GRU_EX_FLOATS : kp == CTK_PRED_MLP_PAIR ? 2 * MLP_PAIR_EX : 0)) * sizeof(float);
    const size_t tail = merge_lds(P * C, CTK_MPPI_FUSE_MAX_BLOCKS);
    return roll > tail ? roll : tail;
}

// LDS of one launch: the rollout carve, or the fused tail's (staged) merge scratch if larger
static size_t rollout_launch_lds(int P, int H, int pred, int N, int blocks, int* stage_ok, int C = 1) {
    size_t lds = ctk_mppi_rollout_lds(P, H, pred, N, C);
    *stage_ok = 0;
    P *= C;
    if (ctk_ll_records_ok(blocks, P) && merge_can_stage(P, blocks)) {
        const size_t st = merge_lds_staged(P, blocks);
        if (st > lds) lds = st;
        *stage_ok = 1;
    }
    return lds;
}

hipError_t ctk_launch_mppi_rollout(hipStream_t st, int pred, const RolloutArgs& a, const EnvK& k, const MppiK& m,
                                   const float* samples, const float* u_nom, const float* wperm, float* parts, bool log,
                                   const MppiFuse& fuse, hipEvent_t e0, hipEvent_t e1, const char** ran) {
    const dim3 grid(ctk_mppi_num_blocks(a.N, pred)), block(MPPI_BLOCK);
    if (ran) *ran = ctk_mppi_rollout_name(pred, log, a.N, a.identity_interp != 0, samples != nullptr, fuse.mode == 3, a.H);
    if (const int tf = throughput_form(pred, a.N, samples != nullptr, a.identity_interp != 0)) {
        if (tf == 2) {
            if (tps_resident(a.H)) {
                const size_t lds_r = (size_t)(64 * (a.H | 1) + 64 + a.H) * sizeof(float);
                if (log) CTK_LAUNCH((ctk_mppi_rollout_tps<true, true>), grid, dim3(64), lds_r, st, e0, e1, a, k, m, samples, u_nom, parts);
                else CTK_LAUNCH((ctk_mppi_rollout_tps<false, true>), grid, dim3(64), lds_r, st, e0, e1, a, k, m, samples, u_nom, parts);
                return hipGetLastError();
            }
            const size_t lds_d = (size_t)(64 * TPS_LD + 64 + a.H) * sizeof(float);
            if (log) CTK_LAUNCH((ctk_mppi_rollout_tps<true>), grid, dim3(64), lds_d, st, e0, e1, a, k, m, samples, u_nom, parts);
            else CTK_LAUNCH((ctk_mppi_rollout_tps<false>), grid, dim3(64), lds_d, st, e0, e1, a, k, m, samples, u_nom, parts);
            return hipGetLastError();
        }
        const size_t lds_tp = (size_t)(64 * tile_stride(a.P) + 64 + 4 * a.H) * sizeof(float);
        if (log) CTK_LAUNCH((ctk_mppi_rollout_tp<true>), grid, dim3(64), lds_tp, st, e0, e1, a, k, m, samples, u_nom, a.interp, parts);
        else CTK_LAUNCH((ctk_mppi_rollout_tp<false>), grid, dim3(64), lds_tp, st, e0, e1, a, k, m, samples, u_nom, a.interp, parts);
        return hipGetLastError();
    }
    FuseArgs fz{};
    size_t lds = rollout_launch_lds(a.P, a.H, pred, a.N, (int)grid.x, &fz.stage_ok);
    if (fuse.mode == 3) {   // the tail also stages the `world` records of the exchange
        const size_t need = merge_lds_staged(a.P, fuse.p2p_world);
        if (need > lds) lds = need;
    }
    fz.mode = fuse.mode; fz.counter = fuse.counter; fz.out_rec = fuse.out_rec;
    fz.ll = (fuse.mode != 0 && fz.stage_ok) ? fuse.ll : nullptr;
    fz.p2p = static_cast<const P2PArgs*>(fuse.p2p); fz.p2p_seq = fuse.p2p_seq;
    fz.up = MppiUpdateArgs{nullptr, nullptr, nullptr, nullptr, a.H, a.interp, u_nom, fuse.u_nom_out, a.lo[0], a.hi[0], fuse.u_dev, fuse.u_host, fuse.seq};
#define CTK_MPPI_LAUNCH(PREDV, LOGV, P2PV) CTK_LAUNCH((ctk_mppi_rollout<CTK_ENV_CARTPOLE, PREDV, LOGV, P2PV>), grid, block, lds, st, e0, e1, samples, u_nom, a.interp, wperm, parts, a.N, a.H, a.P, a.p_magic, a, k, m, fz)
#define CTK_MPPI_LAUNCH_PRED(PREDV)                                                            \
    do {                                                                                       \
        if (fuse.mode == 3) { if (log) CTK_MPPI_LAUNCH(PREDV, true, true); else CTK_MPPI_LAUNCH(PREDV, false, true); } \
        else { if (log) CTK_MPPI_LAUNCH(PREDV, true, false); else CTK_MPPI_LAUNCH(PREDV, false, false); }             \
    } while (0)
    if (pred == CTK_PRED_ODE) CTK_MPPI_LAUNCH_PRED(CTK_PRED_ODE);
    else if (kernel_pred(pred, a.N) == CTK_PRED_MLP_PAIR) CTK_MPPI_LAUNCH_PRED(CTK_PRED_MLP_PAIR);
    else if (pred == CTK_PRED_MLP) CTK_MPPI_LAUNCH_PRED(CTK_PRED_MLP);
    else CTK_MPPI_LAUNCH_PRED(CTK_PRED_GRU);
#undef CTK_MPPI_LAUNCH_PRED
#undef CTK_MPPI_LAUNCH
    return hipGetLastError();
}

// The same 4-wave kernel for any environment's analytic predictor (ctk_env.h): a.P = inducing points, a.C = control inputs,
// a.lo / a.hi per input; the kernel constants are derived here from the environment's parameter table.  fuse as above
// (modes 0, 1, 2; the peer-to-peer tail is CartPole's).
hipError_t ctk_launch_mppi_rollout_env(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a, const MppiK& m,
                                       const float* samples, const float* u_nom, float* parts, bool log, const MppiFuse& fuse,
                                       hipEvent_t e0, hipEvent_t e1) {
    const dim3 grid((a.N + MPPI_TRAJ - 1) / MPPI_TRAJ), block(MPPI_BLOCK);
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        const int PC = a.P * E::C;
        const typename E::K k = E::derive(params, dt, isteps);
        FuseArgs fz{};
        const size_t lds = rollout_launch_lds(a.P, a.H, CTK_PRED_ODE, a.N, (int)grid.x, &fz.stage_ok, E::C);
        fz.mode = fuse.mode; fz.counter = fuse.counter; fz.out_rec = fuse.out_rec;
        fz.ll = (fuse.mode != 0 && fz.stage_ok) ? fuse.ll : nullptr;
        fz.up = MppiUpdateArgs{nullptr, nullptr, nullptr, nullptr, a.H, a.interp, u_nom, fuse.u_nom_out, a.lo[0], a.hi[0], fuse.u_dev, fuse.u_host, fuse.seq};
        fz.up.C = E::C;
        for (int c = 0; c < E::C; ++c) { fz.up.lo_c[c] = a.lo[c]; fz.up.hi_c[c] = a.hi[c]; }
        const uint32_t pmagic = PC >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)PC - 1) / (uint64_t)PC) : 0u;
        if (log) CTK_LAUNCH((ctk_mppi_rollout<EV, CTK_PRED_ODE, true, false>), grid, block, lds, st, e0, e1, samples, u_nom, a.interp, (const float*)nullptr, parts, a.N, a.H, a.P, pmagic, a, k, m, fz);
        else CTK_LAUNCH((ctk_mppi_rollout<EV, CTK_PRED_ODE, false, false>), grid, block, lds, st, e0, e1, samples, u_nom, a.interp, (const float*)nullptr, parts, a.N, a.H, a.P, pmagic, a, k, m, fz);
    });
    return hipGetLastError();
}
hipError_t ctk_launch_mppi_resident(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a, const MppiK& m,
                                    float* u_nom0, float* u_nom1, float* parts, const MppiFuse& fuse, const CtkResidentBox* box_dev, int box_local,
                                    CtkResidentStat* stat_dev, CtkResidentBox* relay, double idle_us, uint32_t first_req, void* args_dev, void* args_host) {
    const dim3 grid((a.N + MPPI_TRAJ - 1) / MPPI_TRAJ), block(MPPI_BLOCK);
    CTK_FOR_ENV(env, EV, {
        using E = Env<EV>;
        const int PC = a.P * E::C;
        const typename E::K k = E::derive(params, dt, isteps);
        FuseArgs fz{};
        const size_t lds = rollout_launch_lds(a.P, a.H, CTK_PRED_ODE, a.N, (int)grid.x, &fz.stage_ok, E::C);
        if (!fz.stage_ok || fuse.ll == nullptr) return hipErrorInvalidValue;          // the resident form is the {value, seq} hand-off only
        fz.mode = 1; fz.ll = fuse.ll;
        fz.up = MppiUpdateArgs{nullptr, nullptr, nullptr, nullptr, a.H, a.interp, u_nom0, u_nom1, a.lo[0], a.hi[0], fuse.u_dev, fuse.u_host, 0u};
        fz.up.C = E::C;
        for (int c = 0; c < E::C; ++c) { fz.up.lo_c[c] = a.lo[c]; fz.up.hi_c[c] = a.hi[c]; }
        const uint32_t pmagic = PC >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)PC - 1) / (uint64_t)PC) : 0u;
        const unsigned long long ticks = (unsigned long long)(idle_us * 100.0);        // wall_clock64: 100 MHz
        using RA = ResidentArgs<typename E::K>;
        static_assert(sizeof(RA) <= CTK_RES_ARGS_BYTES, "resident argument block");
        RA* hostra = static_cast<RA*>(args_host);               // staging that outlives the asynchronous copy (the handle's)
        *hostra = RA{a.interp, parts, a.N, a.H, a.P, pmagic, u_nom0, u_nom1, a, k, m, fz};
        const hipError_t ce = hipMemcpyAsync(args_dev, hostra, sizeof(RA), hipMemcpyHostToDevice, st);
        if (ce != hipSuccess) return ce;
        hipLaunchKernelGGL((ctk_mppi_resident<EV>), grid, block, lds, st, static_cast<const RA*>(args_dev), box_dev, box_local, stat_dev, relay, ticks, first_req);
    });
    return hipGetLastError();
}
const char* ctk_mppi_resident_name(int env) { return ctk_kernel_name("ctk_mppi_resident<%d>", env); }

size_t ctk_mppi_rollout_env_lds(int env, int P, int H, int N) {
    int C = 1;
    CTK_FOR_ENV(env, EV, { C = Env<EV>::C; });
    int stage_ok;
    return rollout_launch_lds(P, H, CTK_PRED_ODE, N, (N + MPPI_TRAJ - 1) / MPPI_TRAJ, &stage_ok, C);
}
const char* ctk_mppi_rollout_env_name(int env, bool log) {
    return ctk_kernel_name("ctk_mppi_rollout<%d, 0, %4$s, false>", env, 0, 0, log ? "true" : "false");
}

hipError_t ctk_launch_mppi_merge_partial(hipStream_t st, const float* parts, int n_parts, int per_block, int P,
                                         float neg_inv_lbd, float* out_rec) {
    const int blocks = (n_parts + per_block - 1) / per_block;
    const bool stage = merge_can_stage(P, per_block);
    hipLaunchKernelGGL(ctk_mppi_merge<false>, dim3(blocks), dim3(MERGE_BLOCK), stage ? merge_lds_staged(P, per_block) : merge_lds(P, per_block),
                       st, parts, n_parts, per_block, P, neg_inv_lbd, out_rec, MppiUpdateArgs{}, stage ? 1 : 0);
    return hipGetLastError();
}

size_t ctk_p2p_buffer_floats(int world, int P) { return (size_t)2 * world * (2 + P) + (size_t)2 * world; }
size_t ctk_p2p_args_bytes() { return sizeof(P2PArgs); }
void ctk_p2p_fill_args(void* dst, float* const* bufs, int rank, int world, int P, uint32_t* err_host, double timeout_s) {
    P2PArgs x{};
    for (int w = 0; w < world; ++w) x.bufs[w] = bufs[w];
    x.rank = rank; x.world = world; x.rs = 2 + P; x.seq = 0; x.err_host = err_host;
    x.timeout_ticks = (unsigned long long)(timeout_s * 1.0e8);
    std::memcpy(dst, &x, sizeof(x));
}
bool ctk_mppi_fusable(int P, int blocks, bool have_ll) {
    if (blocks <= CTK_MPPI_FUSE_MAX_BLOCKS) return true;
    return have_ll && ctk_ll_records_ok(blocks, P) && merge_can_stage(P, blocks);
}
bool ctk_p2p_can_fuse(int P, int world, int blocks) {
    return ctk_ll_records_ok(blocks, P) && merge_can_stage(P, blocks) && merge_can_stage(P, world);
}

hipError_t ctk_launch_mppi_p2p_exchange(hipStream_t st, float* const* bufs, int rank, int world, int P, uint32_t p2p_seq,
                                        uint32_t* err_host, double timeout_s, float neg_inv_lbd, int H, const InterpEntry* interp,
                                        const float* u_nom_in, float* u_nom_out, float lo, float hi, float* u_dev, float* u_host,
                                        uint32_t seq) {
    P2PArgs x{};
    for (int w = 0; w < world; ++w) x.bufs[w] = bufs[w];
    x.rank = rank; x.world = world; x.rs = 2 + P; x.seq = p2p_seq; x.err_host = err_host;
    x.timeout_ticks = (unsigned long long)(timeout_s * 1.0e8);
    const bool stage = merge_can_stage(P, world);
    hipLaunchKernelGGL(ctk_mppi_p2p_exchange, dim3(1), dim3(MERGE_BLOCK), stage ? merge_lds_staged(P, world) : merge_lds(P, world), st,
                       x, P, neg_inv_lbd,
                       MppiUpdateArgs{nullptr, nullptr, nullptr, nullptr, H, interp, u_nom_in, u_nom_out, lo, hi, u_dev, u_host, seq},
                       stage ? 1 : 0);
    return hipGetLastError();
}

hipError_t ctk_launch_mppi_update(hipStream_t st, const float* parts, int n_parts, int P, float neg_inv_lbd, int H,
                                  const InterpEntry* interp, const float* u_nom_in, float* u_nom_out, float lo, float hi,
                                  float* u_dev, float* u_host, uint32_t seq) {
    const bool stage = merge_can_stage(P, n_parts);
    hipLaunchKernelGGL(ctk_mppi_merge<true>, dim3(1), dim3(MERGE_BLOCK), stage ? merge_lds_staged(P, n_parts) : merge_lds(P, n_parts), st,
                       parts, n_parts, n_parts, P, neg_inv_lbd, (float*)nullptr,
                       MppiUpdateArgs{nullptr, nullptr, nullptr, nullptr, H, interp, u_nom_in, u_nom_out, lo, hi, u_dev, u_host, seq}, stage ? 1 : 0);
    return hipGetLastError();
}
