// ctk_launch.h — host-callable launchers implemented in the kernel translation units.
#pragma once
#include <hip/hip_ext.h>
#include "ctk_common.h"

// Launch with the dispatch's own begin/end timestamps attached to (ev_start, ev_stop) when given
// (hipExtLaunchKernelGGL): the measured interval is the kernel, not the queue around it.
#define CTK_LAUNCH(kernel, grid, block, lds, st, e0, e1, ...)                                         \
    do {                                                                                              \
        if (e0) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0, e1, 0, __VA_ARGS__);          \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                           \
    } while (0)

// kernel names as rocprofv3's trace prints them, formatted once per (template, arguments) and kept for the process's lifetime
// (ctk_dominant_kernel returns the pointer); any number of environments
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
inline const char* ctk_kernel_name(const char* fmt, int a = 0, int b = 0, int c = 0, const char* s0 = "", const char* s1 = "") {
    static std::map<std::string, std::string> names;
    static std::mutex mu;
    char buf[160];
    std::snprintf(buf, sizeof buf, fmt, a, b, c, s0, s1);
    std::lock_guard<std::mutex> lock(mu);
    return names.emplace(buf, buf).first->second.c_str();
}

// ---- ctk_mppi.hip ---------------------------------------------------------------------------
const char* ctk_mppi_rollout_name(int pred, bool log, int N, bool identity_interp = false, bool have_samples = true, bool p2p = false, int H = 0);
int ctk_mppi_num_blocks(int N, int pred);   // workgroups = block records of one rollout launch (64 trajectories each; GRU: 16)
bool ctk_mppi_uses_throughput_kernel(int pred, int N);
size_t ctk_mppi_rollout_lds(int P, int H, int pred = 0, int N = 1 << 30, int C = 1);
// MLP, N <= CTK_MPPI_PAIR_MAX_N: workgroups of 32 trajectories with a tile's network step shared by two waves (ctk_mlp.h:
// mlp_step_pair) — at these sizes 16 trajectories per wave leave half of the chip's SIMDs idle.
constexpr int CTK_MPPI_PAIR_MAX_N = 8192;
constexpr int CTK_PRED_MLP_PAIR = 3;   // kernel-variant id (internal to the launchers, not a ctk_predictor value)
// wperm: per-lane permuted MLP weights (ctk_api.hip: permute_mlp_weights), nullptr for the ODE predictor
// In-launch merge by the last block to finish (<= CTK_MPPI_FUSE_MAX_BLOCKS blocks).
constexpr int CTK_MPPI_FUSE_MAX_BLOCKS = 64;     // ticket form: beyond this the last block's serial record fetch costs more than a launch
constexpr int CTK_MPPI_FUSE_MAX_BLOCKS_LL = 128; // {value, seq} form (records staged in LDS): measured 26.7 vs 29.5 us at 128 blocks, 30.8 vs 29.1 at 256
// ... of FULL-width records (cfg2: 2 + 50 words).  What the tail block pays for is WORDS polled and staged, not records: a shard of
// BASELINE configs[4] (N 8 192, period 10: 256 records of 2 + 11 words = 3 328 words, half of cfg2's 128 x 52) hands over in-launch as
// well (round 4: ctk_mppi_step_begin at that size was rollout + merge<false> + merge<true>, three launches)
constexpr int CTK_MPPI_FUSE_MAX_BLOCKS_LL_NARROW = 512, CTK_MPPI_FUSE_MAX_WORDS_LL_NARROW = 4096;
inline bool ctk_ll_records_ok(int blocks, int cols) {
    return blocks <= CTK_MPPI_FUSE_MAX_BLOCKS_LL ||
           (blocks <= CTK_MPPI_FUSE_MAX_BLOCKS_LL_NARROW && blocks * (2 + cols) <= CTK_MPPI_FUSE_MAX_WORDS_LL_NARROW);
}
// can a rollout launch of `blocks` workgroups merge and update in-launch?  (have_ll: the handle owns the LL word buffer)
bool ctk_mppi_fusable(int P, int blocks, bool have_ll);
// From this many rollouts on (ODE predictor) the latency-oriented 4-wave block gives way to the
// throughput-oriented single-wave block (half the LDS, 2x the resident recurrence waves per CU).
constexpr int CTK_MPPI_THROUGHPUT_MIN_N = 32768;
// The resident form's mailbox: one request per MPPI step, written by the HOST only — the payload first, then `req`.  Where it lives:
// in fine-grained DEVICE memory that the host stores into through the PCIe BAR (posted writes + sfence; every workgroup then polls LOCAL
// memory: no PCIe read on the request path), or — where the host cannot reach device memory — in pinned host memory, polled by ONE
// workgroup that relays it to the others through device memory (`relay`).  What the DEVICE reports goes to CtkResidentStat in pinned host
// memory.  ctk_mppi.hip: ctk_mppi_resident.
enum { CTK_RES_CMD_STEP = 0, CTK_RES_CMD_EXIT = 1 };
enum { CTK_RES_IDLE = 0, CTK_RES_RUNNING = 1, CTK_RES_LEAVING = 2, CTK_RES_LEFT = 3 };
struct CtkResidentBox {
    uint32_t req;                       // request number (monotonic within a handle); written LAST
    uint32_t cmd;                       // CTK_RES_CMD_*
    uint32_t seq;                       // the step's sequence number: tag of its record words and of {u, seq}
    uint32_t call;                      // Philox position of the step (in-kernel sampler)
    uint32_t cur;                       // which u_nom buffer holds the current plan
    uint32_t dev_uprev;                 // 1: the previous input is the optimizer's own last output on the device (u_prev == NULL at the API)
    const float* samples;               // this step's draws (device pointer), or nullptr
    float s[CTK_MAX_STATES];
    float u_prev[CTK_MAX_INPUTS];
    const float* next_samples;          // the host's guess at the NEXT step's draws (nullptr with next_known: the in-kernel sampler): the kernel forms
    uint32_t next_known;                // that step's inputs while the host works on this one; a wrong guess costs nothing but that work
    uint32_t tail;                      // == req, written with the payload BEFORE req: a reader that finds req == tail in ONE pass over the box has a
    uint32_t pad;                       //    consistent request (writes arrive in order), and needs no second round trip for the payload
    uint32_t upd;                       // relay only: request number whose update (u_nom, u) block 0 has published
};
struct CtkResidentStat {                // device -> host
    uint32_t state;                     // CTK_RES_*
    uint32_t served;                    // last request taken, written when the kernel leaves
    uint32_t t_relay, t_body;           // diagnostics: wall-clock ticks (10 ns) block 0 spent fetching / relaying the last request, and in its step
    uint32_t c_body;                    // ... and that step in shader-clock cycles (clock64)
    uint32_t stamps[12];                // diagnostic builds (-DCTK_RES_STAMPS): the body's STAMP(i) points, shader cycles since the request
};

struct MppiFuse {
    int mode = 0;              // 0 records only, 1 merge + update u_nom/u, 2 merge into ONE record (sharded step_begin),
                               // 3 = 2 + peer-to-peer exchange + update in the same launch (ctk_p2p_step)
    unsigned* counter = nullptr;
    float* out_rec = nullptr;  // mode 2
    unsigned long long* ll = nullptr;   // [blocks][2+P] {value, seq} words: low-latency hand-off (else ticket + fetch)
    const void* p2p = nullptr;          // mode 3 (needs ll): device-resident P2PArgs; block 0 exchanges with the peers and updates
    uint32_t p2p_seq = 0;
    int p2p_world = 0;
    float* u_nom_out = nullptr, *u_dev = nullptr, *u_host = nullptr;   // mode 1
    uint32_t seq = 0;          // sequence number published with u (mode 1)
};
hipError_t ctk_launch_mppi_rollout(hipStream_t st, int pred, const RolloutArgs& a, const EnvK& k, const MppiK& m,
                                   const float* samples, const float* u_nom, const float* wperm, float* parts, bool log,
                                   const MppiFuse& fuse, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                                   const char** ran = nullptr);   // *ran: name of the kernel this call launched
// the 4-wave kernel for any environment's analytic predictor (a.P inducing points, a.C inputs; constants derived from `params`)
hipError_t ctk_launch_mppi_rollout_env(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a, const MppiK& m,
                                       const float* samples, const float* u_nom, float* parts, bool log, const MppiFuse& fuse,
                                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
size_t ctk_mppi_rollout_env_lds(int env, int P, int H, int N);
// the resident form of the same kernel (fuse mode 1 with the {value, seq} hand-off only): launched once, serves requests from `box`
hipError_t ctk_launch_mppi_resident(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a, const MppiK& m,
                                    float* u_nom0, float* u_nom1, float* parts, const MppiFuse& fuse, const CtkResidentBox* box_dev, int box_local,
                                    CtkResidentStat* stat_dev, CtkResidentBox* relay, double idle_us, uint32_t first_req, void* args_dev, void* args_host);
constexpr size_t CTK_RES_ARGS_BYTES = 1024;   // device + host staging block for the resident kernel's argument struct
const char* ctk_mppi_resident_name(int env);
const char* ctk_mppi_rollout_env_name(int env, bool log);
hipError_t ctk_launch_mppi_merge_partial(hipStream_t st, const float* parts, int n_parts, int per_block, int P,
                                         float neg_inv_lbd, float* out_rec);
// direct peer-to-peer record exchange + merge + update (ctk_mppi.hip: ctk_mppi_p2p_exchange)
constexpr int CTK_P2P_MAX_WORLD = 16;
size_t ctk_p2p_buffer_floats(int world, int P);
size_t ctk_p2p_args_bytes();   // sizeof the device-resident exchange description
// fills `dst` (host memory, ctk_p2p_args_bytes()) for upload
void ctk_p2p_fill_args(void* dst, float* const* bufs, int rank, int world, int P, uint32_t* err_host, double timeout_s);
bool ctk_p2p_can_fuse(int P, int world, int blocks);   // the rollout launch's LDS can stage `world` records too
hipError_t ctk_launch_mppi_p2p_exchange(hipStream_t st, float* const* bufs, int rank, int world, int P, uint32_t p2p_seq,
                                        uint32_t* err_host, double timeout_s, float neg_inv_lbd, int H, const InterpEntry* interp,
                                        const float* u_nom_in, float* u_nom_out, float lo, float hi, float* u_dev, float* u_host,
                                        uint32_t seq);
hipError_t ctk_launch_mppi_update(hipStream_t st, const float* parts, int n_parts, int P, float neg_inv_lbd, int H,
                                  const InterpEntry* interp, const float* u_nom_in, float* u_nom_out, float lo, float hi,
                                  float* u_dev, float* u_host, uint32_t seq);

// up to four device-to-device copies in ONE launch (the step log, ctk_api.hip:log_step); n_i floats each, n_i == 0 skips
struct CopyJob { const float* src; float* dst; unsigned n; };
hipError_t ctk_launch_copy4(hipStream_t st, const CopyJob (&jobs)[4]);

// ---- ctk_sampled.hip : u[n,h] = clip(base[h] + sample[n,h] * scale[h]) rollouts, selection ----
const char* ctk_affine_rollout_name(int pred, bool log);
size_t ctk_affine_rollout_lds(int H, int pred = 0);
// GRU: hidden <- cell(hidden, [s, u]) for the carried state behind the weight table (ctk_gru.h); u_dev NULL: u_val
hipError_t ctk_launch_gru_advance(hipStream_t st, const float* s, const float* u_dev, float u_val, float* wperm);
// samples [N,H] (device) or nullptr (Philox, rng_kind 0 normal / 1 uniform); base/scale [H] device.
// bst (ODE predictor only, <= CTK_AFFINE_BEST_MAX_BLOCKS workgroups): in-launch arg-min — block 0 picks the cheapest
// rollout under (J, index), writes its index to idx_out[0] and publishes its first input {u, seq}
struct AffineBest { unsigned long long* ll; uint32_t seq; float* u_dev; float* u_host; int* idx_out; };
constexpr int CTK_AFFINE_BEST_MAX_BLOCKS = 128;
int ctk_affine_rollout_blocks(int pred, int N);
hipError_t ctk_launch_affine_rollout(hipStream_t st, int pred, const RolloutArgs& a, const EnvK& k, const float* samples,
                                     int rng_kind, const float* base, const float* scale, const float* wperm, bool log,
                                     hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const AffineBest* bst = nullptr);
// the 4-wave kernel for any environment's analytic predictor (constants derived from `params`; bst: 2 + C words per workgroup)
hipError_t ctk_launch_affine_rollout_env(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a,
                                         const float* samples, int rng_kind, const float* base, const float* scale, bool log,
                                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const AffineBest* bst = nullptr);
size_t ctk_affine_rollout_env_lds(int env, int H);
const char* ctk_affine_rollout_env_name(int env, bool log);
// Smallest-K selection under the total order (J, index) and CEM refit
// (optimizer_cem_tf.py:73-78): idx_out[K] ascending, mu/std [H] from Q[idx].
hipError_t ctk_launch_select_topk(hipStream_t st, const float* J, int N, int K, int* idx_out, int ldj = 1);
hipError_t ctk_launch_cem_refit(hipStream_t st, const float* Q, const int* idx, int K, int H, float* mu, float* sd, int ldq);
// plans of one CEM-with-gradient iteration: Q[n,h] = clip(mu[h] + eps[n,h] * std[h]) (optimizer_cem_naive_grad_tf.py:60-62)
hipError_t ctk_launch_sample_plans(hipStream_t st, const RolloutArgs& a, const float* samples, const float* mu, const float* sd, float* Q);
hipError_t ctk_launch_cem_build_population(hipStream_t st, const RolloutArgs& a, int K, int first, const float* Q_prev, const int* idx,
                                           const float* eps_elite, const float* eps_rest, const float* mu, const float* sd, float* Q);
// this shard's best-K records {J, global index, Q[H]} for the sharded selection
hipError_t ctk_launch_pack_candidates(hipStream_t st, const float* J, const float* Q, const int* idx, int K, int H, int global_offset,
                                      float* cand);
// CEM post-loop (optimizer_cem_tf.py:99-102) and u = elite[0,0]
hipError_t ctk_launch_cem_finish(hipStream_t st, const float* Q, const int* idx, int H, float* mu, float* sd,
                                 float std_min, float init_std, float mid, float* u_dev, float* u_host, uint32_t seq, int ldq,
                                 float std_max = 1.0e8f, int u_from_mu = 0);
// random-action: u = Q[argmin J, 0]  (optimizer_random_action_tf.py:65-68)
hipError_t ctk_launch_pick_best_first(hipStream_t st, const float* Q, const int* idx, int H, float* u_dev, float* u_host, uint32_t seq,
                                      int ldq);

// ---- ctk_cem_fused.hip : one CEM step (all outer iterations) in ONE launch, CartPole ODE ---------
constexpr int CTK_CEM_FUSED_MAX_BLOCKS = 128;     // workgroups of 64 rollouts, all co-resident (one per CU)
struct CemFusedLaunch {
    int its, K;                   // outer iterations of this step (optimizer_cem_tf.py:92), cem_best_k
    unsigned long long* ll;       // ctk_cem_fused_ll_words(N, H) hand-off words, zero at allocation
    uint32_t tag0;                // tags tag0 .. tag0 + its - 1: consecutive across launches
    float std_min, std_max, init_std;
    float* mu; float* sd;         // [H*C] the handle's distribution, in / out
    float* u_dev; float* u_host; int* idx_out; uint32_t seq;
    double timeout_s;             // wall-clock bound of every in-launch wait
};
// H: flat columns of a plan (mpc_horizon * control inputs)
bool ctk_cem_fusable(int pred, int N, int H);
size_t ctk_cem_fused_ll_words(int N, int H);
const char* ctk_cem_fused_name(int env, bool log);
// a.H steps, a.C inputs with their limits; constants derived from the environment's parameter table `params`
hipError_t ctk_launch_cem_fused(hipStream_t st, int env, const float* params, float dt, int isteps, const RolloutArgs& a, const float* samples,
                                const CemFusedLaunch& c, bool log, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- ctk_rpgd.hip ---------------------------------------------------------------------------
const char* ctk_rpgd_descent_name(int pred, int N, int H = 0);
bool ctk_rpgd_uses_persistent(int pred, int N, int H);   // the whole descent as one launch: producers + resident Jacobian workers
// host-side state of the persistent form, per handle (the launcher advances both): sequence numbers of the in-launch hand-off
// (64 per launch) and the base of the workers' ticket counter, which lives in the handle's scratch and only ever counts up
struct RpgdPersist { uint32_t seq0, ticket_base; uint32_t* err_word; };
constexpr int CTK_RPGD_WIDE_MAX_N = 4096;
bool ctk_rpgd_uses_wide(int pred, int N);
int ctk_rpgd_fused_max_n(int pred, int N);   // one-launch step (keep-k / warm start as the descent's tail) up to this population   // MLP, small populations: phase launches + grid-wide step Jacobians (ctk_rpgd.hip)
size_t ctk_rpgd_descent_lds(int pred, int H, bool* tape_in_lds);
size_t ctk_rpgd_scratch_floats(int pred, int N, int H);
// All `iters` clipped-gradient Adam iterations + the final cost pass; bc_table[2*(t-1)] = 1-b1^t, [..+1] = 1-b2^t
// Single-workgroup step (N <= 64): keep-k selection + warm start as the tail of the descent launch (ctk_rpgd.hip: FusedWarm).
// Q/m/v of the descent are the old population; the rest describes the new one (arguments of ctk_launch_rpgd_warmstart).
struct RpgdFusedWarm {
    int K; int* idx_out;
    int P, n_new, gather, shift_previous, sampling_distribution, fresh_tail;
    int whole_space;      // sample_whole_control_space: uniform draws span the per-input limits (optimizer_rpgd.py:200-203)
    float sample_stdev, sample_mean, sample_min, sample_max;
    const float* draws; const float* ages_old;
    float* Q_new; float* m_new; float* v_new; float* ages_new;
    const InterpEntry* interp; float* u_nom; float* u_dev; float* u_host; uint32_t seq;
};
constexpr int CTK_RPGD_FUSED_MAX_N = 64;
hipError_t ctk_launch_rpgd_descent(hipStream_t st, int pred, const RolloutArgs& a, const EnvK& k, float lr, float b1, float b2,
                                   float eps, float clip, float* Q, float* m, float* v, const float* bc_table, int bc_len,
                                   int t0, int iters, const float* wperm, float* scratch, hipEvent_t ev_start = nullptr,
                                   hipEvent_t ev_stop = nullptr, int rule = 0, const RpgdFusedWarm* fused = nullptr,
                                   RpgdPersist* pers = nullptr);
// a.C control inputs, a.lo / a.hi the per-input limits; whole_space: uniform samples span [lo[c], hi[c]] (sample_whole_control_space)
hipError_t ctk_launch_rpgd_warmstart(hipStream_t st, const RolloutArgs& a, int N, int H, int P, int n_new, int gather, int shift_previous,
                                     int sampling_distribution, int reset, int whole_space, float sample_stdev,
                                     float sample_mean, float sample_min, float sample_max, const float* draws, const int* idx,
                                     const float* Q_old, const float* m_old, const float* v_old, const float* ages_old,
                                     float* Q_new, float* m_new, float* v_new, float* ages_new, const InterpEntry* interp,
                                     float* u_nom, float* u_dev, float* u_host, uint32_t seq, const float* recs = nullptr,
                                     int rs = 0, int keeper_base = 0, int fresh_tail = 0);
hipError_t ctk_launch_rpgd_pack_keepers(hipStream_t st, const float* J, const float* Q, const float* m, const float* v,
                                        const float* ages, const int* idx, int K, int H, int global_offset, float* out);

// ---- ctk_generic.hip : environment-agnostic template kernels (ctk_env.h) ------------------------------------
constexpr int CTK_G_MODE_MPPI = 0, CTK_G_MODE_AFFINE = 1;
size_t ctk_g_rollout_lds(int cols, int H, int C);
int ctk_g_rollout_blocks(int N);                 // 64 trajectories per workgroup = block records of one MPPI launch
const char* ctk_g_rollout_name(int env, int mode, bool log);
// a.P = inducing points (MPPI) — the launcher turns it into sample columns per row (P*C or H*C); params = the handle's
// primary parameters of `env` (derived constants are formed per launch on the host, double -> fp32 once)
hipError_t ctk_launch_g_rollout(hipStream_t st, int env, int mode, const RolloutArgs& a, const float* params, float dt, int isteps,
                                const MppiK& mk, const float* samples, const float* base, const float* scale, int rng_kind,
                                float* parts, bool log, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
int ctk_g_mppi_update_max_parts();   // records the update launch merges itself
hipError_t ctk_launch_g_mppi_update(hipStream_t st, const float* parts, int n_parts, float neg_inv_lbd, int P, int C, int H,
                                    const InterpEntry* interp, const float* u_nom_in, float* u_nom_out, const RolloutArgs& a, float* u_dev,
                                    float* u_host, uint32_t seq);
hipError_t ctk_launch_g_cem_finish(hipStream_t st, const float* Q, const int* idx, int H, int C, float* mu, float* sd, float std_min,
                                   float init_std, const RolloutArgs& a, float* u_dev, float* u_host, uint32_t seq, int ldq,
                                   float std_max = 1.0e8f, int u_from_mu = 0);
hipError_t ctk_launch_g_pick_best_first(hipStream_t st, const float* Q, const int* idx, int C, float* u_dev, float* u_host, uint32_t seq,
                                        int ldq);
size_t ctk_g_rpgd_descent_lds(int env, int H, bool* tape_in_lds);
size_t ctk_g_rpgd_scratch_floats(int env, int N, int H);
const char* ctk_g_rpgd_descent_name(int env);
hipError_t ctk_launch_g_rpgd_descent(hipStream_t st, int env, const RolloutArgs& a, const float* params, float dt, int isteps, float lr,
                                     float b1, float b2, float eps, float clip, float* Q, float* m, float* v, const float* bc_table,
                                     int bc_len, int t0, int iters, float* scratch, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                                     int rule = 0);

// ---- ctk_generic_net.hip : the template kernels with a network predictor (net = CTK_PRED_MLP | CTK_PRED_GRU; ctk_net.h) -----
// wperm: the policy's per-lane operand tables (forward | reverse), followed by the GRU's carried hidden state [64]
size_t ctk_g_net_table_floats(int net);
size_t ctk_g_net_hidden_floats(int net);
const char* ctk_g_rollout_net_name(int env, int net, int mode, bool log, int N, int P, int H);
int ctk_g_rollout_net_cols(int env, int mode, int P, int H);
int ctk_g_rollout_net_blocks(int env, int net, int mode, int N, int P, int H);   // workgroups = block records of one MPPI launch
size_t ctk_g_rollout_net_lds(int env, int net, int N, int cols, int H, int C);
hipError_t ctk_launch_g_rollout_net(hipStream_t st, int env, int net, int mode, const RolloutArgs& a, const float* params, float dt, int isteps,
                                    const MppiK& mk, const float* samples, const float* base, const float* scale, int rng_kind,
                                    const float* wperm, float* parts, bool log, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                                    const MppiFuse* fuse = nullptr);   // MPPI mode: the in-launch hand-off (modes 1, 2) when it fits
size_t ctk_g_rpgd_descent_net_lds(int env, int net, int N, int H);
size_t ctk_g_rpgd_scratch_floats_net(int net, int N, int H);
const char* ctk_g_rpgd_descent_net_name(int env, int net, int N, int H);
// ctk_net_split.hip: the network predictors with one 16-trajectory tile spread over several waves of a workgroup (GRU: four, forward and
// BPTT; MLP: two), for the populations that leave SIMDs idle with one wave per tile
struct AdamK;
bool ctk_g_rpgd_split_ok(int env, int net, int N, int H);
size_t ctk_g_rpgd_descent_split_lds(int net, int H, int C);
size_t ctk_g_rpgd_scratch_floats_split(int net, int N, int H);
const char* ctk_g_rpgd_descent_split_name(int env, int net);
hipError_t ctk_launch_g_rpgd_descent_split(hipStream_t st, int env, int net, const RolloutArgs& a, const float* params, float dt, int isteps,
                                           const AdamK& ad, float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters,
                                           const float* wperm, const float* wperm_bwd, const float* hidden, float* scratch,
                                           hipEvent_t e0, hipEvent_t e1);
bool ctk_g_rpgd_wide_ok(int env, int net, int N, int H);          // the MLP descent as phase + Jacobian launches (N <= 4 096)
size_t ctk_g_rpgd_scratch_floats_wide(int N, int H);
const char* ctk_g_rpgd_wide_name(int env, int N = 0, int H = 0);
hipError_t ctk_launch_g_rpgd_wide_split(hipStream_t st, int env, const RolloutArgs& a, const float* params, float dt, int isteps, const AdamK& ad,
                                        float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters, const float* wperm,
                                        float* scratch, hipEvent_t e0, hipEvent_t e1, uint32_t* err_word, RpgdPersist* pers = nullptr);
bool ctk_g_rpgd_persist64_ok(int env, int N, int H);          // ... and of the 64-unit network's (no phase-launch form exists for that width)
const char* ctk_g_rpgd_persist64_name(int env);
hipError_t ctk_launch_g_rpgd_persist64(hipStream_t st, int env, const RolloutArgs& a, const float* params, float dt, int isteps, const AdamK& ad,
                                       float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters, const float* wperm,
                                       float* scratch, hipEvent_t e0, hipEvent_t e1, uint32_t* err_word, RpgdPersist* pers);
bool ctk_g_rpgd_persist_ok(int env, int net, int N, int H);   // the wide form as one launch per MPC step (ctk_g_rpgd_persist)
bool ctk_g_rollout_split_ok(int env, int net, int N, int H, int cols);
size_t ctk_g_rollout_split_lds(int net, int cols, int H, int C);
int ctk_g_rollout_split_blocks(int N);
const char* ctk_g_rollout_split_name(int env, int net, int mode, bool log, int N, int H, int cols);
hipError_t ctk_launch_g_rollout_split(hipStream_t st, int env, int net, int mode, const RolloutArgs& a, const float* params, float dt, int isteps,
                                      const MppiK& mk, const float* samples, const float* base, const float* scale, int rng_kind,
                                      const float* wperm, const float* hidden, float* parts, bool log, hipEvent_t e0, hipEvent_t e1,
                                      const MppiFuse* fuse = nullptr);
hipError_t ctk_launch_g_gru_advance4(hipStream_t st, int env, const RolloutArgs& a, const float* u_dev, const float* wperm, float* hidden);
bool ctk_g_rollout_net_fusable(int env, int net, int N, int P, int H);   // may an MPPI step with this network predictor run as ONE launch?
size_t ctk_g_rpgd_descent_gru4_lds(int H, int C);
size_t ctk_g_rpgd_scratch_floats_gru4(int N, int H);
const char* ctk_g_rpgd_descent_gru4_name(int env);
hipError_t ctk_launch_g_rpgd_descent_gru4(hipStream_t st, int env, const RolloutArgs& a, const float* params, float dt, int isteps,
                                          const AdamK& ad, float* Q, float* m, float* v, const float* bc_table, int bc_len, int t0, int iters,
                                          const float* wperm, const float* wperm_bwd, const float* hidden, float* scratch,
                                          hipEvent_t e0, hipEvent_t e1);
hipError_t ctk_launch_g_rpgd_descent_net(hipStream_t st, int env, int net, const RolloutArgs& a, const float* params, float dt, int isteps,
                                         float lr, float b1, float b2, float eps, float clip, float* Q, float* m, float* v,
                                         const float* bc_table, int bc_len, int t0, int iters, const float* wperm, float* scratch,
                                         hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, int rule = 0, uint32_t* err_word = nullptr,
                                         RpgdPersist* pers = nullptr);
// predictor.update(s, Q0) for the GRU under the template kernels: a.s0 = measured state, a.u_prev = applied input (u_dev overrides)
hipError_t ctk_launch_g_gru_advance(hipStream_t st, int env, const RolloutArgs& a, const float* u_dev, float* wperm);
