// ctk_api.hip — the C ABI of libctk_hip.so (include/ctk_hip.h): handle, device state, step
// sequencing.  All compute is in the HIP kernels; there is no CPU fallback.
#include <atomic>
#include <unordered_map>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ctk_launch.h"
#include "ctk_device.h"   // tile_stride
#include "ctk_mlp.h"      // mlp_hid, per-lane weight layout
#include "ctk_gru.h"      // GRU per-lane table layout
#include "ctk_env.h"      // Env<>: S, C of every environment
#include "ctk_net.h"      // GRUG_*: per-lane table layout of the GRU under the template kernels
#include <algorithm>
#include <cstdlib>

namespace {

thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t a, b;
};

}  // namespace

struct ctk_handle {
    ctk_config cfg{};
    int N = 0, H = 0, P = 0;
    int env = CTK_ENV_CARTPOLE, S = CTK_S, C = CTK_C;   // the environment and its dimensions
    int HC = 0, PC = 0;         // floats per row of a [.,H,C] / [.,P,C] tensor
    bool generic = false;       // run the template kernels of ctk_generic.hip (always for environments without tuned kernels)
    int net = 0;                // what the template NETWORK kernels are told to run: cfg.predictor, or NET_MLP64 (hidden widths 33..64)
    int hid = 32;               // units per hidden layer the handle's predictor kernels hold (32; 64 for net == NET_MLP64)
    float params[CTK_MAX_PARAMS]{};
    EnvK k{};
    MppiK mk{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device state
    InterpEntry* d_interp = nullptr;
    float* d_samples = nullptr;  size_t samples_cap = 0;   // staging for host-supplied draws
    float* d_J = nullptr;
    float* d_Q = nullptr;
    float* d_traj = nullptr;
    float* d_parts = nullptr;   size_t parts_cap = 0;
    float* d_parts2 = nullptr;
    float* d_parts3 = nullptr;
    float* d_Jlog = nullptr;     // RPGD with materialised trajectories: cost output of the logging rollout (the step's own J stays in d_J)
    unsigned* d_counter = nullptr;   // ticket counter of the fused in-launch merge
    unsigned long long* d_ll = nullptr;   // {value, seq} record words of the low-latency in-launch hand-off
    unsigned long long* d_cem_ll = nullptr;   // hand-off words of the one-launch CEM step (ctk_cem_fused.hip); nullptr: not fusable
    uint32_t cem_tag = 1;                 // next hand-off tag (one per outer iteration, consecutive across launches)
    bool idx_stale = false;               // the one-launch CEM step leaves only idx[0]; BEST_IDX is rebuilt from J on demand
    float* d_unom[2] = {nullptr, nullptr};   // MPPI u_nom ping-pong / CEM mu in [0]
    int cur = 0;
    float* d_std = nullptr;     // CEM
    float* d_base = nullptr;    // affine rollout base/scale scratch [H] each
    float* d_scale = nullptr;
    int* d_idx = nullptr;       // best indices [N]
    float* d_u = nullptr;       // optimizer's last output (device)
    RpgdPersist rp_pers{64u, 0u, nullptr};   // the one-launch RPGD descent's hand-off numbers (ctk_rpgd.hip: ctk_rpgd_mlp_persistent)
    float* h_u = nullptr;       // pinned, coherent, device-visible host slot: {u, sequence number} written by ONE 8-B store
    float* h_u_dev = nullptr;   // device pointer aliasing h_u
    uint32_t seq = 1;           // sequence number the NEXT publishing kernel will write (the slot starts at 0)
    float* d_weights = nullptr; // raw network weights (MLP [1380] / GRU [10212])
    float* d_wperm = nullptr;   // MLP: per-lane permuted, forward [64][48] then backward [64][28]
                                // GRU: per-lane table [232][64], then the carried hidden state [2][32]
    float mppi_s[CTK_MAX_STATES] = {};   // state of the pending sharded MPPI step (GRU hidden-state advance at step_end)
    float* d_rec = nullptr;     // generic path: the ONE merged soft-min record {rho, a, b[P*C]} of a step
    // RPGD: population, Adam moments, ages (ping-pong), bias-correction table, adjoint scratch
    float* d_pop[2] = {nullptr, nullptr};
    float* d_m[2] = {nullptr, nullptr};
    float* d_v[2] = {nullptr, nullptr};
    float* d_ages[2] = {nullptr, nullptr};
    float* d_bc = nullptr;  int bc_len = 0;
    float* d_scratch = nullptr;
    int rcur = 0;
    int adam_step = 0;
    int variant = 0;            // the optimizer asked for; cfg.optimizer holds its engine family (GRADIENT -> RPGD, CEM_NAIVE_GRAD -> CEM)
    bool rpgd_ready = false;
    int count = 0;              // CEM / RPGD step counter
    uint32_t call = 0;          // Philox call counter
    bool mppi_pending = false;  // between step_begin and step_end
    bool shard_pending = false; // sharded CEM / random-action: between iter_begin and iter_end
    int shard_it = 0;           // iteration index within the current sharded step
    int* d_shard_idx = nullptr; size_t shard_idx_cap = 0;
    const float* shard_last_cands = nullptr;
    float shard_s[CTK_MAX_STATES] = {}; float shard_uprev[CTK_MAX_INPUTS] = {}; bool shard_has_uprev = false;
    bool have_weights = false;  // network weights uploaded
    float hidden_scale = 1.0f;        // GRU: max(1, max |h|) of a caller-set hidden state (units stay within it)
    float net_out_bound = INFINITY;   // GRU: max_g (sum_j |W_o[g,j]| + |b_o[g]|) >= any predicted state component (|h| <= 1)
    // peer-to-peer sharded MPPI (ctk_p2p_*)
    int p2p_rank = -1, p2p_world = 0;
    bool p2p_connected = false;
    float* p2p_bufs[CTK_P2P_MAX_WORLD] = {};   // [rank] = my own uncached buffer, the others IPC mappings
    uint32_t p2p_seq = 1;
    void* d_p2p_args = nullptr;                // device-resident P2PArgs for the in-launch exchange (fuse mode 3)
    double p2p_timeout_s = 5.0;
    // device-resident step log (ctk_log_enable): rings of `log_cap` slots
    size_t log_cap = 0, log_count = 0;
    float* d_log[4] = {nullptr, nullptr, nullptr, nullptr};   // Q, J, TRAJ, AGES
    // profiling
    bool prof = false;
    int prof_every = 1;         // time every n-th dominant-kernel launch (timing a launch costs ~8 us of host time)
    unsigned prof_tick = 0;
    std::vector<EventPair> events;
    size_t ev_used = 0;
    std::string err;
    std::string dominant;
    const char* dominant_ran = nullptr;   // name the MPPI launcher reported last (kernel choice depends on the sample mode)
    // the resident form (ctk_resident_enable): mailbox in pinned host memory, relay in device memory
    bool res_enabled = false, res_running = false;
    double res_idle_us = 200.0;
    CtkResidentBox* res_box = nullptr;    // what the host stores requests into: fine-grained device memory through the BAR (res_local) or pinned host memory
    CtkResidentBox* res_box_dev = nullptr;   // the same memory as the kernel addresses it
    bool res_local = false;
    bool res_stream_isolated = false;     // the handle's own stream has been re-created at the highest priority (its own hardware queue)
    CtkResidentStat* res_stat = nullptr;  // pinned host memory: what the kernel reports
    CtkResidentStat* res_stat_dev = nullptr;
    CtkResidentBox* d_res_relay = nullptr;
    void* d_res_args = nullptr;           // the resident kernel's argument block (device) and its staging copy
    alignas(16) unsigned char res_args_host[CTK_RES_ARGS_BYTES];
    uint32_t res_req = 0;                 // last request number issued
    uint32_t res_relay_prime[2] = {0, 0}; // staging words of the copy that primes the relay before a launch
    uint64_t res_launches = 0, res_steps = 0;
    std::string res_dominant_saved;       // ctk_dominant_kernel of the launched form while the resident kernel runs
    std::unordered_map<const float*, const float*> res_follow;   // sample buffer -> the buffer that followed it last time
    const float* res_prev_samples = nullptr;
    bool res_readahead = false;           // ctk_resident_enable(h, 2, ...): the caller's sample buffers are immutable while enabled, so they may be read ahead
};

namespace {

int fail(ctk_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail((h), CTK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// others/Interpolator.py:79-84
int num_inducing_points(int H, int p) { return (int)std::ceil((double)(H - 1) / (double)p) + 1; }

// Column t of the reference matrix (others/Interpolator.py:53-77), built the same way: fp32
// (step-j)/step and j/step; closing row set to 1 BEFORE the division (quirk kept, see oracle).
std::vector<InterpEntry> build_interp_table(int H, int p, int P) {
    const int rows = (P - 1) * p + 1;
    std::vector<float> M((size_t)rows * P, 0.0f);
    for (int i = 0; i < P - 1; ++i)
        for (int j = 0; j < p; ++j) {
            M[(size_t)(i * p + j) * P + i] = (float)(p - j);
            M[(size_t)(i * p + j) * P + i + 1] = (float)j;
        }
    M[(size_t)(rows - 1) * P + (P - 1)] = 1.0f;
    std::vector<InterpEntry> tab(H);
    for (int t = 0; t < H; ++t) {
        int first = -1;
        for (int c = 0; c < P; ++c)
            if (M[(size_t)t * P + c] != 0.0f) { first = c; break; }
        InterpEntry e{0, 0.0f, 0.0f};
        if (P == 1) {
            e.i0 = 0; e.w0 = M[(size_t)t * P] / (float)p;
        } else {
            e.i0 = first < 0 ? 0 : (first > P - 2 ? P - 2 : first);
            e.w0 = M[(size_t)t * P + e.i0] / (float)p;
            e.w1 = M[(size_t)t * P + e.i0 + 1] / (float)p;
        }
        tab[t] = e;
    }
    return tab;
}

void refresh_constants(ctk_handle* h) {
    // CartPole's derived constants feed the hand-tuned kernels; the template kernels derive theirs per launch
    // (ctk_generic.hip: Env<>::derive) from the same primary parameters
    if (h->env == CTK_ENV_CARTPOLE) h->k = derive_constants(h->params, h->cfg.dt, h->cfg.intermediate_steps);
    const ctk_config& c = h->cfg;
    MppiK m;
    m.stdev = (float)((double)c.SQRTRHOINV * (1.0 / std::sqrt((double)c.dt)));   // optimizer_mppi.py:130
    const float one_m = 1.0f - 1.0f / c.NU;
    m.k_dd = (0.5f * one_m) * c.R;
    m.R = c.R;
    m.k_uu = 0.5f * c.R;
    m.cc = c.cc_weight;
    m.neg_inv_lbd = (float)(-1.0 / (double)c.LBD);
    h->mk = m;
}

// Per-lane MFMA operand layout of the MLP weights (ctk_mlp.h header comment), for a network with I = S + C <= 8 inputs and
// S <= 8 outputs.  Network input / output index k lives in lane group k % 4, k-step (inputs) or register (outputs) k / 4,
// i.e. at row 4*(k%4) + k/4 of a 16-row tile — inputs are the S state components followed by the C control inputs.
// raw: W1[32,I] b1[32] W2[32,32] b2[32] W3[S,32] b3[S].  out: fwd [64][48] | bwd [64][28].
std::vector<float> permute_mlp_weights(const float* raw, int S = CTK_S, int C = CTK_C) {
    const int I = S + C;
    const float* W1 = raw;                 const float* b1 = W1 + 32 * I;
    const float* W2 = b1 + 32;             const float* b2 = W2 + 32 * 32;
    const float* W3 = b2 + 32;             const float* b3 = W3 + S * 32;
    std::vector<float> out((size_t)64 * (MLP_FWD_PER_LANE + MLP_BWD_PER_LANE), 0.0f);
    auto io_of_row = [](int row) { return 4 * (row % 4) + row / 4; };   // network input / output index held at tile row `row`
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, g = l >> 4;
        float* f = out.data() + (size_t)l * MLP_FWD_PER_LANE;
        for (int m = 0; m < 2; ++m)
            for (int ks = 0; ks < 2; ++ks) {
                const int kk = 4 * ks + g;
                f[m * 2 + ks] = kk < I ? W1[(16 * m + i) * I + kk] : 0.0f;
            }
        for (int mo = 0; mo < 2; ++mo)
            for (int j = 0; j < 8; ++j) f[4 + mo * 8 + j] = W2[(16 * mo + i) * 32 + mlp_hid(j, g)];
        const int out_i = (i % 4 < 2 && io_of_row(i) < S) ? io_of_row(i) : -1;     // rows r = 0, 1 of every lane group carry outputs
        for (int j = 0; j < 8; ++j) f[20 + j] = out_i >= 0 ? W3[out_i * 32 + mlp_hid(j, g)] : 0.0f;
        for (int m = 0; m < 2; ++m)
            for (int r = 0; r < 4; ++r) {
                f[28 + m * 4 + r] = b1[16 * m + 4 * g + r];
                f[36 + m * 4 + r] = b2[16 * m + 4 * g + r];
            }
        for (int r = 0; r < 4; ++r) f[44 + r] = (r < 2 && 4 * r + g < S) ? b3[4 * r + g] : 0.0f;
        for (int m = 0; m < 2; ++m) f[65 + m] = 8 + g < I ? W1[(16 * m + i) * I + 8 + g] : 0.0f;   // third k-step of layer 1 (S + C > 8)
        if (S == CTK_S && C == CTK_C) {   // thin-layer form of the CartPole kernels (ctk_mlp.h: MlpFwdT)
            for (int j = 0; j < 8; ++j) f[48 + j] = W3[(l & 3) * 32 + mlp_hid(j, g)];                  // A of the 4x4x1 blocks: output row lane % 4
            for (int m = 0; m < 2; ++m)
                for (int r = 0; r < 4; ++r) f[56 + m * 4 + r] = W1[(16 * m + 4 * g + r) * I + S];     // the input's column at this lane's accumulator rows
            f[64] = b3[g];
        }
        // backward (vector-Jacobian products): A operands of W3^T, W2^T, W1^T
        float* b = out.data() + (size_t)64 * MLP_FWD_PER_LANE + (size_t)l * MLP_BWD_PER_LANE;
        for (int m = 0; m < 2; ++m)
            for (int ks = 0; ks < 2; ++ks)
                b[2 * m + ks] = (4 * ks + g < S) ? W3[(4 * ks + g) * 32 + 16 * m + i] : 0.0f;           // rows: hidden, k: output component 4ks+g
        for (int mi = 0; mi < 2; ++mi)
            for (int j = 0; j < 8; ++j) b[4 + mi * 8 + j] = W2[mlp_hid(j, g) * 32 + 16 * mi + i];       // rows: hidden_in, k: hidden_out
        const int inp = (i % 4 < 3 && io_of_row(i) < I) ? io_of_row(i) : -1;                            // rows: network inputs (rows 4g+2: inputs 8+g)
        for (int j = 0; j < 8; ++j) b[20 + j] = inp >= 0 ? W1[mlp_hid(j, g) * I + inp] : 0.0f;
    }
    return out;
}

// The 64-unit MLP of ctk_mlp_wide.h (NetMlpWideT): the same placement rules with four 16-row tiles per hidden layer.
// raw: W1[64,I] b1[64] W2[64,64] b2[64] W3[S,64] b3[S].  Per lane: forward w1[T][3] | w2[T][KH] | w3[KH] | b1[T][4] | b2[T][4] | b3[4];
// reverse (behind the 64 forward rows) w3t[T][2] | w2t[T][KH] | w1t[KH].
std::vector<float> permute_mlp_weights_wide(const float* raw, int S, int C) {
    constexpr int T = MLPW_T, HID = MLPW_HID, KH = MLPW_KH;
    const int I = S + C;
    const float* W1 = raw;                 const float* b1 = W1 + HID * I;
    const float* W2 = b1 + HID;            const float* b2 = W2 + HID * HID;
    const float* W3 = b2 + HID;            const float* b3 = W3 + S * HID;
    std::vector<float> out((size_t)64 * (MLPW_FWD_PER_LANE + MLPW_BWD_PER_LANE), 0.0f);
    auto io_of_row = [](int row) { return 4 * (row % 4) + row / 4; };
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, g = l >> 4;
        float* f = out.data() + (size_t)l * MLPW_FWD_PER_LANE;
        int o = 0;
        for (int m = 0; m < T; ++m)
            for (int ks = 0; ks < 3; ++ks) { const int kk = 4 * ks + g; f[o++] = kk < I ? W1[(16 * m + i) * I + kk] : 0.0f; }
        for (int m = 0; m < T; ++m)
            for (int j = 0; j < KH; ++j) f[o++] = W2[(16 * m + i) * HID + mlp_hid(j, g)];
        const int out_i = (i % 4 < 2 && io_of_row(i) < S) ? io_of_row(i) : -1;     // rows r = 0, 1 of every lane group carry outputs
        for (int j = 0; j < KH; ++j) f[o++] = out_i >= 0 ? W3[out_i * HID + mlp_hid(j, g)] : 0.0f;
        for (int m = 0; m < T; ++m) for (int r = 0; r < 4; ++r) f[o++] = b1[16 * m + 4 * g + r];
        for (int m = 0; m < T; ++m) for (int r = 0; r < 4; ++r) f[o++] = b2[16 * m + 4 * g + r];
        for (int r = 0; r < 4; ++r) f[o++] = (r < 2 && 4 * r + g < S) ? b3[4 * r + g] : 0.0f;
        float* b = out.data() + (size_t)64 * MLPW_FWD_PER_LANE + (size_t)l * MLPW_BWD_PER_LANE;
        o = 0;
        for (int m = 0; m < T; ++m)
            for (int ks = 0; ks < 2; ++ks) b[o++] = (4 * ks + g < S) ? W3[(4 * ks + g) * HID + 16 * m + i] : 0.0f;    // rows: hidden, k: output component 4ks+g
        for (int m = 0; m < T; ++m)
            for (int j = 0; j < KH; ++j) b[o++] = W2[mlp_hid(j, g) * HID + 16 * m + i];                                // rows: hidden_in, k: hidden_out
        const int inp = (i % 4 < 3 && io_of_row(i) < I) ? io_of_row(i) : -1;                                           // rows: network inputs
        for (int j = 0; j < KH; ++j) b[o++] = inp >= 0 ? W1[mlp_hid(j, g) * I + inp] : 0.0f;
    }
    return out;
}

// Per-lane MFMA operands of the GRU weights for the four waves of a workgroup (ctk_gru.h header comment and
// struct GruW): out[wave][lane][72].  Wave (m, q) = wave index 2m + q owns hidden-unit tile m; its A rows are the
// r (q = 0) or z (q = 1) rows, its B rows the n rows (input products for q = 0, recurrent products for q = 1).
// raw: per layer W_i[96,I] W_h[96,32] b_i[96] b_h[96] (rows r|z|n), then W_o[4,32] b_o[4].
std::vector<float> permute_gru_weights(const float* raw) {
    std::vector<float> out((size_t)GRU_TABLE_FLOATS, 0.0f);
    for (int wv = 0; wv < 4; ++wv)
        for (int l = 0; l < 64; ++l) {
            const int m = wv >> 1, q = wv & 1, i = l & 15, g = l >> 4;
            float* e = out.data() + (size_t)(wv * 64 + l) * GRU_W_PER_LANE;
            const float* p = raw;
            for (int L = 0; L < 2; ++L) {
                const int I = L == 0 ? CTK_MLP_IN : 32, KS = L == 0 ? 2 : 8, base = L == 0 ? 0 : 18;
                const float* Wi = p;              const float* Wh = Wi + 96 * I;
                const float* bi = Wh + 96 * 32;   const float* bh = bi + 96;
                p = bh + 96;
                const int rowA = 32 * q + 16 * m + i, rowN = 64 + 16 * m + i;
                auto kk = [&](int ks) { return L == 0 ? 4 * ks + g : mlp_hid(ks, g); };
                auto wi = [&](int row, int ks) { return (L == 0 && kk(ks) >= CTK_MLP_IN) ? 0.0f : Wi[row * I + kk(ks)]; };
                for (int ks = 0; ks < KS; ++ks) e[base + ks] = wi(rowA, ks);
                for (int j = 0; j < 8; ++j) e[base + KS + j] = Wh[rowA * 32 + mlp_hid(j, g)];
                if (q == 0) for (int ks = 0; ks < KS; ++ks) e[base + KS + 8 + ks] = wi(rowN, ks);
                else for (int j = 0; j < 8; ++j) e[base + KS + 8 + j] = Wh[rowN * 32 + mlp_hid(j, g)];
                for (int r = 0; r < 4; ++r) {
                    const int unit = 16 * m + 4 * g + r;
                    e[50 + 8 * L + r] = bi[32 * q + unit] + bh[32 * q + unit];   // r or z: both biases feed one accumulator
                    e[54 + 8 * L + r] = q == 0 ? bi[64 + unit] : bh[64 + unit];  // n: input part / recurrent part (scaled by r)
                }
            }
            const float* Wo = p; const float* bo = Wo + 4 * 32;
            for (int j = 0; j < 8; ++j) e[42 + j] = (i % 4 == 0) ? Wo[(i / 4) * 32 + mlp_hid(j, g)] : 0.0f;
            for (int r = 0; r < 4; ++r) e[66 + r] = r == 0 ? bo[g] : 0.0f;
        }
    return out;
}

// parameter names (the keys of the cost YAML / dynamics section) and defaults, in enum order:
// oracle/ctk_oracle.py: EnvParams / Quad2DParams
const char* const kCartPoleNames[CTK_P_COUNT] = {
    "g", "m_cart", "m_pole", "L", "u_max", "M_fric", "J_fric", "target_position", "target_equilibrium",
    "dd_weight", "ep_weight", "ekp_weight", "cc_weight", "ccrc_weight", "R", "x_scale", "terminal_weight"};
const float kCartPoleDefaults[CTK_P_COUNT] = {9.81f, 0.230f, 0.087f, 0.1975f, 2.62f, 4.77f, 2.5e-4f, 0.0f, 1.0f,
                                              600.0f, 20000.0f, 80.0f, 1.0f, 1.0f, 1.0f, 0.198f, 0.0f};
const char* const kQuadNames[CTK_Q_COUNT] = {
    "g", "mass", "inertia", "arm", "thrust_gain", "drag_lin", "drag_ang", "target_x", "target_z",
    "pos_weight", "ang_weight", "vel_weight", "angvel_weight", "cc_weight", "ccrc_weight", "R", "pos_scale", "terminal_weight"};
const float kQuadDefaults[CTK_Q_COUNT] = {9.81f, 0.5f, 0.004f, 0.12f, 0.6f, 0.25f, 0.4f, 0.0f, 1.0f,
                                          400.0f, 150.0f, 8.0f, 1.5f, 1.0f, 2.0f, 1.0f, 0.5f, 0.0f};
const char* const kHoverNames[CTK_V_COUNT] = {
    "mass", "inertia", "wheel_inertia", "thrust_max", "lateral_max", "torque_max", "drag_lin", "drag_ang", "wheel_friction", "target_x",
    "target_y", "pos_weight", "ang_weight", "vel_weight", "angvel_weight", "wheel_weight", "cc_weight", "ccrc_weight", "R", "pos_scale",
    "terminal_weight"};
const float kHoverDefaults[CTK_V_COUNT] = {1.2f, 0.05f, 0.01f, 4.0f, 1.5f, 0.2f, 0.3f, 0.2f, 0.05f, 0.0f, 0.0f,
                                           300.0f, 80.0f, 6.0f, 1.0f, 0.02f, 1.0f, 1.5f, 1.0f, 0.5f, 0.0f};
const EnvInfo kEnvs[CTK_ENV_COUNT] = {
    {"CartPole", Env<CTK_ENV_CARTPOLE>::S, Env<CTK_ENV_CARTPOLE>::C, CTK_P_COUNT, kCartPoleNames, kCartPoleDefaults},
    {"Quad2D", Env<CTK_ENV_QUAD2D>::S, Env<CTK_ENV_QUAD2D>::C, CTK_Q_COUNT, kQuadNames, kQuadDefaults},
    {"Hover", Env<CTK_ENV_HOVER>::S, Env<CTK_ENV_HOVER>::C, CTK_V_COUNT, kHoverNames, kHoverDefaults},
#ifdef CTK_USER_ENV_HEADER
    {CtkUserEnv::NAME, CtkUserEnv::S, CtkUserEnv::C, CtkUserEnv::NP, CtkUserEnv::PARAM_NAMES, CtkUserEnv::PARAM_DEFAULTS},    // CTK_ENV_USER
#else
    {nullptr, 0, 0, 0, nullptr, nullptr},              // CTK_ENV_USER: only in a library built with a user model (build_env.py)
#endif
};
const EnvInfo* env_info(int env) { return (env >= 0 && env < CTK_ENV_COUNT && kEnvs[env].name != nullptr) ? &kEnvs[env] : nullptr; }

// Per-lane operand tables of the GRU under the template kernels (ctk_net.h: NetGru — one wave per 16-trajectory tile, operands
// staged in LDS): forward [232][64] then reverse [172][64], for a network with I = S + C <= 8 inputs and S <= 8 outputs.
// raw: per layer W_i[96,I'] W_h[96,32] b_i[96] b_h[96] (rows r|z|n, I' = I then 32), then W_o[S,32] b_o[S].
std::vector<float> permute_gru_weights_g(const float* raw, int S, int C) {
    const int I = S + C;
    std::vector<float> out((size_t)GRUG_TABLE * 64, 0.0f);
    float* F = out.data();
    float* B = F + (size_t)GRUG_FWD * 64;
    float* X = B + (size_t)GRUG_BWD * 64;              // layer 1's third k-step, [gate][tile] (S + C > 8; zeros otherwise)
    const float *Wi[2], *Wh[2], *bi[2], *bh[2];
    const float* p = raw;
    for (int L = 0; L < 2; ++L) {
        const int In = L == 0 ? I : 32;
        Wi[L] = p; Wh[L] = Wi[L] + 96 * In; bi[L] = Wh[L] + 96 * 32; bh[L] = bi[L] + 96;
        p = bh[L] + 96;
    }
    const float* Wo = p; const float* bo = Wo + S * 32;
    auto io_of_row = [](int row) { return 4 * (row % 4) + row / 4; };
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, g = l >> 4;
        for (int L = 0; L < 2; ++L) {
            const int KS = L == 0 ? 2 : 8, In = L == 0 ? I : 32, base = L == 0 ? 0 : GRUG_L1;
            for (int G = 0; G < 3; ++G)
                for (int m = 0; m < 2; ++m) {
                    const int row = G * 32 + 16 * m + i;
                    for (int ks = 0; ks < KS; ++ks) {
                        const int kk = L == 0 ? 4 * ks + g : mlp_hid(ks, g);
                        F[(size_t)(base + (G * 2 + m) * KS + ks) * 64 + l] = (L == 0 && kk >= I) ? 0.0f : Wi[L][row * In + kk];
                    }
                    for (int j = 0; j < 8; ++j) F[(size_t)(base + 6 * KS + (G * 2 + m) * 8 + j) * 64 + l] = Wh[L][row * 32 + mlp_hid(j, g)];
                    if (L == 0) X[(size_t)(G * 2 + m) * 64 + l] = 8 + g < I ? Wi[0][row * I + 8 + g] : 0.0f;
                }
            const int bb = base + 6 * KS + 48;
            for (int m = 0; m < 2; ++m)
                for (int r = 0; r < 4; ++r) {
                    const int unit = 16 * m + 4 * g + r;
                    F[(size_t)(bb + 0 + m * 4 + r) * 64 + l] = bi[L][unit] + bh[L][unit];
                    F[(size_t)(bb + 8 + m * 4 + r) * 64 + l] = bi[L][32 + unit] + bh[L][32 + unit];
                    F[(size_t)(bb + 16 + m * 4 + r) * 64 + l] = bi[L][64 + unit];
                    F[(size_t)(bb + 24 + m * 4 + r) * 64 + l] = bh[L][64 + unit];
                }
        }
        const int ob = GRUG_L1 + GRUG_L2;
        const int out_i = (i % 4 < 2 && io_of_row(i) < S) ? io_of_row(i) : -1;
        for (int j = 0; j < 8; ++j) F[(size_t)(ob + j) * 64 + l] = out_i >= 0 ? Wo[out_i * 32 + mlp_hid(j, g)] : 0.0f;
        for (int r = 0; r < 4; ++r) F[(size_t)(ob + 8 + r) * 64 + l] = (r < 2 && 4 * r + g < S) ? bo[4 * r + g] : 0.0f;
        // reverse: A operands of the transposed products; k-step (G, m, r) has k-slot g = gate neuron G*32 + 16m + 4g + r
        for (int m = 0; m < 2; ++m)
            for (int ks = 0; ks < 2; ++ks) B[(size_t)(m * 2 + ks) * 64 + l] = (4 * ks + g < S) ? Wo[(4 * ks + g) * 32 + 16 * m + i] : 0.0f;
        const int inp = (i % 4 < 3 && io_of_row(i) < I) ? io_of_row(i) : -1;      // rows 4g + 0 / 1 / 2: network inputs g, 4 + g, 8 + g
        for (int G = 0; G < 3; ++G)
            for (int m = 0; m < 2; ++m)
                for (int r = 0; r < 4; ++r) {
                    const int ks = G * 8 + m * 4 + r, neuron = G * 32 + 16 * m + 4 * g + r;
                    for (int mi = 0; mi < 2; ++mi) B[(size_t)(4 + mi * 24 + ks) * 64 + l] = Wi[1][neuron * 32 + 16 * mi + i];
                    for (int mh = 0; mh < 2; ++mh) B[(size_t)(4 + 48 + mh * 24 + ks) * 64 + l] = Wh[1][neuron * 32 + 16 * mh + i];
                    B[(size_t)(4 + 96 + ks) * 64 + l] = inp >= 0 ? Wi[0][neuron * I + inp] : 0.0f;
                    for (int mh = 0; mh < 2; ++mh) B[(size_t)(4 + 96 + 24 + mh * 24 + ks) * 64 + l] = Wh[0][neuron * 32 + 16 * mh + i];
                }
    }
    return out;
}

void default_params(int env, float* p) {
    const EnvInfo* e = env_info(env);
    for (int i = 0; i < e->n_params; ++i) p[i] = e->param_defaults[i];
}

size_t weight_count(int predictor, int S, int C, int hid = 32) {   // include/ctk_hip.h: ctk_set_predictor_weights
    const size_t I = (size_t)S + C, W = (size_t)hid;
    if (predictor == CTK_PRED_MLP) return I * W + W + W * W + W + W * (size_t)S + S;
    if (predictor == CTK_PRED_GRU) return (96 * I + 96 * 32 + 192) + (96 * 32 + 96 * 32 + 192) + (32 * (size_t)S + S);
    return 0;
}

template <class T>
int dev_alloc(ctk_handle* h, T** p, size_t n) {
    HIP_TRY(h, hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
    HIP_TRY(h, hipMemsetAsync(*p, 0, (n ? n : 1) * sizeof(T), h->stream));
    return CTK_OK;
}

int cem_iterations(const ctk_handle* h) {   // optimizer_cem_tf.py:92
    return (h->cfg.warmup && h->count == 0) ? h->cfg.warmup_iterations : h->cfg.cem_outer_it;
}

size_t samples_needed(const ctk_handle* h) {
    const size_t N = h->N, H = h->HC, P = h->PC;   // draws per rollout: [H,C] / [P,C] blocks
    switch (h->cfg.optimizer) {
        case CTK_OPT_MPPI: return N * P;
        case CTK_OPT_CEM:
            if (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ)   // initial elites + (N-K) fresh samples per iteration
                return (size_t)h->cfg.cem_best_k * H + (size_t)cem_iterations(h) * (N - (size_t)h->cfg.cem_best_k) * H;
            return (size_t)cem_iterations(h) * N * H;
        case CTK_OPT_RANDOM_ACTION: return N * H;
        case CTK_OPT_RPGD:
            if (h->variant == CTK_OPT_GRADIENT) return N * h->C;   // the shifted-in tail input of every plan [N,1,C]
            return (h->count % h->cfg.resamp_per == 0 && (size_t)h->cfg.opt_keep_k < N) ? (N - (size_t)h->cfg.opt_keep_k) * P : 0;
    }
    return 0;
}

// resolve a caller sample buffer to a device pointer (nullptr => on-device Philox)
int resolve_samples(ctk_handle* h, const float* samples, int loc, size_t n, const float** out) {
    *out = nullptr;
    if (loc == CTK_LOC_NONE || n == 0) return CTK_OK;
    if (samples == nullptr) return fail(h, CTK_ERR_INVALID_ARGUMENT, "samples pointer is NULL but samples_loc != CTK_LOC_NONE");
    if (loc == CTK_LOC_DEVICE) { *out = samples; return CTK_OK; }
    if (loc != CTK_LOC_HOST) return fail(h, CTK_ERR_INVALID_ARGUMENT, "bad samples_loc");
    if (n > h->samples_cap) {
        if (h->d_samples) HIP_TRY(h, hipFree(h->d_samples));
        h->d_samples = nullptr; h->samples_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->d_samples, n * sizeof(float)));
        h->samples_cap = n;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_samples, samples, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    *out = h->d_samples;
    return CTK_OK;
}

RolloutArgs make_args(ctk_handle* h, const float* s, const float* u_prev, int N, int P) {
    RolloutArgs a{};
    for (int i = 0; i < h->S; ++i) a.s0[i] = s[i];
    for (int c = 0; c < h->C; ++c) { a.u_prev[c] = u_prev ? u_prev[c] : 0.0f; a.lo[c] = h->cfg.action_low[c]; a.hi[c] = h->cfg.action_high[c]; }
    a.u_prev_dev = u_prev ? nullptr : h->d_u;
    a.C = h->C;
    a.N = N; a.H = h->H; a.P = P;
    a.p_magic = P >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)P - 1) / (uint64_t)P) : 0u;
    a.identity_interp = (h->cfg.period_interpolation_inducing_points == 1 && P == h->H) ? 1 : 0;
    a.inv_Hp1 = 1.0f / (float)(h->H + 1);
    a.interp = h->d_interp;
    a.J = h->d_J;
    a.Q_out = h->d_Q;
    a.traj_out = h->cfg.materialize_trajectories ? h->d_traj : nullptr;
    a.seed_lo = (uint32_t)(h->cfg.seed & 0xFFFFFFFFull);
    a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    a.call = h->call;
    a.stream_id = 0;
    a.global_row0 = h->cfg.global_rollout_offset;
    // GRU rollouts: the angle the cost sees is s[2] at h = 0 and a network output afterwards
    a.fast_cos_ok = (h->env == CTK_ENV_CARTPOLE && std::fabs(s[2]) <= CTK_SINCOS_FAST_LIMIT &&
                     h->net_out_bound * h->hidden_scale <= CTK_SINCOS_FAST_LIMIT) ? 1 : 0;
    return a;
}

// Event pair for the next dominant-kernel launch (nullptrs when profiling is off / ring is full).
struct ProfSlot {
    hipEvent_t a = nullptr, b = nullptr;
    explicit ProfSlot(ctk_handle* h) {
        if (h->prof && h->ev_used < h->events.size() && (h->prof_tick++ % (unsigned)h->prof_every) == 0) {
            a = h->events[h->ev_used].a; b = h->events[h->ev_used].b; ++h->ev_used;
        }
    }
};

// Where a readable tensor lives right now (shared by ctk_read and the step log).
int locate_buffer(ctk_handle* h, int which, const float** src_out, size_t* n_out, bool* is_int_out) {
    const size_t N = h->N, H = h->HC;   // [.,H,C] rows
    const float* src = nullptr; size_t n = 0; bool is_int = false;
    switch (which) {
        case CTK_BUF_Q: src = h->cfg.optimizer == CTK_OPT_RPGD ? h->d_pop[h->rcur ^ 1] : (h->variant == CTK_OPT_CEM_NAIVE_GRAD ? h->d_pop[0] : (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ ? h->d_pop[h->rcur] : h->d_Q));
            n = N * H; break;
        case CTK_BUF_PLAN: if (h->cfg.optimizer != CTK_OPT_RPGD) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: PLAN is an RPGD buffer");
            src = h->d_pop[h->rcur]; n = N * H; break;
        case CTK_BUF_ADAM_M:
            if (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ) { src = h->d_m[0]; n = N * H; break; }
            if (h->cfg.optimizer != CTK_OPT_RPGD) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: ADAM_M is an RPGD buffer");
            src = h->d_m[h->rcur]; n = N * H; break;
        case CTK_BUF_ADAM_V:
            if (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ) { src = h->d_v[0]; n = N * H; break; }
            if (h->cfg.optimizer != CTK_OPT_RPGD) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: ADAM_V is an RPGD buffer");
            src = h->d_v[h->rcur]; n = N * H; break;
        case CTK_BUF_AGES: if (h->cfg.optimizer != CTK_OPT_RPGD) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: AGES is an RPGD buffer");
            src = h->d_ages[h->rcur]; n = N; break;
        case CTK_BUF_AGES_LOGGED: if (h->cfg.optimizer != CTK_OPT_RPGD) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: AGES_LOGGED is an RPGD buffer");
            src = h->d_ages[h->rcur ^ 1]; n = N; break;    // the buffer the last step read its ages from (the warm start wrote the other one)
        case CTK_BUF_J: src = h->d_J; n = N; break;
        case CTK_BUF_TRAJ:
            if (!h->d_traj) return fail(h, CTK_ERR_STATE, "ctk_read: trajectories not materialised (cfg.materialize_trajectories == 0)");
            src = h->d_traj; n = N * ((size_t)h->H + 1) * h->S; break;
        case CTK_BUF_U_NOM: src = h->d_unom[h->cfg.optimizer == CTK_OPT_MPPI ? h->cur : 0]; n = H; break;
        case CTK_BUF_STD: src = h->d_std; n = H; break;
        case CTK_BUF_BEST_IDX:
            if (h->idx_stale) {   // one-launch CEM step: the sorted elite indices are materialised from the last iteration's costs
                HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, h->cfg.cem_best_k, h->d_idx));
                h->idx_stale = false;
            }
            src = (const float*)h->d_idx; is_int = true;
            n = h->cfg.optimizer == CTK_OPT_CEM ? (size_t)h->cfg.cem_best_k : (h->cfg.optimizer == CTK_OPT_RPGD ? (size_t)h->cfg.opt_keep_k : 1);
            break;
        default: return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: buffer not available for this optimizer");
    }
    *src_out = src; *n_out = n; *is_int_out = is_int;
    return CTK_OK;
}

// floats per step of ring `i` (Q, J, TRAJ, AGES); 0 = not logged for this handle
size_t log_slot_floats(const ctk_handle* h, int i) {
    const size_t N = h->N;
    switch (i) {
        case 0: return N * h->HC;
        case 1: return N;
        case 2: return h->d_traj ? N * ((size_t)h->H + 1) * h->S : 0;
        default: return h->cfg.optimizer == CTK_OPT_RPGD ? N : 0;
    }
}

// append this step's tensors to the rings: one launch, behind the step on the stream (the result is out already)
int log_step(ctk_handle* h) {
    static const int which[4] = {CTK_BUF_Q, CTK_BUF_J, CTK_BUF_TRAJ, CTK_BUF_AGES_LOGGED};
    const size_t slot = h->log_count % h->log_cap;
    CopyJob jobs[4];
    for (int i = 0; i < 4; ++i) {
        jobs[i] = CopyJob{nullptr, nullptr, 0u};
        const size_t n = log_slot_floats(h, i);
        if (n == 0 || !h->d_log[i]) continue;
        const float* src; size_t cnt; bool is_int;
        if (int rc = locate_buffer(h, which[i], &src, &cnt, &is_int)) return rc;
        jobs[i] = CopyJob{src, h->d_log[i] + slot * n, (unsigned)n};
    }
    HIP_TRY(h, ctk_launch_copy4(h->stream, jobs));
    ++h->log_count;
    return CTK_OK;
}

// Completion of a step = the publishing kernel's single 8-byte system-scope store {u, seq} landing in
// the pinned host slot.  Polling it avoids the completion-signal round trip of
// hipStreamSynchronize (several microseconds per step); the spin is bounded, and on timeout the
// stream is synchronised so that a device fault surfaces as an error instead of a hang.
int finish_step(ctk_handle* h, float* u_out) {
    const uint32_t want = h->seq;
    volatile uint32_t* slot = reinterpret_cast<volatile uint32_t*>(h->h_u) + 1;
    bool seen = false;
    for (int spin = 0; spin < 4000000; ++spin) {
        if (*slot == want) { seen = true; break; }
        __builtin_ia32_pause();
    }
    if (!seen) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (*slot != want) return fail(h, CTK_ERR_HIP, "step finished without publishing its result");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (u_out) {
        u_out[0] = *reinterpret_cast<volatile float*>(h->h_u);
        for (int c = 1; c < h->C; ++c) u_out[c] = reinterpret_cast<volatile float*>(h->h_u)[4 + c];   // publish_u_vec
    }
    ++h->seq;
    ++h->call;
    // error word behind {u, seq}: a bounded device-side wait ran out (1: a peer's record never arrived,
    // ctk_mppi.hip:p2p_exchange_and_update; 2: a block record of the in-launch hand-off never arrived; 3: an RPGD Jacobian record never
    // arrived / non-finite gradient, ctk_net_split.hip: rpgd_jac_worker + the update's tile_bad) — the published
    // result is NaN or built from a stale record, never silently wrong
    volatile uint32_t* errw = reinterpret_cast<volatile uint32_t*>(h->h_u) + 2;
    const uint32_t dev_err = errw[0], bad_tile = errw[1];              // (word 3: a tile of the RPGD update saw a non-finite gradient norm)
    if (dev_err || bad_tile) {
        errw[0] = 0; errw[1] = 0;
        if (dev_err == 3) {
            char msg[512];
            const uint32_t where = errw[6], seen = errw[7], want = errw[8], us = errw[9];
            std::snprintf(msg, sizeof msg, "RPGD: an in-launch Jacobian hand-off timed out (flag %u of step %u, tile %u: held %u, awaited %u, for %u us); "
                          "that iteration's update was skipped for the tile (its plans and Adam moments are as before it) — the returned control "
                          "comes from a population that missed an update", where >> 20, (where >> 10) & 1023u, where & 1023u, seen, want, us);
            return fail(h, CTK_ERR_STATE, msg);
        }
        if (dev_err == 0) {
            char msg[384];
            std::snprintf(msg, sizeof msg, "RPGD: the gradient of a plan of tile %u was not finite (a rollout that diverged); that iteration's update "
                          "was skipped for the tile (its plans and Adam moments are as before it) — the returned control comes from a population "
                          "that missed an update", bad_tile - 1);
            return fail(h, CTK_ERR_STATE, msg);
        }
        return fail(h, CTK_ERR_STATE, dev_err == 1 ? "timed out waiting for a peer's record (a rank is gone or out of step)"
                                                   : "in-launch record hand-off timed out (a workgroup's record never arrived)");
    }
    return h->log_cap ? log_step(h) : CTK_OK;
}

// Every launch attempt consumes a sequence number: if a step fails between its launch and finish_step, the next
// launch must not reuse the number (record words of the failed launch already carry it).
template <class F>
int guarded(ctk_handle* h, F&& body) {
    const uint32_t seq0 = h->seq;
    const int rc = body();
    if (rc != CTK_OK && h->seq == seq0) ++h->seq;
    return rc;
}

// ---- the resident form of the MPPI step (ctk_mppi.hip: ctk_mppi_resident) -------------------------------------------------------
// Ends a running resident kernel at once (cmd = EXIT) and waits for it; every API entry other than ctk_step does this first, so that
// nothing is ever queued behind it and nothing reads state it is still writing.
int resident_quiesce(ctk_handle* h) {
    if (!h->res_running) return CTK_OK;
    if (__atomic_load_n(&h->res_stat->state, __ATOMIC_ACQUIRE) != CTK_RES_LEFT) {
        volatile CtkResidentBox* b = h->res_box;
        b->cmd = CTK_RES_CMD_EXIT;
        b->tail = h->res_req + 1;
        __builtin_ia32_sfence();                         // (write-combined stores into device memory: payload before the number)
        b->req = ++h->res_req;
        __builtin_ia32_sfence();
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));       // bounded on the device side: the kernel leaves on EXIT, or after its idle time-out
    h->res_running = false;
    if (!h->res_dominant_saved.empty()) { h->dominant = h->res_dominant_saved; h->res_dominant_saved.clear(); }
    return CTK_OK;
}

bool resident_ok(const ctk_handle* h, int samples_loc) {
    return h->res_enabled && !h->mppi_pending && samples_loc != CTK_LOC_HOST && h->log_cap == 0 && !h->prof;
}

int resident_launch(ctk_handle* h, uint32_t first_req) {
    float zs[CTK_MAX_STATES] = {};
    RolloutArgs a = make_args(h, zs, nullptr, h->N, h->P);      // per-step fields come from the mailbox; u_prev_dev = d_u stays
    MppiFuse fz;
    fz.mode = 1; fz.ll = h->d_ll; fz.u_dev = h->d_u; fz.u_host = h->h_u_dev;
    __atomic_store_n(&h->res_stat->state, (uint32_t)CTK_RES_RUNNING, __ATOMIC_RELEASE);
    // the relay as the new session expects it: request number = the last one served, cmd = STEP (a session that left has raised EXIT there)
    h->res_relay_prime[0] = first_req - 1u; h->res_relay_prime[1] = CTK_RES_CMD_STEP;
    HIP_TRY(h, hipMemcpyAsync(h->d_res_relay, h->res_relay_prime, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, ctk_launch_mppi_resident(h->stream, h->env, h->params, h->cfg.dt, h->cfg.intermediate_steps, a, h->mk, h->d_unom[0], h->d_unom[1],
                                        h->d_parts, fz, h->res_box_dev, h->res_local ? 1 : 0, h->res_stat_dev, h->d_res_relay, h->res_idle_us, first_req, h->d_res_args, h->res_args_host));
    h->res_running = true;
    ++h->res_launches;
    if (h->res_dominant_saved.empty()) h->res_dominant_saved = h->dominant;      // restored when the kernel is ended
    h->dominant = ctk_mppi_resident_name(h->env);
    return CTK_OK;
}

// one MPPI step through the mailbox; samples: device pointer or nullptr (in-kernel sampler)
int resident_step(ctk_handle* h, const float* s, const float* u_prev, const float* d_samples, float* u_out) {
    volatile CtkResidentBox* b = h->res_box;
    CtkResidentStat* st = h->res_stat;
    if (h->res_running && __atomic_load_n(&st->state, __ATOMIC_ACQUIRE) == CTK_RES_LEFT) {      // it timed out idle since the last step
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->res_running = false;
    }
    const uint32_t req = h->res_req + 1;
    if (!h->res_running) if (int rc = resident_launch(h, req)) return rc;
    b->cmd = CTK_RES_CMD_STEP; b->seq = h->seq; b->call = h->call; b->cur = (uint32_t)h->cur; b->dev_uprev = u_prev ? 0u : 1u;
    b->samples = d_samples;
    for (int i = 0; i < h->S; ++i) b->s[i] = s[i];
    for (int c = 0; c < h->C; ++c) b->u_prev[c] = u_prev ? u_prev[c] : 0.0f;
    // a guess at the NEXT step's draws, so that the kernel can form that step's inputs while the host works on this one: the in-kernel
    // sampler's next position is known; for buffers, "what followed this buffer the last time it was used" (callers cycle through a pool)
    {
        const float* guess = nullptr;
        uint32_t known = d_samples == nullptr ? 1u : 0u;
        // Buffers are read ahead ONLY under the caller's promise that their contents do not change while the resident form is enabled
        // (ctk_resident_enable(h, 2, ...), include/ctk_hip.h).  Without it a caller who refills one buffer in place between steps would be
        // served the previous contents (a pointer cannot tell): every buffer step then forms its inputs at the request.
        if (d_samples != nullptr && h->res_readahead) {
            if (h->res_prev_samples != nullptr) h->res_follow[h->res_prev_samples] = d_samples;
            const auto it = h->res_follow.find(d_samples);
            if (it != h->res_follow.end()) { guess = it->second; known = 1u; }
            if (h->res_follow.size() > 4096) h->res_follow.clear();
        }
        h->res_prev_samples = d_samples;
        b->next_samples = guess; b->next_known = known;
    }
    b->tail = req;
    __builtin_ia32_sfence();                                    // payload before the number (write-combined stores when the box is device memory)
    b->req = req;
    __builtin_ia32_sfence();
    h->res_req = req;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const uint32_t want = h->seq;
    volatile uint32_t* slot = reinterpret_cast<volatile uint32_t*>(h->h_u) + 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
        if (*slot == want) break;
        if ((spin & 63) == 63) {
            if (__atomic_load_n(&st->state, __ATOMIC_ACQUIRE) == CTK_RES_LEFT && *slot != want) {
                // the kernel left without taking this request (idle time-out raced the request, or a device-side wait ran out): launch anew
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                h->res_running = false;
                if (*slot == want) break;
                if (int rc = resident_launch(h, req)) return rc;   // first_req = this request: it is served at once
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
                (void)resident_quiesce(h);
                return fail(h, CTK_ERR_HIP, "resident step: no result within 5 s");
            }
        }
        __builtin_ia32_pause();
    }
    ++h->res_steps;
    h->cur ^= 1;
    return finish_step(h, u_out);        // sees the result at once; error word, sequence / Philox counters as for a launched step
}

#define RES_Q(h) do { if ((h) != nullptr && (h)->res_running) { const int _q = resident_quiesce(h); if (_q != CTK_OK) return _q; } } while (0)

int check_predictor(ctk_handle* h) {
    if (h->cfg.predictor != CTK_PRED_ODE && !h->have_weights)
        return fail(h, CTK_ERR_STATE, "network predictor: call ctk_set_predictor_weights before stepping");
    return CTK_OK;
}

// ---- MPPI ------------------------------------------------------------------------------------
int mppi_block_parts(const ctk_handle* h) {
    if (h->generic && h->cfg.predictor != CTK_PRED_ODE) return ctk_g_rollout_net_blocks(h->env, h->net, CTK_G_MODE_MPPI, h->N, h->P, h->H);
    return h->generic ? ctk_g_rollout_blocks(h->N) : ctk_mppi_num_blocks(h->N, h->cfg.predictor);
}
// template path: the analytic predictor of ANY environment runs the 4-wave kernel of ctk_mppi.hip (in-launch hand-off included)
// below the throughput sizes; its network predictors keep the one-wave kernels of ctk_generic_net.hip
bool mppi_env_kernel(const ctk_handle* h) {
    return h->generic && h->cfg.predictor == CTK_PRED_ODE && !ctk_mppi_uses_throughput_kernel(CTK_PRED_ODE, h->N) &&
           ctk_mppi_rollout_env_lds(h->env, h->P, h->H, h->N) <= 160 * 1024;
}
bool mppi_can_fuse(const ctk_handle* h) {
    if (h->generic && h->cfg.predictor != CTK_PRED_ODE)     // network template kernels: the {value, seq} hand-off only
        return h->d_ll != nullptr && ctk_g_rollout_net_fusable(h->env, h->net, h->N, h->P, h->H);
    if (h->generic) return mppi_env_kernel(h) && ctk_mppi_fusable(h->PC, mppi_block_parts(h), h->d_ll != nullptr);
    return !ctk_mppi_uses_throughput_kernel(h->cfg.predictor, h->N) && ctk_mppi_fusable(h->P, mppi_block_parts(h), h->d_ll != nullptr);
}

// fuse_mode: 0 block records only; 1 the last block also merges + updates (single-GPU step);
//            2 the last block emits this shard's ONE record into partial_dev (sharded step_begin)
int mppi_rollout(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int loc, int fuse_mode,
                 float* partial_dev) {
    const float* d_s = nullptr;
    if (int rc = resolve_samples(h, samples, loc, (size_t)h->N * h->PC, &d_s)) return rc;
    RolloutArgs a = make_args(h, s, u_prev, h->N, h->P);
    const bool log = h->cfg.materialize_trajectories != 0;
    if (int rc = check_predictor(h)) return rc;
    if (mppi_env_kernel(h)) {
        MppiFuse fz;
        fz.mode = fuse_mode; fz.counter = h->d_counter; fz.out_rec = partial_dev; fz.ll = h->d_ll;
        fz.u_nom_out = h->d_unom[h->cur ^ 1]; fz.u_dev = h->d_u; fz.u_host = h->h_u_dev; fz.seq = h->seq;
        ProfSlot ps(h);
        HIP_TRY(h, ctk_launch_mppi_rollout_env(h->stream, h->env, h->params, h->cfg.dt, h->cfg.intermediate_steps, a, h->mk, d_s, h->d_unom[h->cur],
                                               h->d_parts, log, fz, ps.a, ps.b));
        return CTK_OK;
    }
    if (h->generic) {   // template kernels: network predictors hand their records over in-launch when asked to; else block records only
        ProfSlot ps(h);
        if (h->cfg.predictor != CTK_PRED_ODE) {
            MppiFuse fz;
            fz.mode = fuse_mode; fz.out_rec = partial_dev; fz.ll = h->d_ll;
            fz.u_nom_out = h->d_unom[h->cur ^ 1]; fz.u_dev = h->d_u; fz.u_host = h->h_u_dev; fz.seq = h->seq;
            HIP_TRY(h, ctk_launch_g_rollout_net(h->stream, h->env, h->net, CTK_G_MODE_MPPI, a, h->params, h->cfg.dt, h->cfg.intermediate_steps,
                                                h->mk, d_s, h->d_unom[h->cur], nullptr, 0, h->d_wperm, h->d_parts, log, ps.a, ps.b, &fz));
        } else
            HIP_TRY(h, ctk_launch_g_rollout(h->stream, h->env, CTK_G_MODE_MPPI, a, h->params, h->cfg.dt, h->cfg.intermediate_steps, h->mk, d_s,
                                            h->d_unom[h->cur], nullptr, 0, h->d_parts, log, ps.a, ps.b));
        return CTK_OK;
    }
    MppiFuse fz;
    fz.mode = fuse_mode; fz.counter = h->d_counter; fz.out_rec = partial_dev; fz.ll = h->d_ll;
    if (fuse_mode == 3) { fz.p2p = h->d_p2p_args; fz.p2p_seq = h->p2p_seq; fz.p2p_world = h->p2p_world; }
    fz.u_nom_out = h->d_unom[h->cur ^ 1]; fz.u_dev = h->d_u; fz.u_host = h->h_u_dev; fz.seq = h->seq;
    ProfSlot ps(h);
    const char* ran = nullptr;
    HIP_TRY(h, ctk_launch_mppi_rollout(h->stream, h->cfg.predictor, a, h->k, h->mk, d_s, h->d_unom[h->cur], h->d_wperm,
                                       h->d_parts, log, fz, ps.a, ps.b, &ran));
    if (ran != h->dominant_ran) { h->dominant_ran = ran; h->dominant = ran; }   // string literals: pointer compare
    return CTK_OK;
}

// reduce the block records to <= 2048 records (hierarchical when the grid was huge)
int mppi_reduce_blocks(ctk_handle* h, const float** parts, int* n_parts) {
    // tree of 32-way merges (each a grid of 256-thread blocks with the records staged in LDS) until one
    // block can take the rest: N = 65536 -> 1024 -> 32 records; N = 4 M -> 65536 -> 2048 -> 64
    int n = mppi_block_parts(h);
    const float* src = h->d_parts;
    float* bufs[2] = {h->d_parts2, h->d_parts3};
    int b = 0;
    while (n > 64) {
        HIP_TRY(h, ctk_launch_mppi_merge_partial(h->stream, src, n, 32, h->PC, h->mk.neg_inv_lbd, bufs[b]));
        n = (n + 31) / 32;
        src = bufs[b];
        b ^= 1;
    }
    *parts = src; *n_parts = n;
    return CTK_OK;
}

// optimizer_mppi.py:192,195-197: predictor.update(s, u_nom[:, :1]) — behind the update on the stream, off the
// caller's latency path (the step's result is already published when this runs)
// where the GRU's carried hidden state [2,32] lives: behind the per-lane tables of whichever kernel family the handle runs
float* gru_hidden(ctk_handle* h) { return h->d_wperm + (h->generic ? ctk_g_net_table_floats(CTK_PRED_GRU) : (size_t)GRU_TABLE_FLOATS); }

int gru_advance(ctk_handle* h, const float* s, const float* u_host) {   // u_host NULL: the optimizer's last output on the device
    if (h->generic) {
        float zero_u[CTK_MAX_INPUTS] = {};
        const RolloutArgs a = make_args(h, s, u_host ? u_host : zero_u, 1, 1);
        HIP_TRY(h, ctk_launch_g_gru_advance(h->stream, h->env, a, u_host ? nullptr : h->d_u, h->d_wperm));
    } else {
        HIP_TRY(h, ctk_launch_gru_advance(h->stream, s, u_host ? nullptr : h->d_u, u_host ? u_host[0] : 0.0f, h->d_wperm));
    }
    return CTK_OK;
}

int mppi_advance_hidden(ctk_handle* h) {
    if (h->cfg.predictor != CTK_PRED_GRU) return CTK_OK;
    return gru_advance(h, h->mppi_s, nullptr);
}

int mppi_update(ctk_handle* h, const float* parts, int n_parts, float* u_out) {
    const int nxt = h->cur ^ 1;
    if (h->generic) {
        // merge the records into ONE (the same merge kernel the CartPole path uses; it only sees P*C columns), then the
        // per-channel update u_nom <- clip(shift(u_nom) + interp(b)/a)
        const float* rec = parts;
        if (n_parts > ctk_g_mppi_update_max_parts()) {   // (mppi_reduce_blocks leaves <= 64: not reached today)
            HIP_TRY(h, ctk_launch_mppi_merge_partial(h->stream, parts, n_parts, n_parts, h->PC, h->mk.neg_inv_lbd, h->d_rec));
            rec = h->d_rec; n_parts = 1;
        }
        float zero_s[CTK_MAX_STATES] = {};
        const RolloutArgs a = make_args(h, zero_s, nullptr, h->N, h->P);   // limits per input
        HIP_TRY(h, ctk_launch_g_mppi_update(h->stream, rec, n_parts, h->mk.neg_inv_lbd, h->P, h->C, h->H, h->d_interp, h->d_unom[h->cur],
                                            h->d_unom[nxt], a, h->d_u, h->h_u_dev, h->seq));
        h->cur = nxt;
        if (int rc = mppi_advance_hidden(h)) return rc;   // optimizer_mppi.py:192 (RNN hidden state), behind the update on the stream
        return finish_step(h, u_out);
    }
    HIP_TRY(h, ctk_launch_mppi_update(h->stream, parts, n_parts, h->P, h->mk.neg_inv_lbd, h->H, h->d_interp, h->d_unom[h->cur],
                                      h->d_unom[nxt], h->cfg.action_low[0], h->cfg.action_high[0], h->d_u, h->h_u_dev, h->seq));
    h->cur = nxt;
    if (int rc = mppi_advance_hidden(h)) return rc;
    return finish_step(h, u_out);
}

// u[n,h,c] = clip(base[h,c] + sample[n,h,c] * scale[h,c]) rollouts (CEM, random-action, plain), tuned or template kernel
// template path: the analytic predictor of ANY environment runs the 4-wave kernel of ctk_sampled.hip (ctk_affine_rollout<ENV, ODE, .>)
bool affine_env_kernel(const ctk_handle* h) {
    return h->generic && h->cfg.predictor == CTK_PRED_ODE && ctk_affine_rollout_env_lds(h->env, h->H) <= 160 * 1024;
}

int launch_affine(ctk_handle* h, const RolloutArgs& a, const float* d_s, int rng_kind, const float* base, const float* scale, bool log,
                  const AffineBest* bst = nullptr) {
    ProfSlot ps(h);
    if (affine_env_kernel(h))
        HIP_TRY(h, ctk_launch_affine_rollout_env(h->stream, h->env, h->params, h->cfg.dt, h->cfg.intermediate_steps, a, d_s, rng_kind, base, scale,
                                                 log, ps.a, ps.b, bst));
    else if (h->generic && h->cfg.predictor != CTK_PRED_ODE)
        HIP_TRY(h, ctk_launch_g_rollout_net(h->stream, h->env, h->net, CTK_G_MODE_AFFINE, a, h->params, h->cfg.dt, h->cfg.intermediate_steps,
                                            h->mk, d_s, base, scale, rng_kind, h->d_wperm, nullptr, log, ps.a, ps.b));
    else if (h->generic)
        HIP_TRY(h, ctk_launch_g_rollout(h->stream, h->env, CTK_G_MODE_AFFINE, a, h->params, h->cfg.dt, h->cfg.intermediate_steps, h->mk, d_s, base,
                                        scale, rng_kind, nullptr, log, ps.a, ps.b));
    else
        HIP_TRY(h, ctk_launch_affine_rollout(h->stream, h->cfg.predictor, a, h->k, d_s, rng_kind, base, scale, h->d_wperm, log, ps.a, ps.b, bst));
    return CTK_OK;
}

// the Adam / SGD descent launch of the RPGD family, tuned (CartPole) or template kernel
int launch_descent(ctk_handle* h, const RolloutArgs& a, float lr, float b1, float b2, float eps, float* Q, float* m, float* v,
                   const float* bc, int bc_len, int t0, int iters, int rule, const RpgdFusedWarm* fused = nullptr) {
    const ctk_config& c = h->cfg;
    ProfSlot ps(h);
    if (h->generic && c.predictor != CTK_PRED_ODE)
        HIP_TRY(h, ctk_launch_g_rpgd_descent_net(h->stream, h->env, h->net, a, h->params, c.dt, c.intermediate_steps, lr, b1, b2, eps,
                                                 c.gradmax_clip, Q, m, v, bc, bc_len, t0, iters, h->d_wperm, h->d_scratch, ps.a, ps.b, rule,
                                                 reinterpret_cast<uint32_t*>(h->h_u_dev) + 2, &h->rp_pers));     // (the error word behind {u, seq})
    else if (h->generic)
        HIP_TRY(h, ctk_launch_g_rpgd_descent(h->stream, h->env, a, h->params, c.dt, c.intermediate_steps, lr, b1, b2, eps, c.gradmax_clip, Q, m, v,
                                             bc, bc_len, t0, iters, h->d_scratch, ps.a, ps.b, rule));
    else
        HIP_TRY(h, ctk_launch_rpgd_descent(h->stream, c.predictor, a, h->k, lr, b1, b2, eps, c.gradmax_clip, Q, m, v, bc, bc_len, t0, iters,
                                           h->d_wperm, h->d_scratch, ps.a, ps.b, rule, fused, &h->rp_pers));
    return CTK_OK;
}

// random-action: u = first input of the cheapest plan (optimizer_random_action_tf.py:65-68)
int launch_pick_best(ctk_handle* h, const float* Q, const int* idx, int ldq) {
    if (h->generic) HIP_TRY(h, ctk_launch_g_pick_best_first(h->stream, Q, idx, h->C, h->d_u, h->h_u_dev, h->seq, ldq));
    else HIP_TRY(h, ctk_launch_pick_best_first(h->stream, Q, idx, h->H, h->d_u, h->h_u_dev, h->seq, ldq));
    return CTK_OK;
}

// CEM post-loop (optimizer_cem_tf.py:99-102) + publishing u, tuned or template kernel
int launch_cem_finish(ctk_handle* h, const float* Q, const int* idx, int ldq, float std_max, int u_from_mu) {
    const ctk_config& c = h->cfg;
    if (h->generic) {
        float zero_s[CTK_MAX_STATES] = {};
        const RolloutArgs a = make_args(h, zero_s, nullptr, h->N, h->H);
        HIP_TRY(h, ctk_launch_g_cem_finish(h->stream, Q, idx, h->H, h->C, h->d_unom[0], h->d_std, c.cem_stdev_min, c.cem_initial_action_stdev, a,
                                           h->d_u, h->h_u_dev, h->seq, ldq, std_max, u_from_mu));
    } else {
        const float mid = (c.action_low[0] + c.action_high[0]) * 0.5f;
        HIP_TRY(h, ctk_launch_cem_finish(h->stream, Q, idx, h->H, h->d_unom[0], h->d_std, c.cem_stdev_min, c.cem_initial_action_stdev, mid,
                                         h->d_u, h->h_u_dev, h->seq, ldq, std_max, u_from_mu));
    }
    return CTK_OK;
}

// ---- cem-grad-bharadhwaj (variant of the CEM family) ---------------------------------------------
// optimizer_cem_grad_bharadhwaj_tf.py:151-178.  Per outer iteration (:93-120): population = [elites | fresh
// samples], ONE Keras-Adam step on all of it (the optimizer's moments persist by POSITION across iterations
// and MPC steps and are never shifted or reset, exactly like the tf.Variable / Keras pair of the reference),
// cost pass, best K -> next iteration's elites, refit.
int cem_bharadhwaj_step(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int loc, float* u_out) {
    const ctk_config& c = h->cfg;
    const int its = cem_iterations(h), K = c.cem_best_k;
    const size_t H = h->HC, n_el = (size_t)K * H, n_rest = (size_t)(h->N - K) * H;
    const float* d_s = nullptr;
    if (int rc = resolve_samples(h, samples, loc, n_el + its * n_rest, &d_s)) return rc;
    float* mu = h->d_unom[0];
    int cur = h->rcur;
    for (int it = 0; it < its; ++it) {
        RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);
        a.stream_id = (uint32_t)it;
        HIP_TRY(h, ctk_launch_cem_build_population(h->stream, a, K, it == 0 ? 1 : 0, h->d_pop[cur ^ 1], h->d_idx, d_s,
                                                   d_s ? d_s + n_el + it * n_rest : nullptr, mu, h->d_std, h->d_pop[cur]));
        if (int rc = launch_descent(h, a, c.learning_rate, c.adam_beta_1, c.adam_beta_2, c.adam_epsilon, h->d_pop[cur], h->d_m[0], h->d_v[0],
                                    h->d_bc, h->bc_len, h->adam_step, 1, 1)) return rc;
        ++h->adam_step;
        HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, K, h->d_idx));
        HIP_TRY(h, ctk_launch_cem_refit(h->stream, h->d_pop[cur], h->d_idx, K, h->HC, mu, h->d_std, h->HC));
        if (it + 1 < its) cur ^= 1;   // the refined population becomes Q_prev of the next build
    }
    h->rcur = cur;
    if (int rc = launch_cem_finish(h, h->d_pop[cur], h->d_idx, h->HC, 10.0f, 0)) return rc;   // :167-168,:130-141
    ++h->count;
    return finish_step(h, u_out);
}

// ---- CEM --------------------------------------------------------------------------------------
int cem_step(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int loc, float* u_out) {
    if (int rc = check_predictor(h)) return rc;
    if (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ) return cem_bharadhwaj_step(h, s, u_prev, samples, loc, u_out);
    const int its = cem_iterations(h);
    const size_t per_it = (size_t)h->N * h->HC;
    const float* d_s = nullptr;
    if (int rc = resolve_samples(h, samples, loc, per_it * its, &d_s)) return rc;
    const bool log = h->cfg.materialize_trajectories != 0;
    float* mu = h->d_unom[0];
    if (h->variant == CTK_OPT_CEM_NAIVE_GRAD) {
        // optimizer_cem_naive_grad_tf.py:57-87 per outer iteration: sample, ONE clipped-gradient SGD step on every
        // sample (:63-71), roll the moved samples out again (:73-74), elite refit (:77-83)
        const ctk_config& c = h->cfg;
        for (int it = 0; it < its; ++it) {
            RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);
            a.stream_id = (uint32_t)it;
            HIP_TRY(h, ctk_launch_sample_plans(h->stream, a, d_s ? d_s + per_it * it : nullptr, mu, h->d_std, h->d_pop[0]));
            if (int rc = launch_descent(h, a, c.learning_rate, 0.0f, 0.0f, 0.0f, h->d_pop[0], nullptr, nullptr, nullptr, 0, 0, 1, 2)) return rc;
            HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, c.cem_best_k, h->d_idx));
            HIP_TRY(h, ctk_launch_cem_refit(h->stream, h->d_pop[0], h->d_idx, c.cem_best_k, h->HC, mu, h->d_std, h->HC));
        }
        if (int rc = launch_cem_finish(h, h->d_pop[0], h->d_idx, h->HC, 10.0f, 1)) return rc;   // :101-104
        ++h->count;
        return finish_step(h, u_out);
    }
    if (h->d_cem_ll && h->variant == CTK_OPT_CEM && h->cfg.cem_best_k <= h->N) {
        // ONE launch: all outer iterations, selection and refit between workgroups inside it (ctk_cem_fused.hip)
        RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);
        const ctk_config& c = h->cfg;
        if ((uint32_t)(h->cem_tag + (uint32_t)its) < h->cem_tag) h->cem_tag = 1;   // wrapped: tag 0 is the never-written value
        const CemFusedLaunch cl{its, c.cem_best_k, h->d_cem_ll, h->cem_tag, c.cem_stdev_min, 1.0e8f, c.cem_initial_action_stdev,
                                mu, h->d_std, h->d_u, h->h_u_dev, h->d_idx, h->seq, 0.5};
        h->cem_tag += (uint32_t)its;
        ProfSlot ps(h);
        HIP_TRY(h, ctk_launch_cem_fused(h->stream, h->env, h->params, c.dt, c.intermediate_steps, a, d_s, cl, log, ps.a, ps.b));
        h->idx_stale = true;
        ++h->count;
        return finish_step(h, u_out);
    }
    for (int it = 0; it < its; ++it) {
        RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);
        a.stream_id = (uint32_t)it;
        if (int rc = launch_affine(h, a, d_s ? d_s + per_it * it : nullptr, 0, mu, h->d_std, log)) return rc;
        HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, h->cfg.cem_best_k, h->d_idx));
        HIP_TRY(h, ctk_launch_cem_refit(h->stream, h->d_Q, h->d_idx, h->cfg.cem_best_k, h->HC, mu, h->d_std, h->HC));
    }
    if (int rc = launch_cem_finish(h, h->d_Q, h->d_idx, h->HC, 1.0e8f, 0)) return rc;
    ++h->count;
    return finish_step(h, u_out);
}

// ---- random action ----------------------------------------------------------------------------
int random_step(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int loc, float* u_out) {
    if (int rc = check_predictor(h)) return rc;
    const float* d_s = nullptr;
    if (int rc = resolve_samples(h, samples, loc, (size_t)h->N * h->HC, &d_s)) return rc;
    RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);
    const bool log = h->cfg.materialize_trajectories != 0;
    static const bool three_launches = std::getenv("CTK_NO_FUSED_ARGMIN") != nullptr;   // A/B switch
    if ((!h->generic || affine_env_kernel(h)) && h->cfg.predictor == CTK_PRED_ODE && h->d_ll && !three_launches &&
        ctk_affine_rollout_blocks(h->cfg.predictor, h->N) <= CTK_AFFINE_BEST_MAX_BLOCKS) {
        // ONE launch: rollout + arg-min over the block minima + u (optimizer_random_action_tf.py:62-68)
        const AffineBest bst{h->d_ll, h->seq, h->d_u, h->h_u_dev, h->d_idx};
        if (int rc = launch_affine(h, a, d_s, 1, h->d_base, h->d_scale, log, &bst)) return rc;
        return finish_step(h, u_out);
    }
    if (int rc = launch_affine(h, a, d_s, 1, h->d_base, h->d_scale, log)) return rc;
    HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, 1, h->d_idx));
    if (int rc = launch_pick_best(h, h->d_Q, h->d_idx, h->HC)) return rc;
    return finish_step(h, u_out);
}

// ---- RPGD --------------------------------------------------------------------------------------
int rpgd_iterations(const ctk_handle* h) {   // optimizer_rpgd.py:219-221,397-400
    const int first = h->cfg.warmup ? h->cfg.warmup_iterations : h->cfg.outer_its;
    return h->count == 0 ? first : h->cfg.outer_its;
}

int rpgd_warm(ctk_handle* h, const RolloutArgs& a, int n_new, int gather, int reset, const float* d_draws, int from, int to,
              const int* idx = nullptr, const float* recs = nullptr, int keeper_base = 0, int fresh_tail = 0) {
    const ctk_config& c = h->cfg;
    HIP_TRY(h, ctk_launch_rpgd_warmstart(h->stream, a, h->N, h->H, h->P, n_new, gather, c.shift_previous, c.sampling_distribution,
                                         reset, c.sample_whole_control_space, c.sample_stdev, c.sample_mean, c.sample_min,
                                         c.sample_max, d_draws, idx ? idx : h->d_idx, h->d_pop[from], h->d_m[from], h->d_v[from],
                                         h->d_ages[from], h->d_pop[to], h->d_m[to], h->d_v[to], h->d_ages[to], h->d_interp,
                                         h->d_unom[0], h->d_u, h->h_u_dev, h->seq, recs, 3 + 3 * h->HC, keeper_base, fresh_tail));
    return CTK_OK;
}

int rpgd_descent(ctk_handle* h, const float* s, const float* u_prev, const RpgdFusedWarm* fused = nullptr) {
    const ctk_config& c = h->cfg;
    const int iters = rpgd_iterations(h);
    RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);   // p_magic divides by H in the descent kernel
    if (int rc = launch_descent(h, a, c.learning_rate, c.adam_beta_1, c.adam_beta_2, c.adam_epsilon, h->d_pop[h->rcur], h->d_m[h->rcur],
                                h->d_v[h->rcur], h->d_bc, h->bc_len, h->adam_step, iters, (h->variant == CTK_OPT_GRADIENT || c.adam_rule == 1) ? 1 : 0, fused)) return rc;
    h->adam_step += iters;
    return CTK_OK;
}

// optimizer_rpgd.py:340-343,:424,:431: get_action's rollout of the DESCENDED plans is what the reference logs as
// `rollout_trajectories`.  The descent kernels keep their states in LDS / tape scratch; with cfg.materialize_trajectories the same plans
// go through the plain rollout kernel once more (limits as given: u = clip(0 + plan * 1) is the plan), its costs into a scratch vector
// so that the keep-k selection still sees the descent's own J.  Logging mode only: one more launch per step.
int rpgd_materialize(ctk_handle* h, const float* s, const float* u_prev, const float* d_plans) {
    if (!h->cfg.materialize_trajectories || !h->d_traj || !h->d_Jlog) return CTK_OK;
    RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);
    a.J = h->d_Jlog;                                       // (Q_out stays d_Q: the tuned kernels store u_run whenever they log; unused by RPGD)
    return launch_affine(h, a, d_plans, 0, h->d_base, h->d_scale, true);
}

int rpgd_reset(ctk_handle* h, const float* draws, int loc) {
    const float* d_draws = nullptr;
    if (int rc = resolve_samples(h, draws, loc, (size_t)h->N * h->PC, &d_draws)) return rc;
    float zero_s[CTK_MAX_STATES] = {};
    RolloutArgs a = make_args(h, zero_s, nullptr, h->N, h->P);
    if (int rc = rpgd_warm(h, a, h->N, 0, 1, d_draws, h->rcur, h->rcur)) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->count = 0; h->adam_step = 0; h->rpgd_ready = true;
    ++h->call;
    return CTK_OK;
}

// description of the warm start for the single-launch step (N <= 64): the arguments rpgd_warm would pass
RpgdFusedWarm rpgd_fused(ctk_handle* h, int K, int n_new, int gather, const float* d_draws, int from, int to, int fresh_tail) {
    const ctk_config& c = h->cfg;
    return RpgdFusedWarm{K, h->d_idx, h->P, n_new, gather, c.shift_previous, c.sampling_distribution, fresh_tail,
                         c.sample_whole_control_space, c.sample_stdev, c.sample_mean, c.sample_min, c.sample_max, d_draws, h->d_ages[from],
                         h->d_pop[to], h->d_m[to], h->d_v[to], h->d_ages[to], h->d_interp, h->d_unom[0], h->d_u, h->h_u_dev, h->seq};
}
bool rpgd_can_fuse(const ctk_handle* h) {
    static const bool off = std::getenv("CTK_NO_RPGD_FUSED") != nullptr;   // A/B switch
    // (materialised trajectories: the logging rollout goes between the descent and the warm start, which the single launch has no seam for)
    return !off && !h->generic && !h->cfg.materialize_trajectories && h->N <= ctk_rpgd_fused_max_n(h->cfg.predictor, h->N);
}

int rpgd_step(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int loc, float* u_out) {
    if (!h->rpgd_ready) return fail(h, CTK_ERR_STATE, "RPGD: call ctk_reset (optimizer_reset) before the first step");
    if (int rc = check_predictor(h)) return rc;
    const ctk_config& c = h->cfg;
    if (c.opt_keep_k > h->N) return fail(h, CTK_ERR_INVALID_ARGUMENT, "RPGD: opt_keep_k exceeds this handle's rollouts (a shard? use ctk_rpgd_step_begin/end)");
    const int cur = h->rcur, nxt = cur ^ 1;
    if (h->variant == CTK_OPT_GRADIENT) {
        // optimizer_gradient_tf.py:101-173: Keras-Adam descent on all N plans, u = best plan's first input,
        // every plan shifted by one with a FRESH uniform tail input (:137-144), moments shifted by one (:147-166)
        const float* d_tail = nullptr;
        if (int rc = resolve_samples(h, samples, loc, (size_t)h->N * h->C, &d_tail)) return rc;
        if (rpgd_can_fuse(h)) {
            const RpgdFusedWarm fw = rpgd_fused(h, 1, 0, 0, d_tail, cur, nxt, 1);
            if (int rc = rpgd_descent(h, s, u_prev, &fw)) return rc;
        } else {
            if (int rc = rpgd_descent(h, s, u_prev)) return rc;
            if (int rc = rpgd_materialize(h, s, u_prev, h->d_pop[cur])) return rc;
            HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, 1, h->d_idx));
            RolloutArgs ag = make_args(h, s, u_prev, h->N, h->P);
            if (int rc = rpgd_warm(h, ag, 0, 0, 0, d_tail, cur, nxt, nullptr, nullptr, 0, 1)) return rc;
        }
        h->rcur = nxt;
        ++h->count;
        return finish_step(h, u_out);
    }
    const bool resample = (h->count % c.resamp_per) == 0;                      // :449
    const float* d_draws = nullptr;
    if (resample)
        if (int rc = resolve_samples(h, samples, loc, (size_t)(h->N - c.opt_keep_k) * h->PC, &d_draws)) return rc;
    if (rpgd_can_fuse(h)) {   // one workgroup holds the population: descent, keep-k and warm start in ONE launch
        const RpgdFusedWarm fw = rpgd_fused(h, c.opt_keep_k, resample ? h->N - c.opt_keep_k : 0, resample ? 1 : 0, d_draws, cur, nxt, 0);
        if (int rc = rpgd_descent(h, s, u_prev, &fw)) return rc;
    } else {
        if (int rc = rpgd_descent(h, s, u_prev)) return rc;
        if (int rc = rpgd_materialize(h, s, u_prev, h->d_pop[cur])) return rc;
        HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, c.opt_keep_k, h->d_idx));   // :345-346
        RolloutArgs aw = make_args(h, s, u_prev, h->N, h->P);
        if (int rc = rpgd_warm(h, aw, resample ? h->N - c.opt_keep_k : 0, resample ? 1 : 0, 0, d_draws, cur, nxt)) return rc;
    }
    h->rcur = nxt;
    ++h->count;
    return finish_step(h, u_out);
}

// d[i] = vals[i % C] for i < n (a [H,C] tensor whose rows are all `vals`)
int fill_rows(ctk_handle* h, float* d, const float* vals, int C, int n) {
    std::vector<float> tmp((size_t)n);
    for (int i = 0; i < n; ++i) tmp[(size_t)i] = vals[i % C];
    HIP_TRY(h, hipMemcpyAsync(d, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTK_OK;
}

int fill_const(ctk_handle* h, float* d, float v, int n) {
    std::vector<float> tmp((size_t)n, v);
    HIP_TRY(h, hipMemcpyAsync(d, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTK_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int ctk_abi_version(void) { return CTK_ABI_VERSION; }
static_assert(CTK_P_COUNT <= CTK_MAX_PARAMS && CTK_Q_COUNT <= CTK_MAX_PARAMS && CTK_V_COUNT <= CTK_MAX_PARAMS, "parameter table size");
static_assert(Env<CTK_ENV_QUAD2D>::S <= CTK_MAX_STATES && Env<CTK_ENV_QUAD2D>::C <= CTK_MAX_INPUTS, "environment dimensions");
static_assert(Env<CTK_ENV_HOVER>::S <= CTK_MAX_STATES && Env<CTK_ENV_HOVER>::C <= CTK_MAX_INPUTS, "environment dimensions");

const char* ctk_last_error(const ctk_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ctk_create(const ctk_config* cfg, ctk_handle** out) {
    if (out) *out = nullptr;
    if (!cfg || !out) return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: NULL argument");
    if (cfg->struct_size != sizeof(ctk_config))
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: ctk_config size mismatch (ABI)");
    const EnvInfo* einfo = env_info(cfg->environment);
    if (!einfo) return fail(nullptr, CTK_ERR_UNSUPPORTED, "ctk_create: unknown environment (built: CartPole, Quad2D, Hover; CTK_ENV_USER only in a library compiled with a user model, control_toolkit_amd/build_env.py)");
    if (cfg->num_states != einfo->S || cfg->num_control_inputs != einfo->C)
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, std::string("ctk_create: environment ") + einfo->name + " has num_states == " +
                    std::to_string(einfo->S) + ", num_control_inputs == " + std::to_string(einfo->C));
    if (cfg->num_rollouts < 1 || cfg->mpc_horizon < 1 || cfg->mpc_horizon > 1024)
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: need num_rollouts >= 1 and 1 <= mpc_horizon <= 1024");
    if (cfg->period_interpolation_inducing_points < 1 || cfg->intermediate_steps < 1)
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: period_interpolation_inducing_points and intermediate_steps must be >= 1");
    if (!(cfg->dt > 0.0f)) return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: dt must be > 0");
    if ((long long)cfg->num_rollouts * (cfg->mpc_horizon + 1) * std::max(einfo->S, einfo->C) > (1ll << 30))
        return fail(nullptr, CTK_ERR_UNSUPPORTED, "ctk_create: num_rollouts * (mpc_horizon + 1) * num_states must stay below 2^30 (32-bit element indices)");
    for (int c = 0; c < einfo->C; ++c)
        if (!(cfg->action_low[c] <= cfg->action_high[c])) return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: action_low must be <= action_high for every control input");
    if (cfg->optimizer < CTK_OPT_MPPI || cfg->optimizer > CTK_OPT_CEM_GRAD_BHARADHWAJ)
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: unknown optimizer");
    if (cfg->predictor < CTK_PRED_ODE || cfg->predictor > CTK_PRED_GRU) return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: unknown predictor");
    if (cfg->predictor != CTK_PRED_ODE && einfo->S + einfo->C > 12)
        return fail(nullptr, CTK_ERR_UNSUPPORTED, std::string("ctk_create: the network predictors take at most 12 network inputs (num_states + num_control_inputs: three "
                    "layer-1 k-steps of four); ") + einfo->name + " has " + std::to_string(einfo->S + einfo->C));
    // variants run on an engine family: gradient = RPGD machinery without resampling (Keras Adam, fresh tail);
    // cem-naive-grad = CEM machinery with one SGD step on the samples
    ctk_config mapped = *cfg;
    if (cfg->optimizer == CTK_OPT_GRADIENT) {
        mapped.optimizer = CTK_OPT_RPGD;
        mapped.period_interpolation_inducing_points = 1;   // plans are sampled per step, no inducing points (:176-181)
        mapped.sampling_distribution = 0; mapped.sample_whole_control_space = 1;
        mapped.sample_min = cfg->action_low[0]; mapped.sample_max = cfg->action_high[0];
        mapped.shift_previous = 1; mapped.opt_keep_k = 1; mapped.resamp_per = 0x7FFFFFFF;
    } else if (cfg->optimizer == CTK_OPT_CEM_NAIVE_GRAD || cfg->optimizer == CTK_OPT_CEM_GRAD_BHARADHWAJ) {
        mapped.optimizer = CTK_OPT_CEM;
        if (cfg->optimizer == CTK_OPT_CEM_NAIVE_GRAD) mapped.warmup = 0;   // the reference has no warm-up for this optimizer (:96)
        if (cfg->intermediate_steps != 1)
            return fail(nullptr, CTK_ERR_UNSUPPORTED, "ctk_create: the gradient kernels are built for intermediate_steps == 1");
    }
    const int variant = cfg->optimizer;
    cfg = &mapped;
    if (cfg->predictor != CTK_PRED_ODE && cfg->predictor != CTK_PRED_MLP && cfg->predictor != CTK_PRED_GRU)
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: unknown predictor");
    // the template kernels roll the analytic model out; the network predictors have (so far) CartPole-shaped MFMA kernels only
    // template kernels: every environment but CartPole; CartPole on request; and the gradient-based optimizers with the
    // recurrent predictor (reverse mode through the GRU = NetGru::Bwd of ctk_net.h; CartPole's 4-wave GRU kernels are forward only)
    const bool grad_family = cfg->optimizer == CTK_OPT_RPGD || variant != cfg->optimizer;
    // hidden widths of a network predictor (the <h1>H1-<h2>H2 of the reference's network names): 0 = 32; up to 32 the 32-unit kernels;
    // 33..64 (MLP) the 64-unit form of the one-wave template kernels (ctk_mlp_wide.h)
    const int hw1 = cfg->predictor_hidden1 ? cfg->predictor_hidden1 : 32, hw2 = cfg->predictor_hidden2 ? cfg->predictor_hidden2 : 32;
    if (hw1 < 1 || hw2 < 1) return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: predictor_hidden1 / predictor_hidden2 must be >= 1 (0 = 32)");
    const bool wide_net = cfg->predictor != CTK_PRED_ODE && std::max(hw1, hw2) > 32;
    if (wide_net && (cfg->predictor != CTK_PRED_MLP || std::max(hw1, hw2) > MLPW_HID))
        return fail(nullptr, CTK_ERR_UNSUPPORTED, std::string("ctk_create: network ") + std::to_string(einfo->S + einfo->C) + "IN-" + std::to_string(hw1) + "H1-" +
                    std::to_string(hw2) + "H2-" + std::to_string(einfo->S) + "OUT: built are MLPs with hidden layers of up to 64 units and GRUs of up to 32");
    const bool generic = cfg->environment != CTK_ENV_CARTPOLE || cfg->generic_kernels != 0 || (cfg->predictor == CTK_PRED_GRU && grad_family) || wide_net;
    if (cfg->optimizer == CTK_OPT_CEM && (cfg->cem_best_k < 1 || cfg->cem_best_k > cfg->num_rollouts || cfg->cem_outer_it < 1))
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: need 1 <= cem_best_k <= num_rollouts and cem_outer_it >= 1");
    if (cfg->optimizer == CTK_OPT_RPGD) {
        if (cfg->opt_keep_k < 1 || cfg->outer_its < 0 || cfg->resamp_per < 1 ||
            cfg->shift_previous < 0 || (cfg->sampling_distribution != 0 && cfg->sampling_distribution != 1))
            return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: RPGD needs opt_keep_k >= 1 (<= the GLOBAL population), outer_its >= 0, resamp_per >= 1, shift_previous >= 0, sampling_distribution in {0,1}");
        if (cfg->intermediate_steps != 1)
            return fail(nullptr, CTK_ERR_UNSUPPORTED, "ctk_create: the RPGD adjoint is built for intermediate_steps == 1");
        if (cfg->adam_rule != 0 && cfg->adam_rule != 1)
            return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: adam_rule must be 0 (the reference's torch ADAM, optimizer_rpgd.py:56-82) or 1 (tf.keras.optimizers.Adam)");
    }
    if (cfg->optimizer == CTK_OPT_MPPI && !(cfg->LBD > 0.0f && cfg->NU != 0.0f))
        return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: MPPI needs LBD > 0 and NU != 0");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, CTK_ERR_NO_DEVICE, "ctk_create: no HIP device visible (libctk_hip has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, CTK_ERR_INVALID_ARGUMENT, "ctk_create: bad device ordinal");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, CTK_ERR_NO_DEVICE, "ctk_create: hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CTK_ERR_NO_DEVICE, std::string("ctk_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);

    ctk_handle* h = new ctk_handle();
    h->cfg = *cfg;
    h->variant = variant;
    h->N = cfg->num_rollouts; h->H = cfg->mpc_horizon;
    h->env = cfg->environment; h->S = einfo->S; h->C = einfo->C; h->generic = generic;
    h->net = wide_net ? NET_MLP64 : cfg->predictor; h->hid = wide_net ? MLPW_HID : 32;
    const bool interp = (cfg->optimizer == CTK_OPT_MPPI || cfg->optimizer == CTK_OPT_RPGD);
    h->P = interp ? num_inducing_points(h->H, cfg->period_interpolation_inducing_points) : h->H;
    h->HC = h->H * h->C; h->PC = h->P * h->C;
    default_params(h->env, h->params);
    refresh_constants(h);

    auto bail = [&](int rc) { g_create_error = h->err; ctk_destroy(h); return rc; };
#define TRY_CREATE(expr) do { int _rc = (expr); if (_rc) return bail(_rc); } while (0)
#define HIP_CREATE(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(_e); return bail(CTK_ERR_HIP); } } while (0)

    HIP_CREATE(hipSetDevice(cfg->device));
    HIP_CREATE(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;

    const size_t N = h->N, H = h->H, P = h->P, HC = h->HC, PC = h->PC;
    const bool descends = cfg->optimizer == CTK_OPT_RPGD || variant == CTK_OPT_CEM_NAIVE_GRAD || variant == CTK_OPT_CEM_GRAD_BHARADHWAJ;
    // LDS budget of the rollout tiles (one wave per block): 64 * stride * 4 B <= 160 KiB
    {
        size_t lds;
        const int cols = (int)(cfg->optimizer == CTK_OPT_MPPI ? PC : HC);
        if (generic && cfg->predictor != CTK_PRED_ODE)
            lds = descends ? ctk_g_rpgd_descent_net_lds(h->env, h->net, (int)N, (int)H) : ctk_g_rollout_net_lds(h->env, h->net, (int)N, cols, (int)H, h->C);
        else if (generic) lds = descends ? ctk_g_rpgd_descent_lds(h->env, (int)H, nullptr) : ctk_g_rollout_lds(cols, (int)H, h->C);
        else lds = cfg->optimizer == CTK_OPT_MPPI ? ctk_mppi_rollout_lds((int)P, (int)H, cfg->predictor, (int)N)
                 : descends ? ctk_rpgd_descent_lds(cfg->predictor, (int)H, nullptr) : ctk_affine_rollout_lds((int)H, cfg->predictor);
        if (lds > 160 * 1024) { h->err = "horizon too long for the LDS sample tiles (160 KiB per CU)"; return bail(CTK_ERR_UNSUPPORTED); }
    }

    std::vector<InterpEntry> tab = build_interp_table((int)H, interp ? cfg->period_interpolation_inducing_points : 1, (int)P);
    TRY_CREATE(dev_alloc(h, &h->d_interp, H));
    HIP_CREATE(hipMemcpyAsync(h->d_interp, tab.data(), H * sizeof(InterpEntry), hipMemcpyHostToDevice, h->stream));
    HIP_CREATE(hipStreamSynchronize(h->stream));   // tab goes out of scope below

    TRY_CREATE(dev_alloc(h, &h->d_J, N));
    TRY_CREATE(dev_alloc(h, &h->d_Q, N * HC));
    if (cfg->materialize_trajectories) TRY_CREATE(dev_alloc(h, &h->d_traj, N * (H + 1) * h->S));
    const size_t nblk = (size_t)mppi_block_parts(h);
    h->parts_cap = nblk * (2 + PC);
    TRY_CREATE(dev_alloc(h, &h->d_parts, h->parts_cap));
    TRY_CREATE(dev_alloc(h, &h->d_parts2, ((nblk + 31) / 32) * (2 + PC)));
    TRY_CREATE(dev_alloc(h, &h->d_parts3, ((nblk + 1023) / 1024) * (2 + PC)));
    TRY_CREATE(dev_alloc(h, &h->d_rec, 2 + PC));
    TRY_CREATE(dev_alloc(h, &h->d_counter, 1));
    if (ctk_ll_records_ok((int)nblk, (int)PC) && !std::getenv("CTK_NO_LL"))
        TRY_CREATE(dev_alloc(h, &h->d_ll, nblk * (2 + PC)));
    if (cfg->optimizer == CTK_OPT_CEM && ctk_cem_fusable(cfg->predictor, (int)N, (int)HC) && !std::getenv("CTK_NO_CEM_FUSED"))
        TRY_CREATE(dev_alloc(h, &h->d_cem_ll, ctk_cem_fused_ll_words((int)N, (int)HC)));   // tuned and template path alike
    TRY_CREATE(dev_alloc(h, &h->d_unom[0], HC));
    TRY_CREATE(dev_alloc(h, &h->d_unom[1], HC));
    TRY_CREATE(dev_alloc(h, &h->d_std, HC));
    TRY_CREATE(dev_alloc(h, &h->d_base, HC));
    TRY_CREATE(dev_alloc(h, &h->d_scale, HC));
    TRY_CREATE(dev_alloc(h, &h->d_idx, N));
    TRY_CREATE(dev_alloc(h, &h->d_u, CTK_MAX_INPUTS));
    TRY_CREATE(dev_alloc(h, &h->d_weights, std::max<size_t>(1, weight_count(cfg->predictor, h->S, h->C, h->hid))));
    TRY_CREATE(dev_alloc(h, &h->d_wperm, generic ? ctk_g_net_table_floats(wide_net ? NET_MLP64 : cfg->predictor == CTK_PRED_GRU ? CTK_PRED_GRU : CTK_PRED_MLP) + GRU_HIDDEN_FLOATS
                                         : cfg->predictor == CTK_PRED_GRU ? (size_t)GRU_TABLE_FLOATS + GRU_HIDDEN_FLOATS
                                                                          : (size_t)64 * (MLP_FWD_PER_LANE + MLP_BWD_PER_LANE)));
    HIP_CREATE(hipHostMalloc((void**)&h->h_u, 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h->h_u, 0, 64);
    HIP_CREATE(hipHostGetDevicePointer((void**)&h->h_u_dev, h->h_u, 0));
    h->rp_pers.err_word = reinterpret_cast<uint32_t*>(h->h_u_dev) + 2;

    if (descends) {
        for (int b = 0; b < 2; ++b) {
            TRY_CREATE(dev_alloc(h, &h->d_pop[b], N * HC));
            TRY_CREATE(dev_alloc(h, &h->d_m[b], N * HC));
            TRY_CREATE(dev_alloc(h, &h->d_v[b], N * HC));
            TRY_CREATE(dev_alloc(h, &h->d_ages[b], N));
        }
        // bias corrections 1 - beta^t in double, rounded to fp32 (optimizer_rpgd.py:73-74); beyond the
        // table both are exactly 1.0f in fp32
        h->bc_len = 32768;
        std::vector<float> bc((size_t)2 * h->bc_len);
        for (int tt = 1; tt <= h->bc_len; ++tt) {
            bc[2 * (tt - 1)] = (float)(1.0 - std::pow((double)cfg->adam_beta_1, (double)tt));
            bc[2 * (tt - 1) + 1] = (float)(1.0 - std::pow((double)cfg->adam_beta_2, (double)tt));
        }
        if (cfg->materialize_trajectories) {   // the logging rollout of the descended plans (rpgd_materialize): u = 0 + plan * 1
            TRY_CREATE(dev_alloc(h, &h->d_Jlog, N));
            std::vector<float> zo(2 * HC, 0.0f);
            for (size_t i = 0; i < HC; ++i) zo[HC + i] = 1.0f;
            HIP_CREATE(hipMemcpyAsync(h->d_base, zo.data(), HC * sizeof(float), hipMemcpyHostToDevice, h->stream));
            HIP_CREATE(hipMemcpyAsync(h->d_scale, zo.data() + HC, HC * sizeof(float), hipMemcpyHostToDevice, h->stream));
            HIP_CREATE(hipStreamSynchronize(h->stream));
        }
        TRY_CREATE(dev_alloc(h, &h->d_bc, bc.size()));
        HIP_CREATE(hipMemcpyAsync(h->d_bc, bc.data(), bc.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_CREATE(hipStreamSynchronize(h->stream));
        TRY_CREATE(dev_alloc(h, &h->d_scratch, !generic ? ctk_rpgd_scratch_floats(cfg->predictor, (int)N, (int)H)
                                               : cfg->predictor != CTK_PRED_ODE ? ctk_g_rpgd_scratch_floats_net(h->net, (int)N, (int)H)
                                                                                : ctk_g_rpgd_scratch_floats(h->env, (int)N, (int)H)));
    }
    const bool mat = cfg->materialize_trajectories != 0;
    const bool gnet = generic && cfg->predictor != CTK_PRED_ODE;
    if (descends) h->dominant = gnet ? ctk_g_rpgd_descent_net_name(h->env, h->net, (int)N, (int)H) : generic ? ctk_g_rpgd_descent_name(h->env) : ctk_rpgd_descent_name(cfg->predictor, (int)N, (int)H);
    else {
        const int mode = cfg->optimizer == CTK_OPT_MPPI ? CTK_G_MODE_MPPI : CTK_G_MODE_AFFINE;
        h->dominant = gnet ? ctk_g_rollout_net_name(h->env, h->net, mode, mat, (int)N, (int)P, (int)H) : generic ? ctk_g_rollout_name(h->env, mode, mat)
                    : cfg->optimizer == CTK_OPT_MPPI ? ctk_mppi_rollout_name(cfg->predictor, mat, cfg->num_rollouts, cfg->period_interpolation_inducing_points == 1 && h->P == (int)H, false)
                                                     : ctk_affine_rollout_name(cfg->predictor, mat);
    }
    if (h->d_cem_ll && h->variant == CTK_OPT_CEM && cfg->cem_best_k <= (int)N) h->dominant = ctk_cem_fused_name(h->env, mat);
    if (cfg->optimizer == CTK_OPT_MPPI && mppi_env_kernel(h)) h->dominant = ctk_mppi_rollout_env_name(h->env, mat);
    if (!descends && cfg->optimizer != CTK_OPT_MPPI && affine_env_kernel(h) && !(h->d_cem_ll && h->variant == CTK_OPT_CEM)) h->dominant = ctk_affine_rollout_env_name(h->env, mat);
    if (cfg->optimizer != CTK_OPT_RPGD) TRY_CREATE(ctk_reset(h, nullptr, CTK_LOC_NONE));
    HIP_CREATE(hipStreamSynchronize(h->stream));
    *out = h;
    return CTK_OK;
#undef TRY_CREATE
#undef HIP_CREATE
}

void ctk_destroy(ctk_handle* h) {
    if (!h) return;
    hipSetDevice(h->cfg.device);
    (void)resident_quiesce(h);
    if (h->res_box) { if (h->res_local) hipFree(h->res_box); else hipHostFree(h->res_box); }
    if (h->res_stat) hipHostFree(h->res_stat);
    if (h->d_res_relay) hipFree(h->d_res_relay);
    if (h->d_res_args) hipFree(h->d_res_args);
    hipStreamSynchronize(h->stream);   // also correct for the null (default) stream handed in by ctk_set_stream
    for (auto& e : h->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    void* bufs[] = {h->d_interp, h->d_samples, h->d_J, h->d_Q, h->d_traj, h->d_parts, h->d_parts2, h->d_parts3, h->d_unom[0], h->d_unom[1],
                    h->d_std, h->d_base, h->d_scale, h->d_idx, h->d_u, h->d_weights, h->d_wperm, h->d_counter, h->d_ll, h->d_cem_ll, h->d_rec,
                    h->d_pop[0], h->d_pop[1], h->d_m[0], h->d_m[1], h->d_v[0], h->d_v[1], h->d_ages[0], h->d_ages[1], h->d_bc, h->d_scratch, h->d_Jlog};
    for (void* b : bufs) if (b) hipFree(b);
    if (h->d_shard_idx) hipFree(h->d_shard_idx);
    for (float* p : h->d_log) if (p) hipFree(p);
    ctk_p2p_close(h);
    if (h->h_u) hipHostFree(h->h_u);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int ctk_set_stream(ctk_handle* h, void* hip_stream) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t ns = (hipStream_t)hip_stream;
    if (ns == h->stream) return CTK_OK;
    // Order the handle's earlier work before whatever is issued on the new stream WITHOUT stopping the host (a caller that
    // alternates streams paid a full hipStreamSynchronize per step): an event on the old stream, a wait on the new one.  Only the
    // handle's own stream is synchronised, because it is destroyed right after.
    if (h->own_stream && h->stream) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipStreamDestroy(h->stream));
    } else {
        hipEvent_t ev = nullptr;
        HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, h->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ns, ev, 0);
        (void)hipEventDestroy(ev);                            // (released when the recorded work has completed)
        HIP_TRY(h, e);
    }
    h->stream = ns;
    h->own_stream = false;
    return CTK_OK;
}

void* ctk_get_stream(const ctk_handle* h) { return h ? (void*)h->stream : nullptr; }

int ctk_reset(ctk_handle* h, const float* draws, int draws_loc) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    float mid[CTK_MAX_INPUTS], span[CTK_MAX_INPUTS];
    for (int c = 0; c < h->C; ++c) { mid[c] = 0.5f * (h->cfg.action_low[c] + h->cfg.action_high[c]); span[c] = h->cfg.action_high[c] - h->cfg.action_low[c]; }
    h->count = 0;
    h->mppi_pending = false;
    h->shard_pending = false; h->shard_it = 0; h->shard_last_cands = nullptr;
    switch (h->cfg.optimizer) {
        case CTK_OPT_MPPI:   // optimizer_mppi.py:227-231 (self.u is NOT reset there)
            h->cur = 0;
            return fill_rows(h, h->d_unom[0], mid, h->C, h->HC);
        case CTK_OPT_CEM: {  // optimizer_cem_tf.py:113-117 (self.u = 0.0)
            if (int rc = fill_rows(h, h->d_unom[0], mid, h->C, h->HC)) return rc;
            if (int rc = fill_const(h, h->d_std, h->cfg.cem_initial_action_stdev, h->HC)) return rc;
            // only optimizer_cem_tf.py:117 resets self.u; the gradient variants' resets leave it (and Adam) alone
            return h->variant == CTK_OPT_CEM ? fill_const(h, h->d_u, 0.0f, h->C) : CTK_OK;
        }
        case CTK_OPT_RANDOM_ACTION:   // :78-86 draws and discards a sample
            if (int rc = fill_rows(h, h->d_base, h->cfg.action_low, h->C, h->HC)) return rc;
            return fill_rows(h, h->d_scale, span, h->C, h->HC);
        case CTK_OPT_RPGD:
            return rpgd_reset(h, draws, draws_loc);
    }
    return CTK_OK;
}

int ctk_set_param(ctk_handle* h, int id, float value) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    if (id < 0 || id >= env_info(h->env)->n_params) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_set_param: unknown parameter id for this environment");
    if (h->mppi_pending) return fail(h, CTK_ERR_STATE, "ctk_set_param: illegal between step_begin and step_end");
    h->params[id] = value;
    refresh_constants(h);
    return CTK_OK;
}

int ctk_get_param(const ctk_handle* h, int id, float* value) {
    if (!h || !value || id < 0 || id >= env_info(h->env)->n_params) return CTK_ERR_INVALID_ARGUMENT;
    *value = h->params[id];
    return CTK_OK;
}

int ctk_env_info(int environment, int* num_states, int* num_control_inputs, int* n_params) {
    const EnvInfo* e = env_info(environment);
    if (!e) return CTK_ERR_INVALID_ARGUMENT;
    if (num_states) *num_states = e->S;
    if (num_control_inputs) *num_control_inputs = e->C;
    if (n_params) *n_params = e->n_params;
    return CTK_OK;
}

const char* ctk_param_name(int environment, int id) {
    const EnvInfo* e = env_info(environment);
    return (e && id >= 0 && id < e->n_params) ? e->param_names[id] : nullptr;
}

int ctk_param_default(int environment, int id, float* value) {
    const EnvInfo* e = env_info(environment);
    if (!e || !value || id < 0 || id >= e->n_params) return CTK_ERR_INVALID_ARGUMENT;
    *value = e->param_defaults[id];
    return CTK_OK;
}

const char* ctk_environment_name(int environment) {
    const EnvInfo* e = env_info(environment);
    return e ? e->name : nullptr;
}

size_t ctk_predictor_weight_count(const ctk_handle* h) { return h ? weight_count(h->cfg.predictor, h->S, h->C, h->hid) : 0; }

int ctk_set_predictor_weights(ctk_handle* h, const float* w, size_t n) {
    RES_Q(h);
    if (!h || !w) return CTK_ERR_INVALID_ARGUMENT;
    if (h->cfg.predictor == CTK_PRED_ODE) return fail(h, CTK_ERR_STATE, "ctk_set_predictor_weights: the ODE predictor has no weights");
    const bool gru = h->cfg.predictor == CTK_PRED_GRU;
    if (n != weight_count(h->cfg.predictor, h->S, h->C, h->hid))
        return fail(h, CTK_ERR_INVALID_ARGUMENT, std::string("ctk_set_predictor_weights: expected ") +
                    std::to_string(weight_count(h->cfg.predictor, h->S, h->C, h->hid)) + " floats for this predictor and environment (ctk_predictor_weight_count)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(h->d_weights, w, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (gru) HIP_TRY(h, hipMemsetAsync(gru_hidden(h), 0, GRU_HIDDEN_FLOATS * sizeof(float), h->stream));
    const std::vector<float> perm = h->net == NET_MLP64 ? permute_mlp_weights_wide(w, h->S, h->C)
                                  : !gru ? permute_mlp_weights(w, h->S, h->C)
                                  : h->generic ? permute_gru_weights_g(w, h->S, h->C) : permute_gru_weights(w);
    if (gru && !h->generic) {   // |h2| <= 1 (convex mix of tanh values and the previous state, which starts at 0 or at what the caller set)
        const float* Wo = w + GRU_NW_RAW - 4 * 32 - 4; const float* bo = Wo + 4 * 32;
        float bound = 0.0f;
        for (int g = 0; g < 4; ++g) { float r = std::fabs(bo[g]); for (int j = 0; j < 32; ++j) r += std::fabs(Wo[g * 32 + j]); bound = std::max(bound, r); }
        h->net_out_bound = bound;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_wperm, perm.data(), perm.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->have_weights = true;
    return CTK_OK;
}

// Networks of other widths (the reference names a network by its sizes, config_controllers.yml:8 `GRU-6IN-32H1-32H2-5OUT-0`): the
// matrix-core kernels hold 32 units per hidden layer.  A NARROWER layer is embedded exactly — the missing units get zero weights and
// biases: tanh(0) = 0 (MLP) puts exact zeros into every sum it enters; a GRU unit with zero weights has r = z = 1/2, n = tanh(0) = 0 and
// h' = (1 - z) n + z h stays at the 0 it starts from.  A WIDER layer is refused with the sizes in the message.
static size_t shaped_weight_count(int predictor, int S, int C, int h1, int h2) {
    const size_t I = (size_t)S + C, a = (size_t)h1, b = (size_t)h2;
    if (predictor == CTK_PRED_MLP) return I * a + a + a * b + b + b * (size_t)S + S;
    if (predictor == CTK_PRED_GRU) return (3 * a * I + 3 * a * a + 6 * a) + (3 * b * a + 3 * b * b + 6 * b) + (b * (size_t)S + S);
    return 0;
}
size_t ctk_predictor_weight_count_shaped(const ctk_handle* h, int h1, int h2) {
    return (h && h1 >= 1 && h2 >= 1) ? shaped_weight_count(h->cfg.predictor, h->S, h->C, h1, h2) : 0;
}

int ctk_set_predictor_weights_shaped(ctk_handle* h, const float* w, size_t n, int h1, int h2) {
    if (!h || !w) return CTK_ERR_INVALID_ARGUMENT;
    if (h->cfg.predictor == CTK_PRED_ODE) return fail(h, CTK_ERR_STATE, "ctk_set_predictor_weights_shaped: the ODE predictor has no weights");
    if (h1 < 1 || h2 < 1) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_set_predictor_weights_shaped: hidden widths must be >= 1");
    const int S = h->S, I = h->S + h->C, W = h->hid;
    if (h1 > W || h2 > W)
        return fail(h, CTK_ERR_UNSUPPORTED, std::string("network ") + std::to_string(I) + "IN-" + std::to_string(h1) + "H1-" + std::to_string(h2) + "H2-" +
                    std::to_string(S) + "OUT: this handle's predictor kernels hold " + std::to_string(W) + " units per hidden layer (narrower layers are "
                    "embedded exactly; create the handle with cfg.predictor_hidden1/2 = the widths: MLPs up to 64 / 64 are built, GRUs up to 32 / 32)");
    if (n != shaped_weight_count(h->cfg.predictor, h->S, h->C, h1, h2))
        return fail(h, CTK_ERR_INVALID_ARGUMENT, std::string("ctk_set_predictor_weights_shaped: expected ") +
                    std::to_string(shaped_weight_count(h->cfg.predictor, h->S, h->C, h1, h2)) + " floats for " + std::to_string(I) + "IN-" +
                    std::to_string(h1) + "H1-" + std::to_string(h2) + "H2-" + std::to_string(S) + "OUT, got " + std::to_string(n));
    std::vector<float> full(weight_count(h->cfg.predictor, h->S, h->C, W), 0.0f);
    const float* p = w;
    float* q = full.data();
    auto rows = [&](int r_src, int c_src, int r_dst, int c_dst) {      // a [r_src, c_src] matrix into the top-left of a [r_dst, c_dst] one
        for (int r = 0; r < r_src; ++r) std::memcpy(q + (size_t)r * c_dst, p + (size_t)r * c_src, (size_t)c_src * sizeof(float));
        p += (size_t)r_src * c_src; q += (size_t)r_dst * c_dst;
    };
    if (h->cfg.predictor == CTK_PRED_MLP) {
        rows(h1, I, W, I); rows(1, h1, 1, W);          // W1, b1
        rows(h2, h1, W, W); rows(1, h2, 1, W);         // W2, b2
        rows(S, h2, S, W); rows(1, S, 1, S);           // W3, b3
    } else {
        auto gates = [&](int hs, int c_src, int c_dst) {               // [3 hs, c_src] (rows r|z|n) -> [96, c_dst]
            for (int gte = 0; gte < 3; ++gte) {
                for (int r = 0; r < hs; ++r) std::memcpy(q + ((size_t)gte * 32 + r) * c_dst, p + ((size_t)gte * hs + r) * c_src, (size_t)c_src * sizeof(float));
            }
            p += (size_t)3 * hs * c_src; q += (size_t)96 * c_dst;
        };
        gates(h1, I, I); gates(h1, h1, 32); gates(h1, 1, 1); gates(h1, 1, 1);          // layer 1: W_i, W_h, b_i, b_h
        gates(h2, h1, 32); gates(h2, h2, 32); gates(h2, 1, 1); gates(h2, 1, 1);        // layer 2
        rows(S, h2, S, 32); rows(1, S, 1, S);                                           // W_o, b_o
    }
    return ctk_set_predictor_weights(h, full.data(), full.size());
}

size_t ctk_predictor_hidden_size(const ctk_handle* h) { return (h && h->cfg.predictor == CTK_PRED_GRU) ? (size_t)GRU_HIDDEN_FLOATS : 0; }

int ctk_predictor_update(ctk_handle* h, const float* s, const float* u) {
    RES_Q(h);
    if (!h || !s) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_predictor_update: NULL state") : CTK_ERR_INVALID_ARGUMENT;
    if (h->cfg.predictor != CTK_PRED_GRU) return CTK_OK;   // predictor.update is a no-op for stateless predictors
    if (int rc = check_predictor(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return gru_advance(h, s, u);
}

int ctk_predictor_get_hidden(ctk_handle* h, float* dst, size_t cap) {
    RES_Q(h);
    if (!h || !dst) return CTK_ERR_INVALID_ARGUMENT;
    if (h->cfg.predictor != CTK_PRED_GRU) return fail(h, CTK_ERR_STATE, "ctk_predictor_get_hidden: predictor has no hidden state");
    if (cap < (size_t)GRU_HIDDEN_FLOATS) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_predictor_get_hidden: need room for 64 floats");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(dst, gru_hidden(h), GRU_HIDDEN_FLOATS * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTK_OK;
}

int ctk_predictor_set_hidden(ctk_handle* h, const float* src, size_t n) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    if (h->cfg.predictor != CTK_PRED_GRU) return fail(h, CTK_ERR_STATE, "ctk_predictor_set_hidden: predictor has no hidden state");
    if (src && n != (size_t)GRU_HIDDEN_FLOATS) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_predictor_set_hidden: expected 64 floats");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->hidden_scale = 1.0f;
    if (src) for (int i = 0; i < GRU_HIDDEN_FLOATS; ++i) h->hidden_scale = std::max(h->hidden_scale, std::fabs(src[i]));
    if (src) HIP_TRY(h, hipMemcpyAsync(gru_hidden(h), src, GRU_HIDDEN_FLOATS * sizeof(float), hipMemcpyHostToDevice, h->stream));
    else HIP_TRY(h, hipMemsetAsync(gru_hidden(h), 0, GRU_HIDDEN_FLOATS * sizeof(float), h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTK_OK;
}

size_t ctk_samples_needed(const ctk_handle* h) { return h ? samples_needed(h) : 0; }

int ctk_rng_get_position(const ctk_handle* h, uint32_t* call) {
    if (!h || !call) return CTK_ERR_INVALID_ARGUMENT;
    *call = h->call;
    return CTK_OK;
}

int ctk_rng_set_position(ctk_handle* h, uint32_t call) {
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    h->call = call;
    return CTK_OK;
}

int ctk_step(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int samples_loc, float* u_out) {
    if (!h || !s) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_step: NULL state") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (h->mppi_pending) return fail(h, CTK_ERR_STATE, "ctk_step: a sharded step is pending (call ctk_mppi_step_end)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    switch (h->cfg.optimizer) {
        case CTK_OPT_MPPI: {
            for (int i = 0; i < h->S; ++i) h->mppi_s[i] = s[i];
            if (resident_ok(h, samples_loc)) {   // opt-in: the step is served by the kernel that stays on the device
                const float* d_s = nullptr;
                if (int rc = resolve_samples(h, samples, samples_loc, (size_t)h->N * h->PC, &d_s)) return rc;
                return resident_step(h, s, u_prev, d_s, u_out);
            }
            if (int rc = resident_quiesce(h)) return rc;
            if (mppi_can_fuse(h)) {   // one launch: the last block to finish merges and updates
                if (int rc = mppi_rollout(h, s, u_prev, samples, samples_loc, 1, nullptr)) return rc;
                h->cur ^= 1;
                if (int rc = mppi_advance_hidden(h)) return rc;
                return finish_step(h, u_out);
            }
            if (int rc = mppi_rollout(h, s, u_prev, samples, samples_loc, 0, nullptr)) return rc;
            const float* parts; int n_parts;
            if (int rc = mppi_reduce_blocks(h, &parts, &n_parts)) return rc;
            return mppi_update(h, parts, n_parts, u_out);
        }
        case CTK_OPT_CEM: return cem_step(h, s, u_prev, samples, samples_loc, u_out);
        case CTK_OPT_RANDOM_ACTION: return random_step(h, s, u_prev, samples, samples_loc, u_out);
        case CTK_OPT_RPGD: return rpgd_step(h, s, u_prev, samples, samples_loc, u_out);
    }
    return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_step: unknown optimizer");
    });
}

size_t ctk_mppi_partial_size(const ctk_handle* h) { return h ? (size_t)(2 + h->PC) : 0; }

int ctk_mppi_step_begin(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int samples_loc,
                        float* partial_dev) {
    RES_Q(h);
    if (!h || !s || !partial_dev) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_mppi_step_begin: NULL argument") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (h->cfg.optimizer != CTK_OPT_MPPI) return fail(h, CTK_ERR_STATE, "ctk_mppi_step_begin: handle is not MPPI");
    if (h->mppi_pending) return fail(h, CTK_ERR_STATE, "ctk_mppi_step_begin: previous sharded step not ended");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int i = 0; i < h->S; ++i) h->mppi_s[i] = s[i];
    if (mppi_can_fuse(h)) {
        if (int rc = mppi_rollout(h, s, u_prev, samples, samples_loc, 2, partial_dev)) return rc;
    } else {
        if (int rc = mppi_rollout(h, s, u_prev, samples, samples_loc, 0, nullptr)) return rc;
        const float* parts; int n_parts;
        if (int rc = mppi_reduce_blocks(h, &parts, &n_parts)) return rc;
        HIP_TRY(h, ctk_launch_mppi_merge_partial(h->stream, parts, n_parts, n_parts, h->PC, h->mk.neg_inv_lbd, partial_dev));
    }
    h->mppi_pending = true;
    return CTK_OK;
    });
}

int ctk_mppi_step_end(ctk_handle* h, const float* parts_dev, int n_parts, float* u_out) {
    RES_Q(h);
    if (!h || !parts_dev || n_parts < 1) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_mppi_step_end: bad argument") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (!h->mppi_pending) return fail(h, CTK_ERR_STATE, "ctk_mppi_step_end: no sharded step pending");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->mppi_pending = false;
    return mppi_update(h, parts_dev, n_parts, u_out);
    });
}

// ---- sharded CEM / random-action ----------------------------------------------------------------
static int shard_k(const ctk_handle* h) { return h->cfg.optimizer == CTK_OPT_CEM ? h->cfg.cem_best_k : 1; }

size_t ctk_shard_candidates_size(const ctk_handle* h) {
    if (!h || (h->cfg.optimizer != CTK_OPT_CEM && h->cfg.optimizer != CTK_OPT_RANDOM_ACTION)) return 0;
    return (size_t)shard_k(h) * (2 + h->HC);
}

int ctk_shard_iterations(const ctk_handle* h) {
    if (!h) return 0;
    return h->cfg.optimizer == CTK_OPT_CEM ? cem_iterations(h) : (h->cfg.optimizer == CTK_OPT_RANDOM_ACTION ? 1 : 0);
}

int ctk_shard_iter_begin(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int samples_loc, float* cand_dev) {
    RES_Q(h);
    if (!h || !s || !cand_dev) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_shard_iter_begin: NULL argument") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    const bool cem = h->cfg.optimizer == CTK_OPT_CEM;
    if (!cem && h->cfg.optimizer != CTK_OPT_RANDOM_ACTION) return fail(h, CTK_ERR_STATE, "ctk_shard_iter_begin: CEM / random-action handles only");
    if (h->variant != h->cfg.optimizer) return fail(h, CTK_ERR_UNSUPPORTED, "ctk_shard_iter_begin: not built for this optimizer variant");
    if (h->shard_pending) return fail(h, CTK_ERR_STATE, "ctk_shard_iter_begin: previous iteration not ended");
    if (int rc = check_predictor(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const float* d_s = nullptr;
    if (int rc = resolve_samples(h, samples, samples_loc, (size_t)h->N * h->HC, &d_s)) return rc;   // ONE iteration's draws [N,H,C]
    RolloutArgs a = make_args(h, s, u_prev, h->N, h->H);
    a.stream_id = (uint32_t)h->shard_it;
    if (int rc = launch_affine(h, a, d_s, cem ? 0 : 1, cem ? h->d_unom[0] : h->d_base, cem ? h->d_std : h->d_scale,
                               h->cfg.materialize_trajectories != 0)) return rc;
    const int K = shard_k(h);
    HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, K, h->d_idx));
    HIP_TRY(h, ctk_launch_pack_candidates(h->stream, h->d_J, h->d_Q, h->d_idx, K, h->HC, h->cfg.global_rollout_offset, cand_dev));
    h->shard_pending = true;
    return CTK_OK;
    });
}

int ctk_shard_iter_end(ctk_handle* h, const float* cands_all_dev, int n_ranks) {
    RES_Q(h);
    if (!h || !cands_all_dev || n_ranks < 1) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_shard_iter_end: bad argument") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (!h->shard_pending) return fail(h, CTK_ERR_STATE, "ctk_shard_iter_end: no iteration pending");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int K = shard_k(h), rs = 2 + h->HC, M = n_ranks * K;
    if ((size_t)M > h->shard_idx_cap) {
        if (h->d_shard_idx) HIP_TRY(h, hipFree(h->d_shard_idx));
        h->d_shard_idx = nullptr; h->shard_idx_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->d_shard_idx, (size_t)M * sizeof(int)));
        h->shard_idx_cap = (size_t)M;
    }
    // global best K of the union of the per-shard best-K lists: positions are ordered like global indices
    // among equal costs (rank-major, each list sorted), so the positional tie-break is the global one
    HIP_TRY(h, ctk_launch_select_topk(h->stream, cands_all_dev, M, K, h->d_shard_idx, rs));
    if (h->cfg.optimizer == CTK_OPT_CEM)
        HIP_TRY(h, ctk_launch_cem_refit(h->stream, cands_all_dev + 2, h->d_shard_idx, K, h->HC, h->d_unom[0], h->d_std, rs));
    h->shard_last_cands = cands_all_dev;
    h->shard_pending = false;
    ++h->shard_it;
    return CTK_OK;
    });
}

int ctk_shard_finish(ctk_handle* h, float* u_out) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (h->shard_pending || h->shard_it == 0 || !h->shard_last_cands) return fail(h, CTK_ERR_STATE, "ctk_shard_finish: no completed iteration");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int rs = 2 + h->HC;
    if (h->cfg.optimizer == CTK_OPT_CEM) {
        if (int rc = launch_cem_finish(h, h->shard_last_cands + 2, h->d_shard_idx, rs, 1.0e8f, 0)) return rc;
        ++h->count;
    } else {
        if (int rc = launch_pick_best(h, h->shard_last_cands + 2, h->d_shard_idx, rs)) return rc;
    }
    h->shard_it = 0;
    h->shard_last_cands = nullptr;
    return finish_step(h, u_out);
    });
}

// ---- sharded RPGD --------------------------------------------------------------------------------
static int rpgd_local_keep(const ctk_handle* h) { return h->cfg.opt_keep_k < h->N ? h->cfg.opt_keep_k : h->N; }

size_t ctk_rpgd_keepers_size(const ctk_handle* h) {
    if (!h || h->cfg.optimizer != CTK_OPT_RPGD) return 0;
    return (size_t)rpgd_local_keep(h) * (3 + 3 * h->HC);
}

// fresh rows this shard must draw at the next step_end (0 on non-resampling steps), given the world size
size_t ctk_rpgd_fresh_rows(const ctk_handle* h, int n_ranks) {
    if (!h || h->cfg.optimizer != CTK_OPT_RPGD || n_ranks < 1) return 0;
    if (h->count % h->cfg.resamp_per != 0) return 0;
    const long Ng = (long)n_ranks * h->N, first_keeper = Ng - h->cfg.opt_keep_k, off = h->cfg.global_rollout_offset;
    long n = first_keeper - off;
    if (n < 0) n = 0;
    if (n > h->N) n = h->N;
    return (size_t)n;
}

int ctk_rpgd_step_begin(ctk_handle* h, const float* s, const float* u_prev, float* keep_dev) {
    RES_Q(h);
    if (!h || !s || !keep_dev) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_rpgd_step_begin: NULL argument") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (h->cfg.optimizer != CTK_OPT_RPGD || h->variant != CTK_OPT_RPGD) return fail(h, CTK_ERR_STATE, "ctk_rpgd_step_begin: handle is not RPGD");
    if (!h->rpgd_ready) return fail(h, CTK_ERR_STATE, "RPGD: call ctk_reset (optimizer_reset) before the first step");
    if (h->shard_pending) return fail(h, CTK_ERR_STATE, "ctk_rpgd_step_begin: previous sharded step not ended");
    if (int rc = check_predictor(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int i = 0; i < h->S; ++i) h->shard_s[i] = s[i];
    h->shard_has_uprev = u_prev != nullptr;
    for (int c = 0; c < h->C; ++c) h->shard_uprev[c] = u_prev ? u_prev[c] : 0.0f;
    if (int rc = rpgd_descent(h, s, u_prev)) return rc;
    const int kl = rpgd_local_keep(h), cur = h->rcur;
    HIP_TRY(h, ctk_launch_select_topk(h->stream, h->d_J, h->N, kl, h->d_idx));
    HIP_TRY(h, ctk_launch_rpgd_pack_keepers(h->stream, h->d_J, h->d_pop[cur], h->d_m[cur], h->d_v[cur], h->d_ages[cur], h->d_idx, kl,
                                            h->HC, h->cfg.global_rollout_offset, keep_dev));
    h->shard_pending = true;
    return CTK_OK;
    });
}

int ctk_rpgd_step_end(ctk_handle* h, const float* keep_all_dev, int n_ranks, const float* draws, int draws_loc, float* u_out) {
    RES_Q(h);
    if (!h || !keep_all_dev || n_ranks < 1) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_rpgd_step_end: bad argument") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (h->cfg.optimizer != CTK_OPT_RPGD || !h->shard_pending) return fail(h, CTK_ERR_STATE, "ctk_rpgd_step_end: no sharded RPGD step pending");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const ctk_config& c = h->cfg;
    const int kl = rpgd_local_keep(h), M = n_ranks * kl, rs = 3 + 3 * h->HC;
    if (c.opt_keep_k > M) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_rpgd_step_end: opt_keep_k exceeds the gathered candidates");
    if ((size_t)M > h->shard_idx_cap) {
        if (h->d_shard_idx) HIP_TRY(h, hipFree(h->d_shard_idx));
        h->d_shard_idx = nullptr; h->shard_idx_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->d_shard_idx, (size_t)M * sizeof(int)));
        h->shard_idx_cap = (size_t)M;
    }
    // global keep-k (sorted) out of the union of the shards' sorted best lists (positional tie-break == global index)
    HIP_TRY(h, ctk_launch_select_topk(h->stream, keep_all_dev, M, c.opt_keep_k, h->d_shard_idx, rs));
    const bool resample = (h->count % c.resamp_per) == 0;
    const int n_fresh = (int)ctk_rpgd_fresh_rows(h, n_ranks);
    const long first_keeper = (long)n_ranks * h->N - c.opt_keep_k;
    const int keeper_base = (int)(c.global_rollout_offset > first_keeper ? c.global_rollout_offset - first_keeper : 0);
    const float* d_draws = nullptr;
    if (resample && n_fresh > 0)
        if (int rc = resolve_samples(h, draws, draws_loc, (size_t)n_fresh * h->PC, &d_draws)) return rc;
    const float* up = h->shard_has_uprev ? h->shard_uprev : nullptr;
    RolloutArgs aw = make_args(h, h->shard_s, up, h->N, h->P);
    const int cur = h->rcur, nxt = cur ^ 1;
    if (int rc = rpgd_warm(h, aw, resample ? n_fresh : 0, resample ? 1 : 0, 0, d_draws, cur, nxt, h->d_shard_idx, keep_all_dev, keeper_base)) return rc;
    h->rcur = nxt;
    ++h->count;
    h->shard_pending = false;
    return finish_step(h, u_out);
    });
}

int ctk_rollout(ctk_handle* h, const float* s, const float* u_prev, const float* Q, int n, float* traj_out, float* J_out) {
    RES_Q(h);
    if (!h || !s || !Q) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_rollout: NULL argument") : CTK_ERR_INVALID_ARGUMENT;
    if (n < 1 || n > h->N) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_rollout: need 1 <= n <= num_rollouts");
    if (int rc = check_predictor(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t H = h->HC, Hs = h->H;   // H: floats per plan row [H,C]; Hs: horizon steps
    const float* d_s = nullptr;
    if (int rc = resolve_samples(h, Q, CTK_LOC_HOST, (size_t)n * H, &d_s)) return rc;
    float* d_traj = h->d_traj;
    float* tmp_traj = nullptr;
    if (traj_out && !d_traj) { HIP_TRY(h, hipMalloc((void**)&tmp_traj, (size_t)n * (Hs + 1) * h->S * sizeof(float))); d_traj = tmp_traj; }
    // base 0, scale 1, no clipping: the plans are taken as given
    float *d_zero = nullptr, *d_one = nullptr;
    HIP_TRY(h, hipMalloc((void**)&d_zero, 2 * H * sizeof(float)));
    d_one = d_zero + H;
    std::vector<float> zo(2 * H, 0.0f);
    for (size_t i = 0; i < H; ++i) zo[H + i] = 1.0f;
    HIP_TRY(h, hipMemcpyAsync(d_zero, zo.data(), 2 * H * sizeof(float), hipMemcpyHostToDevice, h->stream));
    RolloutArgs a = make_args(h, s, u_prev, n, h->H);
    for (int c = 0; c < h->C; ++c) { a.lo[c] = -INFINITY; a.hi[c] = INFINITY; }
    a.traj_out = traj_out ? d_traj : nullptr;
    hipError_t e = (h->generic && h->cfg.predictor != CTK_PRED_ODE)
        ? ctk_launch_g_rollout_net(h->stream, h->env, h->net, CTK_G_MODE_AFFINE, a, h->params, h->cfg.dt, h->cfg.intermediate_steps, h->mk,
                                   d_s, d_zero, d_one, 0, h->d_wperm, nullptr, traj_out != nullptr)
        : h->generic
        ? ctk_launch_g_rollout(h->stream, h->env, CTK_G_MODE_AFFINE, a, h->params, h->cfg.dt, h->cfg.intermediate_steps, h->mk, d_s, d_zero, d_one,
                               0, nullptr, traj_out != nullptr)
        : ctk_launch_affine_rollout(h->stream, h->cfg.predictor, a, h->k, d_s, 0, d_zero, d_one, h->d_wperm, traj_out != nullptr);
    if (e == hipSuccess && traj_out)
        e = hipMemcpyAsync(traj_out, d_traj, (size_t)n * (Hs + 1) * h->S * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && J_out) e = hipMemcpyAsync(J_out, h->d_J, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    hipFree(d_zero);
    if (tmp_traj) hipFree(tmp_traj);
    HIP_TRY(h, e);
    HIP_TRY(h, e2);
    return CTK_OK;
}

int ctk_read(ctk_handle* h, int which, float* dst, size_t cap, size_t* n_out) {
    RES_Q(h);
    if (!h || !dst) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: NULL destination") : CTK_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const float* src = nullptr; size_t n = 0; bool is_int = false;
    if (int rc = locate_buffer(h, which, &src, &n, &is_int)) return rc;
    if (cap < n) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_read: destination too small");
    HIP_TRY(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (is_int) for (size_t i = 0; i < n; ++i) { int v; std::memcpy(&v, &dst[i], 4); dst[i] = (float)v; }
    if (n_out) *n_out = n;
    return CTK_OK;
}

size_t ctk_state_size(const ctk_handle* h) {
    if (!h) return 0;
    const size_t H = h->HC, C = h->C;   // [H,C] rows; the optimizer's last output u is C floats
    switch (h->cfg.optimizer) {
        case CTK_OPT_MPPI: return H + C;
        case CTK_OPT_CEM: return 2 * H + C + 1 + (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ ? 2 * (size_t)h->N * H + 1 : 0);
        case CTK_OPT_RANDOM_ACTION: return C;
        case CTK_OPT_RPGD: return 3 * (size_t)h->N * H + (size_t)h->N + C + 2;
    }
    return 0;
}

int ctk_get_state(ctk_handle* h, float* dst, size_t cap) {
    RES_Q(h);
    if (!h || !dst) return CTK_ERR_INVALID_ARGUMENT;
    const size_t n = ctk_state_size(h), H = h->HC, C = h->C;
    if (cap < n) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_get_state: destination too small");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    size_t o = 0;
    auto pull = [&](const float* src, size_t cnt) -> hipError_t {
        hipError_t e = hipMemcpyAsync(dst + o, src, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        o += cnt; return e;
    };
    switch (h->cfg.optimizer) {
        case CTK_OPT_MPPI: HIP_TRY(h, pull(h->d_unom[h->cur], H)); HIP_TRY(h, pull(h->d_u, C)); break;
        case CTK_OPT_CEM: HIP_TRY(h, pull(h->d_unom[0], H)); HIP_TRY(h, pull(h->d_std, H)); HIP_TRY(h, pull(h->d_u, C));
            HIP_TRY(h, hipStreamSynchronize(h->stream)); dst[o++] = (float)h->count;
            if (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ) {   // the Keras optimizer's persistent moments and step count
                HIP_TRY(h, pull(h->d_m[0], (size_t)h->N * H)); HIP_TRY(h, pull(h->d_v[0], (size_t)h->N * H));
                HIP_TRY(h, hipStreamSynchronize(h->stream)); dst[o++] = (float)h->adam_step;
            }
            break;
        case CTK_OPT_RANDOM_ACTION: HIP_TRY(h, pull(h->d_u, C)); break;
        case CTK_OPT_RPGD: {
            const size_t NH = (size_t)h->N * H;
            HIP_TRY(h, pull(h->d_pop[h->rcur], NH)); HIP_TRY(h, pull(h->d_m[h->rcur], NH)); HIP_TRY(h, pull(h->d_v[h->rcur], NH));
            HIP_TRY(h, pull(h->d_ages[h->rcur], h->N)); HIP_TRY(h, pull(h->d_u, C));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            dst[o++] = (float)h->adam_step; dst[o++] = (float)h->count;
            break;
        }
        default: return fail(h, CTK_ERR_UNSUPPORTED, "ctk_get_state: not built for this optimizer");
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTK_OK;
}

int ctk_set_state(ctk_handle* h, const float* src, size_t n) {
    RES_Q(h);
    if (!h || !src) return CTK_ERR_INVALID_ARGUMENT;
    if (n != ctk_state_size(h)) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_set_state: wrong state size");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t H = h->HC, C = h->C;
    size_t o = 0;
    auto push = [&](float* dstp, size_t cnt) -> hipError_t {
        hipError_t e = hipMemcpyAsync(dstp, src + o, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream);
        o += cnt; return e;
    };
    switch (h->cfg.optimizer) {
        case CTK_OPT_MPPI: HIP_TRY(h, push(h->d_unom[h->cur], H)); HIP_TRY(h, push(h->d_u, C)); break;
        case CTK_OPT_CEM: HIP_TRY(h, push(h->d_unom[0], H)); HIP_TRY(h, push(h->d_std, H)); HIP_TRY(h, push(h->d_u, C));
            h->count = (int)src[o++];
            if (h->variant == CTK_OPT_CEM_GRAD_BHARADHWAJ) {
                HIP_TRY(h, push(h->d_m[0], (size_t)h->N * H)); HIP_TRY(h, push(h->d_v[0], (size_t)h->N * H));
                h->adam_step = (int)src[o++];
            }
            break;
        case CTK_OPT_RANDOM_ACTION: HIP_TRY(h, push(h->d_u, C)); break;
        case CTK_OPT_RPGD: {
            const size_t NH = (size_t)h->N * H;
            HIP_TRY(h, push(h->d_pop[h->rcur], NH)); HIP_TRY(h, push(h->d_m[h->rcur], NH)); HIP_TRY(h, push(h->d_v[h->rcur], NH));
            HIP_TRY(h, push(h->d_ages[h->rcur], h->N)); HIP_TRY(h, push(h->d_u, C));
            h->adam_step = (int)src[o++]; h->count = (int)src[o++];
            h->rpgd_ready = true;
            break;
        }
        default: return fail(h, CTK_ERR_UNSUPPORTED, "ctk_set_state: not built for this optimizer");
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTK_OK;
}

int ctk_profile_enable(ctk_handle* h, int on) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (on && h->events.empty()) {
        h->events.resize(4096);
        for (auto& e : h->events) { HIP_TRY(h, hipEventCreate(&e.a)); HIP_TRY(h, hipEventCreate(&e.b)); }
    }
    h->prof = on != 0;
    h->prof_every = on > 1 ? on : 1;
    h->prof_tick = 0;
    h->ev_used = 0;
    return CTK_OK;
}

int ctk_profile_read(ctk_handle* h, float* ms_out, size_t cap, size_t* n_out) {
    RES_Q(h);
    if (!h || !ms_out) return CTK_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    size_t n = h->ev_used < cap ? h->ev_used : cap;
    for (size_t i = 0; i < n; ++i) HIP_TRY(h, hipEventElapsedTime(&ms_out[i], h->events[i].a, h->events[i].b));
    if (n_out) *n_out = n;
    h->ev_used = 0;
    return CTK_OK;
}

const char* ctk_dominant_kernel(const ctk_handle* h) { return h ? h->dominant.c_str() : ""; }

int ctk_p2p_close(ctk_handle* h) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    if (h->p2p_world == 0) return CTK_OK;
    hipSetDevice(h->cfg.device);
    hipStreamSynchronize(h->stream);
    for (int w = 0; w < h->p2p_world; ++w) {
        if (!h->p2p_bufs[w]) continue;
        if (w == h->p2p_rank) hipFree(h->p2p_bufs[w]); else hipIpcCloseMemHandle(h->p2p_bufs[w]);
        h->p2p_bufs[w] = nullptr;
    }
    if (h->d_p2p_args) { hipFree(h->d_p2p_args); h->d_p2p_args = nullptr; }
    h->p2p_world = 0; h->p2p_rank = -1; h->p2p_connected = false;
    return CTK_OK;
}

int ctk_p2p_alloc(ctk_handle* h, int rank, int world, void* handle_out) {
    RES_Q(h);
    if (!h || !handle_out) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_p2p_alloc: NULL argument") : CTK_ERR_INVALID_ARGUMENT;
    if (h->cfg.optimizer != CTK_OPT_MPPI) return fail(h, CTK_ERR_STATE, "ctk_p2p_alloc: handle is not MPPI");
    if (h->generic) return fail(h, CTK_ERR_UNSUPPORTED, "ctk_p2p_alloc: the peer-to-peer exchange is built into the CartPole kernels only (use the begin / all-gather / end path)");
    if (world < 1 || world > CTK_P2P_MAX_WORLD || rank < 0 || rank >= world)
        return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_p2p_alloc: need 0 <= rank < world <= 16");
    static_assert(sizeof(hipIpcMemHandle_t) == CTK_P2P_HANDLE_BYTES, "handle size");
    ctk_p2p_close(h);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t bytes = ctk_p2p_buffer_floats(world, h->P) * sizeof(float);
    float* buf = nullptr;
    HIP_TRY(h, hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached));
    HIP_TRY(h, hipMemset(buf, 0, bytes));
    HIP_TRY(h, hipDeviceSynchronize());
    hipIpcMemHandle_t hd;
    hipError_t e = hipIpcGetMemHandle(&hd, buf);
    if (e != hipSuccess) { hipFree(buf); return fail(h, CTK_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e)); }
    std::memcpy(handle_out, &hd, sizeof(hd));
    h->p2p_rank = rank; h->p2p_world = world; h->p2p_bufs[rank] = buf; h->p2p_seq = 1; h->p2p_connected = false;
    if (const char* t = std::getenv("CTK_P2P_TIMEOUT_S")) { const double v = std::atof(t); if (v > 0.0) h->p2p_timeout_s = v; }
    return CTK_OK;
}

int ctk_p2p_connect(ctk_handle* h, const void* handles) {
    RES_Q(h);
    if (!h || !handles) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_p2p_connect: NULL argument") : CTK_ERR_INVALID_ARGUMENT;
    if (h->p2p_world == 0) return fail(h, CTK_ERR_STATE, "ctk_p2p_connect: call ctk_p2p_alloc first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int w = 0; w < h->p2p_world; ++w) {
        if (w == h->p2p_rank) continue;
        hipIpcMemHandle_t hd;
        std::memcpy(&hd, (const char*)handles + (size_t)w * CTK_P2P_HANDLE_BYTES, sizeof(hd));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(h, CTK_ERR_HIP, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(w) + "): " + hipGetErrorString(e));
        h->p2p_bufs[w] = (float*)p;
    }
    {   // description of the exchange for the in-launch form (the rollout launch's block 0 does it)
        std::vector<char> host(ctk_p2p_args_bytes());
        ctk_p2p_fill_args(host.data(), h->p2p_bufs, h->p2p_rank, h->p2p_world, h->P, reinterpret_cast<uint32_t*>(h->h_u_dev) + 2, h->p2p_timeout_s);
        if (!h->d_p2p_args) HIP_TRY(h, hipMalloc(&h->d_p2p_args, host.size()));
        HIP_TRY(h, hipMemcpy(h->d_p2p_args, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    h->p2p_connected = true;
    return CTK_OK;
}

int ctk_p2p_step(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int samples_loc, float* u_out) {
    RES_Q(h);
    if (!h || !s) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_p2p_step: NULL state") : CTK_ERR_INVALID_ARGUMENT;
    return guarded(h, [&]() -> int {
    if (!h->p2p_connected) return fail(h, CTK_ERR_STATE, "ctk_p2p_step: call ctk_p2p_alloc and ctk_p2p_connect first");
    if (h->mppi_pending) return fail(h, CTK_ERR_STATE, "ctk_p2p_step: a begin/end sharded step is pending");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int i = 0; i < h->S; ++i) h->mppi_s[i] = s[i];
    const int W = h->p2p_world, rs = 2 + h->P, par = (int)(h->p2p_seq & 1u);
    float* my_slot = h->p2p_bufs[h->p2p_rank] + (size_t)(par * W + h->p2p_rank) * rs;
    uint32_t* err = reinterpret_cast<uint32_t*>(h->h_u_dev) + 2;
    const int nxt = h->cur ^ 1;
    static const bool two_launches = std::getenv("CTK_P2P_TWO_LAUNCHES") != nullptr;   // A/B switch
    if (mppi_can_fuse(h) && h->d_ll && !two_launches && ctk_p2p_can_fuse(h->P, W, mppi_block_parts(h))) {
        // ONE launch: rollout, local merge, exchange with the peers, global merge, update (fuse mode 3)
        if (int rc = mppi_rollout(h, s, u_prev, samples, samples_loc, 3, my_slot)) return rc;
    } else {
        if (mppi_can_fuse(h)) {
            if (int rc = mppi_rollout(h, s, u_prev, samples, samples_loc, 2, my_slot)) return rc;
        } else {
            if (int rc = mppi_rollout(h, s, u_prev, samples, samples_loc, 0, nullptr)) return rc;
            const float* parts; int n_parts;
            if (int rc = mppi_reduce_blocks(h, &parts, &n_parts)) return rc;
            HIP_TRY(h, ctk_launch_mppi_merge_partial(h->stream, parts, n_parts, n_parts, h->P, h->mk.neg_inv_lbd, my_slot));
        }
        HIP_TRY(h, ctk_launch_mppi_p2p_exchange(h->stream, h->p2p_bufs, h->p2p_rank, W, h->P, h->p2p_seq, err, h->p2p_timeout_s,
                                                h->mk.neg_inv_lbd, h->H, h->d_interp, h->d_unom[h->cur], h->d_unom[nxt],
                                                h->cfg.action_low[0], h->cfg.action_high[0], h->d_u, h->h_u_dev, h->seq));
    }
    h->cur = nxt;
    ++h->p2p_seq;
    if (int rc = mppi_advance_hidden(h)) return rc;
    return finish_step(h, u_out);   // a peer time-out comes back as CTK_ERR_STATE through the error word
    });
}

// ---- the resident form: opt-in per handle --------------------------------------------------------------------------------------------
int ctk_resident_enable(ctk_handle* h, int on, double idle_us) {
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    RES_Q(h);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!on) { h->res_enabled = false; h->res_readahead = false; return CTK_OK; }
    if (on != 1 && on != 2) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_resident_enable: on must be 0, 1 or 2 (2 = 1 + read-ahead of immutable sample buffers)");
    if (h->cfg.optimizer != CTK_OPT_MPPI || h->cfg.predictor != CTK_PRED_ODE)
        return fail(h, CTK_ERR_UNSUPPORTED, "ctk_resident_enable: MPPI with the analytic predictor only");
    if (h->cfg.materialize_trajectories || ctk_mppi_uses_throughput_kernel(CTK_PRED_ODE, h->N) || !mppi_can_fuse(h) || h->d_ll == nullptr ||
        ctk_mppi_rollout_env_lds(h->env, h->P, h->H, h->N) > 160 * 1024)
        return fail(h, CTK_ERR_UNSUPPORTED, "ctk_resident_enable: needs the one-launch step with the {value, seq} hand-off (<= 128 workgroups, no materialised trajectories)");
    if (!(idle_us >= 1.0 && idle_us <= 1.0e6)) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_resident_enable: idle_us must be within [1, 1e6]");
    if (!h->res_box) {
        HIP_TRY(h, hipHostMalloc((void**)&h->res_stat, sizeof(CtkResidentStat), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->res_stat, 0, sizeof(CtkResidentStat));
        HIP_TRY(h, hipHostGetDevicePointer((void**)&h->res_stat_dev, h->res_stat, 0));
        HIP_TRY(h, hipMalloc((void**)&h->d_res_relay, sizeof(CtkResidentBox)));
        HIP_TRY(h, hipMalloc(&h->d_res_args, CTK_RES_ARGS_BYTES));
        HIP_TRY(h, hipMemsetAsync(h->d_res_relay, 0, sizeof(CtkResidentBox), h->stream));
        // the mailbox in device memory the HOST can store into (fine-grained allocation, reached through the PCIe BAR) — probed: a pattern
        // stored by the host must read back through a device-side copy.  CTK_RES_HOST_MAILBOX forces the pinned-host fallback (A/B, tests)
        void* vram = nullptr;
        bool local = false;
        int large_bar = 0;                                  // without a large BAR the host cannot address device memory at all: do not touch it
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->cfg.device) != hipSuccess) large_bar = 0;
        if (large_bar && !std::getenv("CTK_RES_HOST_MAILBOX") && hipExtMallocWithFlags(&vram, 4096, hipDeviceMallocFinegrained) == hipSuccess && vram) {
            hipPointerAttribute_t at{};
            if (hipMemset(vram, 0, 4096) == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipPointerGetAttributes(&at, vram) == hipSuccess) {
                volatile uint32_t* pv = static_cast<volatile uint32_t*>(vram);
                for (int i = 0; i < 32; ++i) pv[i] = 0xC0DE0000u + (uint32_t)i;
                __builtin_ia32_sfence();
                uint32_t back[32] = {};
                if (hipMemcpy(back, vram, sizeof(back), hipMemcpyDeviceToHost) == hipSuccess) {
                    local = true;
                    for (int i = 0; i < 32; ++i) local = local && back[i] == 0xC0DE0000u + (uint32_t)i;
                }
            }
            if (!local) { (void)hipFree(vram); vram = nullptr; }
        }
        (void)hipGetLastError();
        if (local) {
            (void)hipMemset(vram, 0, 4096);
            HIP_TRY(h, hipDeviceSynchronize());
            h->res_box = static_cast<CtkResidentBox*>(vram); h->res_box_dev = h->res_box; h->res_local = true;
        } else {
            HIP_TRY(h, hipHostMalloc((void**)&h->res_box, sizeof(CtkResidentBox), hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(h->res_box, 0, sizeof(CtkResidentBox));
            HIP_TRY(h, hipHostGetDevicePointer((void**)&h->res_box_dev, h->res_box, 0));
            h->res_local = false;
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    // HIP multiplexes a process's streams over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and a queue is in order: work of
    // ANOTHER stream that lands on the resident kernel's queue waits until it leaves (<= idle_us each time — measured: a neighbour's 50 us
    // step became 50 ms beside a resident kernel with idle_us = 50 000 in an 18-stream process).  Streams of another priority are served
    // from other queues, so the handle's own stream is re-created at the highest priority once; a stream handed in by ctk_set_stream is
    // the caller's business (include/ctk_hip.h says so).
    if (h->own_stream && !h->res_stream_isolated) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) {
            hipStream_t ns = nullptr;
            if (hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, hi) == hipSuccess) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                (void)hipStreamDestroy(h->stream);
                h->stream = ns;
            }
        }
        (void)hipGetLastError();
        h->res_stream_isolated = true;
    }
    h->res_idle_us = idle_us;
    h->res_readahead = on == 2;
    h->res_follow.clear(); h->res_prev_samples = nullptr;
    h->res_enabled = true;
    return CTK_OK;
}

int ctk_resident_stop(ctk_handle* h) {
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return resident_quiesce(h);
}

int ctk_resident_stats(const ctk_handle* h, uint64_t* launches, uint64_t* steps, int* running, int* mailbox_in_device_memory) {
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    if (h->res_stat && getenv("CTK_RES_TRACE"))
        fprintf(stderr, "[ctk resident] mailbox in %s memory; last request: fetch%s %.2f us, block 0's step %.2f us, inputs %s\n", h->res_local ? "device" : "pinned host",
                h->res_local ? "" : " + relay", (h->res_stat->t_relay & 0x7FFFFFFFu) * 0.01, h->res_stat->t_body * 0.01,
                (h->res_stat->t_relay & 0x80000000u) ? "prepared ahead" : "formed at the request");
    if (h->res_stat && getenv("CTK_RES_TRACE"))
    {
        fprintf(stderr, "[ctk resident] block 0's step: %u shader cycles -> %.0f MHz; stamps", h->res_stat->c_body, h->res_stat->c_body / (h->res_stat->t_body * 0.01 + 1e-9));
        for (int i = 1; i < 12; ++i) fprintf(stderr, " %d", (int)(h->res_stat->stamps[i] - h->res_stat->stamps[0]));   // 10 ns units since block 0 took the request
        fprintf(stderr, "\n");
    }
    if (launches) *launches = h->res_launches;
    if (steps) *steps = h->res_steps;
    if (running) *running = h->res_running ? 1 : 0;
    if (mailbox_in_device_memory) *mailbox_in_device_memory = h->res_local ? 1 : 0;
    return CTK_OK;
}


int ctk_log_enable(ctk_handle* h, size_t capacity_steps) {
    RES_Q(h);
    if (!h) return CTK_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 4; ++i) { if (h->d_log[i]) HIP_TRY(h, hipFree(h->d_log[i])); h->d_log[i] = nullptr; }
    h->log_cap = 0; h->log_count = 0;
    if (capacity_steps == 0) return CTK_OK;
    for (int i = 0; i < 4; ++i) {
        const size_t n = log_slot_floats(h, i);
        if (n == 0) continue;
        if (hipMalloc((void**)&h->d_log[i], capacity_steps * n * sizeof(float)) != hipSuccess) {
            for (int j = 0; j < 4; ++j) { if (h->d_log[j]) hipFree(h->d_log[j]); h->d_log[j] = nullptr; }
            return fail(h, CTK_ERR_HIP, "ctk_log_enable: not enough device memory for the requested capacity");
        }
    }
    h->log_cap = capacity_steps;
    return CTK_OK;
}

size_t ctk_log_count(const ctk_handle* h) { return h ? h->log_count : 0; }

int ctk_log_read(ctk_handle* h, int which, size_t first_step, size_t n_steps, float* dst, size_t cap, size_t* n_out) {
    RES_Q(h);
    if (!h || !dst) return h ? fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_log_read: NULL destination") : CTK_ERR_INVALID_ARGUMENT;
    if (h->log_cap == 0) return fail(h, CTK_ERR_STATE, "ctk_log_read: logging is not enabled (ctk_log_enable)");
    const int ring = which == CTK_BUF_Q ? 0 : which == CTK_BUF_J ? 1 : which == CTK_BUF_TRAJ ? 2 : (which == CTK_BUF_AGES || which == CTK_BUF_AGES_LOGGED) ? 3 : -1;
    if (ring < 0 || !h->d_log[ring]) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_log_read: this tensor is not logged for this handle");
    if (first_step + n_steps > h->log_count) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_log_read: steps not logged yet");
    if (h->log_count > h->log_cap && first_step < h->log_count - h->log_cap)
        return fail(h, CTK_ERR_STATE, "ctk_log_read: the oldest requested step has been overwritten (ring capacity)");
    const size_t n = log_slot_floats(h, ring);
    if (cap < n_steps * n) return fail(h, CTK_ERR_INVALID_ARGUMENT, "ctk_log_read: destination too small");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    size_t done = 0;
    while (done < n_steps) {   // at most two contiguous runs (ring wrap)
        const size_t slot = (first_step + done) % h->log_cap;
        const size_t run = std::min(n_steps - done, h->log_cap - slot);
        HIP_TRY(h, hipMemcpyAsync(dst + done * n, h->d_log[ring] + slot * n, run * n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        done += run;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (n_out) *n_out = n_steps * n;
    return CTK_OK;
}

}  // extern "C"
