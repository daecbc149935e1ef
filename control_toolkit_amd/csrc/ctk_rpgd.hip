// ctk_rpgd.hip — RPGD on gfx950 (replaces reference Optimizers/optimizer_rpgd.py:298-338, :340-380,
// :449-516 and the in-repo torch ADAM :56-82).
//
//   ctk_rpgd_descent<PRED>   ALL `iters` Adam iterations of one MPC step in ONE launch: the N plans
//        are independent during the descent (loss = sum_n J_n, :325), so a block keeps its 64 plans
//        in LDS across iterations.  Per iteration: forward rollout storing what the adjoint needs,
//        hand-written reverse sweep (the autograd tape of :310-314 / :329-333) giving dJ/dQ,
//        per-plan clip_by_norm (:315,:334), Adam (:56-82) with coalesced m/v traffic, clip to the
//        limits (:319,:336).  After the last iteration one more forward pass yields the costs that
//        get_action sorts (:342).  ODE: one thread per plan (wave 0 of the block), adjoint state in
//        LDS.  MLP: 16 plans per wave on the fp32 matrix cores both ways, activations in an
//        L2-resident global scratch.
//   ctk_rpgd_mlp_wide + ctk_rpgd_mlp_jacobians   MLP predictor, small populations: the reverse sweep's sequential part shrunk to a
//        4 x 5 product per step by forming all step Jacobians at once on the idle chip (a launch per phase);
//   ctk_rpgd_mlp_persistent  the same work as ONE launch per MPC step (up to 512 plans, H <= 64): producers keep plans, Adam moments and
//        weights for all iterations, resident Jacobian workers take (iteration, step, tile) tickets; DESIGN.md 2.5b.
//   ctk_rpgd_warmstart       keep the best k (sorted), shift plans by shift_previous and moments by
//        one, resample the rest at the inducing points, age bookkeeping, u = best plan's first
//        input (:426,:449-516,:523).
#include <atomic>
#include <algorithm>
#include "ctk_rollout.h"
#include "ctk_mlp.h"
#include "ctk_adam.h"
#include "ctk_launch.h"

constexpr int RP_TRAJ = 64;
constexpr int RP_WAVES = 4;
constexpr int RP_BLOCK = RP_TRAJ * RP_WAVES;
constexpr int RP_LD = RP_TRAJ + 1;   // LDS row stride of the [h][plan] tiles (odd: conflict-free both ways)
constexpr int RP_NS = 6;             // floats stored per (plan, step) for the ODE adjoint

// ---------------------------------------------------------------------------------------------
// ODE: forward with tape, reverse sweep
// ---------------------------------------------------------------------------------------------
// tape[(h * RP_NS + i) * 64 + lane], i: 0 x, 1 omega, 2 sin, 3 cos, 4 tmp, 5 thdd
CTK_DEV void rpgd_forward_ode_tape(const RolloutArgs& a, const EnvK& k, const float* q_s, float* tape, int lane, State4& s) {
    s = State4{a.s0[0], a.s0[1], a.s0[2], a.s0[3]};
    const int H = a.H;
    float u_next = q_s[lane];
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = q_s[(h + 1) * RP_LD + lane];
        float sn, cs;
        ctk_sincosf(s.th, &sn, &cs);
        const float F = k.u_max * u;
        const float A = F + k.k_ml * s.om * s.om * sn - k.M_fric * s.v;
        const float tmp = A * k.inv_mt;
        const float D = k.k43l - k.k_mpl_mt * cs * cs;
        const float Nn = k.g * sn - cs * tmp - k.k_jf * s.om;
        const float thdd = fdiv_pos(Nn, D);
        const float xdd = tmp - k.k_mpl_mt * thdd * cs;
        float* tp = tape + (size_t)h * RP_NS * 64 + lane;
        tp[0] = s.x; tp[64] = s.om; tp[128] = sn; tp[192] = cs; tp[256] = tmp; tp[320] = thdd;
        const float nx = s.x + k.dt * s.v, nv = s.v + k.dt * xdd, nth = s.th + k.dt * s.om, nom = s.om + k.dt * thdd;
        s.x = nx; s.v = nv; s.th = nth; s.om = nom;
    }
}

// Reverse sweep: g_s[h][lane] = dJ/dQ[n,h]; returns sum_h g^2.  (oracle: rollout_cost_and_grad)
CTK_DEV float rpgd_backward_ode(const RolloutArgs& a, const EnvK& k, const float* q_s, const float* tape, float* g_s, int lane,
                                const State4& sH, float uprev0) {
    const int H = a.H;
    const float inv = a.inv_Hp1;
    const float two_dd = 2.0f * k.dd_weight * k.inv_xs * k.inv_xs;
    // terminal: terminal_weight * (dd + ep) at s_H
    float snH, csH;
    ctk_sincosf(sH.th, &snH, &csH);
    float lx = k.terminal_weight * two_dd * (sH.x - k.target_position) * inv;
    float lv = 0.0f;
    float lth = k.terminal_weight * 2.0f * k.ep_c * (1.0f - csH) * snH * inv;
    float lom = 0.0f;
    float nrm2 = 0.0f;
    float u_hp1 = 0.0f;                       // u_{h+1}
    float u_h = q_s[(H - 1) * RP_LD + lane];
    for (int h = H - 1; h >= 0; --h) {
        const float u_hm1 = h > 0 ? q_s[(h - 1) * RP_LD + lane] : uprev0;
        const float* tp = tape + (size_t)h * RP_NS * 64 + lane;
        const float x = tp[0], om = tp[64], sn = tp[128], cs = tp[192], tmp = tp[256], thdd = tp[320];
        const float D = k.k43l - k.k_mpl_mt * cs * cs;
        const float dt = k.dt;
        // adjoint of the Euler step (oracle Predictor._ode_vjp)
        const float a_xdd = dt * lv;
        const float a_thdd = dt * lom - k.k_mpl_mt * cs * a_xdd;
        float a_tmp = a_xdd;
        float a_cs = -k.k_mpl_mt * thdd * a_xdd;
        const float a_Nn = fdiv_pos(a_thdd, D);
        const float a_D = -a_Nn * thdd;
        float a_sn = k.g * a_Nn;
        a_cs = a_cs - tmp * a_Nn - 2.0f * k.k_mpl_mt * cs * a_D;
        a_tmp = a_tmp - cs * a_Nn;
        const float a_A = a_tmp * k.inv_mt;
        a_sn = a_sn + k.k_ml * om * om * a_A;
        const float o_x = lx;
        const float o_v = lv + dt * lx - k.M_fric * a_A;
        const float o_th = lth + cs * a_sn - sn * a_cs;
        const float o_om = lom + dt * lth - k.k_jf * a_Nn + 2.0f * k.k_ml * om * sn * a_A;
        const float g_q = k.u_max * a_A;
        // direct input-cost gradient: cc + ccrc towards both neighbours
        float gu = 2.0f * k.ccR * u_h + 2.0f * k.ccrc_weight * (u_h - u_hm1);
        if (h + 1 < H) gu -= 2.0f * k.ccrc_weight * (u_hp1 - u_h);
        const float g = gu * inv + g_q;
        g_s[h * RP_LD + lane] = g;
        nrm2 += g * g;
        // stage-cost state gradient at s_h
        lx = two_dd * (x - k.target_position) * inv + o_x;
        lv = o_v;
        lth = 2.0f * k.ep_c * (1.0f - cs) * sn * inv + o_th;
        lom = 2.0f * k.ekp_weight * om * inv + o_om;
        u_hp1 = u_h; u_h = u_hm1;
    }
    return nrm2;
}

// ---------------------------------------------------------------------------------------------
// MLP: forward with the activations taped to an L2-resident global scratch, reverse sweep on MFMA.
// tape layout per wave: [h][lane][20] floats = {state component, h1[8], h2[8], pad[3]} (5 x float4
// per lane, lane-contiguous: every store/load instruction moves 1 KiB coalesced).
// ---------------------------------------------------------------------------------------------
constexpr int RP_MLP_TAPE = 20;

CTK_DEV float rpgd_forward_mlp_tape(const RolloutArgs& a, const MlpFwdT& w, const float* q_s, float* tape, int col, int g) {
    const int lane = threadIdx.x & 63;
    float sv = lane_state4(a, g);
    const int H = a.H;
    float u_next = q_s[col];
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = q_s[(h + 1) * RP_LD + col];   // a step ahead: the input opens the step (b1 + w1u * u)
        MlpAct act;
        const float nsv = mlp_step(w, sv, u, g, &act);
        float4* tp = reinterpret_cast<float4*>(tape + ((size_t)h * 64 + lane) * RP_MLP_TAPE);
        tp[0] = make_float4(sv, 0.f, 0.f, 0.f);
        tp[1] = make_float4(act.h1[0][0], act.h1[0][1], act.h1[0][2], act.h1[0][3]);
        tp[2] = make_float4(act.h1[1][0], act.h1[1][1], act.h1[1][2], act.h1[1][3]);
        tp[3] = make_float4(act.h2[0][0], act.h2[0][1], act.h2[0][2], act.h2[0][3]);
        tp[4] = make_float4(act.h2[1][0], act.h2[1][1], act.h2[1][2], act.h2[1][3]);
        sv = nsv;
    }
    return sv;
}

// The same forward pass with a tile's step shared by two waves (ctk_mlp.h: mlp_step_pair) — the wide form, where the chip is
// empty and the recurrence's latency is what counts.  Every wave of the workgroup must call this (workgroup barriers inside).
constexpr int RP_PAIR_EX = MLP_PAIR_EX;

CTK_DEV float rpgd_forward_mlp_tape_pair(const RolloutArgs& a, const MlpFwdHalf& w, const float* q_s, int ld, float* tape, int col, int g, int m,
                                         float* ex) {
    const int lane = threadIdx.x & 63;
    float sv = lane_state4(a, g);
    const int H = a.H;
    float u_next = q_s[col];
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = q_s[(h + 1) * ld + col];
        MlpHalfAct act;
        const float nsv = mlp_step_pair(w, sv, u, m, ex, &act);
        // the tape row of (tile, h): {state}, h1[tile 0], h1[tile 1], h2[tile 0], h2[tile 1] — each wave its own halves
        float4* tp = reinterpret_cast<float4*>(tape + ((size_t)h * 64 + lane) * RP_MLP_TAPE);
        if (m == 0) tp[0] = make_float4(sv, 0.f, 0.f, 0.f);
        tp[1 + m] = make_float4(act.h1m[0], act.h1m[1], act.h1m[2], act.h1m[3]);
        tp[3 + m] = make_float4(act.h2m[0], act.h2m[1], act.h2m[2], act.h2m[3]);
        sv = nsv;
    }
    return sv;
}

// terminal adjoint share of lane group g (terminal_weight * (dd + ep) at s_H), scaled like the stage terms
CTK_DEV float rpgd_mlp_terminal_adjoint(const RolloutArgs& a, const EnvK& k, int g, float svH) {
    const float two_dd = 2.0f * k.dd_weight * k.inv_xs * k.inv_xs;
    float snT, csT;
    ctk_sincosf(svH, &snT, &csT);
    return k.terminal_weight * a.inv_Hp1 * (g == 0 ? two_dd * (svH - k.target_position) : (g == 2 ? 2.0f * k.ep_c * (1.0f - csT) * snT : 0.0f));
}

// reverse sweep for the wave's 16 plans; lanes of group 0 write g_s[h][col] and return sum_h g^2
CTK_DEV float rpgd_backward_mlp(const RolloutArgs& a, const EnvK& k, const MlpBwdW& w, const float* q_s, const float* tape,
                                float* g_s, int col, int g, float svH, float uprev0) {
    const int lane = threadIdx.x & 63;
    const int H = a.H;
    const float inv = a.inv_Hp1;
    const float two_dd = 2.0f * k.dd_weight * k.inv_xs * k.inv_xs;
    float lam = rpgd_mlp_terminal_adjoint(a, k, g, svH);
    float nrm2 = 0.0f;
    float u_hp1 = 0.0f, u_h = q_s[(H - 1) * RP_LD + col];
    const float4* tp = reinterpret_cast<const float4*>(tape + ((size_t)(H - 1) * 64 + lane) * RP_MLP_TAPE);
    float4 r0 = tp[0], r1 = tp[1], r2 = tp[2], r3 = tp[3], r4 = tp[4];
    for (int h = H - 1; h >= 0; --h) {
        MlpAct act;
        act.h1[0] = f32x4{r1.x, r1.y, r1.z, r1.w}; act.h1[1] = f32x4{r2.x, r2.y, r2.z, r2.w};
        act.h2[0] = f32x4{r3.x, r3.y, r3.z, r3.w}; act.h2[1] = f32x4{r4.x, r4.y, r4.z, r4.w};
        const float sv = r0.x;
        if (h > 0) {   // prefetch the previous step's tape while this step's MFMAs run
            const float4* tq = reinterpret_cast<const float4*>(tape + ((size_t)(h - 1) * 64 + lane) * RP_MLP_TAPE);
            r0 = tq[0]; r1 = tq[1]; r2 = tq[2]; r3 = tq[3]; r4 = tq[4];
        }
        const float u_hm1 = h > 0 ? q_s[(h - 1) * RP_LD + col] : uprev0;
        float du;
        const float ds = mlp_step_vjp(w, act, lam, &du);
        float gu = 2.0f * k.ccR * u_h + 2.0f * k.ccrc_weight * (u_h - u_hm1);
        if (h + 1 < H) gu -= 2.0f * k.ccrc_weight * (u_hp1 - u_h);
        const float gq = gu * inv + du;
        if (g == 0) { g_s[h * RP_LD + col] = gq; nrm2 += gq * gq; }
        float sn, cs;
        ctk_sincosf(sv, &sn, &cs);
        const float share = g == 0 ? two_dd * (sv - k.target_position)
                          : (g == 2 ? 2.0f * k.ep_c * (1.0f - cs) * sn : (g == 3 ? 2.0f * k.ekp_weight * sv : 0.0f));
        lam = share * inv + ds;
        u_hp1 = u_h; u_h = u_hm1;
    }
    return nrm2;
}

// ---------------------------------------------------------------------------------------------
// MLP, wide form (small populations): the reverse sweep above is a chain of H dependent network products per wave
// while the chip idles (cfg4: 16 waves on 1024 SIMDs, 1.05 us per step).  But lam_h = c_h + J_h^T lam_{h+1} is LINEAR
// in lam once the step Jacobians J_h = d s_{h+1} / d (s_h, u_h) (4 x 5) are known, and those depend on the taped
// activations only: all N x H of them are formed at once by a grid-wide launch (ctk_rpgd_mlp_jacobians, one wave per
// (16-plan tile, step), five tangent passes through the network on the matrix cores), and the sequential part shrinks to
// a 4 x 5 product per step (rpgd_chain_mlp).  Same gradient (optimizer_rpgd.py:310-315), other association of the
// products.
// Forward-mode tangents land component-wise in lane group g (the state layout): lane (c, i) of the Jacobian wave ends with
// ROW i of plan c's Jacobian.  The chain wants a plan in ONE quad of lanes (its products then need only quad-local DPP
// broadcasts): record slot 4c + j of (tile, step) holds {J[0..3][j]}, {J[j][4], c_h[j], 0, 0} — COLUMN j and the stage-cost
// adjoint share — so the Jacobian wave scatters its row into the four slots of its plan.
// ---------------------------------------------------------------------------------------------
constexpr int RP_JAC = 8;

// Tangent j of one step through the network (forward mode on the matrix cores): input basis vector e_j (j < 4: state component j, j == 4: the
// input) -> the lane's row of column j of the step Jacobian, J[g][j] of plan c.  D..: tanh' of the step's activations (1 - h^2).
CTK_DEV float rpgd_mlp_tangent(const MlpFwdT& w, const f32x4& D10, const f32x4& D11, const f32x4& D20, const f32x4& D21, int j, int g) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    // tangent of layer 1's pre-activation for input basis vector e_j: column j of W1 at this lane's accumulator rows
    f32x4 a0, a1;
    if (j < 4) {
        const float ind = g == j ? 1.0f : 0.0f;
        a0 = CTK_MFMA(w.w1s[0], ind, z); a1 = CTK_MFMA(w.w1s[1], ind, z);
    } else { a0 = w.w1u[0]; a1 = w.w1u[1]; }
    f32x4 d1[2];
    d1[0] = a0 * D10; d1[1] = a1 * D11;
    f32x4 c0 = z, c1 = z;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const float b = d1[jj >> 2][jj & 3];
        c0 = CTK_MFMA(w.w2[0][jj], b, c0);
        c1 = CTK_MFMA(w.w2[1][jj], b, c1);
    }
    f32x4 d2[2];
    d2[0] = c0 * D20; d2[1] = c1 * D21;
    f32x4 p0 = z, p1 = z;
#pragma unroll
    for (int jj = 0; jj < 8; jj += 2) {
        p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[jj], d2[jj >> 2][jj & 3], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[jj + 1], d2[(jj + 1) >> 2][(jj + 1) & 3], p1, 0, 0, 0);
    }
    const f32x4 pp = p0 + p1;
    return swap_sum16(swap_sum32(pp[0], pp[2]), swap_sum32(pp[1], pp[3]));
}

// stage-cost adjoint share of lane group g at s_h (the `share` of rpgd_backward_mlp), branch-free
CTK_DEV float rpgd_mlp_stage_share(const RolloutArgs& a, const EnvK& k, int g, float sv) {
    float sn, cs;
    ctk_sincosf(sv, &sn, &cs);
    const float two_dd = 2.0f * k.dd_weight * k.inv_xs * k.inv_xs;
    const float A2 = g == 0 ? two_dd : (g == 3 ? 2.0f * k.ekp_weight : 0.0f);
    const float B = g == 0 ? k.target_position : 0.0f;
    const float E2 = g == 2 ? 2.0f * k.ep_c : 0.0f;
    return (A2 * (sv - B) + E2 * (1.0f - cs) * sn) * a.inv_Hp1;
}

CTK_DEV void rpgd_mlp_jacobian_record(const RolloutArgs& a, const EnvK& k, const MlpFwdT& w, const float* __restrict__ tape_row,
                                      float* __restrict__ jac_step, int c, int g) {
    const float4* tp = reinterpret_cast<const float4*>(tape_row);
    const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2], r3 = tp[3], r4 = tp[4];
    const f32x4 one = f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 h10 = f32x4{r1.x, r1.y, r1.z, r1.w}, h11 = f32x4{r2.x, r2.y, r2.z, r2.w};
    const f32x4 h20 = f32x4{r3.x, r3.y, r3.z, r3.w}, h21 = f32x4{r4.x, r4.y, r4.z, r4.w};
    const f32x4 D10 = one - h10 * h10, D11 = one - h11 * h11, D20 = one - h20 * h20, D21 = one - h21 * h21;   // tanh'
    float Jrow[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) Jrow[j] = rpgd_mlp_tangent(w, D10, D11, D20, D21, j, g);
    const float share = rpgd_mlp_stage_share(a, k, g, r0.x);
    // jac_step: the 64 x RP_JAC floats of this (tile, step); this lane is row g of plan c
    float* slot = jac_step + (size_t)(4 * c) * RP_JAC;
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j * RP_JAC + g] = Jrow[j];
    slot[g * RP_JAC + 4] = Jrow[4];
    slot[g * RP_JAC + 5] = share;
}

// The sequential part of the reverse sweep for the wave's 16 plans, from the Jacobian records of its tile.  Lane 4b + j is
// component j of plan b (QUAD layout):  lam_h[j] = c_h[j] + sum_i lam_{h+1}[i] J_h[i][j]  is four FMAs whose lam operand is a
// quad-local DPP broadcast, and  dJ/du_h = sum_i lam_{h+1}[i] J_h[i][4]  a two-step DPP sum over the quad, + the input-cost
// terms.  Lanes j == 0 write g_s[h][plan] and return sum_h g^2 (as rpgd_backward_mlp); lamH comes in quad layout.
template <bool THROUGH, bool CHECK = false>
struct RpgdChainMlpT {
    // a step is ~20 instructions; the records were written by waves all over the chip (other XCDs' L2s), a round trip to them
    // costs >10 steps' worth: two register chunks of RP_CH steps, one being consumed while the other is in flight
    static constexpr int RP_CH = 8;
    float4 bufA[RP_CH][2], bufB[RP_CH][2];
    const float* jac;
    float lamH;
    // CHECK (the one-launch form, where the records are written during the launch that reads them): every record carries the sequence number
    // of its iteration in a spare word of its second half — same 32-byte sector as everything the chain reads of it — and a copy a cache
    // kept from an earlier iteration shows the earlier number
    uint32_t want = 0, stale = 0;

    CTK_DEV void fetch(float4 (&dst)[RP_CH][2], int top) const {     // steps top, top-1, ..., top-RP_CH+1 (those >= 0)
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < RP_CH; ++i) {
            const int h = max(top - i, 0);                   // unconditional (clamped): straight-line loads keep the wait counts exact
            const float4* jq = reinterpret_cast<const float4*>(jac + ((size_t)h * 64 + lane) * RP_JAC);
            if constexpr (THROUGH) { dst[i][0] = ld4_through(jq); dst[i][1] = ld4_through(jq + 1); }   // (records stored by workers on other XCDs
            else { dst[i][0] = jq[0]; dst[i][1] = jq[1]; }                                               //  in THIS launch: past this XCD's L2)
        }
    }
    // issue the first loads (the launch does this before anything waits on memory)
    CTK_DEV void begin(const float* jac_tile, const float* term, int H) {
        jac = jac_tile;
        lamH = term[threadIdx.x & 63];
        fetch(bufA, H - 1);
    }
    CTK_DEV void begin(const float* jac_tile, float lam_terminal, int H) {
        jac = jac_tile;
        lamH = lam_terminal;
        fetch(bufA, H - 1);
    }
    // Writes du_h = sum_i lam_{h+1}[i] J_h[i][4] to g_s[h][plan] (lanes j == 0, a chunk at a time); the input-cost terms, which do not
    // depend on lam, are added afterwards by rpgd_finish_gradient with the whole workgroup.
    template <bool PARTIAL>
    CTK_DEV void consume(const float4 (&src)[RP_CH][2], int top, float& lam, float* g_col, int ld) {
        float du[RP_CH];
#pragma unroll
        for (int i = 0; i < RP_CH; ++i) {
            if (PARTIAL && top - i < 0) break;
            const float4 c0 = src[i][0], c1 = src[i][1];
            if constexpr (CHECK) stale |= __builtin_bit_cast(uint32_t, c1.z) ^ want;
            float pu = lam * c1.x;
            pu += dpp_mov<DPP_QUAD_XOR1>(pu);
            du[i] = pu + dpp_mov<DPP_QUAD_XOR2>(pu);
            // quad_perm broadcasts of lanes 0..3 of the quad; two short accumulation chains
            const float e = fmaf(dpp_mov<0x00>(lam), c0.x, c1.y), o = dpp_mov<0x55>(lam) * c0.y;
            lam = fmaf(dpp_mov<0xAA>(lam), c0.z, e) + fmaf(dpp_mov<0xFF>(lam), c0.w, o);
        }
        if ((threadIdx.x & 3) == 0) {
#pragma unroll
            for (int i = 0; i < RP_CH; ++i) {
                if (PARTIAL && top - i < 0) break;
                g_col[(top - i) * ld] = du[i];
            }
        }
    }
    // g_tile: &g_s[0][first plan of the tile], ld: row stride of g_s
    CTK_DEV void run(const RolloutArgs& a, float* g_tile, int ld) {
        const int lane = threadIdx.x & 63;
        const int H = a.H;
        float lam = lamH;
        float* g_col = g_tile + (lane >> 2);
        int top = H - 1;
        for (; top >= 2 * RP_CH - 1; top -= 2 * RP_CH) {      // two full chunks per round
            fetch(bufB, top - RP_CH);
            consume<false>(bufA, top, lam, g_col, ld);
            fetch(bufA, top - 2 * RP_CH);
            consume<false>(bufB, top - RP_CH, lam, g_col, ld);
        }
        if (top >= 0) {                                        // the last 1 .. 2 RP_CH - 1 steps
            fetch(bufB, top - RP_CH);
            consume<true>(bufA, top, lam, g_col, ld);
            if (top - RP_CH >= 0) consume<true>(bufB, top - RP_CH, lam, g_col, ld);
        }
    }
};
using RpgdChainMlp = RpgdChainMlpT<false, false>;

// ---------------------------------------------------------------------------------------------
// warm start / resampling / reset.  One thread per (row, h) of the NEW population.
//   new row i <  n_new : fresh sample (sample_actions :275-296), moments 0, age 0
//   new row i >= n_new : keeper idx[i - n_new] (or row i itself when gather == 0): plan shifted
//                        by shift_previous repeating the last input (:377-379), moments shifted by
//                        ONE and zero-filled (:465,:501), age kept; then every age += 1 (:514)
// ---------------------------------------------------------------------------------------------
struct WarmArgs {
    int N, H, P, n_new, gather, shift_previous, sampling_distribution, reset;
    float sample_stdev, sample_mean, sample_min, sample_max;   // sample_min/max < lo/hi only when sample_whole_control_space is off;
                                                               // otherwise the per-channel limits a.lo / a.hi are the range (whole_space)
    int whole_space;
    // sharded step: keepers and the best plan come from the all-gathered keeper records
    // {J, global index, age, Q[H], m[H], v[H]} instead of this handle's own rows
    const float* recs;     // nullptr: single-handle step
    int rs;                // record stride (3 + 3H)
    int keeper_base;       // index (in the global sorted keeper list) of the first keeper this shard hosts
    int fresh_tail;        // gradient_tf: the shifted-in tail input is a fresh U[lo,hi) draw per plan
                           // (optimizer_gradient_tf.py:137-144) instead of a repeat of the last input
};

// pointers of one warm start (old population -> new population)
struct WarmPtrs {
    const float* draws; const int* idx; const float* Q_old; const float* m_old; const float* v_old; const float* ages_old;
    float* Q_new; float* m_new; float* v_new; float* ages_new; const InterpEntry* interp; float* u_nom; float* u_dev; float* u_host;
    uint32_t seq;
};

// element `gid` (= (row * H + h) * C + c) of the new population [N,H,C]; elements 0..H*C-1 also copy the best plan out and
// element 0 publishes u.  C = a.C control inputs: the reference's tensors are [N,H,C] throughout (optimizer_rpgd.py:275-296,
// :377-379, :454-513), a step is C contiguous floats.
CTK_DEV void rpgd_warm_element(const WarmArgs& w, const RolloutArgs& a, const WarmPtrs& p, int gid) {
    const float* __restrict__ draws = p.draws; const int* __restrict__ idx = p.idx;
    const float* __restrict__ Q_old = p.Q_old; const float* __restrict__ m_old = p.m_old; const float* __restrict__ v_old = p.v_old;
    const float* __restrict__ ages_old = p.ages_old;
    float* __restrict__ Q_new = p.Q_new; float* __restrict__ m_new = p.m_new; float* __restrict__ v_new = p.v_new;
    float* __restrict__ ages_new = p.ages_new; const InterpEntry* __restrict__ interp = p.interp;
    float* __restrict__ u_nom = p.u_nom; float* __restrict__ u_dev = p.u_dev; float* __restrict__ u_host = p.u_host;
    const uint32_t seq = p.seq;
    const int H = w.H, C = a.C, HC = H * C, PC = w.P * C;
    if (gid < w.N * HC) {
        const int i = gid / HC, hc = gid - i * HC, h = hc / C, c = hc - h * C;
        float q, mm = 0.0f, vv = 0.0f;
        if (i < w.n_new) {
            const InterpEntry e = interp[h];
            const float smin = w.whole_space ? a.lo[c] : w.sample_min, smax = w.whole_space ? a.hi[c] : w.sample_max;
            float y[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = min(e.i0 + j, w.P - 1) * C + c;      // column of the [P,C] draw block of this row
                float d;
                if (draws != nullptr) {
                    d = draws[(size_t)i * PC + col];
                } else {
                    float d4[4];
                    draw4(a, (uint32_t)(a.global_row0 + i), (uint32_t)(col >> 2), w.sampling_distribution == 0 ? 1 : 0, d4);
                    d = d4[col & 3];
                }
                const float raw = w.sampling_distribution == 0 ? d * (smax - smin) + smin                           // uniform
                                                                : d * w.sample_stdev + w.sample_mean;                // normal
                y[j] = fminf(fmaxf(raw, a.lo[c]), a.hi[c]);                                                          // :292
            }
            q = y[0] * e.w0 + (e.i0 + 1 < w.P ? y[1] * e.w1 : 0.0f);                                                 // :294
        } else if (w.gather && w.recs) {
            const float* rec = w.recs + (size_t)idx[w.keeper_base + i - w.n_new] * w.rs;
            const int hs = min(h + w.shift_previous, H - 1);
            q = rec[3 + hs * C + c];
            if (h + 1 < H) { mm = rec[3 + HC + hc + C]; vv = rec[3 + 2 * HC + hc + C]; }
        } else {
            const int src = w.gather ? idx[i - w.n_new] : i;
            const int hs = min(h + w.shift_previous, H - 1);
            q = Q_old[(size_t)src * HC + hs * C + c];
            if (w.fresh_tail && h + w.shift_previous >= H) {
                float d;
                if (draws != nullptr) {
                    d = draws[(size_t)i * C + c];
                } else {
                    float d4[4];
                    draw4(a, (uint32_t)(a.global_row0 + i), 0u, 1, d4);
                    d = d4[c & 3];
                }
                q = d * (a.hi[c] - a.lo[c]) + a.lo[c];
            }
            if (h + 1 < H) { mm = m_old[(size_t)src * HC + hc + C]; vv = v_old[(size_t)src * HC + hc + C]; }
        }
        Q_new[gid] = q; m_new[gid] = mm; v_new[gid] = vv;
        if (hc == 0) {
            float age = 0.0f;
            if (i >= w.n_new) {
                if (w.gather && w.recs) age = w.recs[(size_t)idx[w.keeper_base + i - w.n_new] * w.rs + 2];
                else age = ages_old[w.gather ? idx[i - w.n_new] : i];
            }
            ages_new[i] = w.reset ? 0.0f : age + 1.0f;
        }
    }
    if (!w.reset && gid < HC) {
        // u_nom = Q_tf[best_idx[0]] BEFORE the warm start (:426)
        const float q = w.recs ? w.recs[(size_t)idx[0] * w.rs + 3 + gid] : Q_old[(size_t)idx[0] * HC + gid];
        u_nom[gid] = q;
        if (C == 1) {
            if (gid == 0) publish_u(u_dev, u_host, q, seq);   // :523
        } else if (gid == 0) {
            // one thread publishes the whole input vector: u[c] first (floats 4..), then the {u[0], seq} word the host polls
            for (int cc = 0; cc < C; ++cc) {
                const float uc = w.recs ? w.recs[(size_t)idx[0] * w.rs + 3 + cc] : Q_old[(size_t)idx[0] * HC + cc];
                u_dev[cc] = uc;
                __hip_atomic_store(u_host + 4 + cc, uc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const unsigned long long pv = ((unsigned long long)seq << 32) | (unsigned long long)__builtin_bit_cast(unsigned, q);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(u_host), pv, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// this shard's best plans with their optimizer state, sorted: {J, global index, age, Q[H], m[H], v[H]}

__global__ __launch_bounds__(256) void ctk_rpgd_warmstart(WarmArgs w, RolloutArgs a, WarmPtrs p) {
    rpgd_warm_element(w, a, p, blockIdx.x * blockDim.x + threadIdx.x);
}

// Single-workgroup RPGD step (N <= 64, the reference's default is 32): keep-k selection and the warm start run as the tail
// of the descent launch — the whole optimizer_rpgd.py:388-524 step in ONE launch instead of three.
struct FusedWarm {
    int enabled, K;
    int* idx_out;          // [K] best indices (ascending cost), as ctk_select_topk writes them
    WarmArgs w;
    WarmPtrs p;            // p.idx is ignored (the tail's own selection is used)
};

// keep-k selection + warm start as the tail of a descent launch whose ONE workgroup holds the whole population
CTK_DEV void rpgd_fused_tail(const RolloutArgs& a, const FusedWarm& fw, float* g_s, int t, int H) {
    __threadfence();
    __syncthreads();                                   // Q, m, v, J of this launch are visible to every thread of the block
    uint32_t* key_s = reinterpret_cast<uint32_t*>(g_s);   // g_s is dead: [64] keys, then [64] indices
    int* idx_s = reinterpret_cast<int*>(g_s) + 64;
    if (t < 64) {
        const float Jt = t < a.N ? __hip_atomic_load(a.J + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INFINITY;
        const uint32_t u = __builtin_bit_cast(uint32_t, Jt);
        key_s[t] = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);   // order-preserving map (ctk_sampled.hip:f32_sortable)
    }
    __syncthreads();
    if (t < a.N) {                                      // rank under the total order (J, index): ctk_select_topk
        const uint32_t ki = key_s[t];
        int rk = 0;
        for (int j = 0; j < a.N; ++j) { const uint32_t kj = key_s[j]; rk += (kj < ki) | ((kj == ki) & (j < t)); }
        if (rk < fw.K) { idx_s[rk] = t; fw.idx_out[rk] = t; }
    }
    __syncthreads();
    WarmPtrs p = fw.p;
    p.idx = idx_s;
    for (int gid = t; gid < max(fw.w.N * H, H); gid += RP_BLOCK) rpgd_warm_element(fw.w, a, p, gid);   // CartPole: C == 1
}

template <int PRED>
__global__ __launch_bounds__(RP_BLOCK) void ctk_rpgd_descent(RolloutArgs a, EnvK k, AdamK ad, float* __restrict__ Q,
                                                             float* __restrict__ m, float* __restrict__ v,
                                                             const float* __restrict__ bc_table, int bc_len, int t0, int iters,
                                                             const float* __restrict__ wperm, float* __restrict__ scratch,
                                                             int tape_in_lds, FusedWarm fw) {
    extern __shared__ float lds[];
    const int H = a.H;
    float* q_s = lds;                       // [H][65]
    float* g_s = q_s + H * RP_LD;           // [H][65]
    float* sc_s = g_s + H * RP_LD;          // [64] clip scale per plan
    float* tape_l = sc_s + RP_TRAJ;         // [H][6][64] (ODE, when it fits)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * RP_TRAJ;
    const int rows = min(RP_TRAJ, a.N - row0);
    const int total = rows * H;
    const size_t gbase = (size_t)row0 * H;
    const float uprev0 = uniform_u_prev0(a);

    // plans of this block -> LDS, transposed to [h][plan] (coalesced global read)
    for (int i = t; i < RP_TRAJ * H; i += RP_BLOCK) {
        const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;   // p_magic = ceil(2^32 / H) here
        q_s[h * RP_LD + r] = i < total ? Q[gbase + i] : 0.0f;
    }
    __syncthreads();

    if constexpr (PRED == CTK_PRED_ODE) {
        float* tape = tape_in_lds ? tape_l : scratch + (size_t)blockIdx.x * H * RP_NS * 64;
        for (int it = 0; it < iters; ++it) {
            if (wave == 0) {
                State4 sH;
                rpgd_forward_ode_tape(a, k, q_s, tape, lane, sH);
                const float nrm2 = rpgd_backward_ode(a, k, q_s, tape, g_s, lane, sH, uprev0);
                // lib.clip_by_norm(g, clip, [1,2]) = g * clip / max(||g||, clip)   (:315,:334)
                sc_s[lane] = ad.clip / fmaxf(sqrtf(nrm2), ad.clip);
            }
            __syncthreads();
            const int ti = t0 + it + 1;                                   // state['step'] += 1 (:59)
            const float bc1 = ti <= bc_len ? bc_table[2 * (ti - 1)] : 1.0f;
            const float bc2 = ti <= bc_len ? bc_table[2 * (ti - 1) + 1] : 1.0f;
            for (int i = t; i < total; i += RP_BLOCK) {
                const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;
                float mm = 0.0f, vv = 0.0f;
                if (ad.rule != 2) { mm = m[gbase + i]; vv = v[gbase + i]; }
                const float g = g_s[h * RP_LD + r] * sc_s[r];
                q_s[h * RP_LD + r] = adam_update(ad, q_s[h * RP_LD + r], g, mm, vv, bc1, bc2, a.lo[0], a.hi[0]);
                if (ad.rule != 2) { m[gbase + i] = mm; v[gbase + i] = vv; }
            }
            __syncthreads();
        }
        // get_action's forward pass (:342): costs of the refined plans
        if (wave == 0) {
            const int n = row0 + lane;
            const bool valid = n < a.N;
            const float J = rollout_ode<false, false>(a, k, n, valid, [&](int h) { return q_s[h * RP_LD + lane]; });
            if (valid) a.J[n] = J;
        }
    }
    else {
        const MlpFwdT wf = mlp_load_fwd_thin(wperm);
        const MlpBwdW wb = mlp_load_bwd(wperm);
        const int g = lane >> 4, col = wave * CTK_MLP_TRAJ_PER_WAVE + (lane & 15);
        float* tape = scratch + (size_t)(blockIdx.x * RP_WAVES + wave) * H * 64 * RP_MLP_TAPE;
        for (int it = 0; it < iters; ++it) {
#if defined(CTK_DIAG_RPGD_NO_FWD)   // timing experiments only (tools/rpgd_split.sh): the sweeps in isolation, results meaningless
            const float svH = it == 0 ? rpgd_forward_mlp_tape(a, wf, q_s, tape, col, g) : 0.1f * g;
#else
            const float svH = rpgd_forward_mlp_tape(a, wf, q_s, tape, col, g);
#endif
#if defined(CTK_DIAG_RPGD_NO_BWD)
            const float nrm2 = svH * svH;
            if (g == 0) for (int h = 0; h < H; ++h) g_s[h * RP_LD + col] = 0.01f * svH;
#else
            const float nrm2 = rpgd_backward_mlp(a, k, wb, q_s, tape, g_s, col, g, svH, uprev0);
#endif
            if (g == 0) sc_s[col] = ad.clip / fmaxf(sqrtf(nrm2), ad.clip);
            __syncthreads();
            const int ti = t0 + it + 1;
            const float bc1 = ti <= bc_len ? bc_table[2 * (ti - 1)] : 1.0f;
            const float bc2 = ti <= bc_len ? bc_table[2 * (ti - 1) + 1] : 1.0f;
            for (int i = t; i < total; i += RP_BLOCK) {
                const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;
                float mm = 0.0f, vv = 0.0f;
                if (ad.rule != 2) { mm = m[gbase + i]; vv = v[gbase + i]; }
                const float gg = g_s[h * RP_LD + r] * sc_s[r];
                q_s[h * RP_LD + r] = adam_update(ad, q_s[h * RP_LD + r], gg, mm, vv, bc1, bc2, a.lo[0], a.hi[0]);
                if (ad.rule != 2) { m[gbase + i] = mm; v[gbase + i] = vv; }
            }
            __syncthreads();
        }
        const float J = rollout_mlp<false, false>(a, k, wf, row0 + wave * CTK_MLP_TRAJ_PER_WAVE, [&](int h) { return q_s[h * RP_LD + col]; });
        if (lane < 16 && row0 + col < a.N) a.J[row0 + col] = J;
    }
    __syncthreads();
    for (int i = t; i < total; i += RP_BLOCK) {
        const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;
        Q[gbase + i] = q_s[h * RP_LD + r];
    }
    if (fw.enabled) rpgd_fused_tail(a, fw, g_s, t, H);   // one workgroup holds the whole population (host guarantees gridDim.x == 1)
}

// ---- MLP, wide form: one launch per phase --------------------------------------------------------------------------
// scratch: [tiles][H][64][RP_MLP_TAPE] tape | [tiles][H][64][RP_JAC] Jacobian records | [tiles][64] terminal adjoints
CTK_DEV size_t rp_wide_jac_off(int tiles, int H) { return (size_t)tiles * H * 64 * RP_MLP_TAPE; }
CTK_DEV size_t rp_wide_term_off(int tiles, int H) { return (size_t)tiles * H * 64 * (RP_MLP_TAPE + RP_JAC); }

// one wave per (tile, step): jobs = live_tiles * H
__global__ __launch_bounds__(RP_BLOCK) void ctk_rpgd_mlp_jacobians(RolloutArgs a, EnvK k, const float* __restrict__ wperm,
                                                                  float* __restrict__ scratch, int tiles, int jobs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int job = blockIdx.x * RP_WAVES + wave;       // = tile * H + step (wave-uniform)
    if (job >= jobs) return;
    const MlpFwdT w = mlp_load_fwd_thin(wperm);
    const float* tape_row = scratch + ((size_t)job * 64 + lane) * RP_MLP_TAPE;
    float* jac_step = scratch + rp_wide_jac_off(tiles, a.H) + (size_t)job * 64 * RP_JAC;
    rpgd_mlp_jacobian_record(a, k, w, tape_row, jac_step, lane & 15, g);
}

// phase launch of the descent: [update from the previous launch's tape + Jacobians] then [forward with tape | final costs]
//   update: reverse chain -> dJ/dQ, clip_by_norm, Adam step `ti` (1-based), as one iteration of ctk_rpgd_descent
//   final : get_action's cost pass (:342) (+ the fused keep-k / warm-start tail when one workgroup holds the population)
// A workgroup holds RP_WTRAJ = 32 plans = 2 tiles; waves (2p, 2p+1) share tile p in the forward pass
// (rpgd_forward_mlp_tape_pair), wave 2p runs its chain.  LDS: q_s[H][33] | g_s[max(H*33, 128)] | sc_s[32] | ex[2][RP_PAIR_EX]
constexpr int RP_WTRAJ = 32, RP_WLD = RP_WTRAJ + 1, RP_WTILES = 2;

CTK_DEV void rpgd_finish_gradient_w(const RolloutArgs& a, const EnvK& k, const AdamK& ad, const float* q_s, float* g_s, float* sc_s, float uprev0) {
    // g_s[h][r] holds du (network part of dJ/du); add the input-cost terms (cc + ccrc towards both neighbours, as
    // rpgd_backward_mlp) and form the per-plan clip scale: 8 threads per plan, each an eighth of the horizon
    const int t = threadIdx.x, r = t >> 3, part = t & 7, H = a.H;
    const float inv = a.inv_Hp1;
    float nrm2 = 0.0f;
    for (int h = part; h < H; h += 8) {
        const float u_h = q_s[h * RP_WLD + r], u_hm1 = h > 0 ? q_s[(h - 1) * RP_WLD + r] : uprev0;
        float gu = 2.0f * k.ccR * u_h + 2.0f * k.ccrc_weight * (u_h - u_hm1);
        if (h + 1 < H) gu -= 2.0f * k.ccrc_weight * (q_s[(h + 1) * RP_WLD + r] - u_h);
        const float gq = gu * inv + g_s[h * RP_WLD + r];
        g_s[h * RP_WLD + r] = gq;
        nrm2 += gq * gq;
    }
    nrm2 += dpp_mov<DPP_QUAD_XOR1>(nrm2);
    nrm2 += dpp_mov<DPP_QUAD_XOR2>(nrm2);
    nrm2 += dpp_mov<DPP_ROW_HALF_MIRROR>(nrm2);           // lanes 0..7 of each half row: the 8 parts of one plan
    if (part == 0) sc_s[r] = ad.clip / fmaxf(sqrtf(nrm2), ad.clip);   // lib.clip_by_norm(g, clip, [1,2]) (:315,:334)
}

__global__ __launch_bounds__(RP_BLOCK) void ctk_rpgd_mlp_wide(RolloutArgs a, EnvK k, AdamK ad, float* __restrict__ Q, float* __restrict__ m,
                                                             float* __restrict__ v, const float* __restrict__ bc_table, int bc_len, int ti,
                                                             const float* __restrict__ wperm, float* __restrict__ scratch, int tiles,
                                                             int update, int final_pass, FusedWarm fw) {
    extern __shared__ float lds[];
    const int H = a.H;
    float* q_s = lds;                                   // [H][33]
    float* g_s = q_s + H * RP_WLD;                      // [H][33] (>= 128 words: the fused tail's keys + indices)
    float* sc_s = g_s + max(H * RP_WLD, 128);           // [32]
    float* ex_s = sc_s + RP_WTRAJ;                      // [2 pairs][RP_PAIR_EX]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, pair = wave >> 1, half = wave & 1;
    const int row0 = blockIdx.x * RP_WTRAJ;
    const int rows = min(RP_WTRAJ, a.N - row0);
    const int total = rows * H;
    const size_t gbase = (size_t)row0 * H;
    const int g = lane >> 4;
    const int tile = blockIdx.x * RP_WTILES + pair;
    const bool live = row0 + pair * CTK_MLP_TRAJ_PER_WAVE < a.N;     // wave-uniform: the tile holds at least one plan
    float* tape = scratch + (size_t)tile * H * 64 * RP_MLP_TAPE;
    float* term = scratch + rp_wide_term_off(tiles, H) + (size_t)tile * 64;
#if defined(CTK_DIAG_WIDE_STAMPS)
    unsigned long long st_[8]; int sn_ = 0;
#define WSTAMP() do { st_[sn_++] = wall_clock64(); } while (0)
#else
#define WSTAMP() do {} while (0)
#endif
    WSTAMP();
    // Everything this launch reads from memory was written by OTHER launches (the Jacobian waves ran on every XCD): each
    // dependent round trip costs ~2 us here, so all independent loads are issued before the first wait —
    // weights, previous input, the chain's first records, the Adam moments of this thread's elements, the plans.
    const MlpFwdT wf = mlp_load_fwd_thin(wperm);
    const float uprev0 = uniform_u_prev0(a);
    RpgdChainMlp chain;
    constexpr int AB = 8;                   // Adam elements per thread and pass (H <= 64: one pass)
    float mm0[AB], vv0[AB];
    float bc1 = 1.0f, bc2 = 1.0f;
    const bool chains = update && live && half == 0;
    if (update) {
        if (chains) chain.begin(scratch + rp_wide_jac_off(tiles, H) + (size_t)tile * H * 64 * RP_JAC, term, H);
#pragma unroll
        for (int j = 0; j < AB; ++j) {
            const int i = t + j * RP_BLOCK;
            mm0[j] = 0.0f; vv0[j] = 0.0f;
            if (i < total && ad.rule != 2) { mm0[j] = m[gbase + i]; vv0[j] = v[gbase + i]; }
        }
        if (ti <= bc_len) { bc1 = bc_table[2 * (ti - 1)]; bc2 = bc_table[2 * (ti - 1) + 1]; }
    }
    {   // the plans -> LDS, transposed to [h][plan]; the first AB elements per thread as one batch of loads
        float q0[AB];
#pragma unroll
        for (int j = 0; j < AB; ++j) { const int i = t + j * RP_BLOCK; q0[j] = i < total ? Q[gbase + i] : 0.0f; }
#pragma unroll
        for (int j = 0; j < AB; ++j) {
            const int i = t + j * RP_BLOCK;
            if (i < RP_WTRAJ * H) { const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H; q_s[h * RP_WLD + r] = q0[j]; }
        }
        for (int i = t + AB * RP_BLOCK; i < RP_WTRAJ * H; i += RP_BLOCK) {
            const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;
            q_s[h * RP_WLD + r] = i < total ? Q[gbase + i] : 0.0f;
        }
    }
    __syncthreads();
    WSTAMP();
    if (update) {
#if defined(CTK_DIAG_WIDE_NO_CHAIN)   // timing experiments only (tools/rpgd_split.sh)
        for (int i = t; i < RP_WTRAJ * H; i += RP_BLOCK) g_s[(i >> 5) * RP_WLD + (i & 31)] = 0.01f * uprev0;
#else
        if (chains) chain.run(a, g_s + pair * CTK_MLP_TRAJ_PER_WAVE, RP_WLD);
        else if (half == 0) for (int h = lane >> 4; h < H; h += 4) g_s[h * RP_WLD + pair * CTK_MLP_TRAJ_PER_WAVE + (lane & 15)] = 0.0f;   // plans beyond N
#endif
        WSTAMP();
        __syncthreads();
        rpgd_finish_gradient_w(a, k, ad, q_s, g_s, sc_s, uprev0);
        __syncthreads();
        WSTAMP();
        auto adam_element = [&](int i, float mmv, float vvv) {
            const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;
            const float gg = g_s[h * RP_WLD + r] * sc_s[r];
            const float qn = adam_update(ad, q_s[h * RP_WLD + r], gg, mmv, vvv, bc1, bc2, a.lo[0], a.hi[0]);
            q_s[h * RP_WLD + r] = qn;
            Q[gbase + i] = qn;
            if (ad.rule != 2) { m[gbase + i] = mmv; v[gbase + i] = vvv; }
        };
#if !defined(CTK_DIAG_WIDE_NO_ADAM)
#pragma unroll
        for (int j = 0; j < AB; ++j) {
            const int i = t + j * RP_BLOCK;
            if (i < total) adam_element(i, mm0[j], vv0[j]);
        }
        for (int i = t + AB * RP_BLOCK; i < total; i += RP_BLOCK) {       // H > 64: the rest, one element at a time
            float mmv = 0.0f, vvv = 0.0f;
            if (ad.rule != 2) { mmv = m[gbase + i]; vvv = v[gbase + i]; }
            adam_element(i, mmv, vvv);
        }
#endif
        __syncthreads();
    }
    WSTAMP();
    const int col = pair * CTK_MLP_TRAJ_PER_WAVE + (lane & 15);
    if (!final_pass) {
        // every wave takes part (workgroup barriers inside), also for a tile beyond N (its tape is never read)
        const MlpFwdHalf wh = mlp_half_of(wf, half);
        const float svH = rpgd_forward_mlp_tape_pair(a, wh, q_s, RP_WLD, tape, col, g, half, ex_s + pair * RP_PAIR_EX);
        if (half == 0) term[4 * (lane & 15) + g] = rpgd_mlp_terminal_adjoint(a, k, g, svH);   // quad layout for the chain
        WSTAMP();
#if defined(CTK_DIAG_WIDE_STAMPS)
        if (blockIdx.x == 0 && t == 0 && ti == 7) {
            printf("wide stamps (10 ns ticks since start): n=%d", sn_);
            for (int q = 1; q < sn_; ++q) printf(" %llu", st_[q] - st_[0]);
            printf("\n");
        }
#endif
    } else {
        if (half == 0) {   // one wave per tile for the cost pass (1 of iters + 1 passes)
            const float J = rollout_mlp<false, false>(a, k, wf, row0 + pair * CTK_MLP_TRAJ_PER_WAVE, [&](int h) { return q_s[h * RP_WLD + col]; });
            if (lane < 16 && row0 + col < a.N) a.J[row0 + col] = J;
        }
        if (fw.enabled) rpgd_fused_tail(a, fw, g_s, t, H);
    }
}

// ---- MLP, wide form as ONE launch per MPC step (round 4) ---------------------------------------------------------------------------
// The phase launches above pay, per Adam iteration, a launch boundary, 2.2 us of reloads, the moments' round trip through memory and a
// separate Jacobian launch (6 us for 1.8 us of matrix work).  Here the first `PB` workgroups are the producers of ctk_rpgd_mlp_wide —
// one 16-plan tile each (RP_PT), plans / moments / weights resident for all `iters` iterations — and the rest are Jacobian WORKERS that stay
// for the whole step too:
//   forward pass  : each step's {state component, seq} (wave 0 of the pair) and {input, seq} (wave 1) go through to memory as one 8-byte
//                   word per lane — value and sequence number in one store, so a reader that sees the number has the value (no flag, no
//                   wait on the recurrence; fewer stores than the tape this replaces);
//   worker        : takes the next (iteration, step, tile) ticket (one counter, step-major: any number of resident workers drains the
//                   queue in the order the forward passes produce), polls that step's words, RECOMPUTES the step's activations (0.45 us;
//                   the chip is idle) and forms one tangent per wave (4 state components; wave 3 the input's too), stores the record
//                   through, raises the step's flag = seq;
//   update        : the chain wave of a tile polls its H flags (lane h <- flag h), reads the records past its L2, then gradient finish
//                   and Adam on registers / LDS as before.
// Tiles never depend on each other (optimizer_rpgd.py:325); producers never wait for a worker that holds no ticket; every poll is
// bounded (200 ms; then the error word, NaN records and a skipped update, as in ctk_net_split.hip).  seq = seq0 + iteration is unique per
// launch and iteration (the host advances seq0 by 64 per launch, iters <= 63).
constexpr int RP_PT = 1;                            // tiles per producer workgroup.  The phase launches hold two (four waves); one — two waves, whose
                                                    // per-step barriers then couple nothing else — measured 693 against 706 us per MPC step at cfg4
constexpr int RP_PBLOCK = RP_BLOCK;                // 4 waves, one per SIMD (five — a wave per tangent — put two on one SIMD: 1.4 us per job against 0.96)
constexpr unsigned long long RP_POLL_TICKS = 20000000ull;   // 200 ms of the 100 MHz clock

struct RpgdPersistK {
    uint32_t seq0, ticket_base;
    uint32_t* err_word;      // behind {u, seq} (nullable)
    int producers, live_tiles, withhold;   // withhold: diagnostic (the step whose words iteration 0 never publishes; -1: none)
};

CTK_DEV size_t rp_pers_flags_off(int tiles, int H) { return rp_wide_term_off(tiles, H); }            // [tiles][64] uint32
CTK_DEV size_t rp_pers_ticket_off(int tiles, int H) { return rp_wide_term_off(tiles, H) + (size_t)tiles * 64; }

CTK_DEV void rp_pers_gave_up(uint32_t* err_word, int what, int h, int tile, uint32_t seen, uint32_t want, unsigned long long t_begin) {
    if (err_word == nullptr) return;
    const unsigned long long us = (wall_clock64() - t_begin) / 100u;
    __hip_atomic_store(err_word + 6, ((uint32_t)what << 20) | ((uint32_t)h << 10) | (uint32_t)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word + 7, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word + 8, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word + 9, (uint32_t)(us > 0xffffffffull ? 0xffffffffull : us), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_word, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the forward pass of a pair with its steps published (every wave of the workgroup's first four takes every step: barriers inside)
CTK_DEV float rpgd_forward_mlp_publish_pair(const RolloutArgs& a, const MlpFwdHalf& w_in, const float* q_s, int ld, unsigned long long* pub_tile,
                                            int col, int g, int m, float* ex, uint32_t seq, bool publish, int withhold) {
    const int lane = threadIdx.x & 63;
    MlpFwdHalf w = w_in;
    mlp_pin(w);                                                          // (as `word` below)
    float sv = lane_state4(a, g);
    const int H = a.H;
    float u_next = q_s[col];
    const unsigned long long hi = (unsigned long long)seq << 32;
    // The lane's word index, pinned to a register HERE: a value the register allocator reloads from scratch just before the loop carries
    // that reload's `s_waitcnt vmcnt(0)` to its first use — inside the loop, where the same wait then also waits for the previous step's
    // store (0.43 us per step: seen in this loop, 1156 against 720 us per MPC step; ctk_common.h: lane_state4 tells the same story).
    uint32_t word = (uint32_t)(lane * 2 + m);                            // + h * 128
    asm volatile("" : "+v"(word));
    const bool pub = __builtin_amdgcn_readfirstlane((int)publish) != 0;  // wave-uniform, as a scalar: a branch, not an exec mask, per step
    for (int h = 0; h < H; ++h) {
        const float u = u_next;
        if (h + 1 < H) u_next = q_s[(h + 1) * ld + col];
        if (pub && h != withhold)
            __hip_atomic_store(pub_tile + (uint32_t)(h * 128) + word, hi | (unsigned long long)__builtin_bit_cast(uint32_t, m == 0 ? sv : u),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sv = mlp_step_pair(w, sv, u, m, ex);
    }
    return sv;
}

__global__ __launch_bounds__(RP_PBLOCK) void ctk_rpgd_mlp_persistent(RolloutArgs a, EnvK k, AdamK ad, float* __restrict__ Q, float* __restrict__ m,
                                                                    float* __restrict__ v, const float* __restrict__ bc_table, int bc_len,
                                                                    int t0, int iters, const float* __restrict__ wperm,
                                                                    float* __restrict__ scratch, int tiles, RpgdPersistK pk) {
    extern __shared__ float lds[];
    const int H = a.H;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, c = lane & 15;
    unsigned long long* pub = reinterpret_cast<unsigned long long*>(scratch);                       // [tiles][H][64][2] words
    float* jac = scratch + rp_wide_jac_off(tiles, H);
    uint32_t* flags = reinterpret_cast<uint32_t*>(scratch + rp_pers_flags_off(tiles, H));
    uint32_t* ticket = reinterpret_cast<uint32_t*>(scratch + rp_pers_ticket_off(tiles, H));
    if ((int)blockIdx.x >= pk.producers) {
        // ------------------------------------------------------------------------------------------ a Jacobian worker
        const MlpFwdT w = mlp_load_fwd_thin(wperm);
        uint32_t* slot_s = reinterpret_cast<uint32_t*>(lds);                                         // [2] ticket hand-down
        const int per_it = pk.live_tiles * H;
        const uint32_t total = (uint32_t)per_it * (uint32_t)iters;
        uint32_t pending = 0;
        if (t == 0) pending = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int n = 0;; ++n) {
            if (t == 0) slot_s[n & 1] = pending - pk.ticket_base;
            __syncthreads();
            const uint32_t job = slot_s[n & 1];
            if (job >= total) break;                                                                 // (one ticket past the end per workgroup)
            if (t == 0) pending = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next one, in this job's shadow
            const int it = (int)(job / (uint32_t)per_it), r = (int)job - it * per_it, h = r / pk.live_tiles, tile = r - h * pk.live_tiles;
            const uint32_t seq = pk.seq0 + (uint32_t)it;
            const unsigned long long* p = pub + ((size_t)(tile * H + h) * 64 + lane) * 2;
            unsigned long long w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool got = (uint32_t)(w0 >> 32) == seq && (uint32_t)(w1 >> 32) == seq;
            if (!__all(got)) {
                const unsigned long long t_begin = wall_clock64();
                while (!__all(got)) {
                    if (wall_clock64() - t_begin > RP_POLL_TICKS) break;
                    __builtin_amdgcn_s_sleep(2);
                    w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    got = (uint32_t)(w0 >> 32) == seq && (uint32_t)(w1 >> 32) == seq;
                }
                if (!__all(got) && wave == 0 && lane == 0) rp_pers_gave_up(pk.err_word, 1, h, tile, (uint32_t)(w0 >> 32), seq, t_begin);
            }
            const bool ok = __all(got);
#if defined(CTK_DIAG_PERS_STAMPS)
            const unsigned long long ws0 = wall_clock64();
#endif
            const float sv = __builtin_bit_cast(float, (uint32_t)w0), u = __builtin_bit_cast(float, (uint32_t)w1);
            MlpAct act;
            mlp_acts_as_pair(w, sv, u, &act);                                                        // bit for bit the forward pass's activations
            const f32x4 one = f32x4{1.f, 1.f, 1.f, 1.f};
            const f32x4 D10 = one - act.h1[0] * act.h1[0], D11 = one - act.h1[1] * act.h1[1];
            const f32x4 D20 = one - act.h2[0] * act.h2[0], D21 = one - act.h2[1] * act.h2[1];
            float Jr = rpgd_mlp_tangent(w, D10, D11, D20, D21, wave, g);                             // wave j: column j of the step Jacobian
            float Ju = wave == 3 ? rpgd_mlp_tangent(w, D10, D11, D20, D21, 4, g) : 0.0f;             // ... and wave 3 the input's (the short one)
            if (!ok) { Jr = __builtin_nanf(""); Ju = Jr; }
            // the record of (tile, step): slot 4c + j = {J[0..3][j]}, {J[j][4], share, seq, -} (rpgd_mlp_jacobian_record + the number the chain
            // checks).  The five waves' pieces meet in LDS, and wave 0 stores each slot as two 16-byte stores through to memory (scattered 4-byte
            // stores of that kind cost ~6 x as much per byte), waits for them, and raises the step's flag itself.
            float* rec_s = lds + 16;                                                                 // [64 slots][RP_JAC]
            rec_s[(4 * c + wave) * RP_JAC + g] = Jr;
            if (wave == 3) {
                const float share = ok ? rpgd_mlp_stage_share(a, k, g, sv) : __builtin_nanf("");
                float4* half2 = reinterpret_cast<float4*>(rec_s + (4 * c + g) * RP_JAC + 4);
                *half2 = make_float4(Ju, share, __builtin_bit_cast(float, seq), 0.0f);
            }
            __syncthreads();
            if (wave == 0) {
                const float4* src = reinterpret_cast<const float4*>(rec_s + lane * RP_JAC);
                float4* dst = reinterpret_cast<float4*>(jac + ((size_t)(tile * H + h) * 64 + lane) * RP_JAC);
                const float4 c0 = src[0], c1 = src[1];
                st4_through(dst, c0);
                st4_through(dst + 1, c1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     // the record has reached memory
                if (lane == 0) __hip_atomic_store(flags + tile * 64 + h, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#if defined(CTK_DIAG_PERS_STAMPS)
            if (it == 7 && tile == 0 && t == 0 && (h == H - 1 || h == H - 2 || h == H / 2 || h == 0))
                printf("pers worker (it 7, tile 0, step %d, job %u of this workgroup %d): words seen abs %llu, flag stored +%llu\n", h, job, n, ws0, wall_clock64() - ws0);
#endif
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------- a producer (RP_PT tiles)
    if (wave >= 2 * RP_PT) return;
    float* q_s = lds;                                   // [H][33]
    float* g_s = q_s + H * RP_WLD;                      // [H][33]
    float* sc_s = g_s + max(H * RP_WLD, 128);           // [32]
    float* ex_s = sc_s + RP_WTRAJ;                      // [2 pairs][RP_PAIR_EX]
    float* term_s = ex_s + 2 * RP_PAIR_EX;              // [2 pairs][64]
    float* m_s = term_s + 128;                          // [H][33] Adam moments, resident for the step (registers would spill around the chain)
    float* v_s = m_s + H * RP_WLD;                      // [H][33]
    const int pair = wave >> 1, half = wave & 1;
    const int row0 = blockIdx.x * (16 * RP_PT);
    const int rows = min(16 * RP_PT, a.N - row0);
    const int total = rows * H;
    const size_t gbase = (size_t)row0 * H;
    const int tile = blockIdx.x * RP_PT + pair;
    const bool live = row0 + pair * CTK_MLP_TRAJ_PER_WAVE < a.N;     // wave-uniform: the tile holds at least one plan
    const int col = pair * CTK_MLP_TRAJ_PER_WAVE + c;
    const float uprev0 = uniform_u_prev0(a);
    constexpr int AB = 8;                               // elements per thread (H <= 64: all of them in one batch)
    {
        float q0[AB], mm0[AB], vv0[AB];
#pragma unroll
        for (int j = 0; j < AB; ++j) {
            const int i = t + j * (128 * RP_PT);
            q0[j] = i < total ? Q[gbase + i] : 0.0f;
            mm0[j] = 0.0f; vv0[j] = 0.0f;
            if (i < total && ad.rule != 2) { mm0[j] = m[gbase + i]; vv0[j] = v[gbase + i]; }
        }
#pragma unroll
        for (int j = 0; j < AB; ++j) {
            const int i = t + j * (128 * RP_PT);
            if (i < 16 * RP_PT * H) {
                const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;
                q_s[h * RP_WLD + r] = q0[j]; m_s[h * RP_WLD + r] = mm0[j]; v_s[h * RP_WLD + r] = vv0[j];
            }
        }
    }
    MlpFwdHalf wh;
    { const MlpFwdT wf = mlp_load_fwd_thin(wperm); wh = mlp_half_of(wf, half); }
    __syncthreads();
    const bool chains = live && half == 0;
#if defined(CTK_DIAG_PERS_STAMPS)      // tools/rpgd_stamps.sh: where one iteration's time goes (block 0, thread 0; a worker's side below)
    unsigned long long ps_[8]; int pn_ = 0;
#define PSTAMP() do { if (it == 7 && pn_ < 8) ps_[pn_++] = wall_clock64(); } while (0)
#else
#define PSTAMP() do {} while (0)
#endif
    for (int it = 0; it < iters; ++it) {
        const uint32_t seq = pk.seq0 + (uint32_t)it;
        PSTAMP();
        const int ti = t0 + it + 1;                                   // state['step'] += 1 (:59); the two scalars arrive in the forward pass's shadow
        float bc1 = 1.0f, bc2 = 1.0f;
        if (ti <= bc_len) { bc1 = bc_table[2 * (ti - 1)]; bc2 = bc_table[2 * (ti - 1) + 1]; }
        const float svH = rpgd_forward_mlp_publish_pair(a, wh, q_s, RP_WLD, pub + (size_t)tile * H * 128, col, g, half, ex_s + pair * RP_PAIR_EX, seq,
                                                        live, it == 0 ? pk.withhold : -1);
        // ---- the update of this iteration: flags of the tile's H records, chain, gradient finish, Adam
        PSTAMP();
        if (chains) {
            term_s[pair * 64 + 4 * c + g] = rpgd_mlp_terminal_adjoint(a, k, g, svH);                 // quad layout for the chain (same wave reads it)
            const uint32_t* fl = flags + tile * 64;
            uint32_t f = lane < H ? __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
            if (!__all(f == seq)) {
                const unsigned long long t_begin = wall_clock64();
                while (!__all(f == seq)) {
                    if (wall_clock64() - t_begin > 2 * RP_POLL_TICKS) break;            // (a worker that gave up still raises its flag: twice its budget)
                    __builtin_amdgcn_s_sleep(1);
                    f = lane < H ? __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
                }
                if (!__all(f == seq)) {
                    const unsigned long long missing = __builtin_amdgcn_ballot_w64(f != seq);
                    if (lane == (int)__builtin_ctzll(missing)) rp_pers_gave_up(pk.err_word, 2, lane, tile, f, seq, t_begin);
                }
            }
            PSTAMP();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                       // flags, then the records they publish
            // The guide's consumer form: one poll, one agent-scope acquire, then PLAIN loads by the acquiring wave (the workers stored every
            // byte through, each wave waited for its stores, a barrier, then the flag).  Loads that bypass the L2 instead cost 2 us per
            // dependent chunk here (15 us per chain against 2.8).  Belt and braces: each record's own sequence number is checked as it is
            // consumed, and a chain that met an older copy is run again past the caches.
            const float* jt = jac + (size_t)tile * H * 64 * RP_JAC;
            RpgdChainMlpT<false, true> chain;
            chain.want = seq;
            chain.begin(jt, term_s[pair * 64 + lane], H);
            chain.run(a, g_s + pair * CTK_MLP_TRAJ_PER_WAVE, RP_WLD);
            if (__any(chain.stale != 0)) {
                RpgdChainMlpT<true, true> again;
                again.want = seq;
                again.begin(jt, term_s[pair * 64 + lane], H);
                again.run(a, g_s + pair * CTK_MLP_TRAJ_PER_WAVE, RP_WLD);
                const bool still = __any(again.stale != 0);
                if (lane == 0) {
                    __hip_atomic_fetch_add(ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // (how often: word 1 behind the ticket counter)
                    if (still) rp_pers_gave_up(pk.err_word, 3, 0, tile, 0u, seq, wall_clock64());
#if defined(CTK_DIAG_PERS_STAMPS)
                    printf("pers chain of tile %d, iteration %d: a record older than its flag, chain run again past the caches (still stale: %d)\n", tile, it, (int)still);
#endif
                }
            }
            PSTAMP();
        } else if (half == 0) {
            for (int h = lane >> 4; h < H; h += 4) g_s[h * RP_WLD + pair * CTK_MLP_TRAJ_PER_WAVE + c] = 0.0f;   // plans beyond N
        }
        __syncthreads();
        // gradient finish (rpgd_finish_gradient_w: the input-cost terms, the per-plan norm over 8 lanes by DPP) and Adam, fused: thread
        // (plan r, eighth `part`) owns steps part, part + 8, ... of its plan in BOTH — the gradients stay in registers, the clip scale needs no
        // LDS, and one barrier (the vote below: every neighbour's plan value has been read before any is rewritten) replaces three.
        {
            const int r = t >> 3, part = t & 7;
            const float inv = a.inv_Hp1;
            float gq[AB];
            float nrm2 = 0.0f;
#pragma unroll
            for (int j = 0; j < AB; ++j) {
                const int h = part + 8 * j;
                gq[j] = 0.0f;
                if (h < H) {
                    const float u_h = q_s[h * RP_WLD + r], u_hm1 = h > 0 ? q_s[(h - 1) * RP_WLD + r] : uprev0;
                    float gu = 2.0f * k.ccR * u_h + 2.0f * k.ccrc_weight * (u_h - u_hm1);
                    if (h + 1 < H) gu -= 2.0f * k.ccrc_weight * (q_s[(h + 1) * RP_WLD + r] - u_h);
                    gq[j] = gu * inv + g_s[h * RP_WLD + r];
                    nrm2 += gq[j] * gq[j];
                }
            }
            nrm2 += dpp_mov<DPP_QUAD_XOR1>(nrm2);
            nrm2 += dpp_mov<DPP_QUAD_XOR2>(nrm2);
            nrm2 += dpp_mov<DPP_ROW_HALF_MIRROR>(nrm2);           // every lane of the plan's 8 holds the total
            const float scl = ad.clip / fmaxf(sqrtf(nrm2), ad.clip);   // lib.clip_by_norm(g, clip, [1,2]) (:315,:334)
            PSTAMP();
            // a gradient that is not finite (a record that never arrived is NaN; so is a rollout that diverged): the workgroup keeps its plans
            // and moments for this iteration, and the step reports CTK_ERR_STATE (h_u word 3: the tile)
            const int bad = __syncthreads_or(!(nrm2 <= 3.0e38f));
            if (bad && t == 0 && pk.err_word != nullptr) __hip_atomic_store(pk.err_word + 1, 1u + (uint32_t)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (!bad && r < rows) {
#pragma unroll
                for (int j = 0; j < AB; ++j) {
                    const int h = part + 8 * j;
                    if (h < H) {   // (reading the moments in one batch ahead of the vote was 1 % faster and no longer bit-identical to the phase launches)
                        float mm = m_s[h * RP_WLD + r], vv = v_s[h * RP_WLD + r];
                        q_s[h * RP_WLD + r] = adam_update(ad, q_s[h * RP_WLD + r], gq[j] * scl, mm, vv, bc1, bc2, a.lo[0], a.hi[0]);
                        m_s[h * RP_WLD + r] = mm; v_s[h * RP_WLD + r] = vv;
                    }
                }
            }
        }
        __syncthreads();
        PSTAMP();
#if defined(CTK_DIAG_PERS_STAMPS)
        if (it == 7 && blockIdx.x == 0 && t == 0) {
            printf("pers producer stamps (10 ns ticks since the iteration's start): forward end %u, flags seen %u, chain end %u, finish end %u, Adam end %u (n = %d)\n",
                   (unsigned)(ps_[1] - ps_[0]), (unsigned)(ps_[2] - ps_[0]), (unsigned)(ps_[3] - ps_[0]), (unsigned)(ps_[4] - ps_[0]),
                   (unsigned)(ps_[5] - ps_[0]), pn_);
        }
#endif
    }
    // get_action's cost pass (:342), then the plans and moments back to memory
    if (half == 0) {   // (one wave per tile, as the phase launches' cost pass: the two forms agree bit for bit; the pair form measured no faster, twice)
        const MlpFwdT wf = mlp_load_fwd_thin(wperm);
        const float J = rollout_mlp<false, false>(a, k, wf, row0 + pair * CTK_MLP_TRAJ_PER_WAVE, [&](int h) { return q_s[h * RP_WLD + col]; });
        if (lane < 16 && row0 + col < a.N) a.J[row0 + col] = J;
    }
#pragma unroll
    for (int j = 0; j < AB; ++j) {
        const int i = t + j * (128 * RP_PT);
        if (i < total) {
            const int r = H >= 2 ? (int)__umulhi((uint32_t)i, a.p_magic) : i, h = i - r * H;
            Q[gbase + i] = q_s[h * RP_WLD + r];
            if (ad.rule != 2) { m[gbase + i] = m_s[h * RP_WLD + r]; v[gbase + i] = v_s[h * RP_WLD + r]; }
        }
    }
}


__global__ __launch_bounds__(256) void ctk_rpgd_pack_keepers(const float* __restrict__ J, const float* __restrict__ Q,
                                                             const float* __restrict__ m, const float* __restrict__ v,
                                                             const float* __restrict__ ages, const int* __restrict__ idx, int K,
                                                             int H, int global_offset, float* __restrict__ out) {
    const int rs = 3 + 3 * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K * rs; i += gridDim.x * blockDim.x) {
        const int kk = i / rs, f = i - kk * rs, src = idx[kk];
        float val;
        if (f == 0) val = J[src];
        else if (f == 1) val = __builtin_bit_cast(float, global_offset + src);
        else if (f == 2) val = ages[src];
        else if (f < 3 + H) val = Q[(size_t)src * H + (f - 3)];
        else if (f < 3 + 2 * H) val = m[(size_t)src * H + (f - 3 - H)];
        else val = v[(size_t)src * H + (f - 3 - 2 * H)];
        out[i] = val;
    }
}

hipError_t ctk_launch_rpgd_pack_keepers(hipStream_t st, const float* J, const float* Q, const float* m, const float* v,
                                        const float* ages, const int* idx, int K, int H, int global_offset, float* out) {
    const int total = K * (3 + 3 * H);
    hipLaunchKernelGGL(ctk_rpgd_pack_keepers, dim3((total + 255) / 256 > 128 ? 128 : (total + 255) / 256), dim3(256), 0, st, J, Q, m, v,
                       ages, idx, K, H, global_offset, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MLP populations up to this size take the wide form (phase launches + grid-wide Jacobians): per iteration
// ~45 us + 6.5 ns per plan against ~87 us of the single-launch form (both measured at cfg4, tools/rpgd_split.sh)
bool ctk_rpgd_uses_wide(int pred, int N) {
    static const bool narrow = getenv("CTK_RPGD_NARROW") != nullptr;   // diagnostic switch: A/B the two forms
    return pred == CTK_PRED_MLP && N <= CTK_RPGD_WIDE_MAX_N && !narrow;
}

// ... and of those, more than one workgroup's worth of plans with H <= 64 runs the whole descent as ONE launch (ctk_rpgd_mlp_persistent)
bool ctk_rpgd_uses_persistent(int pred, int N, int H) {
    static const bool off = getenv("CTK_RPGD_NO_PERSISTENT") != nullptr;   // diagnostic switch: A/B the two forms
    // up to 32 tiles: beyond that the (at most 240) resident workers no longer keep up with the forward passes — a job is ~1.8 us of a
    // workgroup, the tiles produce one each per 0.47 us — and the grid-wide Jacobian launches of the phase form win
    return ctk_rpgd_uses_wide(pred, N) && N > RP_WTRAJ && N <= 512 && H <= 64 && !off;
}

const char* ctk_rpgd_descent_name(int pred, int N, int H) {
    if (H > 0 && ctk_rpgd_uses_persistent(pred, N, H)) return "ctk_rpgd_mlp_persistent";
    if (ctk_rpgd_uses_wide(pred, N)) return "ctk_rpgd_mlp_wide + ctk_rpgd_mlp_jacobians";
    return pred == CTK_PRED_ODE ? "ctk_rpgd_descent<0>" : "ctk_rpgd_descent<1>";
}

size_t ctk_rpgd_descent_lds(int pred, int H, bool* tape_in_lds) {
    const size_t base = (size_t)(2 * H * RP_LD + RP_TRAJ) * sizeof(float);
    const size_t tape = pred == CTK_PRED_ODE ? (size_t)H * RP_NS * 64 * sizeof(float) : 0;
    const bool fits = base + tape <= 160 * 1024;
    if (tape_in_lds) *tape_in_lds = fits && tape > 0;
    return fits ? base + tape : base;
}

static int wide_blocks(int N) { return (N + RP_WTRAJ - 1) / RP_WTRAJ; }

size_t ctk_rpgd_scratch_floats(int pred, int N, int H) {
    const size_t blocks = (N + RP_TRAJ - 1) / RP_TRAJ;
    if (pred == CTK_PRED_ODE) return blocks * H * RP_NS * 64;
    // single-launch form: tape per 16-plan tile; wide form: tape | Jacobian records | terminal adjoints per tile
    const size_t tiles = std::max(blocks * RP_WAVES, (size_t)wide_blocks(N) * RP_WTILES);
    return tiles * (size_t)H * 64 * (RP_MLP_TAPE + RP_JAC) + tiles * 64 + 16;   // (+ the persistent form's ticket counter)
}

// largest population whose step runs as ONE launch with the keep-k / warm-start tail (one workgroup holds it)
int ctk_rpgd_fused_max_n(int pred, int N) { return ctk_rpgd_uses_wide(pred, N) ? RP_WTRAJ : CTK_RPGD_FUSED_MAX_N; }

static size_t wide_lds(int H) { return (size_t)(H * RP_WLD + std::max(H * RP_WLD, 128) + RP_WTRAJ + 2 * RP_PAIR_EX) * sizeof(float); }

template <class K, class... Args>
static void launch_timed(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, hipEvent_t e0, hipEvent_t e1, Args... args) {
    if (e0 || e1) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

hipError_t ctk_launch_rpgd_descent(hipStream_t st, int pred, const RolloutArgs& a, const EnvK& k, float lr, float b1, float b2,
                                   float eps, float clip, float* Q, float* m, float* v, const float* bc_table, int bc_len,
                                   int t0, int iters, const float* wperm, float* scratch, hipEvent_t e0, hipEvent_t e1, int rule,
                                   const RpgdFusedWarm* fused, RpgdPersist* pers) {
    AdamK ad{lr, b1, b2, (float)(1.0 - (double)b1), (float)(1.0 - (double)b2), eps, clip, rule};
    bool tape_in_lds = false;
    const size_t lds = ctk_rpgd_descent_lds(pred, a.H, &tape_in_lds);
    const dim3 grid((a.N + RP_TRAJ - 1) / RP_TRAJ), block(RP_BLOCK);
    auto fw_of = [&](const RpgdFusedWarm* f) {
        FusedWarm x{};
        x.enabled = 1; x.K = f->K; x.idx_out = f->idx_out;
        x.w = WarmArgs{a.N, a.H, f->P, f->n_new, f->gather, f->shift_previous, f->sampling_distribution, 0,
                       f->sample_stdev, f->sample_mean, f->sample_min, f->sample_max, f->whole_space, nullptr, 3 + 3 * a.H, 0, f->fresh_tail};
        x.p = WarmPtrs{f->draws, nullptr, Q, m, v, f->ages_old, f->Q_new, f->m_new, f->v_new, f->ages_new,
                       f->interp, f->u_nom, f->u_dev, f->u_host, f->seq};
        return x;
    };
    FusedWarm fw{};
    if (fused && grid.x == 1) fw = fw_of(fused);
    if (pers != nullptr && iters >= 1 && iters <= 63 && ctk_rpgd_uses_persistent(pred, a.N, a.H)) {
        // ONE launch: producers (two tiles each) + Jacobian workers that stay for all iterations (ctk_rpgd_mlp_persistent)
        static const int diag_step = getenv("CTK_DIAG_RPGD_WITHHOLD_FLAG") ? atoi(getenv("CTK_DIAG_RPGD_WITHHOLD_FLAG")) : -1;
        static std::atomic<int> diag_armed{diag_step >= 0 ? 1 : 0};
        const int PB = (a.N + 16 * RP_PT - 1) / (16 * RP_PT), tiles = PB * RP_PT, live = (a.N + CTK_MLP_TRAJ_PER_WAVE - 1) / CTK_MLP_TRAJ_PER_WAVE;
        const int per_it = live * a.H, W = std::min(240, per_it);
        if (pers->seq0 < 64u || pers->seq0 > 0xffffff00u) pers->seq0 = 64u;
        RpgdPersistK pk{pers->seq0, pers->ticket_base, pers->err_word, PB, live, diag_armed.exchange(0) ? diag_step : -1};
        pers->seq0 += 64u;
        pers->ticket_base += (uint32_t)per_it * (uint32_t)iters + (uint32_t)W;       // every job + one ticket past the end per worker workgroup
        // more than half a CU's LDS: one workgroup per CU — no worker on a producer's SIMDs
        const size_t lds = std::max(wide_lds(a.H) + (128 + 2 * a.H * RP_WLD) * sizeof(float), (size_t)84 * 1024);
        launch_timed(ctk_rpgd_mlp_persistent, dim3(PB + W), dim3(RP_PBLOCK), lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm,
                     scratch, tiles, pk);
        return hipGetLastError();
    }
    if (ctk_rpgd_uses_wide(pred, a.N)) {
        // phase launches: [tape] J [update+tape] J ... [update+final]; the event pair brackets the whole sequence
        const dim3 wgrid(wide_blocks(a.N));
        const int tiles = (int)wgrid.x * RP_WTILES, live = (a.N + CTK_MLP_TRAJ_PER_WAVE - 1) / CTK_MLP_TRAJ_PER_WAVE, jobs = live * a.H;
        FusedWarm wfw{};
        if (fused && wgrid.x == 1) wfw = fw_of(fused);
        const FusedWarm none{};
        for (int it = 0; it <= iters; ++it) {
            const bool last = it == iters;
            launch_timed(ctk_rpgd_mlp_wide, wgrid, block, wide_lds(a.H), st, it == 0 ? e0 : nullptr, last ? e1 : nullptr, a, k, ad, Q, m, v, bc_table,
                         bc_len, t0 + it, wperm, scratch, tiles, it > 0 ? 1 : 0, last ? 1 : 0, last ? wfw : none);
            if (!last)
                hipLaunchKernelGGL(ctk_rpgd_mlp_jacobians, dim3((jobs + RP_WAVES - 1) / RP_WAVES), block, 0, st, a, k, wperm, scratch, tiles, jobs);
        }
        return hipGetLastError();
    }
    if (pred == CTK_PRED_ODE)
        CTK_LAUNCH((ctk_rpgd_descent<CTK_PRED_ODE>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm,
                   scratch, tape_in_lds ? 1 : 0, fw);
    else
        CTK_LAUNCH((ctk_rpgd_descent<CTK_PRED_MLP>), grid, block, lds, st, e0, e1, a, k, ad, Q, m, v, bc_table, bc_len, t0, iters, wperm,
                   scratch, 0, fw);
    return hipGetLastError();
}

hipError_t ctk_launch_rpgd_warmstart(hipStream_t st, const RolloutArgs& a, int N, int H, int P, int n_new, int gather, int shift_previous,
                                     int sampling_distribution, int reset, int whole_space, float sample_stdev,
                                     float sample_mean, float sample_min, float sample_max, const float* draws, const int* idx,
                                     const float* Q_old, const float* m_old, const float* v_old, const float* ages_old,
                                     float* Q_new, float* m_new, float* v_new, float* ages_new, const InterpEntry* interp,
                                     float* u_nom, float* u_dev, float* u_host, uint32_t seq, const float* recs, int rs,
                                     int keeper_base, int fresh_tail) {
    WarmArgs w{N, H, P, n_new, gather, shift_previous, sampling_distribution, reset, sample_stdev, sample_mean, sample_min, sample_max,
               whole_space, recs, rs, keeper_base, fresh_tail};
    const int total = N * H * a.C;
    const WarmPtrs p{draws, idx, Q_old, m_old, v_old, ages_old, Q_new, m_new, v_new, ages_new, interp, u_nom, u_dev, u_host, seq};
    hipLaunchKernelGGL(ctk_rpgd_warmstart, dim3((total + 255) / 256), dim3(256), 0, st, w, a, p);
    return hipGetLastError();
}
