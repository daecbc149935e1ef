// ctk_env.h — the environment interface of the optimizer kernels.
//
// What the reference selects at run time — the plant model behind predictor.predict_core (external SI_Toolkit,
// controller_mpc.py:67-73) and the concrete cost class Control_Toolkit_ASF.Cost_Functions.<environment>.<name>
// (cost_function_wrapper.py:59-66) — is here ONE compile-time policy per environment:
//
//   template <> struct Env<ID> {
//       static constexpr int S, C;                    num_states, num_control_inputs
//       struct / using K;                             derived fp32 constants (host: derive(), double -> fp32 once)
//       step(k, s, u)                                 one predictor step (intermediate_steps Euler sub-steps), in place
//       stage_cost(k, s, u, u_prev) / terminal_cost   the concrete cost terms (aggregation = mean over H+1 is the
//                                                     kernels' job, Cost_Functions/__init__.py:90-93)
//       step_vjp / stage_grad_state / terminal_grad / input_grad     reverse mode for the gradient-based optimizers
//   };
//
// The template kernels of ctk_generic.hip (MPPI / affine-sampled rollouts / RPGD descent) are written against this
// interface only; adding an environment = one more specialisation + one line in CTK_FOR_ENV.  CartPole additionally
// keeps its hand-tuned kernels (ctk_mppi.hip, ctk_sampled.hip, ctk_rpgd.hip); its specialisation here wraps the same
// device functions, so "generic CartPole" and "tuned CartPole" are the same arithmetic (tests/test_gpu_env.py).
#pragma once
#include "ctk_device.h"

template <int ENV>
struct Env;

// host-side description shared by ctk_api.hip (ctk_env_info / ctk_param_name)
struct EnvInfo {
    const char* name;
    int S, C, n_params;
    const char* const* param_names;
    const float* param_defaults;
};

// ---------------------------------------------------------------------------------------------------------------
// CartPole (oracle/ctk_oracle.py: EnvParams, Predictor._ode_step/_ode_vjp, Cost)
// ---------------------------------------------------------------------------------------------------------------
template <>
struct Env<CTK_ENV_CARTPOLE> {
    static constexpr int S = 4, C = 1;
    using K = EnvK;
    static K derive(const float* p, float dt, int isteps) { return derive_constants(p, dt, isteps); }

    CTK_DEV static void step(const K& k, float (&s)[S], const float (&u)[C]) {
        State4 st{s[0], s[1], s[2], s[3]};
        float sn, cs;
        ctk_sincosf(st.th, &sn, &cs);
        ode_step(k, st, u[0], sn, cs);
        s[0] = st.x; s[1] = st.v; s[2] = st.th; s[3] = st.om;
    }
    CTK_DEV static float stage_cost(const K& k, const float (&s)[S], const float (&u)[C], const float (&up)[C]) {
        const State4 st{s[0], s[1], s[2], s[3]};
        return ::stage_cost(k, st, cosf(s[2]), u[0], up[0]);
    }
    CTK_DEV static float terminal_cost(const K& k, const float (&s)[S]) {
        return ::terminal_cost(k, State4{s[0], s[1], s[2], s[3]});
    }
    // ---- hooks of the 4-wave rollout kernels (ctk_mppi.hip): the stage cost splits into a state part and an input-only part
    //      (summed off the recurrence); the recurrence consumes the inputs in the form prep_input() stored them ----------------
    static constexpr bool SEPARABLE = true;
    CTK_DEV static float input_cost(const K& k, const float (&u)[C], const float (&up)[C]) { return stage_cost_input(k, u[0], up[0]); }
    CTK_DEV static float prep_input(const K& k, float u, int /*c*/) { return k.u_max * u; }   // the force
    CTK_DEV static bool fast_ok(const K& k) { return k.intermediate_steps == 1; }
    CTK_DEV static bool out_of_range(float amax) { return !(amax <= CTK_SINCOS_FAST_LIMIT); }
    // stage cost of the state (dd + ep + ekp), then s <- step(s, f).  FAST: unchecked sincos shared by cost and dynamics, one
    // Euler sub-step, max |angle| tracked in amax (the caller re-runs an out-of-range wave with FAST = false)
    template <bool FAST>
    CTK_DEV static void cost_step(const K& k, float (&s)[S], const float (&f)[C], float& csum, float& amax) {
        State4 st{s[0], s[1], s[2], s[3]};
        float sn, cs;
        if constexpr (FAST) { ctk_sincosf_fast(st.th, &sn, &cs); amax = fmaxf(amax, fabsf(st.th)); }
        else ctk_sincosf(st.th, &sn, &cs);
        ode_cost_substep(k, st, f[0], sn, cs, csum);
        if constexpr (!FAST) {
            for (int i = 1; i < k.intermediate_steps; ++i) {
                float sn2, cs2;
                ctk_sincosf(st.th, &sn2, &cs2);
                ode_substep(k, st, f[0], sn2, cs2);
            }
        }
        s[0] = st.x; s[1] = st.v; s[2] = st.th; s[3] = st.om;
    }
    // ---- hooks of the descent kernels (ctk_generic.hip: ctk_g_rpgd_descent): the forward pass of a gradient iteration needs
    //      no cost, only what the reverse sweep will use — NT taped values per step, so that the sweep recomputes nothing
    //      (one Euler sub-step, as step_vjp).  tape: x, omega, sin, cos, tmp, thdd ----------------------------------------------
    static constexpr int NT = 6;
    CTK_DEV static void fwd_tape(const K& k, float (&s)[S], const float (&u)[C], float (&tp)[NT]) {
        float sn, cs;
        ctk_sincosf(s[2], &sn, &cs);
        const float F = k.u_max * u[0];
        const float A = F + k.k_ml * s[3] * s[3] * sn - k.M_fric * s[1];
        const float tmp = A * k.inv_mt;
        const float D = k.k43l - k.k_mpl_mt * cs * cs;
        const float Nn = k.g * sn - cs * tmp - k.k_jf * s[3];
        const float thdd = fdiv_pos(Nn, D);
        const float xdd = tmp - k.k_mpl_mt * thdd * cs;
        tp[0] = s[0]; tp[1] = s[3]; tp[2] = sn; tp[3] = cs; tp[4] = tmp; tp[5] = thdd;
        const float nx = s[0] + k.dt * s[1], nv = s[1] + k.dt * xdd, nth = s[2] + k.dt * s[3], nom = s[3] + k.dt * thdd;
        s[0] = nx; s[1] = nv; s[2] = nth; s[3] = nom;
    }
    // lam: dL/ds_{h+1} in, dL/ds_h out (= stage-cost state gradient * inv + J^T lam); du: dL/du_h through the dynamics
    CTK_DEV static void bwd_tape(const K& k, const float (&tp)[NT], const float (&)[C], float (&lam)[S], float (&du)[C], float inv) {
        const float x = tp[0], om = tp[1], sn = tp[2], cs = tp[3], tmp = tp[4], thdd = tp[5];
        const float lx = lam[0], lv = lam[1], lth = lam[2], lom = lam[3];
        const float two_dd = 2.0f * k.dd_weight * k.inv_xs * k.inv_xs;
        const float D = k.k43l - k.k_mpl_mt * cs * cs;
        const float dt = k.dt;
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
        du[0] = k.u_max * a_A;
        lam[0] = two_dd * (x - k.target_position) * inv + o_x;
        lam[1] = o_v;
        lam[2] = 2.0f * k.ep_c * (1.0f - cs) * sn * inv + o_th;
        lam[3] = 2.0f * k.ekp_weight * om * inv + o_om;
    }
    // adjoint of one Euler step (intermediate_steps == 1): lam = dL/ds' -> ds = dL/ds, du = dL/du
    CTK_DEV static void step_vjp(const K& k, const float (&s)[S], const float (&u)[C], const float (&lam)[S], float (&ds)[S],
                                 float (&du)[C]) {
        const float v = s[1], om = s[3];
        float sn, cs;
        ctk_sincosf(s[2], &sn, &cs);
        const float A = k.u_max * u[0] + k.k_ml * om * om * sn - k.M_fric * v;
        const float tmp = A * k.inv_mt;
        const float D = k.k43l - k.k_mpl_mt * cs * cs;
        const float Nn = k.g * sn - cs * tmp - k.k_jf * om;
        const float thdd = fdiv_pos(Nn, D);
        const float dt = k.dt;
        const float a_xdd = dt * lam[1];
        const float a_thdd = dt * lam[3] - k.k_mpl_mt * cs * a_xdd;
        float a_tmp = a_xdd;
        float a_cs = -k.k_mpl_mt * thdd * a_xdd;
        const float a_Nn = fdiv_pos(a_thdd, D);
        const float a_D = -a_Nn * thdd;
        float a_sn = k.g * a_Nn;
        a_cs = a_cs - tmp * a_Nn - 2.0f * k.k_mpl_mt * cs * a_D;
        a_tmp = a_tmp - cs * a_Nn;
        const float a_A = a_tmp * k.inv_mt;
        a_sn = a_sn + k.k_ml * om * om * a_A;
        ds[0] = lam[0];
        ds[1] = lam[1] + dt * lam[0] - k.M_fric * a_A;
        ds[2] = lam[2] + cs * a_sn - sn * a_cs;
        ds[3] = lam[3] + dt * lam[2] - k.k_jf * a_Nn + 2.0f * k.k_ml * om * sn * a_A;
        du[0] = k.u_max * a_A;
    }
    CTK_DEV static void stage_grad_state(const K& k, const float (&s)[S], float (&g)[S]) {
        float sn, cs;
        ctk_sincosf(s[2], &sn, &cs);
        g[0] = 2.0f * k.dd_weight * k.inv_xs * k.inv_xs * (s[0] - k.target_position);
        g[1] = 0.0f;
        g[2] = 2.0f * k.ep_c * (1.0f - cs) * sn;
        g[3] = 2.0f * k.ekp_weight * s[3];
    }
    CTK_DEV static void terminal_grad(const K& k, const float (&s)[S], float (&g)[S]) {
        float sn, cs;
        ctk_sincosf(s[2], &sn, &cs);
        g[0] = k.terminal_weight * 2.0f * k.dd_weight * k.inv_xs * k.inv_xs * (s[0] - k.target_position);
        g[1] = 0.0f;
        g[2] = k.terminal_weight * 2.0f * k.ep_c * (1.0f - cs) * sn;
        g[3] = 0.0f;
    }
    // stage cost's input-only terms cc + ccrc: gradient w.r.t. this step's input (gu) and the previous one (gp)
    CTK_DEV static void input_grad(const K& k, const float (&u)[C], const float (&up)[C], float (&gu)[C], float (&gp)[C]) {
        const float d = 2.0f * k.ccrc_weight * (u[0] - up[0]);
        gu[0] = 2.0f * k.ccR * u[0] + d;
        gp[0] = -d;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Quad2D — planar quadrotor (oracle/ctk_oracle.py: Quad2DParams, Predictor._quad_step/_quad_vjp, Cost._quad_*)
// state (x, vx, z, vz, theta, omega); inputs (u1, u2) in [-1, 1]; rotor thrust T_i = (m g / 2)(1 + thrust_gain u_i)
// ---------------------------------------------------------------------------------------------------------------
struct QuadK {
    float dt, g, kF, kM, c_v, c_w;                                                       // dynamics
    float tx, tz, pos_c, ang_w, vel_w, angvel_w, ccR, ccrc_weight, terminal_weight;      // cost
    int intermediate_steps;
};

template <>
struct Env<CTK_ENV_QUAD2D> {
    static constexpr int S = 6, C = 2;
    using K = QuadK;
    static K derive(const float* p, float dt, int isteps) {
        auto d = [&](int id) { return (double)p[id]; };
        K k;
        k.dt = (float)((double)dt / isteps);
        k.g = (float)d(CTK_Q_G);
        k.kF = (float)(0.5 * d(CTK_Q_G) * d(CTK_Q_THRUST_GAIN));
        k.kM = (float)(d(CTK_Q_ARM) * 0.5 * d(CTK_Q_MASS) * d(CTK_Q_G) * d(CTK_Q_THRUST_GAIN) / d(CTK_Q_INERTIA));
        k.c_v = (float)d(CTK_Q_DRAG_LIN);
        k.c_w = (float)d(CTK_Q_DRAG_ANG);
        k.tx = p[CTK_Q_TARGET_X]; k.tz = p[CTK_Q_TARGET_Z];
        k.pos_c = (float)(d(CTK_Q_POS_WEIGHT) / (d(CTK_Q_POS_SCALE) * d(CTK_Q_POS_SCALE)));
        k.ang_w = p[CTK_Q_ANG_WEIGHT]; k.vel_w = p[CTK_Q_VEL_WEIGHT]; k.angvel_w = p[CTK_Q_ANGVEL_WEIGHT];
        k.ccR = (float)(d(CTK_Q_CC_WEIGHT) * d(CTK_Q_R));
        k.ccrc_weight = p[CTK_Q_CCRC_WEIGHT];
        k.terminal_weight = p[CTK_Q_TERMINAL_WEIGHT];
        k.intermediate_steps = isteps;
        return k;
    }

    CTK_DEV static void step(const K& k, float (&s)[S], const float (&u)[C]) {
        const float aF = k.g + k.kF * (u[0] + u[1]);      // total thrust / mass
        const float aM = k.kM * (u[0] - u[1]);            // rotor torque / inertia
        for (int i = 0; i < k.intermediate_steps; ++i) {
            float sn, cs;
            ctk_sincosf(s[4], &sn, &cs);
            const float ax = -aF * sn - k.c_v * s[1];
            const float az = aF * cs - k.g - k.c_v * s[3];
            const float al = aM - k.c_w * s[5];
            const float nx = s[0] + k.dt * s[1], nvx = s[1] + k.dt * ax, nz = s[2] + k.dt * s[3], nvz = s[3] + k.dt * az;
            const float nth = s[4] + k.dt * s[5], nom = s[5] + k.dt * al;
            s[0] = nx; s[1] = nvx; s[2] = nz; s[3] = nvz; s[4] = nth; s[5] = nom;
        }
    }
    CTK_DEV static float state_terms(const K& k, const float (&s)[S]) {   // position + attitude (shared by stage and terminal)
        const float dx = s[0] - k.tx, dz = s[2] - k.tz;
        return k.pos_c * (dx * dx + dz * dz) + k.ang_w * (1.0f - cosf(s[4]));
    }
    CTK_DEV static float stage_cost(const K& k, const float (&s)[S], const float (&u)[C], const float (&up)[C]) {
        const float vel = k.vel_w * (s[1] * s[1] + s[3] * s[3]) + k.angvel_w * s[5] * s[5];
        const float d0 = u[0] - up[0], d1 = u[1] - up[1];
        return state_terms(k, s) + vel + k.ccR * (u[0] * u[0] + u[1] * u[1]) + k.ccrc_weight * (d0 * d0 + d1 * d1);
    }
    CTK_DEV static float terminal_cost(const K& k, const float (&s)[S]) { return k.terminal_weight * state_terms(k, s); }
    // hooks of the 4-wave rollout kernels (see CartPole): inputs are consumed as they are; sin / cos of the attitude are shared
    // by the cost's 1 - cos(theta) and the first Euler sub-step
    static constexpr bool SEPARABLE = true;
    CTK_DEV static float input_cost(const K& k, const float (&u)[C], const float (&up)[C]) {
        const float d0 = u[0] - up[0], d1 = u[1] - up[1];
        return k.ccR * (u[0] * u[0] + u[1] * u[1]) + k.ccrc_weight * (d0 * d0 + d1 * d1);
    }
    CTK_DEV static float prep_input(const K&, float u, int /*c*/) { return u; }
    CTK_DEV static bool fast_ok(const K&) { return true; }
    CTK_DEV static bool out_of_range(float amax) { return !(amax <= CTK_SINCOS_FAST_LIMIT); }
    template <bool FAST>
    CTK_DEV static void cost_step(const K& k, float (&s)[S], const float (&u)[C], float& csum, float& amax) {
        float sn, cs;
        if constexpr (FAST) { ctk_sincosf_fast(s[4], &sn, &cs); amax = fmaxf(amax, fabsf(s[4])); }
        else ctk_sincosf(s[4], &sn, &cs);
        const float dx = s[0] - k.tx, dz = s[2] - k.tz;
        csum += k.pos_c * (dx * dx + dz * dz) + k.ang_w * (1.0f - cs) + k.vel_w * (s[1] * s[1] + s[3] * s[3]) + k.angvel_w * s[5] * s[5];
        const float aF = k.g + k.kF * (u[0] + u[1]);
        const float aM = k.kM * (u[0] - u[1]);
        for (int i = 0; i < k.intermediate_steps; ++i) {
            if (i > 0) {
                if constexpr (FAST) { ctk_sincosf_fast(s[4], &sn, &cs); amax = fmaxf(amax, fabsf(s[4])); }
                else ctk_sincosf(s[4], &sn, &cs);
            }
            const float ax = -aF * sn - k.c_v * s[1];
            const float az = aF * cs - k.g - k.c_v * s[3];
            const float al = aM - k.c_w * s[5];
            const float nx = s[0] + k.dt * s[1], nvx = s[1] + k.dt * ax, nz = s[2] + k.dt * s[3], nvz = s[3] + k.dt * az;
            const float nth = s[4] + k.dt * s[5], nom = s[5] + k.dt * al;
            s[0] = nx; s[1] = nvx; s[2] = nz; s[3] = nvz; s[4] = nth; s[5] = nom;
        }
    }

    // hooks of the descent kernels (see CartPole).  tape: the six states, sin, cos of the attitude
    static constexpr int NT = 8;
    CTK_DEV static void fwd_tape(const K& k, float (&s)[S], const float (&u)[C], float (&tp)[NT]) {
        float sn, cs;
        ctk_sincosf(s[4], &sn, &cs);
#pragma unroll
        for (int i = 0; i < S; ++i) tp[i] = s[i];
        tp[6] = sn; tp[7] = cs;
        const float aF = k.g + k.kF * (u[0] + u[1]);
        const float aM = k.kM * (u[0] - u[1]);
        const float ax = -aF * sn - k.c_v * s[1];
        const float az = aF * cs - k.g - k.c_v * s[3];
        const float al = aM - k.c_w * s[5];
        const float nx = s[0] + k.dt * s[1], nvx = s[1] + k.dt * ax, nz = s[2] + k.dt * s[3], nvz = s[3] + k.dt * az;
        const float nth = s[4] + k.dt * s[5], nom = s[5] + k.dt * al;
        s[0] = nx; s[1] = nvx; s[2] = nz; s[3] = nvz; s[4] = nth; s[5] = nom;
    }
    CTK_DEV static void bwd_tape(const K& k, const float (&tp)[NT], const float (&u)[C], float (&lam)[S], float (&du)[C], float inv) {
        const float sn = tp[6], cs = tp[7];
        const float aF = k.g + k.kF * (u[0] + u[1]);
        const float dt = k.dt;
        const float a_ax = dt * lam[1], a_az = dt * lam[3], a_al = dt * lam[5];
        const float a_aF = -sn * a_ax + cs * a_az;
        const float d0 = lam[0], d1 = lam[1] + dt * lam[0] - k.c_v * a_ax, d2 = lam[2], d3 = lam[3] + dt * lam[2] - k.c_v * a_az;
        const float d4 = lam[4] - aF * (cs * a_ax + sn * a_az), d5 = lam[5] + dt * lam[4] - k.c_w * a_al;
        du[0] = k.kF * a_aF + k.kM * a_al;
        du[1] = k.kF * a_aF - k.kM * a_al;
        lam[0] = 2.0f * k.pos_c * (tp[0] - k.tx) * inv + d0;
        lam[1] = 2.0f * k.vel_w * tp[1] * inv + d1;
        lam[2] = 2.0f * k.pos_c * (tp[2] - k.tz) * inv + d2;
        lam[3] = 2.0f * k.vel_w * tp[3] * inv + d3;
        lam[4] = k.ang_w * sn * inv + d4;
        lam[5] = 2.0f * k.angvel_w * tp[5] * inv + d5;
    }
    CTK_DEV static void step_vjp(const K& k, const float (&s)[S], const float (&u)[C], const float (&lam)[S], float (&ds)[S],
                                 float (&du)[C]) {
        float sn, cs;
        ctk_sincosf(s[4], &sn, &cs);
        const float aF = k.g + k.kF * (u[0] + u[1]);
        const float dt = k.dt;
        const float a_ax = dt * lam[1], a_az = dt * lam[3], a_al = dt * lam[5];
        const float a_aF = -sn * a_ax + cs * a_az;
        ds[0] = lam[0];
        ds[1] = lam[1] + dt * lam[0] - k.c_v * a_ax;
        ds[2] = lam[2];
        ds[3] = lam[3] + dt * lam[2] - k.c_v * a_az;
        ds[4] = lam[4] - aF * (cs * a_ax + sn * a_az);
        ds[5] = lam[5] + dt * lam[4] - k.c_w * a_al;
        du[0] = k.kF * a_aF + k.kM * a_al;
        du[1] = k.kF * a_aF - k.kM * a_al;
    }
    CTK_DEV static void stage_grad_state(const K& k, const float (&s)[S], float (&g)[S]) {
        g[0] = 2.0f * k.pos_c * (s[0] - k.tx);
        g[1] = 2.0f * k.vel_w * s[1];
        g[2] = 2.0f * k.pos_c * (s[2] - k.tz);
        g[3] = 2.0f * k.vel_w * s[3];
        g[4] = k.ang_w * sinf(s[4]);
        g[5] = 2.0f * k.angvel_w * s[5];
    }
    CTK_DEV static void terminal_grad(const K& k, const float (&s)[S], float (&g)[S]) {
        g[0] = k.terminal_weight * 2.0f * k.pos_c * (s[0] - k.tx);
        g[1] = 0.0f;
        g[2] = k.terminal_weight * 2.0f * k.pos_c * (s[2] - k.tz);
        g[3] = 0.0f;
        g[4] = k.terminal_weight * k.ang_w * sinf(s[4]);
        g[5] = 0.0f;
    }
    CTK_DEV static void input_grad(const K& k, const float (&u)[C], const float (&up)[C], float (&gu)[C], float (&gp)[C]) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = 2.0f * k.ccrc_weight * (u[c] - up[c]);
            gu[c] = 2.0f * k.ccR * u[c] + d;
            gp[c] = -d;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Hover — planar hovercraft with a reaction wheel (oracle/ctk_oracle.py: HoverParams, Predictor._hover_step/_hover_vjp,
// Cost._hover_*).  state (x, vx, y, vy, theta, omega, wheel speed); inputs (main thruster, lateral thruster, wheel torque) in
// [-1, 1]; the wheel's torque reacts on the body.  S + C = 10: the network predictors need a third layer-1 k-step.
// ---------------------------------------------------------------------------------------------------------------
struct HoverK {
    float dt, aF, aL, kT, kW, c_v, c_w, c_ww;                                                       // dynamics
    float tx, ty, pos_c, ang_w, vel_w, angvel_w, wheel_w, ccR, ccrc_weight, terminal_weight;        // cost
    int intermediate_steps;
};

template <>
struct Env<CTK_ENV_HOVER> {
    static constexpr int S = 7, C = 3;
    using K = HoverK;
    static K derive(const float* p, float dt, int isteps) {
        auto d = [&](int id) { return (double)p[id]; };
        K k;
        k.dt = (float)((double)dt / isteps);
        k.aF = (float)(d(CTK_V_THRUST_MAX) / d(CTK_V_MASS));
        k.aL = (float)(d(CTK_V_LATERAL_MAX) / d(CTK_V_MASS));
        k.kT = (float)(d(CTK_V_TORQUE_MAX) / d(CTK_V_INERTIA));
        k.kW = (float)(d(CTK_V_TORQUE_MAX) / d(CTK_V_WHEEL_INERTIA));
        k.c_v = (float)d(CTK_V_DRAG_LIN); k.c_w = (float)d(CTK_V_DRAG_ANG); k.c_ww = (float)d(CTK_V_WHEEL_FRICTION);
        k.tx = p[CTK_V_TARGET_X]; k.ty = p[CTK_V_TARGET_Y];
        k.pos_c = (float)(d(CTK_V_POS_WEIGHT) / (d(CTK_V_POS_SCALE) * d(CTK_V_POS_SCALE)));
        k.ang_w = p[CTK_V_ANG_WEIGHT]; k.vel_w = p[CTK_V_VEL_WEIGHT]; k.angvel_w = p[CTK_V_ANGVEL_WEIGHT]; k.wheel_w = p[CTK_V_WHEEL_WEIGHT];
        k.ccR = (float)(d(CTK_V_CC_WEIGHT) * d(CTK_V_R));
        k.ccrc_weight = p[CTK_V_CCRC_WEIGHT];
        k.terminal_weight = p[CTK_V_TERMINAL_WEIGHT];
        k.intermediate_steps = isteps;
        return k;
    }
    // one Euler sub-step with sin / cos of the attitude given
    CTK_DEV static void substep(const K& k, float (&s)[S], const float (&u)[C], float sn, float cs) {
        const float fb = k.aF * u[0], fl = k.aL * u[1];
        const float ax = fb * cs - fl * sn - k.c_v * s[1];
        const float ay = fb * sn + fl * cs - k.c_v * s[3];
        const float al = -k.kT * u[2] - k.c_w * s[5];
        const float aw = k.kW * u[2] - k.c_ww * s[6];
        const float nx = s[0] + k.dt * s[1], nvx = s[1] + k.dt * ax, ny = s[2] + k.dt * s[3], nvy = s[3] + k.dt * ay;
        const float nth = s[4] + k.dt * s[5], nom = s[5] + k.dt * al, nw = s[6] + k.dt * aw;
        s[0] = nx; s[1] = nvx; s[2] = ny; s[3] = nvy; s[4] = nth; s[5] = nom; s[6] = nw;
    }
    CTK_DEV static void step(const K& k, float (&s)[S], const float (&u)[C]) {
        for (int i = 0; i < k.intermediate_steps; ++i) {
            float sn, cs;
            ctk_sincosf(s[4], &sn, &cs);
            substep(k, s, u, sn, cs);
        }
    }
    CTK_DEV static float state_terms(const K& k, const float (&s)[S], float cs) {   // position + attitude (shared by stage and terminal)
        const float dx = s[0] - k.tx, dy = s[2] - k.ty;
        return k.pos_c * (dx * dx + dy * dy) + k.ang_w * (1.0f - cs);
    }
    CTK_DEV static float rate_terms(const K& k, const float (&s)[S]) {
        return k.vel_w * (s[1] * s[1] + s[3] * s[3]) + k.angvel_w * s[5] * s[5] + k.wheel_w * s[6] * s[6];
    }
    CTK_DEV static float input_cost(const K& k, const float (&u)[C], const float (&up)[C]) {
        const float d0 = u[0] - up[0], d1 = u[1] - up[1], d2 = u[2] - up[2];
        return k.ccR * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + k.ccrc_weight * (d0 * d0 + d1 * d1 + d2 * d2);
    }
    CTK_DEV static float stage_cost(const K& k, const float (&s)[S], const float (&u)[C], const float (&up)[C]) {
        return state_terms(k, s, cosf(s[4])) + rate_terms(k, s) + input_cost(k, u, up);
    }
    CTK_DEV static float terminal_cost(const K& k, const float (&s)[S]) { return k.terminal_weight * state_terms(k, s, cosf(s[4])); }
    // hooks of the 4-wave rollout kernels
    static constexpr bool SEPARABLE = true;
    CTK_DEV static float prep_input(const K&, float u, int /*c*/) { return u; }
    CTK_DEV static bool fast_ok(const K&) { return true; }
    CTK_DEV static bool out_of_range(float amax) { return !(amax <= CTK_SINCOS_FAST_LIMIT); }
    template <bool FAST>
    CTK_DEV static void cost_step(const K& k, float (&s)[S], const float (&u)[C], float& csum, float& amax) {
        float sn, cs;
        if constexpr (FAST) { ctk_sincosf_fast(s[4], &sn, &cs); amax = fmaxf(amax, fabsf(s[4])); }
        else ctk_sincosf(s[4], &sn, &cs);
        csum += state_terms(k, s, cs) + rate_terms(k, s);
        for (int i = 0; i < k.intermediate_steps; ++i) {
            if (i > 0) {
                if constexpr (FAST) { ctk_sincosf_fast(s[4], &sn, &cs); amax = fmaxf(amax, fabsf(s[4])); }
                else ctk_sincosf(s[4], &sn, &cs);
            }
            substep(k, s, u, sn, cs);
        }
    }
    // hooks of the descent kernels.  tape: the seven states, sin, cos of the attitude
    static constexpr int NT = 9;
    CTK_DEV static void fwd_tape(const K& k, float (&s)[S], const float (&u)[C], float (&tp)[NT]) {
        float sn, cs;
        ctk_sincosf(s[4], &sn, &cs);
#pragma unroll
        for (int i = 0; i < S; ++i) tp[i] = s[i];
        tp[7] = sn; tp[8] = cs;
        substep(k, s, u, sn, cs);
    }
    CTK_DEV static void vjp_core(const K& k, float sn, float cs, const float (&u)[C], const float (&lam)[S], float (&ds)[S], float (&du)[C]) {
        const float fb = k.aF * u[0], fl = k.aL * u[1];
        const float dt = k.dt;
        const float a_ax = dt * lam[1], a_ay = dt * lam[3], a_al = dt * lam[5], a_aw = dt * lam[6];
        ds[0] = lam[0];
        ds[1] = lam[1] + dt * lam[0] - k.c_v * a_ax;
        ds[2] = lam[2];
        ds[3] = lam[3] + dt * lam[2] - k.c_v * a_ay;
        ds[4] = lam[4] + (-fb * sn - fl * cs) * a_ax + (fb * cs - fl * sn) * a_ay;
        ds[5] = lam[5] + dt * lam[4] - k.c_w * a_al;
        ds[6] = lam[6] - k.c_ww * a_aw;
        du[0] = k.aF * (cs * a_ax + sn * a_ay);
        du[1] = k.aL * (-sn * a_ax + cs * a_ay);
        du[2] = -k.kT * a_al + k.kW * a_aw;
    }
    CTK_DEV static void bwd_tape(const K& k, const float (&tp)[NT], const float (&u)[C], float (&lam)[S], float (&du)[C], float inv) {
        float ds[S];
        vjp_core(k, tp[7], tp[8], u, lam, ds, du);
        lam[0] = 2.0f * k.pos_c * (tp[0] - k.tx) * inv + ds[0];
        lam[1] = 2.0f * k.vel_w * tp[1] * inv + ds[1];
        lam[2] = 2.0f * k.pos_c * (tp[2] - k.ty) * inv + ds[2];
        lam[3] = 2.0f * k.vel_w * tp[3] * inv + ds[3];
        lam[4] = k.ang_w * tp[7] * inv + ds[4];
        lam[5] = 2.0f * k.angvel_w * tp[5] * inv + ds[5];
        lam[6] = 2.0f * k.wheel_w * tp[6] * inv + ds[6];
    }
    CTK_DEV static void step_vjp(const K& k, const float (&s)[S], const float (&u)[C], const float (&lam)[S], float (&ds)[S], float (&du)[C]) {
        float sn, cs;
        ctk_sincosf(s[4], &sn, &cs);
        vjp_core(k, sn, cs, u, lam, ds, du);
    }
    CTK_DEV static void stage_grad_state(const K& k, const float (&s)[S], float (&g)[S]) {
        g[0] = 2.0f * k.pos_c * (s[0] - k.tx);
        g[1] = 2.0f * k.vel_w * s[1];
        g[2] = 2.0f * k.pos_c * (s[2] - k.ty);
        g[3] = 2.0f * k.vel_w * s[3];
        g[4] = k.ang_w * sinf(s[4]);
        g[5] = 2.0f * k.angvel_w * s[5];
        g[6] = 2.0f * k.wheel_w * s[6];
    }
    CTK_DEV static void terminal_grad(const K& k, const float (&s)[S], float (&g)[S]) {
        g[0] = k.terminal_weight * 2.0f * k.pos_c * (s[0] - k.tx);
        g[1] = 0.0f;
        g[2] = k.terminal_weight * 2.0f * k.pos_c * (s[2] - k.ty);
        g[3] = 0.0f;
        g[4] = k.terminal_weight * k.ang_w * sinf(s[4]);
        g[5] = 0.0f;
        g[6] = 0.0f;
    }
    CTK_DEV static void input_grad(const K& k, const float (&u)[C], const float (&up)[C], float (&gu)[C], float (&gp)[C]) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = 2.0f * k.ccrc_weight * (u[c] - up[c]);
            gu[c] = 2.0f * k.ccR * u[c] + d;
            gp[c] = -d;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// The recurrence of the 4-wave rollout kernels, for any environment: steps [hb, he) of ONE trajectory per lane.  F: the
// trajectory's prepared inputs [H*C] in LDS (Env::prep_input form), read one step ahead of their use; s / csum / amax are
// carried so that the caller may split the horizon (ctk_mppi_rollout runs the first steps while the other waves still
// prepare the inputs of the later ones).  traj: this trajectory's [H+1, S] rows (WRITE_TRAJ) or nullptr.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
CTK_DEV void store_state(float* dst, const float (&s)[S]) {
    if constexpr (S % 4 == 0) {
#pragma unroll
        for (int i = 0; i < S / 4; ++i) reinterpret_cast<float4*>(dst)[i] = make_float4(s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
    } else if constexpr (S % 2 == 0) {
#pragma unroll
        for (int i = 0; i < S / 2; ++i) reinterpret_cast<float2*>(dst)[i] = make_float2(s[2 * i], s[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < S; ++i) dst[i] = s[i];
    }
}

// PREP: F holds the raw inputs u (the affine-sampled kernels keep them for the plans they write out); Env::prep_input is applied here
template <int ENV, bool WRITE_TRAJ, bool FAST, bool PREP = false>
CTK_DEV void recur_env_range(const typename Env<ENV>::K& k, float* traj, bool valid, const float* F, int hb, int he,
                             float (&s)[Env<ENV>::S], float& csum, float& amax) {
    using E = Env<ENV>;
    constexpr int C = E::C, S = E::S;
    if (hb >= he) return;
    float fn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) fn[c] = F[hb * C + c];
#pragma unroll 2
    for (int h = hb; h < he; ++h) {
        float f[C];
#pragma unroll
        for (int c = 0; c < C; ++c) f[c] = PREP ? E::prep_input(k, fn[c], c) : fn[c];
        if (h + 1 < he) {
#pragma unroll
            for (int c = 0; c < C; ++c) fn[c] = F[(h + 1) * C + c];
        }
        if constexpr (WRITE_TRAJ) {
            if (valid && traj) store_state<S>(traj + (size_t)h * S, s);
        }
        E::template cost_step<FAST>(k, s, f, csum, amax);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A USER environment, compiled in at configure time (include/ctk_user_env.h says what the header must define; the reference imports its
// concrete cost class and plant model at run time, cost_function_wrapper.py:59-66 / controller_mpc.py:43,67-73 — here the model is C++
// that control_toolkit_amd/build_env.py compiles, with every optimizer kernel instantiated for it, into a library of its own).
// EnvFromModel turns the mathematical minimum — step, state / input / terminal cost, and for the gradient optimizers the step's
// vector-Jacobian product and the cost gradients — into the full hook set of the template kernels.
// ---------------------------------------------------------------------------------------------------------------
template <class M>
struct EnvFromModel {
    static constexpr int S = M::S, C = M::C;
    using K = typename M::K;
    static K derive(const float* p, float dt, int isteps) { return M::derive(p, dt, isteps); }
    CTK_DEV static void step(const K& k, float (&s)[S], const float (&u)[C]) { M::step(k, s, u); }
    CTK_DEV static float stage_cost(const K& k, const float (&s)[S], const float (&u)[C], const float (&up)[C]) {
        return M::state_cost(k, s) + M::input_cost(k, u, up);
    }
    CTK_DEV static float terminal_cost(const K& k, const float (&s)[S]) { return M::terminal_cost(k, s); }
    // hooks of the 4-wave rollout kernels: the stage cost is state part + input-only part by construction of the model interface; no
    // range-limited fast path (the checked sin / cos always), inputs consumed as they are
    static constexpr bool SEPARABLE = true;
    CTK_DEV static float input_cost(const K& k, const float (&u)[C], const float (&up)[C]) { return M::input_cost(k, u, up); }
    CTK_DEV static float prep_input(const K&, float u, int) { return u; }
    CTK_DEV static bool fast_ok(const K&) { return false; }
    CTK_DEV static bool out_of_range(float) { return false; }
    template <bool FAST>
    CTK_DEV static void cost_step(const K& k, float (&s)[S], const float (&u)[C], float& csum, float&) {
        csum += M::state_cost(k, s);
        M::step(k, s, u);
    }
    // hooks of the descent kernels: the tape is the state itself
    static constexpr int NT = S;
    CTK_DEV static void fwd_tape(const K& k, float (&s)[S], const float (&u)[C], float (&tp)[NT]) {
#pragma unroll
        for (int i = 0; i < S; ++i) tp[i] = s[i];
        M::step(k, s, u);
    }
    CTK_DEV static void bwd_tape(const K& k, const float (&tp)[NT], const float (&u)[C], float (&lam)[S], float (&du)[C], float inv) {
        float s[S], ds[S], gs[S];
#pragma unroll
        for (int i = 0; i < S; ++i) s[i] = tp[i];
        M::step_vjp(k, s, u, lam, ds, du);
        M::stage_grad_state(k, s, gs);
#pragma unroll
        for (int i = 0; i < S; ++i) lam[i] = gs[i] * inv + ds[i];
    }
    CTK_DEV static void step_vjp(const K& k, const float (&s)[S], const float (&u)[C], const float (&lam)[S], float (&ds)[S], float (&du)[C]) {
        M::step_vjp(k, s, u, lam, ds, du);
    }
    CTK_DEV static void stage_grad_state(const K& k, const float (&s)[S], float (&g)[S]) { M::stage_grad_state(k, s, g); }
    CTK_DEV static void terminal_grad(const K& k, const float (&s)[S], float (&g)[S]) { M::terminal_grad(k, s, g); }
    CTK_DEV static void input_grad(const K& k, const float (&u)[C], const float (&up)[C], float (&gu)[C], float (&gp)[C]) { M::input_grad(k, u, up, gu, gp); }
};

#ifdef CTK_USER_ENV_HEADER
#include CTK_USER_ENV_HEADER                      // struct CtkUserEnv
static_assert(CtkUserEnv::S >= 1 && CtkUserEnv::S <= CTK_MAX_STATES && CtkUserEnv::C >= 1 && CtkUserEnv::C <= CTK_MAX_INPUTS,
              "user environment: 1 <= S <= 8 states, 1 <= C <= 4 control inputs");
static_assert(CtkUserEnv::NP >= 1 && CtkUserEnv::NP <= CTK_MAX_PARAMS, "user environment: 1 <= NP <= 32 parameters");
template <>
struct Env<CTK_ENV_USER> : EnvFromModel<CtkUserEnv> {};
#define CTK_FOR_ENV_USER_BRANCH(id, ENVV, ...) else if ((id) == CTK_ENV_USER) { constexpr int ENVV = CTK_ENV_USER; __VA_ARGS__; }
#else
#define CTK_FOR_ENV_USER_BRANCH(id, ENVV, ...)
#endif

// dispatch on the runtime environment id: CTK_FOR_ENV(id, ENVV, stmt) runs `stmt` with the constant ENVV
#define CTK_FOR_ENV(id, ENVV, ...)                                              \
    do {                                                                        \
        if ((id) == CTK_ENV_CARTPOLE) { constexpr int ENVV = CTK_ENV_CARTPOLE; __VA_ARGS__; } \
        else if ((id) == CTK_ENV_QUAD2D) { constexpr int ENVV = CTK_ENV_QUAD2D; __VA_ARGS__; } \
        CTK_FOR_ENV_USER_BRANCH(id, ENVV, __VA_ARGS__)                          \
        else { constexpr int ENVV = CTK_ENV_HOVER; __VA_ARGS__; }             \
    } while (0)
