// tests/envs/pendulum_env.h — a USER environment (include/ctk_user_env.h): torque-driven pendulum, upright at theta = 0.
// state (theta, omega), input u in [-1, 1]; explicit Euler like the built environments.  Its NumPy counterpart, statement by statement,
// is in tests/test_gpu_user_env.py (the parity oracle for this model).
struct CtkUserEnv {
    static constexpr int S = 2, C = 1;
    static constexpr const char* NAME = "Pendulum";
    static constexpr int NP = 11;
    static constexpr const char* PARAM_NAMES[NP] = {"g", "length", "damping", "torque_gain", "target_angle", "ang_weight", "vel_weight",
                                                    "cc_weight", "ccrc_weight", "R", "terminal_weight"};
    static constexpr float PARAM_DEFAULTS[NP] = {9.81f, 0.5f, 0.1f, 12.0f, 0.0f, 50.0f, 0.5f, 1.0f, 2.0f, 1.0f, 0.0f};
    struct K { float dt, gl, c, kU, target, ang_w, vel_w, ccR, ccrc_weight, terminal_weight; int intermediate_steps; };
    static K derive(const float* p, float dt, int isteps) {
        K k;
        k.dt = (float)((double)dt / isteps);
        k.gl = (float)((double)p[0] / (double)p[1]);
        k.c = p[2]; k.kU = p[3]; k.target = p[4]; k.ang_w = p[5]; k.vel_w = p[6];
        k.ccR = (float)((double)p[7] * (double)p[9]);
        k.ccrc_weight = p[8]; k.terminal_weight = p[10];
        k.intermediate_steps = isteps;
        return k;
    }
    CTK_DEV static void step(const K& k, float (&s)[S], const float (&u)[C]) {
        for (int i = 0; i < k.intermediate_steps; ++i) {
            float sn, cs;
            ctk_sincosf(s[0], &sn, &cs);
            const float al = k.gl * sn - k.c * s[1] + k.kU * u[0];
            const float nth = s[0] + k.dt * s[1], nom = s[1] + k.dt * al;
            s[0] = nth; s[1] = nom;
        }
    }
    CTK_DEV static float state_cost(const K& k, const float (&s)[S]) { return k.ang_w * (1.0f - cosf(s[0] - k.target)) + k.vel_w * s[1] * s[1]; }
    CTK_DEV static float input_cost(const K& k, const float (&u)[C], const float (&up)[C]) {
        const float d = u[0] - up[0];
        return k.ccR * u[0] * u[0] + k.ccrc_weight * d * d;
    }
    CTK_DEV static float terminal_cost(const K& k, const float (&s)[S]) { return k.terminal_weight * k.ang_w * (1.0f - cosf(s[0] - k.target)); }
    CTK_DEV static void step_vjp(const K& k, const float (&s)[S], const float (&u)[C], const float (&lam)[S], float (&ds)[S], float (&du)[C]) {
        float sn, cs;
        ctk_sincosf(s[0], &sn, &cs);
        const float a_al = k.dt * lam[1];
        ds[0] = lam[0] + a_al * k.gl * cs;
        ds[1] = lam[1] + k.dt * lam[0] - k.c * a_al;
        du[0] = k.kU * a_al;
    }
    CTK_DEV static void stage_grad_state(const K& k, const float (&s)[S], float (&g)[S]) {
        g[0] = k.ang_w * sinf(s[0] - k.target);
        g[1] = 2.0f * k.vel_w * s[1];
    }
    CTK_DEV static void terminal_grad(const K& k, const float (&s)[S], float (&g)[S]) {
        g[0] = k.terminal_weight * k.ang_w * sinf(s[0] - k.target);
        g[1] = 0.0f;
    }
    CTK_DEV static void input_grad(const K& k, const float (&u)[C], const float (&up)[C], float (&gu)[C], float (&gp)[C]) {
        const float d = 2.0f * k.ccrc_weight * (u[0] - up[0]);
        gu[0] = 2.0f * k.ccR * u[0] + d;
        gp[0] = -d;
    }
};
