/* ctk_cpu.c — TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported, linked or executed by the product: control_toolkit_amd/).
 *
 * A native multi-core CPU restatement (plain C + OpenMP) of two steps of the reference's sampling-based MPC inner loop on the build-defined
 * CartPole analytic predictor and cost — what SURVEY 8d(iii) calls the native multi-core leg, and BASELINE configs[0]'s "CPU plumbing" case:
 *
 *   ctkc_mppi_step            Optimizers/optimizer_mppi.py:181-193 (step), :170-179 (perturbation at the inducing points, interpolated),
 *                             :154-161 (correction cost + trajectory cost), :163-168 (reward-weighted average), :190-191 (update, u)
 *   ctkc_random_action_step   Optimizers/optimizer_random_action_tf.py:49-76 (sample, rollout, cost, arg-min)
 *   rollout + cost            predictor.predict_core (optimizer_mppi.py:188) and cost_function.get_trajectory_cost (:159; aggregation
 *                             Cost_Functions/__init__.py:90-93: mean over [H stage costs | terminal cost]); the concrete dynamics and cost
 *                             terms are the build's (oracle/ctk_oracle.py: Predictor._ode_step, Cost._get_stage_cost — same expressions in
 *                             the same order, fp32)
 *
 * It follows oracle/ctk_oracle.py (the NumPy restatement that the reference-recorded golden vectors pin) statement by statement and is
 * itself held against those goldens by tests/test_oracle_cpu_port.py; bench.py times it as `cpu_baseline` (kind "port", cores = the OpenMP
 * threads used).  One trajectory per loop iteration, `#pragma omp parallel for` over trajectories; reductions over N in double.
 *
 * build: gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC -o oracle/_build/libctk_cpu.so oracle/ctk_cpu.c -lm   (oracle/ctk_cpu.py does it)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    /* oracle/ctk_oracle.py: derived_constants (fp32) */
    float dt, u_max, g, M_fric, inv_mt, k_ml, k_jf, k43l, k_mpl_mt, inv_xs, ep_c, ccR;
    /* primary cost parameters used as they are */
    float target_position, dd_weight, ekp_weight, ccrc_weight, terminal_weight;
    int intermediate_steps;
} ctkc_env;

int ctkc_abi(void) { return 1; }
int ctkc_env_size(void) { return (int)sizeof(ctkc_env); }
int ctkc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Predictor._ode_step: Euler step(s) of the cart-pole */
static inline void ode_step(const ctkc_env* k, float* x, float* v, float* th, float* om, float q) {
    const float F = k->u_max * q;
    for (int i = 0; i < k->intermediate_steps; ++i) {
        const float sn = sinf(*th), cs = cosf(*th);
        const float A = F + k->k_ml * *om * *om * sn - k->M_fric * *v;
        const float tmp = A * k->inv_mt;
        const float D = k->k43l - k->k_mpl_mt * cs * cs;
        const float Nn = k->g * sn - cs * tmp - k->k_jf * *om;
        const float thdd = Nn / D;
        const float xdd = tmp - k->k_mpl_mt * thdd * cs;
        const float xn = *x + k->dt * *v, vn = *v + k->dt * xdd, thn = *th + k->dt * *om, omn = *om + k->dt * thdd;
        *x = xn; *v = vn; *th = thn; *om = omn;
    }
}

/* Cost._state_terms: dd + ep */
static inline void state_terms(const ctkc_env* k, float x, float th, float* dd, float* ep) {
    const float dxn = (x - k->target_position) * k->inv_xs;
    *dd = k->dd_weight * dxn * dxn;
    const float omc = 1.0f - cosf(th);
    *ep = k->ep_c * omc * omc;
}

/* rollout of one plan u[H] from s with the trajectory cost of Cost.get_trajectory_cost (mean over H stage costs and the terminal cost) */
static float rollout_cost(const ctkc_env* k, const float* s, const float* u, int H, float u_prev) {
    float x = s[0], v = s[1], th = s[2], om = s[3];
    float sum = 0.0f, prev = u_prev;
    for (int h = 0; h < H; ++h) {
        float dd, ep;
        state_terms(k, x, th, &dd, &ep);
        const float ekp = k->ekp_weight * om * om;
        const float cc = k->ccR * u[h] * u[h];
        const float du = u[h] - prev;
        const float ccrc = k->ccrc_weight * du * du;
        sum += (dd + ep + ekp + cc + ccrc);
        ode_step(k, &x, &v, &th, &om, u[h]);
        prev = u[h];
    }
    float dd, ep;
    state_terms(k, x, th, &dd, &ep);
    sum += k->terminal_weight * (dd + ep);
    return sum / (float)(H + 1);
}

/* One MPPI iteration.  noise: standard-normal draws [N, P]; (i0, w0, w1)[H]: the two non-zeros of column t of the reference's
 * interpolation matrix (oracle: interpolation_table); u_nom [H] in/out; J [N] out; u_out [1].  Returns 0, or -1 on allocation failure. */
int ctkc_mppi_step(const ctkc_env* k, int N, int H, int P, const int* i0, const float* w0, const float* w1, float stdev, float lo, float hi,
                   float cc_weight, float R, float NU, float LBD, const float* s, float u_prev, const float* noise, float* u_nom, float* J,
                   float* u_out, int threads) {
    float* shifted = (float*)malloc(sizeof(float) * (size_t)H);
    double* b_acc = (double*)calloc((size_t)H, sizeof(double));
    float* e = (float*)malloc(sizeof(float) * (size_t)N);
    if (!shifted || !b_acc || !e) { free(shifted); free(b_acc); free(e); return -1; }
    for (int h = 0; h < H; ++h) shifted[h] = u_nom[h + 1 < H ? h + 1 : H - 1];          /* :184 */
    const float kdd = 0.5f * (1.0f - 1.0f / NU) * R, kuu = 0.5f * R;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        float* du = (float*)malloc(sizeof(float) * (size_t)H * 2);
        float* u = du + H;
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            const float* y = noise + (size_t)n * P;
            float corr = 0.0f;
            for (int h = 0; h < H; ++h) {
                const int a = i0[h], b = a + 1 < P ? a + 1 : P - 1;
                du[h] = (y[a] * stdev) * w0[h] + (y[b] * stdev) * w1[h];                 /* :170-179 */
                u[h] = fminf(fmaxf(shifted[h] + du[h], lo), hi);                         /* :186-187 */
                corr += cc_weight * (kdd * (du[h] * du[h]) + R * u[h] * du[h] + kuu * (u[h] * u[h]));   /* :154-155 */
            }
            J[n] = rollout_cost(k, s, u, H, u_prev) + corr;                              /* :158-161, :188 */
        }
        free(du);
    }
    float rho = INFINITY;                                                                /* :163-168 */
    for (int n = 0; n < N; ++n) rho = fminf(rho, J[n]);
    double a = 0.0;
    const float nil = -1.0f / LBD;
    for (int n = 0; n < N; ++n) { e[n] = expf(nil * (J[n] - rho)); a += (double)e[n]; }
#pragma omp parallel num_threads(threads)
    {
        double* loc = (double*)calloc((size_t)H, sizeof(double));
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            const float* y = noise + (size_t)n * P;
            for (int h = 0; h < H; ++h) {
                const int ia = i0[h], ib = ia + 1 < P ? ia + 1 : P - 1;
                const float d = (y[ia] * stdev) * w0[h] + (y[ib] * stdev) * w1[h];
                loc[h] += (double)(e[n] * d);
            }
        }
#pragma omp critical
        for (int h = 0; h < H; ++h) b_acc[h] += loc[h];
        free(loc);
    }
    const float af = (float)a;
    for (int h = 0; h < H; ++h) u_nom[h] = fminf(fmaxf(shifted[h] + (float)b_acc[h] / af, lo), hi);   /* :190 */
    *u_out = u_nom[0];                                                                   /* :191 */
    free(shifted); free(b_acc); free(e);
    return 0;
}

/* optimizer_random_action_tf.py:49-76.  u01: U[0,1) draws [N, H]; J [N] out; best: index of the cheapest plan (ties: the lowest index);
 * u_out = its first input. */
int ctkc_random_action_step(const ctkc_env* k, int N, int H, float lo, float hi, const float* s, float u_prev, const float* u01, float* J,
                            int* best, float* u_out, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        float* u = (float*)malloc(sizeof(float) * (size_t)H);
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int h = 0; h < H; ++h) u[h] = u01[(size_t)n * H + h] * (hi - lo) + lo;  /* :56-61 */
            J[n] = rollout_cost(k, s, u, H, u_prev);                                     /* :43-45 */
        }
        free(u);
    }
    int b = 0;
    for (int n = 1; n < N; ++n) if (J[n] < J[b]) b = n;                                   /* :65-66 */
    *best = b;
    *u_out = u01[(size_t)b * H] * (hi - lo) + lo;                                         /* :68 */
    return 0;
}
