/* ctk_user_env.h — what a USER ENVIRONMENT model header must define (documentation; nothing includes this file).
 *
 * The reference selects the plant model and the concrete cost at RUN time: `PredictorWrapper()` configured with a
 * `predictor_specification` (Controllers/controller_mpc.py:43,67-73) and `Control_Toolkit_ASF.Cost_Functions.<environment>.<name>`
 * imported by name (Cost_Functions/cost_function_wrapper.py:59-66).  In this build the rollout and the cost are fused into the optimizer
 * kernels, so a new environment is C++: ONE header defining `struct CtkUserEnv` as below, compiled at configure time into a library of
 * its own with every optimizer kernel instantiated for it —
 *
 *     from control_toolkit_amd.build_env import register_environment
 *     register_environment("my_env.h")                  # hipcc, ~1-2 min the first time, cached by content (in-tree, next to the package)
 *     CtkEngine("mppi", "ODE", environment="MyEnv", ...)     # or `environment_name: MyEnv` in the optimizer's YAML entry
 *
 * (the reference's own precedent for compile-at-configure: Controllers/controller_C.py:140-248).  No file under csrc/ is edited.
 * The header is included by csrc/ctk_env.h behind the device helpers: CTK_DEV, ctk_sincosf(x, &s, &c), f32 math are available.
 * A complete example with its NumPy counterpart: tests/envs/pendulum_env.h, tests/test_gpu_user_env.py.
 *
 * struct CtkUserEnv {
 *     static constexpr int S = ..., C = ...;                      // num_states (<= 8), num_control_inputs (<= 4)
 *     static constexpr const char* NAME = "MyEnv";                // environment_name the Python side resolves
 *     static constexpr int NP = ...;                              // parameters (<= 32): dynamics constants and cost weights, settable per
 *     static constexpr const char* PARAM_NAMES[NP] = {...};       //   step by name (ctk_set_param: update_attributes / cost-YAML hot reload)
 *     static constexpr float PARAM_DEFAULTS[NP] = {...};
 *     struct K { ... };                                           // derived fp32 constants the kernels take by value (keep it small)
 *     static K derive(const float* p, float dt, int intermediate_steps);         // HOST: parameter table -> K (dt = mpc_timestep)
 *     // forward (every optimizer):
 *     CTK_DEV static void  step(const K&, float (&s)[S], const float (&u)[C]);    // one predictor step in place (all Euler sub-steps)
 *     CTK_DEV static float state_cost(const K&, const float (&s)[S]);             // stage cost = state_cost(s_h) + input_cost(u_h, u_{h-1})
 *     CTK_DEV static float input_cost(const K&, const float (&u)[C], const float (&u_prev)[C]);
 *     CTK_DEV static float terminal_cost(const K&, const float (&s)[S]);          // of s_H; J = mean over [H stage costs | terminal]
 *     // reverse mode (RPGD, gradient-tf, the CEM gradient variants; intermediate_steps == 1):
 *     CTK_DEV static void step_vjp(const K&, const float (&s)[S], const float (&u)[C], const float (&lam)[S],    // lam = dL/ds'
 *                                  float (&ds)[S], float (&du)[C]);                                              // -> dL/ds, dL/du through the step
 *     CTK_DEV static void stage_grad_state(const K&, const float (&s)[S], float (&g)[S]);       // d state_cost / ds
 *     CTK_DEV static void terminal_grad(const K&, const float (&s)[S], float (&g)[S]);          // d terminal_cost / ds
 *     CTK_DEV static void input_grad(const K&, const float (&u)[C], const float (&u_prev)[C],   // d input_cost / du, d input_cost / du_prev
 *                                    float (&gu)[C], float (&gp)[C]);
 * };
 *
 * Network predictors (MLP / GRU) of a user environment come for free: they see S + C inputs and S outputs like the built environments.
 */
