/* ctk_hip.h — C ABI of libctk_hip.so, the MI355X (gfx950) batched-rollout engine behind the
 * sampling-based MPC optimizers of SensorsINI/Control_Toolkit.
 *
 * The reference is 100 % Python and has no FFI for this path; the boundary it defines is the
 * `template_optimizer` contract (reference Optimizers/__init__.py:10-79) that
 * `controller_mpc` drives (reference Controllers/controller_mpc.py:57-65, :84-89, :104).  Each
 * entry point below names the reference call it replaces.  The reference's own precedent for a
 * C-ABI + ctypes boundary is Controllers/controller_C.py:261-274 (float* in/out, no torch
 * types).  The ctypes binding a reference maintainer would add is shown in INTEGRATION.md and
 * implemented in control_toolkit_amd/_capi.py.
 *
 * Conventions: all tensors fp32, row-major, batch first ([N,H,C], [N,H+1,S]); S = num_states and
 * C = num_control_inputs are properties of the environment (ctk_env_info): the reference's optimizers
 * take them at run time (Optimizers/__init__.py:52-63) from the predictor (controller_mpc.py:87-88).
 * Every function that returns int returns 0 on success and a ctk_status otherwise;
 * ctk_last_error() gives the message.  A handle is NOT thread-safe (the reference caller is a
 * single-threaded loop, controller_server/controller_server.py:55-86); all work of a handle
 * is issued on one HIP stream.  Device pointers handed to the begin/end entry points (records that a
 * collective fills or reads) are accessed on THAT stream: a caller that runs the collective on another
 * stream must first ctk_set_stream() the handle onto it (control_toolkit_amd/dist.py does), or order the
 * two streams with events.  The library never falls back to a CPU path: without a usable
 * gfx950 device ctk_create() fails.
 */
#ifndef CTK_HIP_H
#define CTK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTK_ABI_VERSION 6   /* v6: CTK_BUF_AGES_LOGGED; ctk_resident_enable on = 2 (read-ahead opt-in); the log's AGES ring holds the ages as logged */
#define CTK_MAX_STATES 8   /* S <= 8 */
#define CTK_MAX_INPUTS 4   /* C <= 4 */

typedef struct ctk_handle ctk_handle;

typedef enum ctk_status {
    CTK_OK = 0,
    CTK_ERR_INVALID_ARGUMENT = 1, /* ValueError in the Python wrapper           */
    CTK_ERR_UNSUPPORTED = 2,      /* NotImplementedError                        */
    CTK_ERR_HIP = 3,              /* a HIP runtime call failed (RuntimeError)   */
    CTK_ERR_NO_DEVICE = 4,        /* no gfx950 device / extension unusable      */
    CTK_ERR_STATE = 5             /* call illegal in the current state          */
} ctk_status;

/* reference: which Optimizers/optimizer_<name>.py the handle replaces */
typedef enum ctk_optimizer {
    CTK_OPT_MPPI = 0,          /* Optimizers/optimizer_mppi.py              */
    CTK_OPT_CEM = 1,           /* Optimizers/optimizer_cem_tf.py            */
    CTK_OPT_RPGD = 2,          /* Optimizers/optimizer_rpgd.py              */
    CTK_OPT_RANDOM_ACTION = 3, /* Optimizers/optimizer_random_action_tf.py  */
    /* thin variants over the same kernels */
    CTK_OPT_GRADIENT = 4,      /* Optimizers/optimizer_gradient_tf.py: Keras-Adam descent on N plans, no resampling;
                                  uses outer_its (= gradient_steps), learning_rate, adam_*, gradmax_clip, warmup*   */
    CTK_OPT_CEM_NAIVE_GRAD = 5,/* Optimizers/optimizer_cem_naive_grad_tf.py: CEM whose samples take one clipped-gradient
                                  SGD step before selection; uses cem_*, learning_rate, gradmax_clip               */
    CTK_OPT_CEM_GRAD_BHARADHWAJ = 6 /* Optimizers/optimizer_cem_grad_bharadhwaj_tf.py: population = [elites | fresh
                                  samples], one Keras-Adam step per outer iteration; uses cem_*, learning_rate,
                                  adam_*, gradmax_clip, warmup*                                                    */
} ctk_optimizer;

/* The plant + its concrete cost: what the reference selects by `environment_name` (Controllers/__init__.py:32-37),
 * `predictor_specification` (controller_mpc.py:67-73: which model of that plant) and `cost_function_specification`
 * (cost_function_wrapper.py:59-66: Control_Toolkit_ASF.Cost_Functions.<environment>.<name>).  Both are external to
 * the reference; this build defines two, written against ONE device interface (csrc/ctk_env.h: step, stage /
 * terminal cost and their adjoints), so that the optimizer kernels are environment-agnostic templates.        */
typedef enum ctk_environment {
    CTK_ENV_CARTPOLE = 0, /* S 4 (position, positionD, angle, angleD), C 1 (normalised motor force);
                             parameters: enum ctk_param.  Also has hand-tuned kernels (ctk_mppi.hip ...).      */
    CTK_ENV_QUAD2D = 1,   /* planar quadrotor: S 6 (x, vx, z, vz, theta, omega), C 2 (normalised rotor commands);
                             parameters: enum ctk_param_quad2d                                                  */
    CTK_ENV_HOVER = 2,    /* planar hovercraft with a reaction wheel: S 7 (x, vx, y, vy, theta, omega, wheel speed), C 3 (main
                             thruster, lateral thruster, wheel torque); S + C = 10 network inputs: the environment that
                             exercises a third layer-1 k-step of the network predictors; parameters: enum ctk_param_hover */
    CTK_ENV_USER = 3,     /* an environment supplied by the USER as a C++ model header (include/ctk_user_env.h) and compiled into a
                             library of its own at configure time (control_toolkit_amd/build_env.py) — the reference imports the concrete
                             cost class and the plant model at run time (cost_function_wrapper.py:59-66, controller_mpc.py:43,67-73), and
                             its own precedent for compile-at-configure is Controllers/controller_C.py:140-248.  Present only in a
                             library built with such a header (ctk_env_info reports it); parameters: the header's own list        */
    CTK_ENV_COUNT
} ctk_environment;

/* reference: predictor_specification passed to PredictorWrapper.configure
 * (Controllers/controller_mpc.py:67-73): "ODE" or a network name                              */
typedef enum ctk_predictor {
    CTK_PRED_ODE = 0, /* the environment's analytic model (explicit Euler), VALU, one thread per trajectory  */
    CTK_PRED_MLP = 1, /* (S+C)-32-32-S tanh MLP (CartPole 5-32-32-4), fp32 MFMA (v_mfma_f32_16x16x4_f32), 16
                         trajectories per wave, forward and reverse mode                                       */
    CTK_PRED_GRU = 2  /* 2x32 GRU + dense 32->S (network-name convention 'GRU-..-32H1-32H2-..',
                         Control_Toolkit_ASF_Template/config_controllers.yml:8), fp32 MFMA; carries a hidden state
                         across MPC steps (ctk_predictor_update).  Forward AND reverse mode (back-propagation through
                         time over the horizon): every optimizer takes it.  CartPole's sampling optimizers run the
                         4-wave forward kernels of ctk_gru.h; the gradient-based optimizers and the other
                         environments run csrc/ctk_net.h:NetGru under the template kernels.                     */
} ctk_predictor;

/* Environment / cost parameters.  Replaces template_controller.update_attributes
 * (Controllers/__init__.py:106-107, per-step targets) and the cost-YAML hot reload
 * (Cost_Functions/cost_function_wrapper.py:71-74).  Legal only between steps.              */
typedef enum ctk_param {
    CTK_P_G = 0, CTK_P_M_CART, CTK_P_M_POLE, CTK_P_L, CTK_P_U_MAX, CTK_P_M_FRIC, CTK_P_J_FRIC,
    CTK_P_TARGET_POSITION, CTK_P_TARGET_EQUILIBRIUM,
    CTK_P_DD_WEIGHT, CTK_P_EP_WEIGHT, CTK_P_EKP_WEIGHT, CTK_P_CC_WEIGHT, CTK_P_CCRC_WEIGHT,
    CTK_P_R, CTK_P_X_SCALE, CTK_P_TERMINAL_WEIGHT,
    CTK_P_COUNT
} ctk_param;

/* parameters of CTK_ENV_QUAD2D (same role; ids are per environment, names via ctk_param_name) */
typedef enum ctk_param_quad2d {
    CTK_Q_G = 0, CTK_Q_MASS, CTK_Q_INERTIA, CTK_Q_ARM, CTK_Q_THRUST_GAIN, CTK_Q_DRAG_LIN, CTK_Q_DRAG_ANG,
    CTK_Q_TARGET_X, CTK_Q_TARGET_Z,
    CTK_Q_POS_WEIGHT, CTK_Q_ANG_WEIGHT, CTK_Q_VEL_WEIGHT, CTK_Q_ANGVEL_WEIGHT, CTK_Q_CC_WEIGHT, CTK_Q_CCRC_WEIGHT,
    CTK_Q_R, CTK_Q_POS_SCALE, CTK_Q_TERMINAL_WEIGHT,
    CTK_Q_COUNT
} ctk_param_quad2d;
/* parameters of CTK_ENV_HOVER */
typedef enum ctk_param_hover {
    CTK_V_MASS = 0, CTK_V_INERTIA, CTK_V_WHEEL_INERTIA, CTK_V_THRUST_MAX, CTK_V_LATERAL_MAX, CTK_V_TORQUE_MAX, CTK_V_DRAG_LIN,
    CTK_V_DRAG_ANG, CTK_V_WHEEL_FRICTION,
    CTK_V_TARGET_X, CTK_V_TARGET_Y,
    CTK_V_POS_WEIGHT, CTK_V_ANG_WEIGHT, CTK_V_VEL_WEIGHT, CTK_V_ANGVEL_WEIGHT, CTK_V_WHEEL_WEIGHT, CTK_V_CC_WEIGHT, CTK_V_CCRC_WEIGHT,
    CTK_V_R, CTK_V_POS_SCALE, CTK_V_TERMINAL_WEIGHT,
    CTK_V_COUNT
} ctk_param_hover;
#define CTK_MAX_PARAMS 32

/* Device-resident tensors readable with ctk_read(); replaces the to_numpy() calls that fill
 * `logging_values` (optimizer_mppi.py:214-220, optimizer_cem_tf.py:96,104-108,
 * optimizer_rpgd.py:428-435).                                                                */
typedef enum ctk_buffer {
    CTK_BUF_Q = 0,      /* [N,H,C]   inputs rolled out in the last step (u_run / Q)            */
    CTK_BUF_J = 1,      /* [N]       trajectory costs of the last rollout                       */
    CTK_BUF_TRAJ = 2,   /* [N,H+1,S] rollout trajectories (only if cfg.materialize_trajectories)*/
    CTK_BUF_U_NOM = 3,  /* [H,C]     MPPI nominal plan / RPGD best plan / CEM mean              */
    CTK_BUF_STD = 4,    /* [H,C]     CEM stdev                                                  */
    CTK_BUF_ADAM_M = 5, /* [N,H,C]   RPGD Adam first moment                                     */
    CTK_BUF_ADAM_V = 6, /* [N,H,C]   RPGD Adam second moment                                    */
    CTK_BUF_AGES = 7,   /* [N]       RPGD trajectory ages                                       */
    CTK_BUF_BEST_IDX = 8,/* [K] as fp32: indices of the best K rollouts, ascending cost         */
    CTK_BUF_PLAN = 9,   /* [N,H,C]   RPGD population after warm start (next step's Q_tf)        */
    CTK_BUF_AGES_LOGGED = 10, /* [N] RPGD trajectory ages as the last step's get_action saw them, BEFORE its keep-k gather and
                                 +1 (what optimizer_rpgd.py:432 logs; the device log's AGES ring holds these)   */
    CTK_BUF_COUNT
} ctk_buffer;

/* where a `const float*` sample buffer lives */
typedef enum ctk_loc {
    CTK_LOC_NONE = 0,   /* no buffer: draw with the on-device Philox4x32-10 generator          */
    CTK_LOC_HOST = 1,   /* host memory (parity mode: caller's generator, copied H2D)           */
    CTK_LOC_DEVICE = 2  /* already resident in HBM (device pointer)                            */
} ctk_loc;

/* Constructor arguments.  Field names are the reference's YAML keys / ctor argument names:
 * common  Optimizers/__init__.py:13-24 ; MPPI optimizer_mppi.py:16-34 ; CEM
 * optimizer_cem_tf.py:16-34 ; RPGD optimizer_rpgd.py:148-179 ; random
 * optimizer_random_action_tf.py:15-27 ; dt = config "mpc_timestep" (controller_mpc.py:69). */
typedef struct ctk_config {
    uint32_t struct_size; /* = sizeof(ctk_config), ABI guard                                   */
    int32_t optimizer;    /* ctk_optimizer                                                     */
    int32_t predictor;    /* ctk_predictor                                                     */
    int32_t device;       /* HIP device ordinal                                                */
    int32_t num_rollouts; /* N — rollouts owned by THIS handle (the local shard)               */
    int32_t mpc_horizon;  /* H                                                                 */
    int32_t num_states;   /* S, must equal the environment's (ctk_env_info)                    */
    int32_t num_control_inputs; /* C, must equal the environment's                             */
    int32_t period_interpolation_inducing_points;
    int32_t intermediate_steps; /* Euler sub-steps per dt (ODE predictor), >= 1               */
    int32_t materialize_trajectories; /* optimizer_logging / calculate_optimal_trajectory      */
    int32_t global_rollout_offset;    /* first global rollout index of this shard (Philox)     */
    uint64_t seed;
    float dt;
    int32_t environment;     /* ctk_environment                                                */
    int32_t generic_kernels; /* 1: run the environment-agnostic template kernels (csrc/ctk_generic.hip) even where
                                a hand-tuned kernel exists (CartPole); 0: fastest available                      */
    float action_low[CTK_MAX_INPUTS], action_high[CTK_MAX_INPUTS]; /* control_limits, one pair per input       */
    /* MPPI */
    float cc_weight, R, LBD, NU, SQRTRHOINV;
    /* CEM */
    int32_t cem_outer_it, cem_best_k, warmup, warmup_iterations;
    float cem_initial_action_stdev, cem_stdev_min;
    /* RPGD */
    int32_t outer_its, resamp_per, shift_previous, opt_keep_k;
    int32_t sampling_distribution; /* 0 = uniform, 1 = normal                                 */
    int32_t sample_whole_control_space; /* 1: uniform samples span [action_low[c], action_high[c]] per input
                                           (optimizer_rpgd.py:200-203); 0: [sample_min, sample_max] for every input  */
    float sample_stdev, sample_mean, sample_min, sample_max;
    float learning_rate, gradmax_clip, adam_beta_1, adam_beta_2, adam_epsilon;
    int32_t adam_rule;   /* RPGD update rule: 0 = the in-repo torch Adam (optimizer_rpgd.py:56-82); 1 = tf.keras.optimizers.Adam,
                            what the reference's TensorFlow branch wraps (:38-43,:306-320; the YAML entry `rpgd-tf`) — third-party
                            arithmetic, published rule: lr_t = lr sqrt(1-b2^t)/(1-b1^t), epsilon not bias-corrected.  Parity of rule 1 is
                            pinned only against the oracle's restatement of that rule (no reference recording: it needs
                            TensorFlow itself).  Any other value: CTK_ERR_INVALID_ARGUMENT at ctk_create.                     */
    int32_t predictor_hidden1, predictor_hidden2;   /* (ABI v6, appended) hidden widths of the MLP / GRU predictor, the <h1>H1-<h2>H2 of the
                            reference's network names (config_controllers.yml:8).  0 = 32.  1..32: the 32-unit kernels (narrower layers
                            embedded exactly by ctk_set_predictor_weights_shaped).  33..64 (MLP only): the handle is built on the
                            64-unit form of the template kernels (ctk_mlp_wide.h; one wave per 16-trajectory tile, ~3x the matrix
                            work per step), ctk_set_predictor_weights then expects the 64 / 64 layout.  > 64, or a GRU wider
                            than 32: CTK_ERR_UNSUPPORTED with the sizes in ctk_last_error.                                      */
} ctk_config;

/* -------------------------------------------------------------------------------------------
 * lifetime
 * ----------------------------------------------------------------------------------------- */
int ctk_abi_version(void);

/* Optimizer.__init__ + configure (controller_mpc.py:57-65,84-89; Optimizers/__init__.py:13-63).
 * Allocates all device state, then performs optimizer_reset() except for RPGD, whose reset
 * needs draws (call ctk_reset).  On failure *out is NULL and ctk_last_error(NULL) explains.  */
int ctk_create(const ctk_config* cfg, ctk_handle** out);
void ctk_destroy(ctk_handle* h);

/* optimizer_reset() (optimizer_mppi.py:227-231, optimizer_cem_tf.py:113-117,
 * optimizer_rpgd.py:527-548).  `draws`: RPGD only — [N,P,C] raw draws (U[0,1) or N(0,1)) for
 * sample_actions (:275-296); loc NONE draws them on device.                                  */
int ctk_reset(ctk_handle* h, const float* draws, int draws_loc);

const char* ctk_last_error(const ctk_handle* h); /* h may be NULL: last create() error       */

/* Use an existing HIP stream (e.g. torch's current stream) for all work of this handle.  The handle's earlier work is ordered before
 * what follows on the new stream by an event (no host synchronisation; the handle's OWN stream, if it is being replaced, is drained and
 * destroyed).  ctk_get_stream: the stream the handle issues on right now — after ctk_resident_enable that is a high-priority stream of
 * its own (see there), which a caller who queues dependent work elsewhere must order against.                                      */
int ctk_set_stream(ctk_handle* h, void* hip_stream);
void* ctk_get_stream(const ctk_handle* h);

/* -------------------------------------------------------------------------------------------
 * parameters
 * ----------------------------------------------------------------------------------------- */
int ctk_set_param(ctk_handle* h, int id, float value);
int ctk_get_param(const ctk_handle* h, int id, float* value);
/* What an environment is made of: S, C and its parameter list (ids 0 .. n_params-1, names as in the cost YAML /
 * the dynamics section).  Any output pointer may be NULL.                                                      */
int ctk_env_info(int environment, int* num_states, int* num_control_inputs, int* n_params);
const char* ctk_param_name(int environment, int id);       /* NULL if out of range */
const char* ctk_environment_name(int environment);         /* "CartPole", "Quad2D", ...; the model's NAME for CTK_ENV_USER; NULL if unknown */
int ctk_param_default(int environment, int id, float* value);  /* the value a new handle starts with (ABI v6)                         */
/* Network weights, flat fp32, I = S + C inputs, S outputs (CartPole: I 5, S 4):
 *   MLP: W1[32,I] b1[32] W2[32,32] b2[32] W3[S,32] b3[S]                       (CartPole 1380 floats)
 *   GRU: per layer W_i[96,I'] W_h[96,32] b_i[96] b_h[96] (rows r|z|n, I' = I then 32), then
 *        W_o[S,32] b_o[S]                                                        (CartPole 10212 floats)
 * Uploading GRU weights also zeroes the carried hidden state.                                 */
size_t ctk_predictor_weight_count(const ctk_handle* h);    /* floats ctk_set_predictor_weights expects (0: ODE) */
int ctk_set_predictor_weights(ctk_handle* h, const float* w, size_t n);
/* Networks of other widths.  The reference names a network by its sizes — `Dense-<I>IN-<h1>H1-<h2>H2-<O>OUT-<n>`, `GRU-6IN-32H1-32H2-5OUT-0`
 * (Control_Toolkit_ASF_Template/config_controllers.yml:8) — and hands the specification to the predictor
 * (Controllers/controller_mpc.py:67-73).  The matrix-core kernels hold 32 units per hidden layer: hidden widths 1..32 are embedded
 * EXACTLY (absent units get zero weights and biases: tanh(0) = 0 enters every sum as an exact zero; a GRU unit with zero weights stays
 * at 0), wider ones are refused with CTK_ERR_UNSUPPORTED and the sizes in ctk_last_error.  Layouts as ctk_set_predictor_weights with
 * 32 replaced by h1 / h2: MLP W1[h1,I] b1[h1] W2[h2,h1] b2[h2] W3[S,h2] b3[S]; GRU per layer W_i[3h,in] W_h[3h,h] b_i[3h] b_h[3h] (rows
 * r|z|n; layer 2's input is h1), then W_o[S,h2] b_o[S].  I = num_states + num_control_inputs, outputs = num_states.  (ABI v6)      */
size_t ctk_predictor_weight_count_shaped(const ctk_handle* h, int h1, int h2);
int ctk_set_predictor_weights_shaped(ctk_handle* h, const float* w, size_t n, int h1, int h2);

/* Recurrent predictor state (GRU): [2,32] fp32, the state every rollout starts from.
 * ctk_predictor_update = predictor.update(s, Q0) (optimizer_mppi.py:195-197): advance it by the measured
 * state s[S] and the applied input u[C] (NULL: the optimizer's last output, still on the device).
 * ctk_step of an MPPI handle does this itself after the nominal-plan update (optimizer_mppi.py:192),
 * CEM / random-action never do (optimizer_cem_tf.py, optimizer_random_action_tf.py have no such call).
 * Non-recurrent predictors: size 0, update is a no-op, get/set reject.                          */
size_t ctk_predictor_hidden_size(const ctk_handle* h);
int ctk_predictor_update(ctk_handle* h, const float* s, const float* u);
int ctk_predictor_get_hidden(ctk_handle* h, float* dst, size_t cap);
int ctk_predictor_set_hidden(ctk_handle* h, const float* src, size_t n); /* src NULL: zeros */

/* -------------------------------------------------------------------------------------------
 * the hot path: optimizer.step(s, time) -> u   (Optimizers/__init__.py:67)
 *   MPPI   optimizer_mppi.py:205-225 (+ :181-193)   samples: N(0,1)   [N,P,C]
 *   CEM    optimizer_cem_tf.py:83-111               samples: N(0,1)   [iters,N,H,C]
 *   RPGD   optimizer_rpgd.py:388-524                samples: raw draws [N-k,P,C], used only on
 *                                                   resampling steps (count % resamp_per == 0)
 *   random optimizer_random_action_tf.py:49-76      samples: U[0,1)   [N,H,C]
 *   gradient optimizer_gradient_tf.py:101-173       samples: U[0,1)   [N,1,C] (the shifted-in tail input, :137-142)
 *   cem-naive-grad optimizer_cem_naive_grad_tf.py:89-115  samples: N(0,1) [cem_outer_it,N,H,C]
 *   cem-grad-bharadhwaj optimizer_cem_grad_bharadhwaj_tf.py:151-178  samples: N(0,1) [K,H,C] (initial elites, :158)
 *                                                   then [iters, N-K, H, C] (:94)
 * s: host [S].  u_prev: host [C] previous applied input (cost `previous_input`); NULL = the
 * optimizer's own last output, as the reference passes self.u.  u_out: host [C].
 * Synchronous: returns when u_out is valid.
 * CTK_ERR_STATE from a step = a bounded device-side wait ran out (ctk_last_error says which): a peer's record (sharded MPPI over
 * p2p), a workgroup's record of an in-launch hand-off (MPPI / CEM), or — RPGD with a network predictor — a step Jacobian of the
 * in-launch hand-off / a non-finite gradient.  In the RPGD case the affected 16-plan tile SKIPPED that iteration's update: plans and
 * Adam moments are finite and as they were before it, u_out comes from a population that missed an update, the handle stays usable
 * and the next ctk_step is a normal step (ctk_reset / ctk_set_state re-pin it if the caller wants a defined population).
 * ----------------------------------------------------------------------------------------- */
int ctk_step(ctk_handle* h, const float* s, const float* u_prev,
             const float* samples, int samples_loc, float* u_out);

/* Number of raw draws (floats) the NEXT ctk_step consumes from `samples` (0 if none).        */
size_t ctk_samples_needed(const ctk_handle* h);

/* Position of the on-device Philox stream = the number of completed steps/resets that drew (the `call` word of
 * the counter).  Together with ctk_get_state (and ctk_predictor_get_hidden for a recurrent predictor) it makes a
 * handle resumable bit-for-bit: reference create_rng(seed) returns a generator object whose state the caller can
 * save and restore (others/globals_and_utils.py:86-99).                                                          */
int ctk_rng_get_position(const ctk_handle* h, uint32_t* call);
int ctk_rng_set_position(ctk_handle* h, uint32_t call);

/* predictor.predict_core(s, Q) + cost_function.get_trajectory_cost(traj, Q, u_prev)
 * (call sites optimizer_mppi.py:188,199-202; Cost_Functions/__init__.py:74-93) for n <= N
 * caller-supplied plans Q [n,H,C] (host).  traj_out [n,H+1,S] and J_out [n] may be NULL.     */
int ctk_rollout(ctk_handle* h, const float* s, const float* u_prev, const float* Q, int n,
                float* traj_out, float* J_out);

/* -------------------------------------------------------------------------------------------
 * sharded MPPI (one handle per GPU).  ctk_step == begin + end with one part.
 *   begin: sample, roll out and reduce this shard to one partial record
 *          [rho_r, a_r, b_r[P*C]] written to `partial_dev` (device pointer, 2+P*C floats);
 *   (caller all-gathers the records over RCCL)
 *   end:   merge `n_parts` records (device pointer, contiguous) and update u_nom; u_out host.
 * ----------------------------------------------------------------------------------------- */
size_t ctk_mppi_partial_size(const ctk_handle* h); /* floats per record                      */
int ctk_mppi_step_begin(ctk_handle* h, const float* s, const float* u_prev,
                        const float* samples, int samples_loc, float* partial_dev);
int ctk_mppi_step_end(ctk_handle* h, const float* parts_dev, int n_parts, float* u_out);

/* -------------------------------------------------------------------------------------------
 * sharded CEM / random-action: the global best K is the best K of the union of the
 * shards' best-K lists, so ONE all-gather of K records {J, global index, Q[H]} per outer iteration
 * replaces tf.argsort over all N (optimizer_cem_tf.py:73-75, optimizer_random_action_tf.py:65-66).
 *   for it in range(ctk_shard_iterations(h)):
 *       ctk_shard_iter_begin(h, s, u_prev, samples_it, loc, cand_dev)   roll out + local best K -> cand_dev
 *       (caller all-gathers the ctk_shard_candidates_size(h) floats of every rank)
 *       ctk_shard_iter_end(h, cands_all_dev, n_ranks)                   global best K (+ CEM refit, :77-78)
 *   ctk_shard_finish(h, u_out)                                          CEM :99-102 / random :68
 * samples_it: this iteration's draws only ([N,H,C]); cand_dev / cands_all_dev are device pointers
 * and cands_all_dev must stay valid until ctk_shard_finish.  K = cem_best_k (CEM) or 1.
 * ----------------------------------------------------------------------------------------- */
size_t ctk_shard_candidates_size(const ctk_handle* h);
int ctk_shard_iterations(const ctk_handle* h);
int ctk_shard_iter_begin(ctk_handle* h, const float* s, const float* u_prev,
                         const float* samples, int samples_loc, float* cand_dev);
int ctk_shard_iter_end(ctk_handle* h, const float* cands_all_dev, int n_ranks);
int ctk_shard_finish(ctk_handle* h, float* u_out);

/* -------------------------------------------------------------------------------------------
 * sharded RPGD.  The descent needs no exchange (loss = sum_n J_n, optimizer_rpgd.py:325);
 * the keep-k / resample step (:345-346, :449-516) needs the global best k plans WITH their Adam
 * moments and ages: one all-gather of ctk_rpgd_keepers_size(h) floats per rank — records
 * {J, global index, age, Q[H], m[H], v[H]} of the shard's best min(k, N_local) plans, sorted.
 *   ctk_rpgd_step_begin(h, s, u_prev, keep_dev)          descent + cost pass + local best list -> keep_dev
 *   (all-gather)
 *   ctk_rpgd_step_end(h, keep_all_dev, n_ranks, draws, loc, u_out)
 * Every rank then rebuilds ITS rows of the global population [fresh | keepers sorted] (:454-455): global
 * row g < N_global - k is resampled (Philox by global row, or `draws` = raw draws for this shard's
 * ctk_rpgd_fresh_rows() fresh rows, [rows,P,C]); the others take keeper g - (N_global - k) from the
 * records.  cfg.opt_keep_k is the GLOBAL k; equal shard sizes are assumed (N_global = n_ranks * N).
 * ----------------------------------------------------------------------------------------- */
size_t ctk_rpgd_keepers_size(const ctk_handle* h);
size_t ctk_rpgd_fresh_rows(const ctk_handle* h, int n_ranks);
int ctk_rpgd_step_begin(ctk_handle* h, const float* s, const float* u_prev, float* keep_dev);
int ctk_rpgd_step_end(ctk_handle* h, const float* keep_all_dev, int n_ranks,
                      const float* draws, int draws_loc, float* u_out);

/* -------------------------------------------------------------------------------------------
 * state access
 * ----------------------------------------------------------------------------------------- */
/* Copies buffer `which` to host `dst` (capacity `cap` floats); *n_out = floats written.      */
int ctk_read(ctk_handle* h, int which, float* dst, size_t cap, size_t* n_out);

/* Warm-start state (SURVEY.md 5 'checkpoint/resume'): MPPI u_nom,u ; CEM mu,std,count,u ;
 * RPGD Q,m,v,ages,adam_step,count,u.  ctk_state_size() floats.                               */
size_t ctk_state_size(const ctk_handle* h);
int ctk_get_state(ctk_handle* h, float* dst, size_t cap);
int ctk_set_state(ctk_handle* h, const float* src, size_t n);

/* -------------------------------------------------------------------------------------------
 * measurement: HIP-event timing of the dominant (rollout) kernel on the handle's stream.
 * enable, run steps, then read the durations in milliseconds.  `on` = n > 0 times every n-th launch of
 * the dominant kernel (timing a launch through its dispatch timestamps costs several microseconds of
 * host time, so a sparse sample perturbs the timed region less); 0 disables.
 * ----------------------------------------------------------------------------------------- */
int ctk_profile_enable(ctk_handle* h, int on);
int ctk_profile_read(ctk_handle* h, float* ms_out, size_t cap, size_t* n_out);
/* name of the dominant kernel (as it appears in rocprofv3 --kernel-trace) */
const char* ctk_dominant_kernel(const ctk_handle* h);

/* -------------------------------------------------------------------------------------------
 * sharded MPPI over peer-to-peer stores (no counterpart in the reference).  Alternative to
 * ctk_mppi_step_begin / all-gather / ctk_mppi_step_end for ranks on ONE node: every rank owns an uncached
 * exchange buffer that all peers map through HIP IPC; per step each rank's exchange kernel stores its
 * (2+P)-float record into every peer's buffer over xGMI, raises a flag, waits for all flags in its own
 * buffer (bounded by a timeout) and merges — no host round trip and no collective-library launch between the
 * rollout and the update.  Protocol: every rank calls ctk_p2p_alloc (returns a CTK_P2P_HANDLE_BYTES handle),
 * the caller exchanges the handles by any means (torch.distributed all_gather_object), every rank calls
 * ctk_p2p_connect with all handles in rank order, then ctk_p2p_step in lockstep (same number of calls on
 * every rank).  world <= 16.  A timeout (peer gone) surfaces as CTK_ERR_STATE.
 * ----------------------------------------------------------------------------------------- */
#define CTK_P2P_HANDLE_BYTES 64
int ctk_p2p_alloc(ctk_handle* h, int rank, int world, void* handle_out);
int ctk_p2p_connect(ctk_handle* h, const void* handles);
int ctk_p2p_step(ctk_handle* h, const float* s, const float* u_prev, const float* samples, int samples_loc, float* u_out);
int ctk_p2p_close(ctk_handle* h);

/* -------------------------------------------------------------------------------------------
 * device-resident step log.  Replaces the per-step to_numpy() of `optimizer_logging`
 * (optimizer_mppi.py:214-218, optimizer_cem_tf.py:104-108, optimizer_rpgd.py:428-433) + the per-step copies of
 * template_controller.update_logs (Controllers/__init__.py:170-178): after every completed step ONE copy
 * kernel appends Q [N,H,C], J [N], the rollout trajectories [N,H+1,S] (if materialised) and, for RPGD, the
 * trajectory ages [N] to a ring in HBM (what ctk_read would return for CTK_BUF_Q / J / TRAJ / AGES_LOGGED at that
 * moment); the host fetches any run of steps in one transfer when it wants them
 * (template_controller.get_outputs, Controllers/__init__.py:159-168).
 * capacity_steps > 0 allocates the ring (capacity * (N*H + 2N + N*(H+1)*S) floats) and starts logging;
 * 0 stops and frees.  Steps older than `capacity` are overwritten; reading them is CTK_ERR_STATE.
 * ----------------------------------------------------------------------------------------- */
int ctk_log_enable(ctk_handle* h, size_t capacity_steps);
size_t ctk_log_count(const ctk_handle* h); /* steps logged since ctk_log_enable */
/* which: CTK_BUF_Q | CTK_BUF_J | CTK_BUF_TRAJ | CTK_BUF_AGES (= CTK_BUF_AGES_LOGGED: the ages as logged); steps [first_step, first_step + n_steps)   */
int ctk_log_read(ctk_handle* h, int which, size_t first_step, size_t n_steps, float* dst, size_t cap, size_t* n_out);

/* -------------------------------------------------------------------------------------------
 * resident MPPI step (opt-in; ABI v5).  The reference's loop calls optimizer.step once per control period
 * (Controllers/controller_mpc.py:104, controller_server/controller_server.py:55-86): one kernel launch per call costs ~6 us of
 * runtime + command processor on top of the kernel.  With ctk_resident_enable(h, 1, idle_us) the FIRST ctk_step launches a kernel that
 * stays on the device and serves every following ctk_step from a pinned mailbox — same arguments, same results, same error
 * behaviour.  MPPI with the analytic predictor in the one-launch regime only (<= 128 workgroups, trajectories not materialised);
 * steps that hand over HOST samples, logging or per-launch timing run the launched form (the resident kernel is ended first).
 *  - every device-side wait is bounded by the wall clock: without a request for `idle_us` microseconds (1 .. 1e6; 200 is a good
 *    value for a closed loop) the kernel leaves by itself and the next ctk_step launches it again;
 *  - it therefore holds the device at most idle_us beyond its last step: a device-wide synchronize issued by anybody else
 *    (another handle, torch.cuda.synchronize) waits at most that long.  ctk_resident_stop ends it at once; so does every
 *    other call on the handle that touches device state (read, reset, set_param, set_state, rollout, ...), and ctk_destroy;
 *  - the mailbox lives in fine-grained device memory that the host stores into through the PCIe BAR (every workgroup polls local memory;
 *    no PCIe read on the request path), or — where a probe at enable time finds the host cannot reach device memory — in pinned host
 *    memory polled by one workgroup that relays the request to the others;
 *  - hardware queues: HIP multiplexes a process's streams over a few in-order hardware queues (GPU_MAX_HW_QUEUES, 4 by default); work of
 *    another stream that lands on the resident kernel's queue waits until it leaves (<= idle_us each time).  ctk_resident_enable
 *    therefore re-creates the handle's OWN stream at the highest stream priority (served from other queues than default-priority
 *    streams).  A stream handed in with ctk_set_stream is used as it is: give it a priority of its own, or keep idle_us short;
 *  - sample buffers (samples_loc = CTK_LOC_DEVICE): between two steps the kernel prepares the NEXT step's inputs.  With the in-kernel
 *    sampler (samples = NULL) that includes the draws.  Caller-owned buffers are read ahead ONLY with on = 2, which is the caller's promise
 *    that the contents of every buffer it hands over do not change while the resident form is enabled (a static pool, cycled in any order):
 *    the library predicts the next buffer from the order seen so far and a changed buffer would be served with its OLD contents.  With
 *    on = 1 (default) a buffer is read when its request arrives — refilling one buffer in place between steps is then as safe as with
 *    the launched form (tests/test_gpu_resident.py: test_resident_refilled_buffer_is_read_at_the_request), at ~2 us more per step;
 *  - ctk_resident_stats: kernel launches and steps served so far, whether the kernel is believed to be running, where the mailbox is.
 * ----------------------------------------------------------------------------------------- */
int ctk_resident_enable(ctk_handle* h, int on /* 0 off | 1 on | 2 on + read-ahead of immutable sample buffers */, double idle_us);
int ctk_resident_stop(ctk_handle* h);
int ctk_resident_stats(const ctk_handle* h, uint64_t* launches, uint64_t* steps, int* running, int* mailbox_in_device_memory);

#ifdef __cplusplus
}
#endif
#endif /* CTK_HIP_H */
