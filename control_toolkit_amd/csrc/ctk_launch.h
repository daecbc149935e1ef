// ctk_launch.h — host-callable launchers implemented in the kernel translation units.
#pragma once
#include "ctk_common.h"

// ---- ctk_mppi.hip ---------------------------------------------------------------------------
const char* ctk_mppi_rollout_ode_name(bool log);
int ctk_mppi_num_blocks_ode(int N);
size_t ctk_mppi_rollout_ode_lds(int P, int H);
hipError_t ctk_launch_mppi_rollout_ode(hipStream_t st, const RolloutArgs& a, const EnvK& k, const MppiK& m,
                                       const float* samples, const float* u_nom, float* parts, bool log);
hipError_t ctk_launch_mppi_merge_partial(hipStream_t st, const float* parts, int n_parts, int per_block, int P,
                                         float neg_inv_lbd, float* out_rec);
hipError_t ctk_launch_mppi_update(hipStream_t st, const float* parts, int n_parts, int P, float neg_inv_lbd, int H,
                                  const InterpEntry* interp, const float* u_nom_in, float* u_nom_out, float lo, float hi,
                                  float* u_dev, float* u_host);

// ---- ctk_sampled.hip : u[n,h] = clip(base[h] + sample[n,h] * scale[h]) rollouts, selection ----
const char* ctk_affine_rollout_ode_name(bool log);
// samples [N,H] (device) or nullptr (Philox, rng_kind 0 normal / 1 uniform); base/scale [H] device.
hipError_t ctk_launch_affine_rollout_ode(hipStream_t st, const RolloutArgs& a, const EnvK& k, const float* samples,
                                         int rng_kind, const float* base, const float* scale, bool log);
// Smallest-K selection under the total order (J, index) and CEM refit
// (optimizer_cem_tf.py:73-78): idx_out[K] ascending, mu/std [H] from Q[idx].
hipError_t ctk_launch_select_topk(hipStream_t st, const float* J, int N, int K, int* idx_out, unsigned* scratch);
hipError_t ctk_launch_cem_refit(hipStream_t st, const float* Q, const int* idx, int K, int H, float* mu, float* sd);
// CEM post-loop (optimizer_cem_tf.py:99-102) and u = elite[0,0]
hipError_t ctk_launch_cem_finish(hipStream_t st, const float* Q, const int* idx, int H, float* mu, float* sd,
                                 float std_min, float init_std, float mid, float* u_dev, float* u_host);
// random-action: u = Q[argmin J, 0]  (optimizer_random_action_tf.py:65-68)
hipError_t ctk_launch_pick_best_first(hipStream_t st, const float* Q, const int* idx, int H, float* u_dev, float* u_host);
