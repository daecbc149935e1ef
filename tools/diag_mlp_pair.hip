// diag_mlp_pair.hip — the two-waves-per-tile MLP step (ctk_mlp.h: mlp_step_pair) in isolation: ns per step of a workgroup of two
// pairs, with and without the stage-cost work of ctk_mppi_rollout<3,.>, against the one-wave step.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1 -I control_toolkit_amd/csrc -o tools/diag_mlp_pair tools/diag_mlp_pair.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ctk_mlp.h"

template <int MODE>   // 0: one-wave step; 1: pair step; 2: pair step + the stage-cost terms of ctk_mppi_rollout<3,.> (cos on wave 0 of the pair)
__global__ __launch_bounds__(256) void k_run(const float* wperm, float* out, int H) {
    __shared__ __attribute__((aligned(16))) float ex[2 * MLP_PAIR_EX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const MlpFwdT wt = mlp_load_fwd_thin(wperm);
    float sv = 0.01f * (lane & 15) + 0.1f * g, acc = 0.f;
    if constexpr (MODE == 0) {
        for (int h = 0; h < H; ++h) sv = mlp_step(wt, sv, 0.1f, g);
    } else {
        const int pair = wave >> 1, m = wave & 1;
        const MlpFwdHalf w = mlp_half_of(wt, m);
        for (int h = 0; h < H; ++h) {
            const float now = sv;
            if constexpr (MODE == 2) { if (m == 0) { const float o = 1.0f - ctk_cosf_fast(now); acc += o * o; } else { acc += now * now; } }
            sv = mlp_step_pair(w, now, 0.1f, m, ex + pair * MLP_PAIR_EX);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sv + acc;
}

template <class F>
float time_ms(F&& launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    std::vector<float> w(64 * (MLP_FWD_PER_LANE + MLP_BWD_PER_LANE));
    for (size_t i = 0; i < w.size(); ++i) w[i] = 0.05f * (float)((i * 7919) % 13 - 6);
    float *dw, *dout;
    hipMalloc(&dw, w.size() * 4); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, 4096 * 256 * 4);
    const int H = 200;
    for (int blocks : {64, 256, 512}) {
        const float a = time_ms([&] { hipLaunchKernelGGL(k_run<0>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        const float b = time_ms([&] { hipLaunchKernelGGL(k_run<1>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        const float c = time_ms([&] { hipLaunchKernelGGL(k_run<2>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        printf("blocks %4d x 4 waves, H %d: ns per step  one-wave %6.1f (4 tiles/block) | pair %6.1f | pair + cost terms %6.1f (2 tiles/block)\n",
               blocks, H, a * 1e6f / H, b * 1e6f / H, c * 1e6f / H);
    }
    return 0;
}
