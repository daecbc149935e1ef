// diag_mlp_step.hip — where does a wave's MLP step go?  Times H steps of the 5-32-32-4 tanh MLP step of ctk_mlp.h in
// isolation (no cost, no inputs from LDS), one wave per SIMD-ish launch shapes, in several variants:
//   full      mlp_step as the rollout kernels run it (round 2 start: 28 MFMA + 16 tanh; now the thin-layer form, see diag_mlp_l3.hip)
//   mfma      the same 28 MFMAs with the tanh replaced by a copy (dependent chain kept)
//   tanh      the 16 tanh alone (dependent chain kept)
//   split     two-wave form (14 MFMA + 8 tanh + 2 LDS exchanges with barriers)
//   split_nb  the same without the barriers (WRONG results; isolates the synchronisation cost)
// build: hipcc -O3 --offload-arch=gfx950 -I control_toolkit_amd/csrc -o tools/diag_mlp_step tools/diag_mlp_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "diag_mlp_split.h"

template <int MODE>
__global__ __launch_bounds__(256) void k_unsplit(const float* wperm, float* out, int H) {
    const MlpFwdW w = mlp_load_fwd(wperm);
    const MlpFwdT wt = mlp_load_fwd_thin(wperm);
    const int g = (threadIdx.x & 63) >> 4;
    float sv = 0.01f * (threadIdx.x & 15);
    for (int h = 0; h < H; ++h) {
        if constexpr (MODE == 0) sv = mlp_step(wt, sv, 0.1f, g);
        else if constexpr (MODE == 1) {   // MFMAs only
            const float x1 = (g == 0) ? 0.1f : 0.0f;
            f32x4 a0 = w.b1[0], a1 = w.b1[1];
            a0 = CTK_MFMA(w.w1[0][0], sv, a0); a1 = CTK_MFMA(w.w1[1][0], sv, a1);
            a0 = CTK_MFMA(w.w1[0][1], x1, a0); a1 = CTK_MFMA(w.w1[1][1], x1, a1);
            f32x4 c0 = w.b2[0], c1 = w.b2[1];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float b = (j >> 2) ? a1[j & 3] : a0[j & 3]; c0 = CTK_MFMA(w.w2[0][j], b, c0); c1 = CTK_MFMA(w.w2[1][j], b, c1); }
            f32x4 o0 = w.b3, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; j += 2) { o0 = CTK_MFMA(w.w3[j], (j >> 2) ? c1[j & 3] : c0[j & 3], o0); o1 = CTK_MFMA(w.w3[j + 1], ((j + 1) >> 2) ? c1[(j + 1) & 3] : c0[(j + 1) & 3], o1); }
            sv = (o0[0] + o1[0]) * 1e-3f;
        } else {                          // tanh only: 16 per lane and step, chained
            f32x4 a0 = f32x4{sv, sv + 1.f, sv + 2.f, sv + 3.f}, a1 = a0 * 0.5f;
            f32x4 h0 = ctk_tanhf4(a0), h1 = ctk_tanhf4(a1);
            f32x4 c0 = ctk_tanhf4(h0 + h1), c1 = ctk_tanhf4(h0 - h1);
            sv = c0[0] + c1[1] + c0[2] + c1[3] + c0[1] + c1[0] + c0[3] + c1[2];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sv;
}

template <bool BARRIER>
__global__ __launch_bounds__(256) void k_split(const float* wperm, float* out, int H) {
    __shared__ __attribute__((aligned(16))) float ex[2 * MLP_PAIR_EX_FLOATS];
    const int wave = threadIdx.x >> 6, pair = wave >> 1, m = wave & 1, lane = threadIdx.x & 63, g = lane >> 4;
    const MlpFwdWS w = mlp_load_fwd_split(wperm, m);
    float* e = ex + pair * MLP_PAIR_EX_FLOATS;
    float sv = 0.01f * (threadIdx.x & 15);
    for (int h = 0; h < H; ++h) {
        if constexpr (BARRIER) sv = mlp_step_split(w, sv, g == 0 ? 0.1f : 0.0f, e, m).lo;
        else {
            float4* ex_h = reinterpret_cast<float4*>(e);
            float2* ex_o = reinterpret_cast<float2*>(e + 2 * 64 * 4);
            f32x4 a = w.b1;
            a = CTK_MFMA(w.w1[0], sv, a); a = CTK_MFMA(w.w1[1], g == 0 ? 0.1f : 0.0f, a);
            const f32x4 h1m = ctk_tanhf4(a);
            ex_h[m * 64 + lane] = make_float4(h1m[0], h1m[1], h1m[2], h1m[3]);
            const float4 o4 = ex_h[(m ^ 1) * 64 + lane];
            const f32x4 h1o = f32x4{o4.x, o4.y, o4.z, o4.w};
            const f32x4 h1a = m == 0 ? h1m : h1o, h1b = m == 0 ? h1o : h1m;
            f32x4 c0 = w.b2, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; j += 2) { c0 = CTK_MFMA(w.w2[j], (j >> 2) ? h1b[j & 3] : h1a[j & 3], c0); c1 = CTK_MFMA(w.w2[j + 1], ((j + 1) >> 2) ? h1b[(j + 1) & 3] : h1a[(j + 1) & 3], c1); }
            const f32x4 h2m = ctk_tanhf4(c0 + c1);
            f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = p0;
            p0 = CTK_MFMA(w.w3[0], h2m[0], p0); p1 = CTK_MFMA(w.w3[1], h2m[1], p1); p0 = CTK_MFMA(w.w3[2], h2m[2], p0); p1 = CTK_MFMA(w.w3[3], h2m[3], p1);
            const f32x4 pm = p0 + p1;
            ex_o[m * 64 + lane] = make_float2(pm[0], pm[1]);
            const float2 po = ex_o[(m ^ 1) * 64 + lane];
            sv = pm[0] + po.x + w.b3[0];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sv;
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
    const int H = 100;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device clock rate attribute: %d kHz\n", clk);
    for (int blocks : {64, 128, 256, 512, 1024, 2048}) {
        const float f = time_ms([&] { hipLaunchKernelGGL(k_unsplit<0>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        const float mf = time_ms([&] { hipLaunchKernelGGL(k_unsplit<1>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        const float th = time_ms([&] { hipLaunchKernelGGL(k_unsplit<2>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        const float sp = time_ms([&] { hipLaunchKernelGGL(k_split<true>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        const float nb = time_ms([&] { hipLaunchKernelGGL(k_split<false>, dim3(blocks), dim3(256), 0, 0, dw, dout, H); });
        printf("blocks %5d (x4 waves) H %d: per step ns  full %7.1f  mfma-only %7.1f  tanh-only %7.1f | split %7.1f  split-no-barrier %7.1f   (unsplit block = 64 traj, split block = 32 traj)\n",
               blocks, H, f * 1e6f / H, mf * 1e6f / H, th * 1e6f / H, sp * 1e6f / H, nb * 1e6f / H);
    }
    // one wave per block: a lone wave on a CU
    {
        const float f = time_ms([&] { hipLaunchKernelGGL(k_unsplit<0>, dim3(256), dim3(64), 0, 0, dw, dout, H); });
        const float mf = time_ms([&] { hipLaunchKernelGGL(k_unsplit<1>, dim3(256), dim3(64), 0, 0, dw, dout, H); });
        const float th = time_ms([&] { hipLaunchKernelGGL(k_unsplit<2>, dim3(256), dim3(64), 0, 0, dw, dout, H); });
        printf("256 blocks of ONE wave: per step ns  full %7.1f  mfma-only %7.1f  tanh-only %7.1f\n", f * 1e6f / H, mf * 1e6f / H, th * 1e6f / H);
    }
    return 0;
}
