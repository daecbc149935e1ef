// diag_mlp_l3.hip — candidate re-shaping of the MLP step's thin layers (correctness vs mlp_step, then time per step):
// (what csrc/ctk_mlp.h: mlp_step became after this measurement; the 16x16x4 form is kept here as the reference)
//   layer 3 (32 -> 4): eight v_mfma_f32_4x4x1 (2 passes each) on per-lane-group partial sums + a reduce-scatter over
//                      the four lane groups with v_permlane32_swap / v_permlane16_swap, instead of eight 16x16x4 MFMAs
//                      (8 passes each) of which 12 of 16 output rows are padding;
//   layer 1 (5 -> 32): the control input's column enters the accumulator on the VALU (b1 + w1u*u) so the MFMA part is
//                      one k-step (the four state components) instead of two.
// build: hipcc -O3 --offload-arch=gfx950 -I control_toolkit_amd/csrc -o tools/diag_mlp_l3 tools/diag_mlp_l3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "ctk_mlp.h"

struct ThinW { float w3n[8]; f32x4 w1u[2]; float b3g; };

// the round-1 form: both thin layers on 16x16x4 tiles (28 MFMAs)
CTK_DEV float mlp_step_wide(const MlpFwdW& w, float sv, float u, int g) {
    const float x1 = (g == 0) ? u : 0.0f;
    f32x4 a0 = w.b1[0], a1 = w.b1[1];
    a0 = CTK_MFMA(w.w1[0][0], sv, a0); a1 = CTK_MFMA(w.w1[1][0], sv, a1);
    a0 = CTK_MFMA(w.w1[0][1], x1, a0); a1 = CTK_MFMA(w.w1[1][1], x1, a1);
    f32x4 h1[2];
    h1[0] = ctk_tanhf4(a0); h1[1] = ctk_tanhf4(a1);
    f32x4 c0 = w.b2[0], c1 = w.b2[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float b = h1[j >> 2][j & 3]; c0 = CTK_MFMA(w.w2[0][j], b, c0); c1 = CTK_MFMA(w.w2[1][j], b, c1); }
    f32x4 h2[2];
    h2[0] = ctk_tanhf4(c0); h2[1] = ctk_tanhf4(c1);
    f32x4 o0 = w.b3, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) { o0 = CTK_MFMA(w.w3[j], h2[j >> 2][j & 3], o0); o1 = CTK_MFMA(w.w3[j + 1], h2[(j + 1) >> 2][(j + 1) & 3], o1); }
    return o0[0] + o1[0];
}

template <int MODE>   // 0: both changes, 1: layer 3 only, 2: layer 1 only
CTK_DEV float mlp_step_thin(const MlpFwdW& w, const ThinW& t, float sv, float u, int g) {
    f32x4 a0, a1;
    if constexpr (MODE != 1 && MODE != 3) {
        a0 = t.w1u[0] * u + w.b1[0]; a1 = t.w1u[1] * u + w.b1[1];
        a0 = CTK_MFMA(w.w1[0][0], sv, a0);
        a1 = CTK_MFMA(w.w1[1][0], sv, a1);
    } else {
        const float x1 = (g == 0) ? u : 0.0f;
        a0 = w.b1[0]; a1 = w.b1[1];
        a0 = CTK_MFMA(w.w1[0][0], sv, a0); a1 = CTK_MFMA(w.w1[1][0], sv, a1);
        a0 = CTK_MFMA(w.w1[0][1], x1, a0); a1 = CTK_MFMA(w.w1[1][1], x1, a1);
    }
    f32x4 h1[2];
    h1[0] = ctk_tanhf4(a0); h1[1] = ctk_tanhf4(a1);
    f32x4 c0 = w.b2[0], c1 = w.b2[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = h1[j >> 2][j & 3];
        c0 = CTK_MFMA(w.w2[0][j], b, c0);
        c1 = CTK_MFMA(w.w2[1][j], b, c1);
    }
    f32x4 h2[2];
    h2[0] = ctk_tanhf4(c0); h2[1] = ctk_tanhf4(c1);
    if constexpr (MODE == 2) {
        f32x4 o0 = w.b3, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            o0 = CTK_MFMA(w.w3[j], h2[j >> 2][j & 3], o0);
            o1 = CTK_MFMA(w.w3[j + 1], h2[(j + 1) >> 2][(j + 1) & 3], o1);
        }
        return o0[0] + o1[0];
    } else {
        f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = p0;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(t.w3n[j], h2[j >> 2][j & 3], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(t.w3n[j + 1], h2[(j + 1) >> 2][(j + 1) & 3], p1, 0, 0, 0);
        }
        const f32x4 p = p0 + p1;
        if constexpr (MODE == 3) {
            const int c = threadIdx.x & 15;
            float tot = 0.f;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float q0 = __shfl(p[0], 16 * gg + c), q1 = __shfl(p[1], 16 * gg + c), q2 = __shfl(p[2], 16 * gg + c), q3 = __shfl(p[3], 16 * gg + c);
                tot += g == 0 ? q0 : g == 1 ? q1 : g == 2 ? q2 : q3;
            }
            return tot + t.b3g;
        }
        const float s02 = swap_sum32(p[0], p[2]), s13 = swap_sum32(p[1], p[3]);
        return swap_sum16(s02, s13) + t.b3g;
    }
}

template <int MODE>   // -1: reference mlp_step
__global__ __launch_bounds__(256) void k_run(const float* wperm, const float* thin, float* out, int H, int write_all) {
    const MlpFwdW w = mlp_load_fwd(wperm);
    const int lane = threadIdx.x & 63, g = lane >> 4;
    ThinW t;
    const float* tp = thin + lane * 20;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.w3n[j] = tp[j];
    t.w1u[0] = f32x4{tp[8], tp[9], tp[10], tp[11]}; t.w1u[1] = f32x4{tp[12], tp[13], tp[14], tp[15]};
    t.b3g = tp[16];
    // state: component g of trajectory c
    float sv = 0.05f * (float)((threadIdx.x & 15) - 8) + 0.3f * g;
    for (int h = 0; h < H; ++h) {
        const float u = 0.02f * (float)((lane & 15) + h % 5) - 0.2f;
        if constexpr (MODE < 0) sv = mlp_step_wide(w, sv, u, g);
        else sv = mlp_step_thin<MODE>(w, t, sv, u, g);
        if (write_all) out[(size_t)h * 64 + lane] = sv;
    }
    if (!write_all) out[blockIdx.x * 256 + threadIdx.x] = sv;
}

// host tables: the library's forward layout (csrc/ctk_api.hip: permute_mlp_weights, S = 4, C = 1) + the thin-layer additions
static void build(const std::vector<float>& raw, std::vector<float>& fwd, std::vector<float>& thin) {
    const int S = 4, I = 5;
    const float* W1 = raw.data();          const float* b1 = W1 + 32 * I;
    const float* W2 = b1 + 32;             const float* b2 = W2 + 32 * 32;
    const float* W3 = b2 + 32;             const float* b3 = W3 + S * 32;
    fwd.assign(64 * MLP_FWD_PER_LANE, 0.f); thin.assign(64 * 20, 0.f);
    auto io_of_row = [](int row) { return 4 * (row % 4) + row / 4; };
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, g = l >> 4;
        float* f = fwd.data() + (size_t)l * MLP_FWD_PER_LANE;
        for (int m = 0; m < 2; ++m)
            for (int ks = 0; ks < 2; ++ks) { const int kk = 4 * ks + g; f[m * 2 + ks] = kk < I ? W1[(16 * m + i) * I + kk] : 0.f; }
        for (int mo = 0; mo < 2; ++mo)
            for (int j = 0; j < 8; ++j) f[4 + mo * 8 + j] = W2[(16 * mo + i) * 32 + mlp_hid(j, g)];
        const int out_i = (i % 4 < 2 && io_of_row(i) < S) ? io_of_row(i) : -1;
        for (int j = 0; j < 8; ++j) f[20 + j] = out_i >= 0 ? W3[out_i * 32 + mlp_hid(j, g)] : 0.f;
        for (int m = 0; m < 2; ++m)
            for (int r = 0; r < 4; ++r) { f[28 + m * 4 + r] = b1[16 * m + 4 * g + r]; f[36 + m * 4 + r] = b2[16 * m + 4 * g + r]; }
        for (int r = 0; r < 4; ++r) f[44 + r] = (r < 2 && 4 * r + g < S) ? b3[4 * r + g] : 0.f;
        float* t = thin.data() + (size_t)l * 20;
        for (int j = 0; j < 8; ++j) t[j] = W3[(l & 3) * 32 + mlp_hid(j, g)];          // A of the 4x4x1 blocks: output row l % 4, this lane group's hidden units
        for (int m = 0; m < 2; ++m)
            for (int r = 0; r < 4; ++r) t[8 + m * 4 + r] = W1[(16 * m + 4 * g + r) * I + 4];   // the input's column at the accumulator rows of this lane
        t[16] = b3[g];
    }
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
    std::mt19937 gen(3); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> raw(5 * 32 + 32 + 32 * 32 + 32 + 4 * 32 + 4);
    for (auto& v : raw) v = 0.25f * nd(gen);
    std::vector<float> fwd, thin;
    build(raw, fwd, thin);
    float *dw, *dt, *dout;
    hipMalloc(&dw, fwd.size() * 4); hipMemcpy(dw, fwd.data(), fwd.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dt, thin.size() * 4); hipMemcpy(dt, thin.data(), thin.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, 4096 * 256 * 4);
    const int Hc = 20;
    std::vector<float> ref(Hc * 64), got(Hc * 64);
    hipLaunchKernelGGL(k_run<-1>, dim3(1), dim3(64), 0, 0, dw, dt, dout, Hc, 1);
    hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost);
    auto check = [&](const char* name) {
        hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0; for (size_t i = 0; i < ref.size(); ++i) worst = std::fmax(worst, std::fabs((double)got[i] - ref[i]));
        printf("%-14s max |diff| vs mlp_step over %d steps: %.3g   (sample: %g vs %g)\n", name, Hc, worst, got[5 * 64 + 37], ref[5 * 64 + 37]);
    };
    hipLaunchKernelGGL(k_run<0>, dim3(1), dim3(64), 0, 0, dw, dt, dout, Hc, 1); check("both");
    hipLaunchKernelGGL(k_run<1>, dim3(1), dim3(64), 0, 0, dw, dt, dout, Hc, 1); check("layer 3 only");
    hipLaunchKernelGGL(k_run<2>, dim3(1), dim3(64), 0, 0, dw, dt, dout, Hc, 1); check("layer 1 only");
    hipLaunchKernelGGL(k_run<3>, dim3(1), dim3(64), 0, 0, dw, dt, dout, Hc, 1); check("layer 3 shfl");
    const int H = 100;
    for (int blocks : {128, 256, 512, 2048}) {
        const float r = time_ms([&] { hipLaunchKernelGGL(k_run<-1>, dim3(blocks), dim3(256), 0, 0, dw, dt, dout, H, 0); });
        const float b = time_ms([&] { hipLaunchKernelGGL(k_run<0>, dim3(blocks), dim3(256), 0, 0, dw, dt, dout, H, 0); });
        const float l3 = time_ms([&] { hipLaunchKernelGGL(k_run<1>, dim3(blocks), dim3(256), 0, 0, dw, dt, dout, H, 0); });
        const float l1 = time_ms([&] { hipLaunchKernelGGL(k_run<2>, dim3(blocks), dim3(256), 0, 0, dw, dt, dout, H, 0); });
        printf("blocks %5d (x4 waves) H %d: per step ns  mlp_step %7.1f | thin layer 3 %7.1f | thin layer 1 %7.1f | both %7.1f\n",
               blocks, H, r * 1e6f / H, l3 * 1e6f / H, l1 * 1e6f / H, b * 1e6f / H);
    }
    return 0;
}
