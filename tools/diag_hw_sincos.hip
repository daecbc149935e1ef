// diag_hw_sincos.hip — accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) against double, for the rollout recurrence's
// unchecked sincos: max |error| over a dense sweep of [-R, R], (a) t = x * fl(1/2pi) directly, (b) with a two-term product
// (t = x*hi + x*lo) to see how much of the error is the argument's rounding.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/diag_hw_sincos tools/diag_hw_sincos.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s1, float* c1, float* s2, float* c2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = x[i] * 0.15915494309189535f;
    s1[i] = __builtin_amdgcn_sinf(t); c1[i] = __builtin_amdgcn_cosf(t);
    const float hi = 0.15915494f, lo = 0.15915494309189535 - (double)0.15915494f;
    const float t2 = fmaf(x[i], lo, x[i] * hi);
    const float tf = t2 - rintf(t2);
    s2[i] = __builtin_amdgcn_sinf(tf); c2[i] = __builtin_amdgcn_cosf(tf);
}
int main() {
    const int n = 1 << 22;
    for (double R : {3.2, 10.0, 100.0, 1000.0}) {
        std::vector<float> x(n);
        for (int i = 0; i < n; ++i) x[i] = (float)(-R + 2.0 * R * (i + 0.37) / n);
        float *dx, *d[4];
        hipMalloc(&dx, n * 4); hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        for (auto& p : d) hipMalloc(&p, n * 4);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], n);
        std::vector<float> o[4];
        for (int j = 0; j < 4; ++j) { o[j].resize(n); hipMemcpy(o[j].data(), d[j], n * 4, hipMemcpyDeviceToHost); }
        double e[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const double xs = (double)x[i];
            e[0] = std::fmax(e[0], std::fabs(o[0][i] - std::sin(xs))); e[1] = std::fmax(e[1], std::fabs(o[1][i] - std::cos(xs)));
            e[2] = std::fmax(e[2], std::fabs(o[2][i] - std::sin(xs))); e[3] = std::fmax(e[3], std::fabs(o[3][i] - std::cos(xs)));
        }
        printf("|x| <= %7.1f: max abs error  v_sin %.3e  v_cos %.3e   | two-term argument + rint: v_sin %.3e  v_cos %.3e\n", R, e[0], e[1], e[2], e[3]);
        hipFree(dx); for (auto& p : d) hipFree(p);
    }
    return 0;
}
