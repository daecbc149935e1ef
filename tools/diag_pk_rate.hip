// diag_pk_rate.hip — issue rate of v_fma_f32 vs v_pk_fma_f32 on a full chip (8 independent chains per lane, 4 waves per SIMD)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/diag_pk_rate tools/diag_pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 v = {x[i], x[i + 1]};
                v = __builtin_elementwise_fma(v, f2{a, a}, f2{b, b});
                x[i] = v.x; x[i + 1] = v.y;
            }
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    const int iters = 20000, blocks = 1024;   // 4 waves per SIMD
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fma = (double)blocks * 256 * iters * 16;
        printf("%s: %.3f ms, %.1f TFLOP/s (fp32 FMA = 2 flop), %.2f cycles per wave-instruction-slot at 2.4 GHz\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ", ms,
               2 * fma / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)iters * (mode ? 8 : 16) * 4));
    }
    return 0;
}
