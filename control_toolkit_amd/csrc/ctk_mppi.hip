// ctk_mppi.hip — MPPI on gfx950 (replaces reference Optimizers/optimizer_mppi.py:181-193).
//
//   ctk_mppi_rollout_ode   one thread per trajectory, 64-thread (one-wave) blocks:
//        LDS tile of the block's [64,P] perturbations (coalesced HBM read, scaled by stdev)
//        -> H fused steps {interpolate, add nominal, clip, stage cost, MPPI correction, Euler step}
//        -> J[n]; block-local soft-min partial (rho_b, a_b, b_b[P]) by wave shuffles + LDS
//           column sums (sum_n e_n * delta_u_n is linear in the inducing points, so the
//           reduction runs over P values per trajectory instead of H).
//   ctk_mppi_merge         merges partial records {rho, a, b[P]} (blocks of one GPU, or the
//        all-gathered records of several GPUs — SURVEY.md 8e) and either emits one record or
//        applies the update u_nom <- clip(shift(u_nom) + interp(b)/a)   (:163-168,:184,:190).
#include "ctk_rollout.h"
#include "ctk_launch.h"

constexpr int MPPI_BLOCK = 64;

template <bool LOG>
__global__ __launch_bounds__(MPPI_BLOCK) void ctk_mppi_rollout_ode(RolloutArgs a, EnvK k, MppiK m,
                                                                   const float* __restrict__ samples,
                                                                   const float* __restrict__ u_nom,
                                                                   float* __restrict__ parts) {
    extern __shared__ float lds[];
    const int P = a.P, stride = tile_stride(P);
    float* tile = lds;                        // [64][stride]
    float* e_s = lds + MPPI_BLOCK * stride;   // [64]
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * MPPI_BLOCK;
    const int n = row0 + lane;
    const bool valid = n < a.N;

    load_tile<MPPI_BLOCK>(tile, samples, a, row0, m.stdev, /*normal*/ 0);
    __syncthreads();

    const float* my = tile + lane * stride;
    const int H = a.H;
    float corr = 0.0f;
    float J = rollout_ode<LOG, LOG>(a, k, n, valid, [&](int h) {
        const InterpEntry e = a.interp[h];                       // wave-uniform -> scalar loads
        const float du = my[e.i0] * e.w0 + my[e.i0 + 1] * e.w1;  // Interpolator.py:97-106
        const float un = u_nom[min(h + 1, H - 1)];               // optimizer_mppi.py:184 (shift)
        const float u = fminf(fmaxf(un + du, a.lo), a.hi);       // :186-187
        corr += m.cc * (m.k_dd * (du * du) + m.R * u * du + m.k_uu * (u * u));  // :154-155
        return u;
    });
    J += corr;
    if (valid) a.J[n] = J;

    // block-local soft-min partial (optimizer_mppi.py:163-168 restricted to this block)
    const float Jv = valid ? J : INFINITY;
    const float rho = wave_min(Jv);
    const float e = valid ? expf(m.neg_inv_lbd * (J - rho)) : 0.0f;
    const float asum = wave_sum(e);
    e_s[lane] = e;
    __syncthreads();
    float* rec = parts + (size_t)blockIdx.x * (2 + P);
    if (lane == 0) { rec[0] = rho; rec[1] = asum; }
    for (int p = lane; p < P; p += MPPI_BLOCK) {
        float acc = 0.0f;
#pragma unroll 8
        for (int r = 0; r < MPPI_BLOCK; ++r) acc += e_s[r] * tile[r * stride + p];
        rec[2 + p] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// merge of partial records.  grid.x blocks; block b merges records [b*per_block, ...).
// FINAL=false: writes one record per block to `out_rec`.
// FINAL=true (grid.x == 1): applies the MPPI update and publishes u.
// ---------------------------------------------------------------------------------------------
constexpr int MERGE_BLOCK = 256;

template <bool FINAL>
__global__ __launch_bounds__(MERGE_BLOCK) void ctk_mppi_merge(const float* __restrict__ parts, int n_parts, int per_block,
                                                             int P, float neg_inv_lbd, float* __restrict__ out_rec,
                                                             // FINAL only:
                                                             int H, const InterpEntry* __restrict__ interp,
                                                             const float* __restrict__ u_nom_in,
                                                             float* __restrict__ u_nom_out, float lo, float hi,
                                                             float* __restrict__ u_dev, float* __restrict__ u_host) {
    extern __shared__ float lds[];
    float* red = lds;                 // [MERGE_BLOCK / 64] cross-wave scratch
    float* b_s = lds + 8;             // [P] merged numerator
    float* sc_s = b_s + P + 1;        // [chunk] per-record rescale factors
    constexpr int CHUNK = 1024;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int first = blockIdx.x * per_block;
    const int cnt = min(per_block, n_parts - first);
    const int rs = 2 + P;
    const float* base = parts + (size_t)first * rs;

    // rho = min over records
    float r = INFINITY;
    for (int i = t; i < cnt; i += MERGE_BLOCK) r = fminf(r, base[(size_t)i * rs]);
    r = wave_min(r);
    if (lane == 0) red[wave] = r;
    __syncthreads();
    float rho = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    __syncthreads();

    float a_acc = 0.0f;
    float b_acc[4] = {0.f, 0.f, 0.f, 0.f};   // thread t owns columns t, t+256, ... (P <= 1024)
    for (int c0 = 0; c0 < cnt; c0 += CHUNK) {
        const int cn = min(CHUNK, cnt - c0);
        for (int i = t; i < cn; i += MERGE_BLOCK) {
            const float* rec = base + (size_t)(c0 + i) * rs;
            const float sc = expf(neg_inv_lbd * (rec[0] - rho));   // e^{-(rho_r - rho)/lambda}
            sc_s[i] = sc;
            a_acc += rec[1] * sc;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = t + j * MERGE_BLOCK;
            if (p < P) {
                float acc = b_acc[j];
                for (int i = 0; i < cn; ++i) acc += base[(size_t)(c0 + i) * rs + 2 + p] * sc_s[i];
                b_acc[j] = acc;
            }
        }
        __syncthreads();
    }
    a_acc = wave_sum(a_acc);
    if (lane == 0) red[wave] = a_acc;
    __syncthreads();
    const float a_tot = red[0] + red[1] + red[2] + red[3];

    if constexpr (!FINAL) {
        float* rec = out_rec + (size_t)blockIdx.x * rs;
        if (t == 0) { rec[0] = rho; rec[1] = a_tot; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = t + j * MERGE_BLOCK;
            if (p < P) rec[2 + p] = b_acc[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = t + j * MERGE_BLOCK;
            if (p < P) b_s[p] = b_acc[j];
        }
        if (t == 0) b_s[P] = 0.0f;   // pad read by i0+1 when P == 1
        __syncthreads();
        for (int h = t; h < H; h += MERGE_BLOCK) {
            const InterpEntry e = interp[h];
            const float w = (b_s[e.i0] * e.w0 + b_s[e.i0 + 1] * e.w1) / a_tot;
            const float un = u_nom_in[min(h + 1, H - 1)];
            const float o = fminf(fmaxf(un + w, lo), hi);   // optimizer_mppi.py:190
            u_nom_out[h] = o;
            if (h == 0) { *u_dev = o; *u_host = o; }        // :191 u = u_nom[0,0,:]
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
const char* ctk_mppi_rollout_ode_name(bool log) {
    return log ? "ctk_mppi_rollout_ode<true>" : "ctk_mppi_rollout_ode<false>";
}

int ctk_mppi_num_blocks_ode(int N) { return (N + MPPI_BLOCK - 1) / MPPI_BLOCK; }

size_t ctk_mppi_rollout_ode_lds(int P) { return (size_t)(MPPI_BLOCK * tile_stride(P) + MPPI_BLOCK) * sizeof(float); }

hipError_t ctk_launch_mppi_rollout_ode(hipStream_t st, const RolloutArgs& a, const EnvK& k, const MppiK& m,
                                       const float* samples, const float* u_nom, float* parts, bool log) {
    const int blocks = ctk_mppi_num_blocks_ode(a.N);
    const size_t lds = ctk_mppi_rollout_ode_lds(a.P);
    if (log)
        hipLaunchKernelGGL(ctk_mppi_rollout_ode<true>, dim3(blocks), dim3(MPPI_BLOCK), lds, st, a, k, m, samples, u_nom, parts);
    else
        hipLaunchKernelGGL(ctk_mppi_rollout_ode<false>, dim3(blocks), dim3(MPPI_BLOCK), lds, st, a, k, m, samples, u_nom, parts);
    return hipGetLastError();
}

static size_t merge_lds(int P) { return (size_t)(8 + P + 1 + 1024) * sizeof(float); }

hipError_t ctk_launch_mppi_merge_partial(hipStream_t st, const float* parts, int n_parts, int per_block, int P,
                                         float neg_inv_lbd, float* out_rec) {
    const int blocks = (n_parts + per_block - 1) / per_block;
    hipLaunchKernelGGL(ctk_mppi_merge<false>, dim3(blocks), dim3(MERGE_BLOCK), merge_lds(P), st, parts, n_parts, per_block, P,
                       neg_inv_lbd, out_rec, 0, (const InterpEntry*)nullptr, (const float*)nullptr, (float*)nullptr, 0.f, 0.f,
                       (float*)nullptr, (float*)nullptr);
    return hipGetLastError();
}

hipError_t ctk_launch_mppi_update(hipStream_t st, const float* parts, int n_parts, int P, float neg_inv_lbd, int H,
                                  const InterpEntry* interp, const float* u_nom_in, float* u_nom_out, float lo, float hi,
                                  float* u_dev, float* u_host) {
    hipLaunchKernelGGL(ctk_mppi_merge<true>, dim3(1), dim3(MERGE_BLOCK), merge_lds(P), st, parts, n_parts, n_parts, P, neg_inv_lbd,
                       (float*)nullptr, H, interp, u_nom_in, u_nom_out, lo, hi, u_dev, u_host);
    return hipGetLastError();
}
