// Can the HOST store a request straight into DEVICE memory (fine-grained VRAM through the PCIe BAR) so that a resident kernel polls LOCAL memory?
// Round trip of one word: host stores cmd into (c) fine-grained device memory / (b) pinned host memory; resident kernel answers into pinned host memory.
//   hipcc -O3 --offload-arch=gfx950 tools/diag_mailbox_vram.hip -o tools/diag_mailbox_vram
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void resident(const unsigned* cmd, unsigned* ack, int blocks_answering) {
    const unsigned long long t0 = wall_clock64();
    unsigned last = 0;
    while (true) {
        const unsigned c = __hip_atomic_load(cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (c == 0xFFFFFFFFu) break;
        if (c != last) {
            last = c;
            if ((int)blockIdx.x < blocks_answering && threadIdx.x == 0) __hip_atomic_store(ack + blockIdx.x, c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (wall_clock64() - t0 > 200000000ull) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

int main() {
    unsigned *ack, *ack_d, *cmd_h, *cmd_hd, *cmd_v = nullptr;
    CK(hipHostMalloc((void**)&ack, 256, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&ack_d, ack, 0));
    CK(hipHostMalloc((void**)&cmd_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&cmd_hd, cmd_h, 0));
    hipError_t e = hipExtMallocWithFlags((void**)&cmd_v, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s  ptr %p\n", hipGetErrorString(e), (void*)cmd_v);
    hipPointerAttribute_t at{};
    if (e == hipSuccess && hipPointerGetAttributes(&at, cmd_v) == hipSuccess) printf("  type %d  hostPointer %p devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int iters = 3000;
    auto stats = [&](std::vector<double>& v, const char* name) {
        std::sort(v.begin(), v.end());
        printf("%-70s median %6.2f us  p10 %6.2f  p90 %6.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
    };
    for (int mode = 0; mode < 2; ++mode) {
        unsigned* cmd_host_view = mode == 0 ? cmd_h : cmd_v;       // what the host stores to
        unsigned* cmd_dev_view = mode == 0 ? cmd_hd : cmd_v;
        if (mode == 1 && (e != hipSuccess || cmd_v == nullptr)) break;
        if (mode == 1) { CK(hipMemset(cmd_v, 0, 64)); CK(hipDeviceSynchronize()); printf("host store into device memory ...\n"); fflush(stdout); }
        for (int blocks : {1, 16}) {
            __atomic_store_n(cmd_host_view, 0u, __ATOMIC_RELEASE);
            for (int i = 0; i < 64; ++i) ack[i] = 0;
            hipLaunchKernelGGL(resident, dim3(16), dim3(64), 0, st, cmd_dev_view, ack_d, blocks);
            std::vector<double> v;
            for (int i = 1; i <= iters; ++i) {
                auto t0 = std::chrono::steady_clock::now();
                __atomic_store_n(cmd_host_view, (unsigned)i, __ATOMIC_RELEASE); __builtin_ia32_sfence();
                bool all = false;
                long spins = 0;
                while (!all && spins < 50000000) { all = true; for (int b = 0; b < blocks; ++b) all &= (((volatile unsigned*)ack)[b] == (unsigned)i); ++spins; }
                if (!all) { printf("  no answer to request %d\n", i); break; }
                v.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            __atomic_store_n(cmd_host_view, 0xFFFFFFFFu, __ATOMIC_RELEASE); __builtin_ia32_sfence();
            CK(hipStreamSynchronize(st));
            char nm[128]; snprintf(nm, sizeof nm, "mailbox in %s, %d block(s) answer", mode == 0 ? "pinned HOST memory" : "fine-grained DEVICE memory", blocks);
            if (!v.empty()) stats(v, nm);
        }
    }
    return 0;
}
