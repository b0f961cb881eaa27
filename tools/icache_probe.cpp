// icache_probe.cpp -- is a launch's critical path sensitive to CODE SIZE (instruction fetch) or only to the number of
// instructions executed?  Same dependent-FMA count, straight-line vs rolled loop, 1 wave per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/icache_probe.cpp -o tools/_build/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s @%d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <int N, int CHAINS, bool ROLLED>
__global__ __launch_bounds__(64) void k_fma(const float* __restrict__ in, float* __restrict__ out) {
    float v[CHAINS];
    const float a = in[threadIdx.x], b = in[64 + threadIdx.x];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = a + c;
    if constexpr (ROLLED) {
#pragma unroll 1
        for (int i = 0; i < N / 64; ++i) {
#pragma unroll
            for (int j = 0; j < 64 / CHAINS; ++j)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) v[c] = __builtin_fmaf(v[c], a, b);
            asm volatile("" ::: "memory");
        }
    } else {
#pragma unroll
        for (int i = 0; i < N / CHAINS; ++i)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) v[c] = __builtin_fmaf(v[c], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int N, int CHAINS, bool ROLLED>
static void run(const char* name, float* in, float* out, hipStream_t s, int NB) {
    const int NL = 128;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NL; ++i) hipLaunchKernelGGL((k_fma<N, CHAINS, ROLLED>), dim3(NB), dim3(64), 0, s, in, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("NB=%3d N=%5d chains=%d %-8s %-10s: %6.2f us/launch  (%.2f ns per FMA)\n", NB, N, CHAINS, ROLLED ? "rolled" : "straight", name,
           best * 1000.f / NL, (best * 1e6f / NL - 1640.f) / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(in, 0, 4096));
    for (int NB : {96, 1}) {
        run<256, 1, false>("", in, out, s, NB);
        run<1024, 1, false>("", in, out, s, NB);
        run<4096, 1, false>("", in, out, s, NB);
        run<1024, 1, true>("", in, out, s, NB);
        run<4096, 1, true>("", in, out, s, NB);
        run<1024, 8, false>("", in, out, s, NB);
        run<4096, 8, false>("", in, out, s, NB);
        run<4096, 8, true>("", in, out, s, NB);
        run<16384, 8, false>("", in, out, s, NB);
        run<16384, 8, true>("", in, out, s, NB);
    }
    return 0;
}
