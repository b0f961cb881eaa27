// loadpath_probe.cpp -- how fast can ONE workgroup pull N KB through its CU's load path as 1 KB wave loads (the shape of the
// latency kernels' load phases: k_attn_mid pulls ~272 KB per workgroup), as a function of the number of waves that issue them?
//   hipcc --offload-arch=gfx950 -O3 tools/loadpath_probe.cpp -o tools/_build/loadpath_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s @%d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// every wave issues `per_wave` independent 1 KB loads (all in flight at once, like the kernels do), then reduces them
template <int PER_WAVE>
__global__ void k_pull(const f32x4* src, float* out, int frag_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* p = src + ((size_t)blockIdx.x * frag_per_wg + (size_t)wave * PER_WAVE) * 64 + lane;
    f32x4 v[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) v[i] = p[(size_t)i * 64];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) s += v[i];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[threadIdx.x] = s[0];
}
template <int PER_WAVE>
static void run(const char* what, int waves, const f32x4* src, float* out, int n_wg, hipStream_t st) {
    const int frag_per_wg = waves * PER_WAVE;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_pull<PER_WAVE>, dim3(n_wg), dim3(64 * waves), 0, st, src, out, frag_per_wg);
    CK(hipEventRecord(e0, st));
    const int N = 400;
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_pull<PER_WAVE>, dim3(n_wg), dim3(64 * waves), 0, st, src, out, frag_per_wg);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %2d waves x %3d loads = %3d KB per workgroup, %3d workgroups: %6.2f us per launch\n", what, waves, PER_WAVE, frag_per_wg, n_wg, 1000.f * ms / N);
}
int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t frags = 96 * 512;                  // 48 MB: every workgroup its own region (L2 / Infinity-Cache resident after warm-up)
    f32x4* src; CK(hipMalloc(&src, frags * 1024)); CK(hipMemset(src, 0, frags * 1024));
    float* out; CK(hipMalloc(&out, 4096));
    hipLaunchKernelGGL(k_pull<1>, dim3(1), dim3(64), 0, st, src, out, 1);
    CK(hipStreamSynchronize(st));
    for (int n_wg : {96, 6}) {
        run<1>("launch floor (1 KB)", 1, src, out, n_wg, st);
        run<68>("k_attn_mid shape", 4, src, out, n_wg, st);        // 4 x 68 = 272 KB
        run<34>("same bytes, 8 waves", 8, src, out, n_wg, st);
        run<17>("same bytes, 16 waves", 16, src, out, n_wg, st);
        run<34>("half the bytes, 4 waves", 4, src, out, n_wg, st);
        run<16>("64 KB, 4 waves", 4, src, out, n_wg, st);
    }
    return 0;
}
