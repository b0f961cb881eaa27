// device side of tools/aql_probe.cpp:  hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 tools/aql_kernels.hip -o tools/_build/aql_kernels.hsaco
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void k_null(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
// every workgroup reads 8 KB of what the previous launch wrote (neighbour slot -> another XCD) + 32 KB of weights,
// writes 8 KB; `check`: out[i] must equal in[i'] + 1 for every launch in the chain (verifies cross-launch visibility)
extern "C" __global__ __launch_bounds__(256) void k_dep(const f32x4* __restrict__ act, const f32x4* __restrict__ w,
                                                          f32x4* __restrict__ out, int nb) {
    // nb is an explicit argument: gridDim.x would come from the HIDDEN kernel arguments the HIP runtime appends, which
    // a hand-written AQL packet does not provide
    const int b = blockIdx.x, t = threadIdx.x;
    f32x4 wv[8], av[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = w[((size_t)b * 8 + i) * 256 + t];
    const int src = (b + 1) % nb;
#pragma unroll
    for (int i = 0; i < 2; ++i) av[i] = act[((size_t)src * 2 + i) * 256 + t];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += wv[i];
    out[((size_t)b * 2) * 256 + t] = av[0] + 1.0f + acc;       // weights are zero: acc == 0
    out[((size_t)b * 2 + 1) * 256 + t] = av[1] + 1.0f + acc;
}

// same, but the output is written with agent-scope (write-through, sc1) stores: is the packet's RELEASE fence (the L2
// write-back, ~0.66 us of the 2.11 us per dependent packet) still needed then?
extern "C" __global__ __launch_bounds__(256) void k_dep_wt(const f32x4* __restrict__ act, const f32x4* __restrict__ w,
                                                             f32x4* __restrict__ out, int nb) {
    const int b = blockIdx.x, t = threadIdx.x;
    f32x4 wv[8], av[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = w[((size_t)b * 8 + i) * 256 + t];
    const int src = (b + 1) % nb;
#pragma unroll
    for (int i = 0; i < 2; ++i) av[i] = act[((size_t)src * 2 + i) * 256 + t];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += wv[i];
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const u64x2 v = __builtin_bit_cast(u64x2, av[i] + 1.0f + acc);
        unsigned long long* p = (unsigned long long*)(out + ((size_t)b * 2 + i) * 256 + t);
        __hip_atomic_store(p, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
