// dep_probe.cpp -- what does a producer->consumer dependency between two back-to-back launches cost on MI355X?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dep_probe.cpp -o tools/_build/dep_probe && tools/_build/dep_probe
// A chain of NL launches replayed from a hipGraph.  Each launch: NB workgroups x 256 lanes; every workgroup reads
// 32 KB of static "weights", 8 KB of activations and writes 4 KB.  Variants differ in where the activations come
// from (static buffer / what the previous launch wrote) and in the cache policy of the stores and loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s @%d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ST_PLAIN = 0, ST_NT = 1, ST_SC = 2 };
enum { LD_PLAIN = 0, LD_NT = 1, LD_SC = 2 };

template <int ST> __device__ __forceinline__ void st4(f32x4* p, f32x4 v) {
    if constexpr (ST == ST_PLAIN) *p = v;
    else if constexpr (ST == ST_NT) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ f32x4 ld4(const f32x4* p) {
    if constexpr (LD == LD_PLAIN) return *p;
    else if constexpr (LD == LD_NT) return __builtin_nontemporal_load(p);
    else { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory"); return v; }
}

// act: [NB][512] f32x4 (8 KB per workgroup)   w: [NB][2048] f32x4 (32 KB per workgroup)   out: [NB][512] f32x4, only first 256 written... 
template <int ST, int LD>
__global__ __launch_bounds__(256) void k_step(const f32x4* __restrict__ act, const f32x4* __restrict__ w, f32x4* __restrict__ out, int shift) {
    const int b = blockIdx.x, t = threadIdx.x, nb = gridDim.x;
    f32x4 wv[8], av[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = w[((size_t)b * 8 + i) * 256 + t];
    const int src = (b + shift) % nb;                 // shift 0: same workgroup slot (same XCD); 1: neighbour (other XCD)
#pragma unroll
    for (int i = 0; i < 2; ++i) av[i] = ld4<LD>(act + ((size_t)src * 2 + i) * 256 + t);
    if constexpr (LD == LD_SC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 acc = av[0] + av[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += wv[i] * 1e-3f;
    st4<ST>(out + ((size_t)b * 2) * 256 + t, acc);
    st4<ST>(out + ((size_t)b * 2 + 1) * 256 + t, acc * 0.5f);
}

template <int ST, int LD>
static float run(const char* name, int NB, bool dep, int shift, int NL, f32x4* a0, f32x4* a1, f32x4* stat, f32x4* w, hipStream_t s, int nsets = 1) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NL; ++i) {
        f32x4* in = dep ? ((i & 1) ? a1 : a0) : stat;
        f32x4* out = (i & 1) ? a0 : a1;
        hipLaunchKernelGGL((k_step<ST, LD>), dim3(NB), dim3(256), 0, s, in, w + (size_t)(i % nsets) * NB * 2048, out, shift);
    }
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
    printf("NB=%3d %-46s : %6.2f us/launch\n", NB, name, best * 1000.f / NL);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int NBMAX = 128, NL = 256;
    f32x4 *a0, *a1, *stat, *w;
    CK(hipMalloc(&a0, NBMAX * 8192)); CK(hipMalloc(&a1, NBMAX * 8192)); CK(hipMalloc(&stat, NBMAX * 8192));
    const int MAXSETS = 128;
    CK(hipMalloc(&w, (size_t)MAXSETS * NBMAX * 32768));
    CK(hipMemset(a0, 0, NBMAX * 8192)); CK(hipMemset(a1, 0, NBMAX * 8192)); CK(hipMemset(stat, 0, NBMAX * 8192));
    CK(hipMemset(w, 0, (size_t)MAXSETS * NBMAX * 32768));
    for (int NB : {24, 96}) {
        run<ST_PLAIN, LD_PLAIN>("static activations (no dependency)", NB, false, 0, NL, a0, a1, stat, w, s);
        run<ST_PLAIN, LD_PLAIN>("dependent, same slot", NB, true, 0, NL, a0, a1, stat, w, s);
        run<ST_PLAIN, LD_PLAIN>("dependent, neighbour slot (other XCD)", NB, true, 1, NL, a0, a1, stat, w, s);
        run<ST_NT, LD_PLAIN>("dependent, other XCD, nontemporal stores", NB, true, 1, NL, a0, a1, stat, w, s);
        run<ST_SC, LD_PLAIN>("dependent, other XCD, sc0 sc1 stores", NB, true, 1, NL, a0, a1, stat, w, s);
        run<ST_PLAIN, LD_NT>("dependent, other XCD, nontemporal loads", NB, true, 1, NL, a0, a1, stat, w, s);
        run<ST_PLAIN, LD_SC>("dependent, other XCD, sc0 sc1 loads", NB, true, 1, NL, a0, a1, stat, w, s);
        run<ST_SC, LD_SC>("dependent, other XCD, sc0 sc1 both", NB, true, 1, NL, a0, a1, stat, w, s);
        run<ST_NT, LD_NT>("dependent, other XCD, nontemporal both", NB, true, 1, NL, a0, a1, stat, w, s);
    }
    // weight-footprint sweep: 96 workgroups (12 per XCD) x 32 KB = 384 KB per XCD per launch, cycling over nsets sets
    for (int ns : {1, 2, 4, 8, 10, 12, 16, 32, 64, 128}) {
        char nm[96]; snprintf(nm, sizeof nm, "static act, weights cycle %3d sets = %5.1f MB/XCD", ns, ns * 0.375);
        run<ST_PLAIN, LD_PLAIN>(nm, 96, false, 0, NL, a0, a1, stat, w, s, ns);
    }
    for (int ns : {1, 8, 32}) {
        char nm[96]; snprintf(nm, sizeof nm, "dep other XCD, weights cycle %3d sets = %5.1f MB/XCD", ns, ns * 0.375);
        run<ST_PLAIN, LD_PLAIN>(nm, 96, true, 1, NL, a0, a1, stat, w, s, ns);
    }
    return 0;
}
