// device side of tools/overlap_probe.cpp
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 tools/overlap_kernels.hip -o tools/_build/overlap_kernels.hsaco
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct OvArgs {
    const f32x4* act;        // [nb][2][256] what the previous packet wrote (uncached memory)
    const f32x4* w;          // [nb][WCH][256] "weights" (cached memory, constant)
    f32x4* out;              // [nb][2][256] (uncached memory)
    unsigned* wait_ctr;      // overlapped mode: the predecessor's arrival counter(s), 16 words apart per shard; null: plain dependent kernel
    unsigned* arrive_ctr;    // overlapped mode: this packet's arrival counter(s)
    unsigned* err;           // [0] = number of workgroups that gave up waiting
    int nb;                  // workgroups
    int wait_target;         // arrivals to wait for (0: the first packet of a chain)
    int shards;              // 1: one counter; 8: one per XCD (arrive on shard XCC_ID, wait for the sum)
    int sleep;               // s_sleep argument between polls
    int work;                // dependent FMA chain length between the loads and the stores (~4 cycles each): the kernel's "busy" time
    unsigned* l2_ctr;        // two-level arrival (shards == 0): per-XCD counters in CACHED memory (agent-scope atomics execute in the XCD's L2),
                             // 16 words apart; the last arriver of an XCD (workgroup b is assumed on XCD b % 8 -- checked, a miss arrives
                             // directly) bumps the one uncached counter the consumers poll: 8 uncached atomics per link instead of nb
};

// one link of a dependent chain shaped like a batch-1 GEMM of the step: 32 KB of weights + 8 KB of the predecessor's output per
// workgroup in, 8 KB out.  Overlapped mode: the packet carries no barrier bit; the weights are requested FIRST, then lane 0 polls the
// predecessor's arrival counter, then the activations are loaded; at the end the workgroup drains its stores and arrives.
extern "C" __global__ __launch_bounds__(256) void k_link(const OvArgs a) {
    const int b = blockIdx.x, t = threadIdx.x;
    constexpr int WCH = 8;
    f32x4 wv[WCH], av[2];
#pragma unroll
    for (int i = 0; i < WCH; ++i) wv[i] = a.w[((size_t)b * WCH + i) * 256 + t];
    if (a.wait_ctr && a.wait_target > 0) {
        if (t == 0) {
            int spins = 0;
            for (;;) {
                unsigned got = 0;
                for (int s = 0; s < (a.shards ? a.shards : 1); ++s) got += __hip_atomic_load(a.wait_ctr + 16 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (got >= (unsigned)a.wait_target) break;
                if (++spins > (1 << 14)) { __hip_atomic_fetch_add(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                if (a.sleep == 1) __builtin_amdgcn_s_sleep(1); else if (a.sleep == 4) __builtin_amdgcn_s_sleep(4); else if (a.sleep == 16) __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
    }
    const int src = (b + 1) % a.nb;
#pragma unroll
    for (int i = 0; i < 2; ++i) av[i] = a.act[((size_t)src * 2 + i) * 256 + t];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WCH; ++i) acc += wv[i];
    float d = av[0][0] * 0.f;                       // a dependent chain on the loaded data: the "MFMA phases" of a real kernel
    for (int i = 0; i < a.work; ++i) d = __builtin_fmaf(d, 0.5f, acc[0]);
    a.out[((size_t)b * 2) * 256 + t] = av[0] + 1.0f + acc + d;          // weights are zero: acc == 0, d == 0
    a.out[((size_t)b * 2 + 1) * 256 + t] = av[1] + 1.0f + acc + d;
    if (a.arrive_ctr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's stores have been acknowledged (uncached memory: they are visible)
        __syncthreads();
        if (t == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;      // HW_REG_XCC_ID, bits [3:0]
            if (a.shards == 0) {
                const unsigned home = (unsigned)b & 7u;                          // where dispatch order normally puts workgroup b
                const unsigned mine = ((unsigned)a.nb + 7u - home) >> 3;         // workgroups with this home XCD
                if (xcc != home) {                                               // displaced: arrive for my home group directly
                    const unsigned old = __hip_atomic_fetch_add(a.l2_ctr + 16 * (8 + home), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    (void)old;
                    __hip_atomic_fetch_add(a.arrive_ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // counted separately: [1] = displaced
                } else {
                    const unsigned old = __hip_atomic_fetch_add(a.l2_ctr + 16 * home, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + 1 == mine) __hip_atomic_fetch_add(a.arrive_ctr, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                __hip_atomic_fetch_add(a.arrive_ctr + 16 * (a.shards > 1 ? xcc : 0u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
