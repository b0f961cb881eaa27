// xcd_probe.cpp -- MI355X micro-probes that size the design of a persistent single-XCD step kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_probe.cpp -o gpurun_out/xcd_probe && gpurun_out/xcd_probe
// 1. workgroup -> XCD placement for a 1-workgroup-per-CU grid
// 2. dependent-launch floor: chain of null kernels / tiny dependent kernels (eager and hipGraph)
// 3. persistent kernel on ONE XCD: barrier latency (L2 atomics, relaxed sc1 polling) for P workgroups, and a
//    producer->consumer visibility check: plain stores + s_waitcnt vmcnt(0) + barrier + sc1 loads, every word checked
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s @%d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

__global__ void k_where(unsigned* out) {
    __shared__ char pad[90 * 1024];          // 1 workgroup per CU
    if (threadIdx.x == 0) { pad[0] = 1; out[blockIdx.x * 2] = xcc_id(); out[blockIdx.x * 2 + 1] = hw_id(); }
}
__global__ void k_null(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void k_dep(const float* __restrict__ in, float* __restrict__ out, int n) {
    // every block reads a 32 KB slice of the previous kernel's output and writes 4 KB
    const int b = blockIdx.x, t = threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 v = ((const float4*)in)[((b * 8 + i) * 256 + t) % (n / 4)];
        acc += v.x + v.y + v.z + v.w;
    }
    out[(b * 256 + t) % n] = acc;
}

// ---- persistent single-XCD kernel -----------------------------------------------------------------------
struct PArgs {
    unsigned* ticket;        // [1]
    unsigned* bar;           // [1] monotonically increasing arrival counter
    unsigned long long* data;// [P][512] slots
    unsigned* errors;        // [1]
    unsigned long long* cyc; // [2] start/end s_memtime of worker 0
    int xcd, P, iters, mode; // mode 0: barrier only; 1: + publish/check 4 KB per workgroup
    unsigned* info;          // [P] hw ids of the participants
};
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_persist(PArgs a) {
    __shared__ char pad[90 * 1024];
    __shared__ unsigned s_rank;
    const int t = threadIdx.x;
    if (t == 0) {
        pad[0] = 0;
        unsigned r = 0xffffffffu;
        if ((int)xcc_id() == a.xcd) r = atomicAdd(a.ticket, 1u);
        s_rank = r;
    }
    __syncthreads();
    const unsigned rank = s_rank;
    if (rank >= (unsigned)a.P) return;
    if (t == 0) a.info[rank] = hw_id();
    unsigned long long t0 = 0;
    unsigned target = 0;
    for (int it = 0; it < a.iters; ++it) {
        if (a.mode == 1) {
            // publish: 512 x 8 B per workgroup, plain stores
            a.data[(size_t)rank * 512 + t] = ((unsigned long long)(it + 1) << 32) | (rank * 1000 + t);
            a.data[(size_t)rank * 512 + 256 + t] = ((unsigned long long)(it + 1) << 32) | (rank * 1000 + 256 + t);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        target += a.P;
        if (t == 0) {
            __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (ld_sc1(a.bar) < target) { if (++spins > 50000000u) { atomicAdd(a.errors, 1000000u); break; } }
        }
        __syncthreads();
        if (it == 0 && rank == 0 && t == 0) t0 = __builtin_readcyclecounter();
        if (a.mode == 1) {
            // consume: read the slot of 2 other workgroups with sc1 loads and check every word
            for (int k = 1; k <= 2; ++k) {
                const unsigned src = (rank + k * 7) % a.P;
                const unsigned long long v0 = ld_sc1(&a.data[(size_t)src * 512 + t]);
                const unsigned long long v1 = ld_sc1(&a.data[(size_t)src * 512 + 256 + t]);
                const unsigned long long e0 = ((unsigned long long)(it + 1) << 32) | (src * 1000 + t);
                const unsigned long long e1 = ((unsigned long long)(it + 1) << 32) | (src * 1000 + 256 + t);
                if (v0 != e0 || v1 != e1) atomicAdd(a.errors, 1u);
            }
            // a second barrier so nobody overwrites a slot that is still being read
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            target += a.P;
            if (t == 0) {
                __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (ld_sc1(a.bar) < target) { if (++spins > 50000000u) { atomicAdd(a.errors, 1000000u); break; } }
            }
            __syncthreads();
        }
    }
    if (rank == 0 && t == 0) { a.cyc[0] = t0; a.cyc[1] = __builtin_readcyclecounter(); }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s CUs %d clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    // ---- 1. placement
    {
        const int G = 512;
        unsigned* d; CK(hipMalloc(&d, G * 2 * sizeof(unsigned)));
        hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, st, d);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> h(G * 2);
        CK(hipMemcpy(h.data(), d, G * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
        int match = 0; int cnt[16] = {0};
        for (int b = 0; b < G; ++b) { if ((int)h[b * 2] == b % 8) ++match; cnt[h[b * 2] & 15]++; }
        printf("placement: %d/%d blocks on XCD (block %% 8); per-XCD counts:", match, G);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n first 16 blocks xcc:");
        for (int b = 0; b < 16; ++b) printf(" %u", h[b * 2]);
        printf("\n");
        CK(hipFree(d));
    }
    // ---- 2. launch floors
    {
        const int N = 2000;
        float *a, *b; const int n = 1 << 20;
        CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
        CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int variant = 0; variant < 2; ++variant) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < N; ++i) {
                    if (variant == 0) hipLaunchKernelGGL(k_null, dim3(96), dim3(256), 0, st, (int*)nullptr);
                    else hipLaunchKernelGGL(k_dep, dim3(96), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
                }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("eager chain %s: %.2f us/launch\n", variant ? "dependent 32KB-read/4KB-write" : "null", 1000.f * ms / N);
            }
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 200; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_null, dim3(96), dim3(256), 0, st, (int*)nullptr);
                else hipLaunchKernelGGL(k_dep, dim3(96), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("graph chain %s: %.2f us/launch\n", variant ? "dependent" : "null", 1000.f * ms / 2000);
            }
        }
    }
    // ---- 3. persistent single-XCD barrier + visibility
    for (int mode = 0; mode < 2; ++mode)
        for (int P : {8, 16, 32}) {
            PArgs a;
            CK(hipMalloc(&a.ticket, 4)); CK(hipMalloc(&a.bar, 4)); CK(hipMalloc(&a.errors, 4));
            CK(hipMalloc(&a.data, (size_t)P * 512 * 8)); CK(hipMalloc(&a.cyc, 16)); CK(hipMalloc(&a.info, P * 4));
            CK(hipMemset(a.ticket, 0, 4)); CK(hipMemset(a.bar, 0, 4)); CK(hipMemset(a.errors, 0, 4));
            CK(hipMemset(a.data, 0, (size_t)P * 512 * 8)); CK(hipMemset(a.info, 0, P * 4));
            a.xcd = 3; a.P = P; a.iters = 2000; a.mode = mode;
            const double w0 = now_us();
            hipLaunchKernelGGL(k_persist, dim3(512), dim3(256), 0, st, a);
            CK(hipStreamSynchronize(st));
            const double w1 = now_us();
            unsigned err, tk; unsigned long long cyc[2];
            CK(hipMemcpy(&err, a.errors, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&tk, a.ticket, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(cyc, a.cyc, 16, hipMemcpyDeviceToHost));
            const int nb = mode == 1 ? 2 : 1;
            printf("persist mode %d P=%2d: tickets on XCD %d = %u, errors = %u, wall %.1f us total, %.3f us per barrier "
                   "(%.0f clk per iteration)\n", mode, P, a.xcd, tk, err, w1 - w0, (w1 - w0) / (a.iters * nb),
                   (double)(cyc[1] - cyc[0]) / (a.iters - 1));
            CK(hipFree(a.ticket)); CK(hipFree(a.bar)); CK(hipFree(a.errors)); CK(hipFree(a.data)); CK(hipFree(a.cyc)); CK(hipFree(a.info));
        }
    printf("done\n");
    return 0;
}
