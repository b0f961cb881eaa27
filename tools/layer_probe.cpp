// layer_probe.cpp -- round-5 verdict item 1(b): can ONE workgroup per 16-row tile (6 workgroups at batch 1) pull a whole encoder
// layer's weights (W_o + W1 + W2 + W_qkv = 1.5 MB in bf16) through its CU's load path fast enough for a one-dispatch layer?
// Every wave streams its share of the layer as 1 KB fragment loads (the packed-weight order of the library) through a register
// ring RD fragments deep and feeds each fragment to one v_mfma_f32_16x16x32_bf16 -- the data flow of a weight-streaming layer kernel
// with nothing else in it (no attention chain, no LayerNorm, no barriers): a LOWER bound on such a kernel's time.
//   layers cycling 0..7 (12 MB: misses the 4 MB L2 of the XCD like the real step) vs one layer (L2 resident)
//   shared addresses (all workgroups read the same weights, as the real kernel) vs private regions
// Also prices the alternative for the hidden-slice reduction of linear2 inside k_attn_mid: 96 workgroups adding their partial
// 16 x 256 tile into one accumulator with 64-bit fixed-point atomics (deterministic) in uncached memory, against plain slab stores.
//   hipcc --offload-arch=gfx950 -O3 tools/layer_probe.cpp -o tools/_build/layer_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s @%d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k_null(float* out) { if (threadIdx.x == 1234567) out[0] = 1.f; }

// NW waves, ring of RD fragments per wave, NF fragments per wave
template <int NW, int RD, int NF>
__global__ __launch_bounds__(NW * 64) void k_stream(const f32x4* w, float* out, size_t wg_stride_frags) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const f32x4* p = w + ((size_t)blockIdx.x * wg_stride_frags + (size_t)wave * NF) * 64 + lane;
    f32x4 ring[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i) ring[i] = p[(size_t)i * 64];
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a = (f32x4){1.f, 2.f, 3.f, 4.f};
    static_assert(NF % RD == 0, "ring");
#pragma unroll 1
    for (int i0 = 0; i0 < NF; i0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[j]), __builtin_bit_cast(bf16x8, a), acc[j & 3], 0, 0, 0);
            const int nx = i0 + RD + j;
            ring[j] = p[(size_t)(nx < NF ? nx : NF - 1) * 64];
            __builtin_amdgcn_sched_barrier(0);          // keep the order MFMA(j), load(j): the ring stays RD deep across iterations
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[threadIdx.x] = s[0];
}

// the hidden-slice reduction: workgroup (slice, row tile) holds a 16 x 256 fp32 tile (4 waves x 4 column tiles x f32x4 per lane)
// MODE 0: plain 16-byte stores into its own slab; 1: 64-bit fixed-point atomic adds into the row tile's accumulator; 2: fp32 atomic adds
template <int MODE>
__global__ __launch_bounds__(256) void k_reduce(float* slab, unsigned long long* acc64, float* acc32, int nslice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
    const int s = blockIdx.x, mt = blockIdx.y;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = (wave * 4 + t) * 16 + 4 * lg;
        f32x4 v = (f32x4){1.f + lane, 2.f, 3.f, 4.f + s};
        if constexpr (MODE == 0) {
            *(f32x4*)(slab + (((size_t)s * 6 + mt) * 16 + lr) * 256 + col) = v;
        } else if constexpr (MODE == 1) {
            unsigned long long* q = acc64 + ((size_t)mt * 16 + lr) * 256 + col;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __hip_atomic_fetch_add(q + e, (unsigned long long)(long long)((double)v[e] * 1073741824.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            float* q = acc32 + ((size_t)mt * 16 + lr) * 256 + col;
#pragma unroll
            for (int e = 0; e < 4; ++e) __hip_atomic_fetch_add(q + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// DSG+ at batch 1 (round-5 verdict item 4): `out_proj + LayerNorm1 + linear1` with W_o split by COLUMNS over the NS hidden-slice workgroups of a
// row tile needs the row statistics of all NS column slices before any of them can normalise: a barrier among the NS workgroups of a row tile
// INSIDE the kernel.  Deterministic form: every workgroup stores its partial (sum, sum of squares) of its 16 rows into its own slot of uncached
// memory, bumps the row tile's arrival counter (agent scope), spins until all NS have arrived, then reads the NS slots in slot order.  `epoch`
// makes the counters reusable across launches without a reset.
__global__ __launch_bounds__(256) void k_exchange(float* slots, unsigned* counters, unsigned epoch, int NS, float* out) {
    const int s = blockIdx.x, mt = blockIdx.y, lane = threadIdx.x;
    float part = 1.0f + s + lane;                               // (sum, sumsq) of 16 rows: 32 floats per workgroup
    if (lane < 32) slots[((size_t)mt * NS + s) * 32 + lane] = part;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (lane == 0) {
        __hip_atomic_fetch_add(&counters[mt * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&counters[mt * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * (unsigned)NS) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    float tot = 0.f;
    if (lane < 32)
        for (int k = 0; k < NS; ++k) tot += __hip_atomic_load(&slots[((size_t)mt * NS + k) * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tot == 123.456f) out[lane] = tot;
}

template <class F>
static float time_chain(hipStream_t st, int N, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 40; ++i) launch(i);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) launch(i);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1000.f * ms / N;
}

template <int NW, int RD, int KBYTES = 1536>
static void run_stream(const char* what, const f32x4* w, float* out, int n_wg, bool cycle, bool shared, hipStream_t st, float floor_us) {
    constexpr int NF = KBYTES / NW;                     // 1.5 MB per workgroup (the whole layer), or 512 KB (W2 alone: the linear2 + LayerNorm2 + QKV fusion)
    const size_t layer_frags = 1536 * 6;                // private: 6 regions per layer
    const float us = time_chain(st, 400, [&](int i) {
        const f32x4* base = w + (size_t)(cycle ? (i & 7) : 0) * layer_frags * 64;
        hipLaunchKernelGGL((k_stream<NW, RD, NF>), dim3(n_wg), dim3(NW * 64), 0, st, base, out, shared ? (size_t)0 : (size_t)1536);
    });
    printf("%-34s %2d waves, ring %2d (%3d KB in flight), %4d KB per workgroup, %2d workgroups, %s, %s: %6.2f us per launch, %6.2f above the floor = %5.0f GB/s per CU\n", what, NW, RD,
           NW * RD, KBYTES, n_wg, cycle ? "8 layers cycling" : "one layer      ", shared ? "shared " : "private", us, us - floor_us, KBYTES * 1024.0 / ((us - floor_us) * 1e3));
}

int main(int argc, char** argv) {
    const bool dsgplus_only = argc > 1 && std::string(argv[1]) == "dsgplus";
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t bytes = (size_t)8 * 6 * 1536 * 1024;   // 8 layers x 6 private regions x 1.5 MB = 72 MB
    f32x4* w; CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 0, bytes));
    float* out; CK(hipMalloc(&out, 1 << 16));
    CK(hipStreamSynchronize(st));
    const float floor6 = time_chain(st, 400, [&](int) { hipLaunchKernelGGL(k_null, dim3(6), dim3(512), 0, st, out); });
    const float floor96 = time_chain(st, 400, [&](int) { hipLaunchKernelGGL(k_null, dim3(96, 6), dim3(256), 0, st, out); });
    printf("# launch floor of this harness (HIP launches back to back): 6 x 512 threads %.2f us, 96 x 6 x 256 threads %.2f us\n", floor6, floor96);
    {   // ---- DSG+ widths at batch 1 (BEAT: latent 384, 96-dim heads; TWH: latent 512, 128-dim heads; 151 tokens = 10 row tiles; round 6)
        const float floor20 = time_chain(st, 400, [&](int) { hipLaunchKernelGGL(k_null, dim3(20), dim3(512), 0, st, out); });
        const float floor160 = time_chain(st, 400, [&](int) { hipLaunchKernelGGL(k_null, dim3(16, 10), dim3(256), 0, st, out); });
        printf("# DSG+ batch 1: launch floor 20 x 512 threads %.2f us, 16 x 10 x 256 threads %.2f us\n", floor20, floor160);
        // (A) LayerNorm + QKV + attention per (head, query-tile pair): 4 heads x 5 pairs = 20 workgroups, each pulling the clip's rows (151 x D bf16)
        //     and its head's W_q / W_k / W_v slices (3 x hd x D bf16): 116 + 221 = 337 KB (BEAT), 155 + 393 = 548 KB (TWH)
        for (int cyc = 0; cyc < 2; ++cyc) {
            run_stream<8, 14, 336>("(A) BEAT rows + head's W_qkv", w, out, 20, cyc, true, st, floor20);
            run_stream<8, 17, 544>("(A) TWH  rows + head's W_qkv", w, out, 20, cyc, true, st, floor20);
        }
        // (B) out_proj + LayerNorm1 + linear1, W_o split by columns over 16 hidden-slice workgroups per row tile: W_o / 16 + W1 slice + the rows
        //     (18 + 48 + 12 = 78 KB BEAT, 32 + 64 + 16 = 112 KB TWH) -- and the exchange of the row statistics among the 16 workgroups
        run_stream<4, 20, 80>("(B) BEAT W_o/16 + W1 slice + rows", w, out, 160, true, true, st, floor160);
        run_stream<4, 28, 112>("(B) TWH  W_o/16 + W1 slice + rows", w, out, 160, true, true, st, floor160);
        float* slots; unsigned* counters;
        CK(hipExtMallocWithFlags((void**)&slots, (size_t)10 * 16 * 32 * 4, hipDeviceMallocUncached));
        CK(hipExtMallocWithFlags((void**)&counters, (size_t)10 * 32 * 4, hipDeviceMallocUncached));
        CK(hipMemset(counters, 0, (size_t)10 * 32 * 4)); CK(hipStreamSynchronize(st)); CK(hipDeviceSynchronize());
        unsigned epoch = 0;
        const float ex = time_chain(st, 400, [&](int) { ++epoch; hipLaunchKernelGGL(k_exchange, dim3(16, 10), dim3(256), 0, st, slots, counters, epoch, 16, out); });
        printf("(B) exchange of the LayerNorm statistics among the 16 column-slice workgroups of a row tile (slots + arrival counter, uncached memory, "
               "160 workgroups): %6.2f us per launch, %6.2f above the floor\n", ex, ex - floor160);
    }
    if (dsgplus_only) return 0;
    for (int cyc = 0; cyc < 2; ++cyc) {
        run_stream<8, 16>("layer stream", w, out, 6, cyc, true, st, floor6);
        run_stream<8, 32>("layer stream", w, out, 6, cyc, true, st, floor6);
        run_stream<8, 48>("layer stream", w, out, 6, cyc, true, st, floor6);
        run_stream<16, 16>("layer stream", w, out, 6, cyc, true, st, floor6);
        run_stream<16, 24>("layer stream", w, out, 6, cyc, true, st, floor6);
        run_stream<4, 48>("layer stream", w, out, 6, cyc, true, st, floor6);
    }
    run_stream<8, 32>("layer stream, private regions", w, out, 6, true, false, st, floor6);
    run_stream<8, 32>("layer stream, 12 workgroups", w, out, 12, true, true, st, floor6);
    run_stream<8, 32>("layer stream, 24 workgroups", w, out, 24, true, true, st, floor6);
    // (a) linear2 + LayerNorm2 + next QKV in one kernel: every (row tile, column group) workgroup needs all of W2 (512 KB) for its full rows
    for (int cyc = 0; cyc < 2; ++cyc)
        for (int n_wg : {18, 36, 72}) {
            run_stream<8, 32, 512>("W2 per (row tile, column group)", w, out, n_wg, cyc, true, st, floor6);
            run_stream<8, 16, 512>("W2 per (row tile, column group)", w, out, n_wg, cyc, true, st, floor6);
        }
    // what k_attn_mid pulls today (280 KB: K / V^T 96 + W_o 128 + W1 slice 32 + rows), 96 workgroups, 4 waves
    run_stream<4, 35, 280>("k_attn_mid's bytes", w, out, 96, false, true, st, floor96);
    run_stream<4, 35, 280>("k_attn_mid's bytes", w, out, 96, true, true, st, floor96);

    // ---- the reduction alternatives (uncached memory, like the library's loop buffers)
    float* slab; unsigned long long* acc64; float* acc32;
    CK(hipExtMallocWithFlags((void**)&slab, (size_t)16 * 96 * 256 * 4, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&acc64, (size_t)96 * 256 * 8, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&acc32, (size_t)96 * 256 * 4, hipDeviceMallocUncached));
    CK(hipMemset(acc64, 0, (size_t)96 * 256 * 8)); CK(hipMemset(acc32, 0, (size_t)96 * 256 * 4));
    const float r0 = time_chain(st, 400, [&](int) { hipLaunchKernelGGL(k_reduce<0>, dim3(16, 6), dim3(256), 0, st, slab, acc64, acc32, 16); });
    const float r1 = time_chain(st, 400, [&](int) { hipLaunchKernelGGL(k_reduce<1>, dim3(16, 6), dim3(256), 0, st, slab, acc64, acc32, 16); });
    const float r2 = time_chain(st, 400, [&](int) { hipLaunchKernelGGL(k_reduce<2>, dim3(16, 6), dim3(256), 0, st, slab, acc64, acc32, 16); });
    printf("# 96 workgroups (16 hidden slices x 6 row tiles) each leaving a 16 x 256 fp32 tile, uncached memory:\n");
    printf("plain 16-byte stores to 16 slabs          %6.2f us per launch\n", r0);
    printf("64-bit fixed-point atomic adds (16 -> 1)  %6.2f us per launch\n", r1);
    printf("fp32 atomic adds (16 -> 1)                %6.2f us per launch\n", r2);
    return 0;
}
