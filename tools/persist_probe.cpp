// persist_probe.cpp -- feasibility of a persistent single-XCD step kernel: time per PHASE when each phase is
//   [sc1 loads of activations written by other workgroups in the previous phase] + [plain weight loads] + MFMA chain
//   + plain stores + L2-atomic barrier.   hipcc --offload-arch=gfx950 -O3 tools/persist_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s @%d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }
__device__ __forceinline__ f32x4 ld_coh16(const void* p) {
    const u64 a = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 b = __hip_atomic_load((const u64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    u64x2 v; v[0] = a; v[1] = b;
    return __builtin_bit_cast(f32x4, v);
}
struct Args {
    unsigned* ticket; unsigned* bar; unsigned* err;
    const unsigned short* W;      // weights: [layers][NT][KB][64][8] bf16
    unsigned short* act[2];       // activations ping-pong: [96][256] bf16
    int xcd, P, steps, phases, tnw, layers;
};
// each phase: out[96 x 256] = act_in[96 x 256] . W_l^T  (bf16 MFMA), tiles: 6 m-tiles x 16 n-tiles = 96 wave tiles
__global__ __launch_bounds__(256) void k_probe(Args a) {
    __shared__ char pad[90 * 1024];
    __shared__ unsigned s_rank;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lg = lane >> 4;
    if (tid == 0) { pad[0] = 0; s_rank = ((int)xcc_id() == a.xcd) ? atomicAdd(a.ticket, 1u) : 0xffffffffu; }
    __syncthreads();
    const unsigned rank = s_rank;
    if (rank >= (unsigned)a.P) return;
    unsigned target = 0;
    int cur = 0;
    for (int st = 0; st < a.steps; ++st)
        for (int ph = 0; ph < a.phases; ++ph) {
            const unsigned short* in = a.act[cur];
            unsigned short* out = a.act[cur ^ 1];
            const unsigned short* W = a.W + (size_t)(ph % a.layers) * 16 * 8 * 64 * 8;
            // wave-tile ids: 96 tiles over P*4 waves
            for (int t = rank * 4 + wave; t < 96; t += a.P * 4) {
                const int mt = t % 6, nt = t / 6;
                f32x4 af[8], bf[8];
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    bf[kb] = *(const f32x4*)(W + (((size_t)nt * 8 + kb) * 64 + lane) * 8);
                    af[kb] = ld_coh16(in + (size_t)(mt * 16 + lr) * 256 + kb * 32 + 8 * lg);
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 8; ++kb)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[kb]), __builtin_bit_cast(bf16x8, af[kb]), acc, 0, 0, 0);
                typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { unsigned u = __builtin_bit_cast(unsigned, acc[e] * 0.05f); o[e] = (unsigned short)(u >> 16); }
                *(u16x4*)(out + (size_t)(mt * 16 + lr) * 256 + nt * 16 + 4 * lg) = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            target += a.P;
            if (tid == 0) {
                __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
                    if (++spins > 20000000u) { atomicAdd(a.err, 1u); break; }
            }
            __syncthreads();
            cur ^= 1;
        }
}
// same arithmetic as ordinary dependent launches (one launch per phase), for comparison and for checking the result
__global__ __launch_bounds__(256) void k_phase(const unsigned short* W, const unsigned short* in, unsigned short* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const int t = blockIdx.x * 4 + wave;
    const int mt = t % 6, nt = t / 6;
    f32x4 af[8], bf[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        bf[kb] = *(const f32x4*)(W + (((size_t)nt * 8 + kb) * 64 + lane) * 8);
        af[kb] = *(const f32x4*)(in + (size_t)(mt * 16 + lr) * 256 + kb * 32 + 8 * lg);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[kb]), __builtin_bit_cast(bf16x8, af[kb]), acc, 0, 0, 0);
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned u = __builtin_bit_cast(unsigned, acc[e] * 0.05f); o[e] = (unsigned short)(u >> 16); }
    *(u16x4*)(out + (size_t)(mt * 16 + lr) * 256 + nt * 16 + 4 * lg) = o;
}
int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int layers = 8, phases = 40, steps = 50;
    std::vector<unsigned short> hW((size_t)layers * 16 * 8 * 64 * 8), hA(96 * 256);
    srand(1);
    for (auto& v : hW) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    for (auto& v : hA) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    Args a;
    unsigned short* dW; CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    a.W = dW;
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&a.act[i], 96 * 256 * 2));
    CK(hipMalloc(&a.ticket, 4)); CK(hipMalloc(&a.bar, 4)); CK(hipMalloc(&a.err, 4));
    // reference: ordinary launches
    unsigned short* r[2]; for (int i = 0; i < 2; ++i) CK(hipMalloc(&r[i], 96 * 256 * 2));
    CK(hipMemcpy(r[0], hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    int cur = 0;
    for (int s = 0; s < steps; ++s) for (int ph = 0; ph < phases; ++ph) {
        hipLaunchKernelGGL(k_phase, dim3(24), dim3(256), 0, st, dW + (size_t)(ph % layers) * 16 * 8 * 64 * 8, r[cur], r[cur ^ 1]);
        cur ^= 1;
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("ordinary launches: %.3f us/phase\n", 1000.f * ms / (steps * phases));
    std::vector<unsigned short> ref(96 * 256), got(96 * 256);
    CK(hipMemcpy(ref.data(), r[cur], ref.size() * 2, hipMemcpyDeviceToHost));
    for (int P : {8, 16, 24, 32}) {
        CK(hipMemset(a.ticket, 0, 4)); CK(hipMemset(a.bar, 0, 4)); CK(hipMemset(a.err, 0, 4));
        CK(hipMemcpy(a.act[0], hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        a.xcd = 2; a.P = P; a.steps = steps; a.phases = phases; a.layers = layers; a.tnw = 1;
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), 0, st, a);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), a.act[(steps * phases) & 1], got.size() * 2, hipMemcpyDeviceToHost));
        int bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
        printf("persistent P=%2d: %.3f us/phase, timeouts %u, mismatching outputs vs ordinary launches: %d / %zu\n", P,
               1000.f * ms / (steps * phases), err, bad, ref.size());
    }
    return 0;
}
