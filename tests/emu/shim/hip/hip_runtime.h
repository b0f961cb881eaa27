// TEST INFRASTRUCTURE ONLY -- a stand-in <hip/hip_runtime.h> that lets the product sources
// (diffusestylegesture_amd/csrc/dsg_hip.cpp + dsg_kernels.h) compile, UNCHANGED, for the host CPU, where each
// workgroup is executed by cooperative fibers (one per work-item) with faithful __syncthreads / wave64 shuffle /
// MFMA fragment semantics (tests/emu/emu_rt.cpp).  It exists because the dev container has no GPU and GPU time is
// rationed: index / layout / host-sequencing bugs are caught here (tests/test_emu_*.py, CPU) before the real
// MI355X parity tests (tests/test_gpu_*.py, -m gpu).  It is NEVER loaded by the package: the product path only
// dlopens libdsg_hip.so and fails loudly without it.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Ctx { dim3 tid, bid, bdim, gdim; };
Ctx* cur();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
unsigned shfl_xor_u32(unsigned v, int mask);
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
f32x4_t mfma_16x16x32_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c);
f32x4_t mfma_16x16x4_f32(float a, float b, f32x4_t c);
typedef float f32x16_t __attribute__((ext_vector_type(16)));
f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c);
bool is_device(const void* p);
void* dmalloc(size_t n);
void dfree(void* p);
}  // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::cur()->bid)
#define blockDim (emu::cur()->bdim)
#define gridDim (emu::cur()->gdim)

static inline void __syncthreads() { emu::syncthreads(); }
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
static inline void __threadfence() {}
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
#define __builtin_amdgcn_s_barrier() emu::syncthreads()
static inline int emu_update_dpp(int src, int ctrl) {
    // exchange across the wave, then pick the source lane the DPP control selects (row = 16 lanes)
    const int lane = (int)(threadIdx.x & 63);
    int srcl = lane;
    if (ctrl == 0xB1) srcl = lane ^ 1;                       // quad_perm [1,0,3,2]
    else if (ctrl == 0x4E) srcl = lane ^ 2;                  // quad_perm [2,3,0,1]
    else if (ctrl == 0x141) srcl = (lane & ~7) | (7 - (lane & 7));      // row_half_mirror
    else if (ctrl == 0x140) srcl = (lane & ~15) | (15 - (lane & 15));   // row_mirror
    return (int)emu::shfl_xor_u32((unsigned)src, lane ^ srcl);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rmask, bmask, bc) emu_update_dpp(src, ctrl)
#define __expf(x) std::exp((float)(x))
#define __builtin_amdgcn_rcpf(x) (1.0f / (float)(x))
static inline float __shfl_xor(float v, int m) {
    return __builtin_bit_cast(float, emu::shfl_xor_u32(__builtin_bit_cast(unsigned, v), m));
}
static inline int __shfl_xor(int v, int m) { return (int)emu::shfl_xor_u32((unsigned)v, m); }
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu::mfma_16x16x32_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_16x16x4_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16_bf16(a, b, c)

static inline void sincospif(float x, float* s, float* c) {
    const double a = 3.14159265358979323846 * (double)x;
    *s = (float)std::sin(a);
    *c = (float)std::cos(a);
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

// ---- runtime API subset ---------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801, hipErrorInvalidValue = 1 };
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipStreamCaptureModeThreadLocal = 1 };
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t { int type; };

static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = emu::dmalloc(n); return *p ? hipSuccess : 2; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { emu::dfree(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    a->type = emu::is_device(p) ? hipMemoryTypeDevice : hipMemoryTypeUnregistered;
    return hipSuccess;
}
// graphs are a device-side launch optimisation; the emulator reports "not supported" and the library runs eagerly
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
