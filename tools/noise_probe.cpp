// noise_probe: accuracy of the Box-Muller pieces of the noise stream (csrc/dsg_kernels.h: philox_normal4) against double arithmetic ON THE DEVICE,
// over ALL 2^24 arguments of each piece, and the distance between the libm form of rounds 1-5 and the round-6 form over 2^26 Philox calls.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/noise_probe.cpp -o tools/_build/noise_probe && tools/_build/noise_probe
#include "../diffusestylegesture_amd/csrc/dsg_kernels.h"
#include <cstdio>
#include <vector>
using namespace dsg;
struct Res { double r_abs, r_rel, s_abs, c_abs, z_abs, z_rel, zl_abs, zl_rel; };
// the libm form of rounds 1-5 (logf, sqrtf, sincospif), kept here for the comparison
__device__ f32x4 philox_normal4_libm(unsigned q, unsigned draw, NoiseKey key) {
    unsigned x[4];
    philox4x32_10(q, draw, key.s0, key.s1, key.k0, key.k1, x);
    const float sc = 5.9604644775390625e-08f;
    float u1a = (float)((x[0] >> 8) + 1u) * sc, u2a = (float)(x[1] >> 8) * sc;
    float u1b = (float)((x[2] >> 8) + 1u) * sc, u2b = (float)(x[3] >> 8) * sc;
    float ra = sqrtf(-2.0f * logf(u1a)), rb = sqrtf(-2.0f * logf(u1b));
    float sa, ca, sb, cb;
    sincospif(2.0f * u2a, &sa, &ca);
    sincospif(2.0f * u2b, &sb, &cb);
    f32x4 z; z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cb; z[3] = rb * sb;
    return z;
}
__device__ void amax(double* p, double v) {
    unsigned long long* a = (unsigned long long*)p; unsigned long long old = *a, assumed;
    do { assumed = old; if (__longlong_as_double((long long)assumed) >= v) break; old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v)); } while (assumed != old);
}
__global__ void k_pieces(Res* out) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;        // 0 .. 2^24 - 1
    const float u1 = (float)(k + 1u) * 5.9604644775390625e-08f;
    const float r = dsg_sqrtf(-1.3862943611198906f * dsg_log2f(u1));
    const double rd = sqrt(-2.0 * log((double)u1));
    const double ea = fabs((double)r - rd);
    amax(&out->r_abs, ea);
    if (rd > 0) amax(&out->r_rel, ea / rd);
    const float t = 2.0f * ((float)k * 5.9604644775390625e-08f);
    float s, c; dsg_sincospi_02(t, s, c);
    amax(&out->s_abs, fabs((double)s - sinpi((double)t)));
    amax(&out->c_abs, fabs((double)c - cospi((double)t)));
}
__global__ void k_normals(Res* out, unsigned draw0) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    const NoiseKey key = {123456u, 7u, 3u, 0u};
    unsigned x[4];
    philox4x32_10(q, draw0, key.s0, key.s1, key.k0, key.k1, x);
    const f32x4 z = philox_normal4(q, draw0, key), zl = philox_normal4_libm(q, draw0, key);
    for (int p = 0; p < 2; ++p) {
        const double u1 = ((double)(x[2 * p] >> 8) + 1.0) * 0x1p-24, u2 = (double)(x[2 * p + 1] >> 8) * 0x1p-24;
        const double r = sqrt(-2.0 * log(u1));
        const double zc = r * cospi(2.0 * u2), zs = r * sinpi(2.0 * u2);
        const double e0 = fabs((double)z[2 * p] - zc), e1 = fabs((double)z[2 * p + 1] - zs);
        amax(&out->z_abs, fmax(e0, e1));
        if (fabs(zc) > 1e-3) amax(&out->z_rel, e0 / fabs(zc));
        if (fabs(zs) > 1e-3) amax(&out->z_rel, e1 / fabs(zs));
        const double l0 = fabs((double)zl[2 * p] - zc), l1 = fabs((double)zl[2 * p + 1] - zs);
        amax(&out->zl_abs, fmax(l0, l1));
        if (fabs(zc) > 1e-3) amax(&out->zl_rel, l0 / fabs(zc));
        if (fabs(zs) > 1e-3) amax(&out->zl_rel, l1 / fabs(zs));
    }
}
int main() {
    Res* d; hipMalloc(&d, sizeof(Res)); hipMemset(d, 0, sizeof(Res));
    k_pieces<<<(1 << 24) / 256, 256>>>(d);
    for (unsigned dr = 0; dr < 4; ++dr) k_normals<<<(1 << 24) / 256, 256>>>(d, dr);
    Res h; hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("r = sqrt(-2 ln u1), all 2^24 u1: max abs err %.3e, max rel err %.3e (float eps 5.96e-8)\n", h.r_abs, h.r_rel);
    printf("sincospi(2 u2), all 2^24 u2: max abs err sin %.3e cos %.3e\n", h.s_abs, h.c_abs);
    printf("normals of 2^26 Philox calls vs double Box-Muller: max abs err %.3e, max rel err (|z| > 1e-3) %.3e\n", h.z_abs, h.z_rel);
    printf("... the libm form of rounds 1-5 (logf, sqrtf, sincospif) on the same calls:  max abs err %.3e, max rel err (|z| > 1e-3) %.3e\n", h.zl_abs, h.zl_rel);
    return 0;
}
