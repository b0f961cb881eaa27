// Which XCD does a workgroup run on?  Prints HW_REG_XCC_ID per block for 1-D and 3-D grids (the pinned lanes of dsg_fused.h
// assume XCD = linear workgroup id % 8 with x fastest).   hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.cpp -o tools/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned full = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned f4 = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    if (threadIdx.x == 0) { out[2 * lin] = full; out[2 * lin + 1] = f4; }
}
static void run(dim3 g) {
    const unsigned n = g.x * g.y * g.z;
    unsigned* d; hipMalloc(&d, n * 8); hipMemset(d, 0xff, n * 8);
    hipLaunchKernelGGL(k, g, dim3(256), 0, 0, d);
    std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    unsigned bad = 0; for (unsigned i = 0; i < n; ++i) bad += (h[2 * i] & 7u) != (i & 7u);
    printf("grid (%u,%u,%u): %u of %u blocks NOT on XCC lin%%8; first 24 (full reg / 4-bit field): ", g.x, g.y, g.z, bad, n);
    for (unsigned i = 0; i < 24 && i < n; ++i) printf("%x/%x ", h[2 * i], h[2 * i + 1]);
    printf("\n");
    hipFree(d);
}
int main() { run(dim3(64)); run(dim3(64, 8, 2)); run(dim3(128, 7, 1)); run(dim3(96, 6, 1)); run(dim3(8, 3, 5)); run(dim3(1024)); return 0; }
