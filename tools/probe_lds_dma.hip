// buffer_load ... lds (LDS-DMA) semantics on gfx950: lane -> LDS mapping, out-of-range lanes, vmcnt accounting.
// hipcc --offload-arch=gfx950 -O3 tools/probe_lds_dma.hip -o tools/_tune_ldsdma.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const float *x, int n, float *out) {
    __shared__ float ring[4][64];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) ring[i][lane] = -1.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, n * 4, 0x00020000);
    // slot 0: lane l reads x[l]; slot 1: stride-2 gather; slot 2: odd lanes out of range; slot 3: scalar offset 64 floats
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, &ring[0][0], 4, lane * 4, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, &ring[1][0], 4, lane * 8, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, &ring[2][0], 4, (lane & 1) ? 0x80000000u : lane * 4, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, &ring[3][0], 4, lane * 4, 64 * 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[i * 64 + lane] = ring[i][lane];
}

int main() {
    const int n = 256;
    float h[n], *dx, *dout, o[256];
    for (int i = 0; i < n; ++i) h[i] = 100.0f + i;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dout, 256 * 4);
    hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, n, dout);
    hipMemcpy(o, dout, 256 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) {
        printf("slot %d:", i);
        for (int l = 0; l < 8; ++l) printf(" %.0f", o[i * 64 + l]);
        printf(" ... %.0f %.0f\n", o[i * 64 + 62], o[i * 64 + 63]);
    }
    return 0;
}
