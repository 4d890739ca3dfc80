// buffer addressing probes (gfx950): (1) voffset + inst_offset wrap-around, (2) unaligned dword load, (3) imm folding by hipcc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const float *x, const uint8_t *b, float *out, uint32_t *outu, int n) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, n * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(b), 0, n, 0x00020000);
    uint32_t off = 0xFFFFFFFCu + threadIdx.x * 0;    // -4
    float v0, v1;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:4\n\ts_waitcnt vmcnt(0)" : "=v"(v0) : "v"(off), "s"(r));
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:8\n\ts_waitcnt vmcnt(0)" : "=v"(v1) : "v"(off), "s"(r));
    out[0] = v0; out[1] = v1;
    // folded by the compiler?
    uint32_t base = threadIdx.x * 4;
    out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, base + 8, 0, 0));
    // unaligned dword from the byte buffer at byte offset 1, 2, 3
    for (int o = 1; o < 4; ++o) {
        uint32_t w;
        uint32_t bo = o;
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(bo), "s"(rb));
        outu[o] = w;
    }
    // dword at the end: bytes n-3..n (1 beyond)
    { uint32_t w; uint32_t bo = n - 3; asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(bo), "s"(rb)); outu[4] = w; }
    { uint32_t w; uint32_t bo = n - 4; asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(bo), "s"(rb)); outu[5] = w; }
}
int main() {
    const int n = 64;
    float hx[n]; uint8_t hb[n];
    for (int i = 0; i < n; ++i) { hx[i] = 100 + i; hb[i] = i + 1; }
    float *x, *out; uint8_t *b; uint32_t *outu;
    hipMalloc(&x, n * 4); hipMalloc(&b, n); hipMalloc(&out, 64); hipMalloc(&outu, 64);
    hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n, hipMemcpyHostToDevice);
    hipMemset(out, 0, 64); hipMemset(outu, 0, 64);
    k<<<1, 64>>>(x, b, out, outu, n);
    float ho[16]; uint32_t hu[16];
    hipMemcpy(ho, out, 64, hipMemcpyDeviceToHost); hipMemcpy(hu, outu, 64, hipMemcpyDeviceToHost);
    printf("wrap: voffset=-4 offset:4 -> %g (100 = wraps to element 0, 0 = out of range); offset:8 -> %g (101 / 0)\n", ho[0], ho[1]);
    printf("compiler-folded base+8: %g (expect 102)\n", ho[2]);
    printf("unaligned dwords at byte 1,2,3: %08x %08x %08x (expect 05040302 06050403 07060504)\n", hu[1], hu[2], hu[3]);
    printf("dword at n-3 (1 byte beyond): %08x ; at n-4: %08x (expect 403f3e3d)\n", hu[4], hu[5]);
    return 0;
}
