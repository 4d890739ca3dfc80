// Issue cost of VALU instruction kinds on gfx950 (round 4): N instructions per lane in 8 independent chains, 1 or 2 waves per
// SIMD, wall time -> ns per instruction per SIMD.  (v_add_f32 = the unit; v_cndmask_b32 turned out to be the surprise.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define BODY(ASM, ...)                                   \
    for (int it = 0; it < iters; ++it) {                 \
        _Pragma("unroll") for (int r = 0; r < 8; ++r)    \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : __VA_ARGS__); \
    }
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed) {
    f32x2 a[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = (f32x2){seed + i + threadIdx.x, seed - i}; u[i] = threadIdx.x * 7 + i; }
    f32x2 b = {seed * 0.5f, seed * 0.25f};
    unsigned long long m = 0x5555555555555555ull;
    unsigned c3 = 3;
    if (MODE == 0) BODY("v_add_f32 %0, %0, %1", "+v"(a[i].x) : "v"(b.x))
    if (MODE == 1) BODY("v_cndmask_b32 %0, %0, %1, vcc", "+v"(a[i].x) : "v"(b.x))
    if (MODE == 2) BODY("v_cndmask_b32_e64 %0, %0, %1, %2", "+v"(a[i].x) : "v"(b.x), "s"(m))
    if (MODE == 3) BODY("v_cmp_eq_u32 vcc, %0, %1", : "v"(u[i]), "v"(c3) : "vcc")
    if (MODE == 4) BODY("v_cmp_eq_u32_e64 %0, %1, %2", "=s"(m) : "v"(u[i]), "v"(c3))
    if (MODE == 5) BODY("v_bfe_i32 %0, %0, %1, 1", "+v"(u[i]) : "v"(c3))
    if (MODE == 6) BODY("v_and_b32 %0, %0, %1", "+v"(u[i]) : "v"(c3))
    if (MODE == 7) BODY("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(u[i]) : "v"(c3))
    if (MODE == 8) BODY("v_max3_f32 %0, %0, %1, %1", "+v"(a[i].x) : "v"(b.x))
    if (MODE == 9) BODY("v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc", "+v"(a[i].x) : "v"(u[i]), "v"(c3), "v"(b.x) : "vcc")
    if (MODE == 10) BODY("v_cmp_eq_u32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %3, vcc", "+v"(a[i].x) : "v"(u[i]), "v"(c3), "v"(b.x) : "vcc")
    if (MODE == 11) BODY("v_maximum3_f32 %0, %0, %1, %1", "+v"(a[i].x) : "v"(b.x))
    if (MODE == 12) BODY("v_cndmask_b32 %0, 0, %1, vcc", "+v"(a[i].x) : "v"(b.x))
    if (MODE == 13) BODY("v_mul_lo_u32 %0, %0, %1", "+v"(u[i]) : "v"(c3))
    if (MODE == 14) BODY("v_add_u32 %0, %0, %1", "+v"(u[i]) : "v"(c3))
    if (MODE == 15) BODY("v_lshlrev_b32 %0, 2, %0", "+v"(u[i]) :)
    if (MODE == 16) BODY("v_sub_f32 %0, %0, %1", "+v"(a[i].x) : "v"(b.x))
    if (MODE == 17) BODY("v_mul_f32 %0, %0, %1", "+v"(a[i].x) : "v"(b.x))
    if (MODE == 18) BODY("v_bfi_b32 %0, %0, %1, %1", "+v"(u[i]) : "v"(c3))
    if (MODE == 19) BODY("v_perm_b32 %0, %0, %1, %1", "+v"(u[i]) : "v"(c3))
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(m & 1);
}
template <int MODE>
void run(const char *name, float *out, int per) {
    for (int threads : {256, 512}) {
        const int iters = 2048, blocks = 256;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<blocks, threads>>>(out, 16, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<blocks, threads>>>(out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * 64 * per;
        const int w = threads / 256;
        printf("%-44s %d wave(s)/SIMD: %7.3f ms -> %6.3f ns per instruction per SIMD\n", name, w, ms, ms * 1e6 / (instr_per_wave * w));
    }
}
int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run<0>("v_add_f32", out, 1);
    run<16>("v_sub_f32", out, 1);
    run<17>("v_mul_f32", out, 1);
    run<1>("v_cndmask_b32 v, v, v, vcc", out, 1);
    run<12>("v_cndmask_b32 v, 0, v, vcc", out, 1);
    run<2>("v_cndmask_b32_e64 v, v, v, s[..]", out, 1);
    run<3>("v_cmp_eq_u32 vcc", out, 1);
    run<4>("v_cmp_eq_u32_e64 s[..]", out, 1);
    run<9>("v_cmp vcc + v_cndmask vcc (pair, /2)", out, 2);
    run<10>("v_cmp vcc + s_nop 1 + v_cndmask (pair, /2)", out, 2);
    run<5>("v_bfe_i32", out, 1);
    run<6>("v_and_b32", out, 1);
    run<18>("v_bfi_b32", out, 1);
    run<19>("v_perm_b32", out, 1);
    run<7>("v_mov_b32_dpp row_shr:1", out, 1);
    run<8>("v_max3_f32", out, 1);
    run<11>("v_maximum3_f32", out, 1);
    run<13>("v_mul_lo_u32", out, 1);
    run<14>("v_add_u32", out, 1);
    run<15>("v_lshlrev_b32", out, 1);
    return 0;
}
