// Issue cost of packed fp32 VALU instructions relative to scalar ones on gfx950 (round 4): the same number of instructions
// per lane, 8 independent chains, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed) {
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = (f32x2){seed + i + threadIdx.x, seed - i};
    f32x2 b = {seed * 0.5f, seed * 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]" : "+v"(a[i]) : "v"(b));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
                if (MODE == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x));
                if (MODE == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (MODE == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, float *out, int threads) {
    const int iters = 4096, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64;
    const int waves_per_simd = threads / 256;
    // ns per instruction per SIMD (all waves of a SIMD together issue waves_per_simd * instr_per_wave instructions)
    printf("%-44s %d wave(s)/SIMD: %7.3f ms  -> %.3f ns per instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / (instr_per_wave * waves_per_simd));
}
int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<0>("v_add_f32", out, threads);
        run<1>("v_pk_add_f32", out, threads);
        run<2>("v_pk_add_f32 op_sel/neg", out, threads);
        run<4>("v_fma_f32", out, threads);
        run<3>("v_pk_fma_f32", out, threads);
        run<7>("v_pk_mul_f32", out, threads);
        run<5>("v_cndmask_b32", out, threads);
        run<6>("v_mov_b32", out, threads);
    }
    return 0;
}
