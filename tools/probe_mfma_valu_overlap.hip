// Does the fp32 matrix pipe run beside the VALU when the two instruction streams come from DIFFERENT waves of one SIMD?
// 256 workgroups x 8 waves (two per SIMD).  mode 1: waves 0-3 issue matrix instructions, waves 4-7 exit;
// mode 2: waves 4-7 issue VALU FMAs, waves 0-3 exit; mode 3: both; mode 4: every wave alternates the two (same totals).
// hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_valu_overlap.hip -o tools/_tune_overlap.bin && tools/_tune_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MFMA_KIND>
__device__ __forceinline__ void mfma_burst(f32x16 (&acc)[4], float a, float b, int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
}

// kind 0: v_fma_f32; 1: v_add_u32 / v_xor (integer); 2: v_cmp + v_cndmask; 3: v_pk_fma_f32; 4: v_add_f32
__device__ __forceinline__ void valu_burst(float (&v)[16], float a, int n, int kind) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    if (kind == 0) {
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], a, 1.0f);
    } else if (kind == 1) {
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned u = __builtin_bit_cast(unsigned, v[j]);
                u = (u + 0x9e3779b9u) ^ (unsigned)i;
                asm volatile("" : "+v"(u));
                v[j] = __builtin_bit_cast(float, u);
            }
    } else if (kind == 2) {
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v[j] = v[j] > a ? v[(j + 1) & 15] : v[j];
                asm volatile("" : "+v"(v[j]));
            }
    } else if (kind == 3) {
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                f32x2 t = {v[j], v[j + 1]};
                t = __builtin_elementwise_fma(t, (f32x2){a, a}, (f32x2){1.0f, 1.0f});
                asm volatile("" : "+v"(t));
                v[j] = t.x;
                v[j + 1] = t.y;
            }
    } else {
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) { v[j] = v[j] + a; asm volatile("" : "+v"(v[j])); }
    }
}

__global__ __launch_bounds__(512) void probe(float *out, int mode, int n_mfma, int n_valu, int rounds, int kind) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (float)j;
    const float a = out[0], b = out[1];
    const bool do_m = mode == 4 || ((mode & 1) && wave < 4), do_v = mode == 4 || ((mode & 2) && wave >= 4);
    if (mode >= 5) {
        // every wave: one matrix instruction, then `per` VALU FMAs, repeated (same totals per SIMD as modes 3 / 4)
        const int total_mfma = rounds * n_mfma * 4 / 2;
        if (mode == 5) {
            for (int i = 0; i < total_mfma; i += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], a, 1.0f);
                }
            }
        } else {
            // mode 6: two matrix instructions then 32 FMAs; mode 7: 4 then 64
            const int grp = mode == 6 ? 2 : 4;
            for (int i = 0; i < total_mfma; i += 4) {
#pragma unroll
                for (int q = 0; q < 4; q += 1) {
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
                    if ((q + 1) % grp == 0) {
#pragma unroll
                        for (int rr = 0; rr < grp; ++rr)
#pragma unroll
                            for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], a, 1.0f);
                    }
                }
            }
        }
    } else
    for (int r = 0; r < rounds; ++r) {
        if (do_m) mfma_burst<0>(acc, a, b, mode == 4 ? n_mfma / 2 : n_mfma);
        if (do_v) valu_burst(v, a, mode == 4 ? n_valu / 2 : n_valu, kind);
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
    if (s == 123.456f) out[2] = s;
}

int main() {
    float *d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n_mfma = 26, n_valu = 104, rounds = 64;   // per round: 104 MFMA (6656 cycles) vs 1664 VALU (6656 cycles)
    for (int kind = 0; kind < 5; ++kind)
    for (int mode = 1; mode <= 7; ++mode) {
        if (kind > 0 && (mode == 1 || mode >= 5)) continue;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, mode, n_mfma, n_valu, rounds, kind);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("valu kind %d mode %d: %.1f us\n", kind, mode, ms * 1e3f);
        }
    }
    return 0;
}
