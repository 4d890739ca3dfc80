// probe: v_mfma_f32_4x4x1_16B_f32 — operand / result lane mapping and issue cost next to v_mfma_f32_16x16x4_f32 (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_4x4.hip -o /tmp/probe_mfma_4x4 && /tmp/probe_mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void map_kernel(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

template <int MODE>
__global__ void time_kernel(float *out, int iters) {
    f32x4 acc[16], t4[16];
    for (int i = 0; i < 16; ++i) { acc[i] = (f32x4){0, 0, 0, 0}; t4[i] = (f32x4){0, 0, 0, 0}; }
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) t4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, t4[i], 0, 0, 0);
            if (MODE == 2) t4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, t4[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3] + t4[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) { a[l] = 1.0f + l; b[l] = 100.0f * (1 + l); }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    map_kernel<<<1, 64>>>(da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane l = 4 * blk + j ; D[l][r] = A[4 * blk + r] * B[4 * blk + j]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float want = a[4 * (l / 4) + r] * b[l];
            if (d[l * 4 + r] != want) ++bad;
        }
    printf("mapping D[lane][r] == A[4*(lane/4)+r] * B[lane]: %s (%d mismatches); lane 5: %g %g %g %g\n", bad ? "NO" : "yes", bad,
           d[20], d[21], d[22], d[23]);
    float *out;
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) time_kernel<0><<<1024, 256>>>(out, iters);
            if (mode == 1) time_kernel<1><<<1024, 256>>>(out, iters);
            if (mode == 2) time_kernel<2><<<1024, 256>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.3f ms\n", mode, mode == 0 ? "16 x 16x16x4" : mode == 1 ? "32 x 16x16x4" : "16 x 16x16x4 + 16 x 4x4x1_16B", ms);
    }
    return 0;
}
