// Standalone tuning harness for the headline kernel (advstep_pgd_linf_step_f32): same arithmetic, different
// streaming shapes.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/tune_pgd_step.hip -o /tmp/tune
// Prints GB/s (16 B/sample algorithmic) for each variant, hot (one buffer set) and cold (rotating sets > 256 MB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float sgn(float g) { return (float)(0.0f < g) - (float)(g < 0.0f); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { v = (v < lo) ? lo : v; return (v > hi) ? hi : v; }
__device__ __forceinline__ float step1(float a, float g, float x, float alpha, float eps) {
    a = a + alpha * sgn(g);
    float d = clampf(a - x, -eps, eps);
    return clampf(x + d, 0.0f, 1.0f);
}
__device__ __forceinline__ float4 step4(float4 a, float4 g, float4 x, float alpha, float eps) {
    return make_float4(step1(a.x, g.x, x.x, alpha, eps), step1(a.y, g.y, x.y, alpha, eps),
                       step1(a.z, g.z, x.z, alpha, eps), step1(a.w, g.w, x.w, alpha, eps));
}

typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld(const float4 *p) {
    if (NT) {
        const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st(float4 *p, float4 v) {
    if (NT) {
        vf4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<vf4 *>(p));
    } else {
        *p = v;
    }
}

// one tile (BLOCK * VECS float4) per workgroup, tile-strided
template <int BLOCK, int VECS, bool NTLD, bool NTST>
__global__ __launch_bounds__(BLOCK) void k_tile(const float4 *__restrict__ adv, const float4 *__restrict__ grad,
                                                const float4 *__restrict__ orig, float4 *out, long n4, long ntiles,
                                                float alpha, float eps) {
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long base = tile * (BLOCK * VECS) + threadIdx.x;
        float4 a[VECS], g[VECS], x[VECS];
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const long i = base + (long)j * BLOCK;
            if (i < n4) { a[j] = ld<NTLD>(adv + i); g[j] = ld<NTLD>(grad + i); x[j] = ld<NTLD>(orig + i); }
        }
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const long i = base + (long)j * BLOCK;
            if (i < n4) st<NTST>(out + i, step4(a[j], g[j], x[j], alpha, eps));
        }
    }
}

// nt on the two cold streams (adv, orig) only; grad (just produced by the backward pass) and the store stay default
template <int BLOCK, int VECS>
__global__ __launch_bounds__(BLOCK) void k_mixed(const float4 *__restrict__ adv, const float4 *__restrict__ grad,
                                                 const float4 *__restrict__ orig, float4 *out, long n4, long ntiles,
                                                 float alpha, float eps) {
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long base = tile * (BLOCK * VECS) + threadIdx.x;
        float4 a[VECS], g[VECS], x[VECS];
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const long i = base + (long)j * BLOCK;
            if (i < n4) { a[j] = ld<true>(adv + i); g[j] = ld<false>(grad + i); x[j] = ld<true>(orig + i); }
        }
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const long i = base + (long)j * BLOCK;
            if (i < n4) st<false>(out + i, step4(a[j], g[j], x[j], alpha, eps));
        }
    }
}
template <int BLOCK, int VECS>
void launch_mixed(const float4 *a, const float4 *g, const float4 *x, float4 *o, long n4, hipStream_t s) {
    long ntiles = (n4 + BLOCK * VECS - 1) / (BLOCK * VECS);
    hipLaunchKernelGGL((k_mixed<BLOCK, VECS>), dim3((int)ntiles), dim3(BLOCK), 0, s, a, g, x, o, n4, ntiles, 2.0f / 255, 0.003f);
}

struct Variant { const char *name; void (*launch)(const float4 *, const float4 *, const float4 *, float4 *, long, hipStream_t); };

template <int BLOCK, int VECS, bool NTLD, bool NTST, int MAXGRID>
void launch_tile(const float4 *a, const float4 *g, const float4 *x, float4 *o, long n4, hipStream_t s) {
    long ntiles = (n4 + BLOCK * VECS - 1) / (BLOCK * VECS);
    int grid = (int)(MAXGRID > 0 && ntiles > MAXGRID ? MAXGRID : ntiles);
    hipLaunchKernelGGL((k_tile<BLOCK, VECS, NTLD, NTST>), dim3(grid), dim3(BLOCK), 0, s, a, g, x, o, n4, ntiles, 2.0f / 255, 0.003f);
}

int main(int argc, char **argv) {
    const long B = 128, T = 64600, n = B * T, n4 = n / 4;
    const int sets = 6, launches = 200;
    std::vector<float *> adv(sets), grad(sets), orig(sets), out(sets);
    std::vector<float> h(n);
    for (long i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000003) / 1000003.0f;
    for (int s = 0; s < sets; ++s) {
        CK(hipMalloc(&adv[s], n * 4)); CK(hipMalloc(&grad[s], n * 4)); CK(hipMalloc(&orig[s], n * 4)); CK(hipMalloc(&out[s], n * 4));
        CK(hipMemcpy(adv[s], h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(orig[s], h.data(), n * 4, hipMemcpyHostToDevice));
        for (long i = 0; i < n; ++i) h[i] -= 0.5f;
        CK(hipMemcpy(grad[s], h.data(), n * 4, hipMemcpyHostToDevice));
        for (long i = 0; i < n; ++i) h[i] += 0.5f;
    }
    Variant vs[] = {
        {"b256 v4 (shipped)        ", launch_tile<256, 4, false, false, 4096>},
        {"b256 v2                  ", launch_tile<256, 2, false, false, 0>},
        {"b256 v8                  ", launch_tile<256, 8, false, false, 0>},
        {"b512 v4                  ", launch_tile<512, 4, false, false, 0>},
        {"b512 v2                  ", launch_tile<512, 2, false, false, 0>},
        {"b1024 v2                 ", launch_tile<1024, 2, false, false, 0>},
        {"b256 v4 nt-load          ", launch_tile<256, 4, true, false, 0>},
        {"b256 v4 nt-store         ", launch_tile<256, 4, false, true, 0>},
        {"b256 v4 nt-load nt-store ", launch_tile<256, 4, true, true, 0>},
        {"b256 v4 grid 1024 strided", launch_tile<256, 4, false, false, 1024>},
        {"b256 v4 grid 512 strided ", launch_tile<256, 4, false, false, 512>},
        {"b256 v2 grid 2048 strided", launch_tile<256, 2, false, false, 2048>},
        {"b256 v1                  ", launch_tile<256, 1, false, false, 0>},
        {"b256 v1 nt-load          ", launch_tile<256, 1, true, false, 0>},
        {"b256 v1 nt adv+orig only ", launch_mixed<256, 1>},
        {"b256 v2 nt adv+orig only ", launch_mixed<256, 2>},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        for (auto &v : vs) {
            float ms_hot, ms_cold;
            for (int i = 0; i < 5; ++i) v.launch((float4 *)adv[0], (float4 *)grad[0], (float4 *)orig[0], (float4 *)out[0], n4, 0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < launches; ++i) v.launch((float4 *)adv[0], (float4 *)grad[0], (float4 *)orig[0], (float4 *)out[0], n4, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_hot, e0, e1));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < launches; ++i) { int s = i % sets; v.launch((float4 *)adv[s], (float4 *)grad[s], (float4 *)orig[s], (float4 *)out[s], n4, 0); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_cold, e0, e1));
            // in-situ-like: adv / orig / out cold (rotating), grad hot (one buffer)
            float ms_mix;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < launches; ++i) { int s = i % sets; v.launch((float4 *)adv[s], (float4 *)grad[0], (float4 *)orig[s], (float4 *)out[s], n4, 0); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_mix, e0, e1));
            const double bytes = 16.0 * n;
            printf("%s hot %7.2f us %6.0f GB/s | cold %7.2f us %6.0f GB/s | grad-hot %7.2f us %6.0f GB/s\n", v.name, 1e3 * ms_hot / launches,
                   bytes / (ms_hot / launches * 1e-3) / 1e9, 1e3 * ms_cold / launches, bytes / (ms_cold / launches * 1e-3) / 1e9,
                   1e3 * ms_mix / launches, bytes / (ms_mix / launches * 1e-3) / 1e9);
        }
        printf("--\n");
    }
    return 0;
}
