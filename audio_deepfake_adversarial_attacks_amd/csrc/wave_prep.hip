// wave_prep.hip — gfx950 kernels for the waveform batches either side of the attack loop
// (C ABI: include/advstep_dataset.h).
//
//  * wave_pad_tile_kernel: the reference's per-utterance CPU chain  torchaudio.load(normalize) -> waveform[:1] ->
//    PadDataset.apply_pad  (src/datasets/base_dataset.py:165,104-105,344-355) and the CPU round trip of
//    wavefake_preprocessing_on_batch (:122-148) as ONE pass over a ragged batch: each thread produces four consecutive
//    output samples (one 16-B store), reading channel 0 of frame (t mod len).  Pure HBM streaming: 4 B written per
//    sample, 2-4 B read (PCM16 / float32) — the tiled re-reads of short utterances hit L2.
//  * qual_select_kernel / wave_gather_rows_kernel: AttackAnalyser's selection (attacks_analysis.py:78-84,100-106) and
//    the packing of just the selected rows, so the host copy is n_selected * T floats instead of 2 * B * T.
// All exact (index arithmetic and a power-of-two scale): parity is bit-exact.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "advstep_dataset.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 4;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float decode(const float *p, int64_t i) { return p[i]; }
// torchaudio.load(normalize=True) for 16-bit PCM: sample / 2^15 (exact in float32)
__device__ __forceinline__ float decode(const int16_t *p, int64_t i) { return (float)p[i] * (1.0f / 32768.0f); }

// grid (ceil(cut / 1024), B).  VEC: cut % 4 == 0 and dst 16-B aligned (then every row is).
template <typename SRC, bool VEC>
__global__ __launch_bounds__(kBlock) void wave_pad_tile_kernel(const SRC *__restrict__ src,
                                                               const int64_t *__restrict__ offsets,
                                                               const int64_t *__restrict__ lengths,
                                                               const int32_t *__restrict__ channels,
                                                               float *__restrict__ dst, int64_t cut) {
    const int64_t b = blockIdx.y;
    const int64_t t0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread;
    if (t0 >= cut) return;
    const int64_t len = lengths[b];
    const int64_t ch = channels ? channels[b] : 1;
    const SRC *row = src + offsets[b];
    float v[kPerThread];
    if (len <= 0) {
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) v[k] = 0.0f;
    } else {
        // frame of the first output; the next three advance with wrap-around (len >= cut never wraps)
        int64_t f = len >= cut ? t0 : t0 % len;
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            v[k] = (t0 + k < cut) ? decode(row, f * ch) : 0.0f;
            f = (f + 1 == len) ? 0 : f + 1;
        }
    }
    float *out = dst + b * cut + t0;
    if (VEC) {
        *reinterpret_cast<float4 *>(out) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < kPerThread; ++k)
            if (t0 + k < cut) out[k] = v[k];
    }
}

// One wave: ordered compaction with ballots.  Pass 0 lists the flipped spoof rows (y == 0), pass 1 the flipped bonafide
// rows (y == 1), each ascending — np.where order.
__global__ __launch_bounds__(64) void qual_select_kernel(const int64_t *__restrict__ y,
                                                         const int32_t *__restrict__ clean,
                                                         const int32_t *__restrict__ attacked, int64_t B,
                                                         int32_t *__restrict__ rows, int32_t *__restrict__ counts) {
    const int lane = threadIdx.x;
    int32_t written = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int32_t start = written;
        for (int64_t base = 0; base < B; base += 64) {
            const int64_t i = base + lane;
            bool hit = false;
            if (i < B) {
                const int64_t yi = y[i];
                const int32_t c = clean[i];
                hit = yi == pass && (int64_t)c == yi && c != attacked[i];
            }
            const unsigned long long m = __ballot(hit);
            if (hit) rows[written + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
            written += __popcll(m);
        }
        if (lane == 0) counts[pass] = written - start;
    }
}

// grid (ceil(T / 1024), n)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void wave_gather_rows_kernel(const float *__restrict__ src,
                                                                  const int32_t *__restrict__ rows,
                                                                  float *__restrict__ dst, int64_t T) {
    const int64_t i = blockIdx.y;
    const int64_t t0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread;
    if (t0 >= T) return;
    const float *in = src + (int64_t)rows[i] * T + t0;
    float *out = dst + i * T + t0;
    if (VEC) {
        *reinterpret_cast<float4 *>(out) = *reinterpret_cast<const float4 *>(in);
    } else {
#pragma unroll
        for (int k = 0; k < kPerThread; ++k)
            if (t0 + k < T) out[k] = in[k];
    }
}

}  // namespace

extern "C" int advstep_wave_pad_tile_f32(const void *src, int src_kind, const int64_t *offsets, const int64_t *lengths,
                                         const int32_t *channels, float *dst, int64_t B, int64_t cut,
                                         advstep_stream_t stream) {
    if (B < 0 || cut < 0 || (src_kind != ADVSTEP_WAVE_F32 && src_kind != ADVSTEP_WAVE_PCM16)) return ADVSTEP_EINVAL;
    if (B == 0 || cut == 0) return ADVSTEP_OK;
    if (!src || !offsets || !lengths || !dst || B > 65535) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)ceil_div(cut, (int64_t)kBlock * kPerThread), (unsigned)B);
    const bool vec = (cut % kPerThread) == 0 && aligned16(dst);
    hipStream_t s = as_stream(stream);
    if (src_kind == ADVSTEP_WAVE_F32) {
        const float *p = static_cast<const float *>(src);
        if (vec)
            hipLaunchKernelGGL((wave_pad_tile_kernel<float, true>), grid, dim3(kBlock), 0, s, p, offsets, lengths,
                               channels, dst, cut);
        else
            hipLaunchKernelGGL((wave_pad_tile_kernel<float, false>), grid, dim3(kBlock), 0, s, p, offsets, lengths,
                               channels, dst, cut);
    } else {
        const int16_t *p = static_cast<const int16_t *>(src);
        if (vec)
            hipLaunchKernelGGL((wave_pad_tile_kernel<int16_t, true>), grid, dim3(kBlock), 0, s, p, offsets, lengths,
                               channels, dst, cut);
        else
            hipLaunchKernelGGL((wave_pad_tile_kernel<int16_t, false>), grid, dim3(kBlock), 0, s, p, offsets, lengths,
                               channels, dst, cut);
    }
    return status_after_launch();
}

extern "C" int advstep_qual_select(const int64_t *y, const int32_t *pred_noattack_label, const int32_t *pred_label,
                                   int64_t B, int32_t *rows, int32_t *counts, advstep_stream_t stream) {
    if (B < 0 || B > INT32_MAX || !counts) return ADVSTEP_EINVAL;
    if (B > 0 && (!y || !pred_noattack_label || !pred_label || !rows)) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(qual_select_kernel, dim3(1), dim3(64), 0, as_stream(stream), y, pred_noattack_label, pred_label,
                       B, rows, counts);
    return status_after_launch();
}

extern "C" int advstep_wave_gather_rows_f32(const float *src, const int32_t *rows, float *dst, int64_t n, int64_t T,
                                            advstep_stream_t stream) {
    if (n < 0 || T < 0) return ADVSTEP_EINVAL;
    if (n == 0 || T == 0) return ADVSTEP_OK;
    if (!src || !rows || !dst || n > 65535) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)ceil_div(T, (int64_t)kBlock * kPerThread), (unsigned)n);
    const bool vec = (T % kPerThread) == 0 && aligned16(src) && aligned16(dst);
    if (vec)
        hipLaunchKernelGGL((wave_gather_rows_kernel<true>), grid, dim3(kBlock), 0, as_stream(stream), src, rows, dst, T);
    else
        hipLaunchKernelGGL((wave_gather_rows_kernel<false>), grid, dim3(kBlock), 0, as_stream(stream), src, rows, dst,
                           T);
    return status_after_launch();
}
