/*
 * advstep_frontend.h — C ABI of the fused LFCC tail in libadvstep.so.
 *
 * Reference: src/frontends.py:24-32 `LFCC_FN` = torchaudio.transforms.LFCC(n_lfcc=80, n_filter=128, n_fft=512,
 * win 400, hop 160), called inside `model(adv)` on every attack step (src/models/lcnn.py:233-237).  torchaudio's forward
 * after the STFT is:  |X|^2 -> matmul(linear filterbank 257x128) -> 10*log10(clamp(., 1e-10)) -> max(., batch_max - 80)
 * -> matmul(DCT 128x80), with four transposes in between; under autograd that is ~25 small kernels per direction.
 * Here the STFT (reflect pad + framing + rocFFT) stays with PyTorch and everything after it is two kernels forward and
 * three backward:
 *   bands   : power + SPARSE triangular filterbank (each band touches <= `span` bins) + dB, per-block maxima
 *   project : batch-wide floor + DCT, written FRAME-MAJOR (B, frames, n_lfcc) — the layout LCNN's first block reads,
 *             so the (B, 80, 404) -> (B, 404, 80) transpose copy disappears
 *   backward: DCT^T + floor mask (+ the floored gradients, which torchaudio routes to the batch maximum through
 *             `amax`) + d(dB); then filterbank^T and d|X|^2 back to the complex spectrum.
 * PARITY UNPINNED against torchaudio (absent from the reference tree); pinned against this repository's own torch
 * restatement (frontends.LFCC) within float tolerance (tests/test_gpu_frontend_ops.py).
 * Conventions as in advstep.h.  Layouts are frame-major with the spectral index fastest — the STFT's native one
 * (torch.stft returns a (B, F, NF) view of a (B, NF, F) buffer): spec (B, NF, F) complex64 as float pairs,
 * band_db / dband (B, NF, M), out / dout (B, NF, K).  M <= 128.
 */
#ifndef ADVSTEP_FRONTEND_H_
#define ADVSTEP_FRONTEND_H_

#include <stddef.h>
#include <stdint.h>

#include "advstep.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of per-block maxima `advstep_lfcc_bands_f32` writes (size of `block_max`, floats). */
size_t advstep_lfcc_block_count(int64_t B, int64_t M, int64_t NF);

/* band_db[b, t, m] = 10*log10(max(sum_j fb_w[m, j] * |spec[b, t, fb_start[m] + j]|^2, 1e-10));
 * block_max[i] = maximum of band_db over workgroup i (NaN propagates). fb_w is (M, span), zero padded. */
int advstep_lfcc_bands_f32(const float *spec, const int32_t *fb_start, const float *fb_w, int64_t span, float *band_db,
                           float *block_max, int64_t B, int64_t F, int64_t M, int64_t NF, advstep_stream_t stream);

/* stats[0] = max(block_max[0..n)), stats[1] = 0 (tie counter, filled by the projection), stats[2] = 0 (floored-gradient
 * sum, filled by the backward pass).  stats holds 4 floats. */
int advstep_lfcc_reduce_max_f32(const float *block_max, int64_t n, float *stats, advstep_stream_t stream);

/* out[b, t, k] = sum_m max(band_db[b, t, m], stats[0] - top_db) * dct[m, k]  (dct is (M, K)); also counts, in
 * stats[1], the elements equal to the batch maximum (normally 1).  K must be a multiple of 4 and <= 128. */
int advstep_lfcc_project_f32(const float *band_db, const float *dct, float *stats, float top_db, float *out, int64_t B,
                             int64_t M, int64_t NF, int64_t K, advstep_stream_t stream);

/* dband[b, t, m] = d loss / d (band power) from dout (B, NF, K): DCT^T, floor mask of torch.max (1 / 0.5 / 0), and
 * d(10 log10 clamp) ; the floored share of the gradient is summed into stats[2]. */
int advstep_lfcc_project_backward_f32(const float *dout, const float *dct, const float *band_db, float *stats,
                                      float top_db, float *dband, int64_t B, int64_t M, int64_t NF, int64_t K,
                                      advstep_stream_t stream);

/* ---- the projection pair on the matrix cores (M = 128 bands, K = 80 coefficients: the reference's LFCC; other sizes take
 * the launches above) ------------------------------------------------------------------------------------------------------
 * advstep_lfcc_project_prepare_f32: the DCT matrix re-ordered into the operand fragments of the two kernels, once per weight
 * version (`frag`: advstep_lfcc_project_fragment_floats(M, K) floats, 16-byte aligned, caller-owned; 0 floats = this size has
 * no matrix-core path and `frag` may be NULL everywhere below). */
size_t advstep_lfcc_project_fragment_floats(int64_t M, int64_t K);
int advstep_lfcc_project_prepare_f32(const float *dct, int64_t M, int64_t K, float *frag, advstep_stream_t stream);

/* advstep_lfcc_reduce_max_f32 + advstep_lfcc_project_f32 as ONE launch (src/frontends.py:24-32: the `amplitude_to_DB` floor
 * and the DCT of torchaudio's LFCC): every workgroup of the projection reduces the n block maxima itself and stats becomes
 * {max, 0, 0, 1} - stats[3] == 1 asks advstep_lfcc_project_backward_zero_f32 to count the elements equal to the maximum into
 * stats[1] (so ONE backward pass per forward pass may run on a given stats buffer; re-arm it by zeroing stats[1..2]).
 * frag == NULL or another size: the two launches.  band_db, out: 16-byte aligned. */
int advstep_lfcc_max_project_f32(const float *band_db, const float *dct, const float *frag, const float *block_max, int64_t n,
                                 float *stats, float top_db, float *out, int64_t B, int64_t M, int64_t NF, int64_t K,
                                 advstep_stream_t stream);

/* advstep_lfcc_project_backward_f32 on the matrix cores, which also zero-fills `zero` (zero_n floats; may be NULL / 0) in the
 * same launch: the waveform gradient the overlap-add of advstep_stft_bands_backward_fixup_f32(dx_is_zero = 1) accumulates
 * into (replaces the memset node in front of that kernel).  frag == NULL or another size: memset + the plain call. */
int advstep_lfcc_project_backward_zero_f32(const float *dout, const float *dct, const float *frag, const float *band_db,
                                           float *stats, float top_db, float *dband, int64_t B, int64_t M, int64_t NF,
                                           int64_t K, float *zero, int64_t zero_n, advstep_stream_t stream);

/* torchaudio's floor is `amax(x_db) - top_db`, so the floored gradients flow to the batch maximum: adds
 * stats[2] / stats[1] * d(dB) at every element of band_db equal to stats[0].  No-op when stats[2] == 0. */
int advstep_lfcc_floor_fixup_f32(const float *band_db, const float *stats, float *dband, int64_t n,
                                 advstep_stream_t stream);

/* dspec[b, t, f] = 2 * spec[b, t, f] * sum_j fbt_w[f, j] * dband[b, t, fbt_start[f] + j]  (complex, as float pairs).
 * hermitian_half != 0 pre-scales it for a c2r inverse FFT (interior bins * 1/2, DC / Nyquist imaginary parts 0):
 * the gradient of a one-sided real FFT is then irfft(dspec, norm="forward"). */
int advstep_lfcc_bands_backward_f32(const float *dband, const float *spec, const int32_t *fbt_start, const float *fbt_w,
                                    int64_t span_t, float *dspec, int64_t B, int64_t F, int64_t M, int64_t NF,
                                    int hermitian_half, advstep_stream_t stream);

/* STFT framing of torch.stft(center=True, pad_mode="reflect") without the padded copy and the strided-frame clone:
 * frames[b, f, n] = window[n] * x[b, reflect(f*hop + n - nfft/2)], window (nfft) already centred/zero-padded.
 * frames (B, NF, nfft) is what a batched r2c FFT consumes; requires T > nfft/2. */
int advstep_stft_frames_f32(const float *x, const float *window, float *frames, int64_t B, int64_t T, int64_t NF,
                            int64_t hop, int64_t nfft, advstep_stream_t stream);

/* Transpose of the framing (the backward of the line above): windowed overlap-add with the reflected borders folded
 * back, as a gather (no atomics, deterministic): dx (B, T) from dframes (B, NF, nfft). */
int advstep_stft_overlap_add_f32(const float *dframes, const float *window, float *dx, int64_t B, int64_t T, int64_t NF,
                                 int64_t hop, int64_t nfft, advstep_stream_t stream);

/* ---- the STFT end fused around an in-LDS FFT (n_fft = 512) -------------------------------------------------------------
 * torch.stft(center=True, reflect, window) -> |.|^2 -> filterbank -> 10 log10 in ONE kernel: band_db (B, NF, M) and the
 * per-workgroup maxima (advstep_stft_bands_block_count() floats; feed them to advstep_lfcc_reduce_max_f32).  Frames and
 * spectrum never touch memory.  x (B, T), window (512: the analysis window centred / zero-padded), fb_* as above. */
size_t advstep_stft_bands_block_count(int64_t B, int64_t NF);
int advstep_stft_bands_supported(int64_t nfft, int64_t hop, int64_t T);
int advstep_stft_bands_f32(const float *x, const float *window, const int32_t *fb_start, const float *fb_w, int64_t span,
                           float *band_db, float *block_max, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft,
                           int64_t M, advstep_stream_t stream);

/* Its backward: dx (B, T) from dband (B, NF, M).  The spectrum is recomputed from x (bit-identical to the forward pass:
 * nothing was saved), d|X|^2 -> inverse FFT -> window -> overlap-add, summed in a fixed order (deterministic). */
int advstep_stft_bands_backward_f32(const float *x, const float *window, const float *dband, const int32_t *fbt_start,
                                    const float *fbt_w, int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF,
                                    int64_t hop, int64_t nfft, int64_t M, advstep_stream_t stream);

/* advstep_lfcc_floor_fixup_f32 + advstep_stft_bands_backward_f32 as ONE launch: the fix-up (stats[2] / stats[1] * d(dB) added
 * at every element of band_db equal to stats[0]; nothing when stats[2] == 0) is applied to the band gradients as the kernel
 * reads them - dband itself is not modified unless ADVSTEP_STFT_REG=0 selects the radix-4 kernels, which fix it up in place
 * first.  dx_is_zero != 0: dx has been zero-filled on this stream already (advstep_lfcc_project_backward_zero_f32). */
int advstep_stft_bands_backward_fixup_f32(const float *x, const float *window, float *dband, const float *band_db,
                                          const float *stats, const int32_t *fbt_start, const float *fbt_w, int64_t span_t,
                                          float *dx, int dx_is_zero, int64_t B, int64_t T, int64_t NF, int64_t hop,
                                          int64_t nfft, int64_t M, advstep_stream_t stream);

/* ---- mel-spec frontend (src/frontends.py:53-79) around the same in-LDS FFT --------------------------------------------
 * torch.stft (window 512 = the rectangular 400-sample window centred / zero-padded) -> MelScale applied to the real and
 * the imaginary part -> magnitude and phase:  out (B, 2, M, NF), plane 0 = |Y|, plane 1 = angle(Y),
 * Y[m] = sum_j fb_w[m, j] X[fb_start[m] + j].  M <= 80, span <= 48 (a lane keeps its bands' taps in registers). */
int advstep_stft_mel_f32(const float *x, const float *window, const int32_t *fb_start, const float *fb_w, int64_t span,
                         float *out, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft, int64_t M,
                         advstep_stream_t stream);

/* dx (B, T) from dout (B, 2, M, NF): spectrum and mel bands recomputed, torch's abs / angle backward (0 at Y = 0),
 * filterbank transpose, inverse FFT, overlap-add (deterministic, as above). */
int advstep_stft_mel_backward_f32(const float *x, const float *window, const float *dout, const int32_t *fb_start,
                                  const float *fb_w, int64_t span, const int32_t *fbt_start, const float *fbt_w,
                                  int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft,
                                  int64_t M, advstep_stream_t stream);

/* The same gradient from the forward OUTPUT instead of the waveform: Y = fb X is linear, so dx needs only dY, and dY needs only
 * Y = out[:, 0] * exp(i out[:, 1]) — no framing, forward FFT or band projection is recomputed (about 45 % less work).  `out` is
 * what advstep_stft_mel_f32 wrote for the same x (any rounding of it enters dx at 1e-7 relative). */
int advstep_stft_mel_backward_from_output_f32(const float *window, const float *dout, const float *out, const int32_t *fbt_start,
                                              const float *fbt_w, int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF,
                                              int64_t hop, int64_t nfft, int64_t M, advstep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVSTEP_FRONTEND_H_ */
