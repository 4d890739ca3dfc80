// lfcc_stft.hip — the STFT end of the LFCC frontend fused around an in-LDS FFT (C ABI: include/advstep_frontend.h).
//
// Reference op chain (src/frontends.py:24-32 -> torchaudio LFCC): torch.stft(n_fft 512, hop 160, hann 400 centred,
// center=True, reflect) -> |.|^2 -> linear filterbank (257 -> 128) -> 10 log10, and its backward.  Done with library
// FFTs this is framing kernel -> rocFFT r2c (2 kernels) -> filterbank kernel, with the (B, NF, 512) frames and the
// (B, NF, 257) complex spectrum (106 MB each at B = 128) written and re-read in between — and the mirror image on the
// way back.  Here ONE kernel per direction: a wave owns a frame, gathers its 512 windowed samples straight from the
// waveform (33 MB, L2-resident), runs a 256-point complex radix-4 Stockham FFT in LDS (the real 512-point transform
// by even/odd packing), and
//   forward : power -> sparse filterbank -> dB -> band_db row (512 B, coalesced) + the workgroup maximum;
//   backward: recomputes the frame's spectrum (same code, same bits: nothing was saved), forms
//             d|X|^2 = 2 X (fb^T dband), inverse-transforms, applies the window and overlap-adds.  A workgroup owns 4
//             consecutive frames: their windowed gradients meet in LDS and every output sample is summed in frame
//             order (deterministic); the <= 2 workgroups that share a border sample combine with one float atomic
//             each (two operands: commutative, so still deterministic).  dx must be zeroed by the caller.
// HBM traffic per direction: waveform + band rows (33 + 26 MB) instead of ~0.5 GB.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_frontend.h"

namespace {

constexpr int kN = 256;            // complex FFT length (n_fft = 512 real samples)
constexpr int kNfft = 2 * kN;
constexpr int kBins = kN + 1;      // one-sided spectrum
constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * 64;
constexpr int kFramesPerBlockBwd = 4;   // one per wave: 35 KB of LDS per workgroup = 4 workgroups (16 waves) per CU
constexpr float kAmin = 1e-10f;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 conjf2(float2 a) { return make_float2(a.x, -a.y); }

// The FFT buffers of a wave are private to it, and a wave's LDS instructions execute in program order: ordering one
// stage's writes before the next stage's reads needs no s_barrier, only that the compiler keeps the order and waits for
// the outstanding LDS operations (workgroup-scope fences on LDS lower to s_waitcnt lgkmcnt(0)).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ float max_nan(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    return a > b ? a : b;
}

// 256-point complex FFT of one wave's frame, radix 4, Stockham autosort: 4 stages ping-ponging between `a` and `b`
// (float2[272] each, padded, private to the wave).  tw: the per-stage twiddle tables (see fill_twiddles).  INV conjugates the twiddles and the
// butterfly (unnormalised inverse).  Result in `a` (4 stages = even number of swaps).  One wave-level sync per stage
// orders this stage's writes before the next stage's reads (and, the buffers alternating, the next-but-one stage's
// writes after this stage's reads).  `a` must be complete (and synced) on entry.
// LDS index swizzle (/opt/skills/guides/MI355X_MICROARCH.md, LDS table): a ds_read_b64 is served in two groups of 32 lanes
// over 64 four-byte banks — conflict-free when the 32 float2 slot numbers differ in their low 5 bits; a ds_write_b64 in four
// groups of 16 lanes over 32 banks — conflict-free when the 16 slot numbers differ in their low 4 bits.  Across such a group
// the transform's accesses vary these index bits: reads, the last stage's writes and the frame load bits {0..4} / {0..3};
// the Stockham writes of stage 0 (stride 4) bits {2,3,4,5}, of stage 1 (groups of 4 every 16) bits {0,1,4,5}, of stage 2
// bits {0..3}.  XOR-ing bit 4 into bits 0 and 2 and bit 5 into bits 1 and 3 makes the low 4 (reads: 5) bits a bijection of
// every one of those sets.  (Round 1 padded one slot every 16, which is conflict-free for none of the write patterns: 37 %
// of the kernels' LDS cycles were bank conflicts.)  Every access to an FFT buffer goes through P().
__device__ __forceinline__ int P(int i) { return i ^ (((i >> 4) & 3) * 5); }
constexpr int kNPad = kN + kN / 16;
constexpr int kStageTw = 3 * (4 + 16 + 64);   // per-stage twiddle tables of the radix-4 stages 1..3

template <bool INV>
__device__ __forceinline__ void fft256(float2 *a, float2 *b, const float2 *__restrict__ tw, int lane) {
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
        const int Ns = 1 << (2 * stage);
        const int k = lane & (Ns - 1);
        float2 v0 = a[P(lane)], v1 = a[P(lane + 64)], v2 = a[P(lane + 128)], v3 = a[P(lane + 192)];
        if (stage > 0) {
            // per-stage tables, [t - 1][k] contiguous in k: the lanes of a stage read consecutive entries (a shared
            // 256-entry table would be read at strides of 4, 8, 12 entries in stage 2: 4- to 8-way bank conflicts)
            const float2 *ts = tw + (stage == 1 ? 0 : (stage == 2 ? 12 : 60));
            float2 w1 = ts[k], w2 = ts[Ns + k], w3 = ts[2 * Ns + k];
            if (INV) w1 = conjf2(w1), w2 = conjf2(w2), w3 = conjf2(w3);
            v1 = cmul(v1, w1), v2 = cmul(v2, w2), v3 = cmul(v3, w3);
        }
        const float2 s02 = cadd(v0, v2), d02 = csub(v0, v2), s13 = cadd(v1, v3), d13 = csub(v1, v3);
        // forward: -i * d13 = (d13.y, -d13.x); inverse: +i * d13 = (-d13.y, d13.x)
        const float2 r = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
        const int j0 = ((lane - k) << 2) + k;
        b[P(j0)] = cadd(s02, s13);
        b[P(j0 + Ns)] = cadd(d02, r);
        b[P(j0 + 2 * Ns)] = csub(s02, s13);
        b[P(j0 + 3 * Ns)] = csub(d02, r);
        wave_lds_sync();
        float2 *t = a;
        a = b;
        b = t;
    }
}

// frame f of utterance xb -> z[n] = w[2n] x[.] + i w[2n+1] x[.] (even / odd packing), reflect padding of nfft/2
__device__ __forceinline__ void load_frame(const float *__restrict__ xb, const float *__restrict__ w, int T, int f,
                                           int hop, float2 *z, int lane) {
    const int q_first = f * hop - kN;
    if (q_first >= 0 && q_first + kNfft <= T && ((q_first & 1) == 0) && ((reinterpret_cast<uintptr_t>(xb) & 7u) == 0)) {
        // interior frame (all but the first and last two at hop 160): no reflection, 8-byte loads
        const float2 *x2 = reinterpret_cast<const float2 *>(xb + q_first);
        const float2 *w2 = reinterpret_cast<const float2 *>(w);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = lane + 64 * i;
            const float2 xv = x2[n], wv = w2[n];
            z[P(n)] = make_float2(wv.x * xv.x, wv.y * xv.y);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = lane + 64 * i;           // complex index; real samples 2n, 2n + 1
        int q0 = f * hop + 2 * n - kN, q1 = q0 + 1;
        q0 = q0 < 0 ? -q0 : q0;
        q0 = q0 >= T ? 2 * (T - 1) - q0 : q0;
        q1 = q1 < 0 ? -q1 : q1;
        q1 = q1 >= T ? 2 * (T - 1) - q1 : q1;
        z[P(n)] = make_float2(w[2 * n] * xb[q0], w[2 * n + 1] * xb[q1]);
    }
}

// Z (256-point FFT of the packed frame) -> X[k], k = 0 .. 256, written to xs (float2[257])
__device__ __forceinline__ void unpack_real(const float2 *Z, const float2 *__restrict__ tw512, float2 *xs, int lane) {
    // tw512[k] = exp(-2 pi i k / 512), k = 0 .. 255
    for (int k = lane; k < kN; k += 64) {
        const float2 zk = Z[P(k)], zc = conjf2(Z[P((kN - k) & (kN - 1))]);
        const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
        const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
        // X[k] = E + W^k * (-i d)
        const float2 o = cmul(tw512[k], make_float2(d.y, -d.x));
        xs[k] = cadd(e, o);
    }
    if (lane == 0) xs[kN] = make_float2(Z[P(0)].x - Z[P(0)].y, 0.0f);
}

struct Lds {
    float2 a[kWavesPerBlock][kNPad];
    float2 b[kWavesPerBlock][kNPad];   // second FFT buffer; between transforms its first 257 entries hold the half spectrum
    __device__ __forceinline__ float2 *xs_of(int wave) { return b[wave]; }
    __device__ __forceinline__ const float2 *xs_of(int wave) const { return b[wave]; }
    float2 tw[kStageTw];  // stage s = 1, 2, 3 (Ns = 4^s): [t - 1][k] = exp(-2 pi i t k / (4 Ns)), t = 1..3, k < Ns
    float2 tw512[kN];     // exp(-2 pi i k / 512)
    float red[kWavesPerBlock];
};

// twiddle tables (device global, written once per process by the first launch): [0, 252) the per-stage tables,
// [256, 512) exp(-2 pi i k / 512)
__device__ float2 g_twiddles[2 * kN];

__global__ void stft_twiddle_kernel() {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= 2 * kN) return;
    double ang = 0.0;
    if (m >= kN) {
        ang = -2.0 * M_PI * (m - kN) / 512.0;
    } else if (m < kStageTw) {
        const int stage = m < 12 ? 1 : (m < 60 ? 2 : 3), base = stage == 1 ? 0 : (stage == 2 ? 12 : 60), Ns = 1 << (2 * stage);
        const int t = (m - base) / Ns + 1, k = (m - base) % Ns;
        ang = -2.0 * M_PI * t * k / (4.0 * Ns);
    }
    g_twiddles[m] = make_float2((float)cos(ang), (float)sin(ang));
}

__device__ __forceinline__ void fill_twiddles(Lds &L) {
    for (int m = threadIdx.x; m < kN; m += kThreads) {
        if (m < kStageTw) L.tw[m] = g_twiddles[m];
        L.tw512[m] = g_twiddles[kN + m];
    }
}

// grid (ceil(NF / 16), B): wave w of workgroup g owns frames 16 g + 4 r + w, r = 0 .. 3
constexpr int kFramesPerBlockFwd = 16;

__global__ __launch_bounds__(kThreads) void stft_bands_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const int32_t *__restrict__ fb_start,
                                                              const float *__restrict__ fb_w, int span,
                                                              float *__restrict__ band_db, float *__restrict__ bmax, int T,
                                                              int NF, int hop, int M) {
    __shared__ Lds L;
    fill_twiddles(L);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = blockIdx.y;
    float vmax = -INFINITY;
    for (int r = 0; r < kFramesPerBlockFwd / kWavesPerBlock; ++r) {
        const int f = blockIdx.x * kFramesPerBlockFwd + r * kWavesPerBlock + wave;
        if (f >= NF) break;                       // wave-uniform; nothing below needs the other waves
        load_frame(x + b * T, w, T, f, hop, L.a[wave], lane);
        wave_lds_sync();
        fft256<false>(L.a[wave], L.b[wave], L.tw, lane);
        unpack_real(L.a[wave], L.tw512, L.xs_of(wave), lane);
        wave_lds_sync();
        float *row = band_db + (b * NF + f) * M;
        for (int m = lane; m < M; m += 64) {
            const int f0 = fb_start[m];
            float band = 0.0f;
            for (int j = 0; j < span; ++j) {
                const int k = f0 + j;
                if (k < kBins) {
                    const float2 z = L.xs_of(wave)[k];
                    band = fmaf(fb_w[m * span + j], fmaf(z.x, z.x, z.y * z.y), band);
                }
            }
            // 10 log10(v) = 3.0103 log2(v): the hardware log2 (1 ulp) keeps the dB value within 1e-6 relative
            const float db = 3.010299956639812f * __log2f(band > kAmin ? band : (band != band ? band : kAmin));
            row[m] = db;
            vmax = max_nan(vmax, db);
        }
        wave_lds_sync();                          // xs / a are rewritten by the next frame
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = max_nan(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) L.red[wave] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = L.red[0];
        for (int i = 1; i < kWavesPerBlock; ++i) r = max_nan(r, L.red[i]);
        bmax[b * gridDim.x + blockIdx.x] = r;
    }
}

constexpr int kMaxSpanT = 8, kMaxBands = 128;   // bands per bin: 2 for the linear bank, up to 6 for 128 mel bands at low frequencies

// ---- shared pieces of the two backward kernels ------------------------------------------------------------------------
// frame f -> X[0 .. 256] in L.xs_of(wave) (bit-identical to the forward pass: same code, same inputs)
__device__ __forceinline__ void frame_spectrum(const float *__restrict__ xb, const float *__restrict__ w, int T, int f,
                                               int hop, Lds &L, int wave, int lane) {
    load_frame(xb, w, T, f, hop, L.a[wave], lane);
    wave_lds_sync();
    fft256<false>(L.a[wave], L.b[wave], L.tw, lane);
    unpack_real(L.a[wave], L.tw512, L.xs_of(wave), lane);
    wave_lds_sync();
}

// L.xs_of(wave) = gradient w.r.t. the one-sided spectrum (dL/dRe, dL/dIm per bin) -> windowed frame gradient in dst[512].
// The one-sided inverse counts interior bins twice: they are halved, imaginary parts of DC / Nyquist dropped; then
// Z'[k] = (G[k] + G*[N-k]) + i W^-k (G[k] - G*[N-k]) is the packed input of the unnormalised 256-point inverse.
__device__ __forceinline__ float2 one_sided_scale(float2 gk, int k) {
    if (k == 0 || k == kN) gk.y = 0.0f;
    else gk.x *= 0.5f, gk.y *= 0.5f;
    return gk;
}

template <bool PRESCALED>
__device__ __forceinline__ void spectrum_grad_to_frame(Lds &L, int wave, int lane, const float *__restrict__ w, float *dst,
                                                       bool live) {
    if (!PRESCALED) {
        for (int k = lane; k < kBins; k += 64) L.xs_of(wave)[k] = one_sided_scale(L.xs_of(wave)[k], k);
        wave_lds_sync();
    }
    for (int k = lane; k < kN; k += 64) {
        const float2 gk = L.xs_of(wave)[k], gc = conjf2(L.xs_of(wave)[kN - k]);
        const float2 e = cadd(gk, gc), d = csub(gk, gc);
        const float2 wd = cmul(conjf2(L.tw512[k]), d);
        L.a[wave][P(k)] = make_float2(e.x - wd.y, e.y + wd.x);    // e + i * wd
    }
    wave_lds_sync();
    fft256<true>(L.a[wave], L.b[wave], L.tw, lane);
    // z'[n] = dframe[2n] + i dframe[2n + 1]; window, park in LDS
    for (int n = lane; n < kN; n += 64) {
        const float2 z = L.a[wave][P(n)];
        const float2 wv = reinterpret_cast<const float2 *>(w)[n];
        reinterpret_cast<float2 *>(dst)[n] = live ? make_float2(wv.x * z.x, wv.y * z.y) : make_float2(0.0f, 0.0f);
    }
    wave_lds_sync();
}

// Overlap-add of a workgroup's windowed frame gradients (dframe) into dx: every sample the frames touch is summed over
// (positions that read it) x (frames) in a fixed order; positions are padded coordinates p = q + nfft/2.  Reflections
// only exist next to the two ends of the signal.
__device__ __forceinline__ void overlap_add_block(const float (*dframe)[kNfft], float *__restrict__ dxb, int f_base, int NF,
                                                  int hop, int T) {
    const int p_lo = f_base * hop;                                          // first padded position of the first frame
    const int frames_here = (NF - f_base) < kFramesPerBlockBwd ? (NF - f_base) : kFramesPerBlockBwd;
    const int p_hi = p_lo + (frames_here - 1) * hop + kNfft;                // one past the last
    auto add_from = [&](int p, float &acc) {     // frames in ascending order: a fixed summation order
        if (p < p_lo || p >= p_hi) return;
#pragma unroll
        for (int fl = 0; fl < kFramesPerBlockBwd; ++fl) {
            const int n = p - p_lo - fl * hop;
            if (fl < frames_here && n >= 0 && n < kNfft) acc += dframe[fl][n];
        }
    };
    const bool near_left = p_lo < 2 * kN, near_right = p_hi > T;            // wave-uniform, almost always false
    const int t_first = p_lo - kN, span_all = p_hi - p_lo;
    for (int i = threadIdx.x; i < span_all; i += kThreads) {
        const int t = t_first + i;
        if (t < 0 || t >= T) continue;            // padded positions outside the signal are reached by reflection below
        float acc = 0.0f;
        add_from(t + kN, acc);
        if (near_left && t >= 1 && t <= kN) add_from(kN - t, acc);                               // left reflection
        if (near_right && t <= T - 2 && t >= T - 1 - kN) add_from(2 * (T - 1) - t + kN, acc);    // right reflection
        if (acc != 0.0f) atomicAdd(dxb + t, acc);
    }
    if (!(near_left || near_right)) return;
    // samples reached ONLY through a reflection from this workgroup's range (their direct position belongs to another
    // workgroup's range or to none): left edge t = pad - p for p < pad, right edge t = 2 (T - 1) - (p - pad) for p - pad >= T
    for (int i = threadIdx.x; i < span_all; i += kThreads) {
        const int p = p_lo + i, q = p - kN;
        int t = -1;
        if (q < 0) t = -q;
        else if (q >= T) t = 2 * (T - 1) - q;
        if (t < 0 || t >= T) continue;
        const int pd = t + kN;
        if (pd >= p_lo && pd < p_hi) continue;    // the first loop already took this contribution
        float acc = 0.0f;
        add_from(p, acc);
        if (acc != 0.0f) atomicAdd(dxb + t, acc);
    }
}

template <int SPAN_CAP>
struct LdsBwd {
    Lds c;
    float dframe[kFramesPerBlockBwd][kNfft];   // windowed frame gradients of this workgroup's frames
    float fbt_w[kBins * SPAN_CAP];             // transposed filterbank, staged once per workgroup
    int32_t fbt_start[kBins];
    float drow[kWavesPerBlock][kMaxBands];     // the current frame's band gradients
};

// grid (ceil(NF / 4), B): a workgroup owns 4 consecutive frames (one per wave).  SPAN_CAP >= span_t sizes the LDS copy of
// the transposed filterbank.
template <int SPAN_CAP>
__global__ __launch_bounds__(kThreads) void stft_bands_backward_kernel(const float *__restrict__ x,
                                                                       const float *__restrict__ w,
                                                                       const float *__restrict__ dband,
                                                                       const int32_t *__restrict__ fbt_start,
                                                                       const float *__restrict__ fbt_w, int span_t,
                                                                       float *__restrict__ dx, int T, int NF, int hop,
                                                                       int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LdsBwd<SPAN_CAP> &S = *reinterpret_cast<LdsBwd<SPAN_CAP> *>(raw);
    Lds &L = S.c;
    {
        // the workgroup's tables -> LDS: every load issued before the first LDS write (seven dependent load -> store
        // round trips were ~40 % of a workgroup's 4-frame lifetime)
        static_assert(kN == kThreads, "one twiddle pair per thread");
        constexpr int kWIt = (kBins * SPAN_CAP + kThreads - 1) / kThreads;
        const int m = threadIdx.x, nw = kBins * span_t;
        const float2 t0 = g_twiddles[m < kStageTw ? m : 0], t1 = g_twiddles[kN + m];
        const int32_t s0 = fbt_start[m], s1 = fbt_start[kThreads + (m < kBins - kThreads ? m : 0)];
        float wq[kWIt];
#pragma unroll
        for (int j = 0; j < kWIt; ++j) {
            const int i = m + j * kThreads;
            wq[j] = fbt_w[i < nw ? i : nw - 1];
        }
        if (m < kStageTw) L.tw[m] = t0;
        L.tw512[m] = t1;
        S.fbt_start[m] = s0;
        if (m < kBins - kThreads) S.fbt_start[kThreads + m] = s1;
#pragma unroll
        for (int j = 0; j < kWIt; ++j) {
            const int i = m + j * kThreads;
            if (i < nw) S.fbt_w[i] = wq[j];
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = blockIdx.y;
    const int f_base = blockIdx.x * kFramesPerBlockBwd;
    const float *xb = x + b * T;
    for (int r = 0; r < kFramesPerBlockBwd / kWavesPerBlock; ++r) {
        const int fl = r * kWavesPerBlock + wave;          // local frame index
        const int f = f_base + fl;
        const bool live = f < NF;
        // band gradients of the frame -> LDS (needed after the FFT; the load latency hides behind it)
        const float *drow_g = dband + (b * NF + (live ? f : NF - 1)) * M;
        for (int m = lane; m < M; m += 64) S.drow[wave][m] = drow_g[m];
        load_frame(xb, w, T, live ? f : NF - 1, hop, L.a[wave], lane);
        wave_lds_sync();
        fft256<false>(L.a[wave], L.b[wave], L.tw, lane);
        // X[k] from the packed transform and, in the same pass, G[k] = 2 X[k] * sum_j fbt_w[k, j] dband[fbt_start[k] + j],
        // already scaled for the one-sided inverse: the spectrum itself is never stored
        const float *drow = S.drow[wave];
        auto grad_of_bin = [&](int k, float2 xk) {
            const int m0 = S.fbt_start[k];
            float dp = 0.0f;
            for (int j = 0; j < span_t; ++j) {
                const int m = m0 + j;
                if (m < M) dp = fmaf(S.fbt_w[k * span_t + j], drow[m], dp);
            }
            return one_sided_scale(make_float2(2.0f * xk.x * dp, 2.0f * xk.y * dp), k);
        };
        {
            const float2 *Z = L.a[wave];
            for (int k = lane; k < kN; k += 64) {
                const float2 zk = Z[P(k)], zc = conjf2(Z[P((kN - k) & (kN - 1))]);
                const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
                const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
                const float2 o = cmul(L.tw512[k], make_float2(d.y, -d.x));
                L.xs_of(wave)[k] = grad_of_bin(k, cadd(e, o));
            }
            if (lane == 0) L.xs_of(wave)[kN] = grad_of_bin(kN, make_float2(Z[P(0)].x - Z[P(0)].y, 0.0f));
        }
        wave_lds_sync();
        spectrum_grad_to_frame<true>(L, wave, lane, w, S.dframe[fl], live);
    }
    __syncthreads();   // all frames' windowed gradients are in LDS
    overlap_add_block(S.dframe, dx + b * T, f_base, NF, hop, T);
}

// ---- mel-spec frontend (src/frontends.py:53-79): STFT -> MelScale on the real and imaginary parts -> |.|, angle ----------
// out (B, 2, M, NF): plane 0 the magnitude, plane 1 the phase of  Y[m] = sum_k fb[k, m] X[k].
constexpr int kMelMax = 80;
constexpr int kMelMaxSpan = 48;     // longest run of bins under one band the kernels take (80 HTK bands over 257 bins: 16)

struct LdsMel {
    Lds c;
    float2 y[kWavesPerBlock][kMelMax];                       // a frame's complex mel bands
    float stage[2][kMelMax][kFramesPerBlockFwd + 1];         // (plane, band, frame) tile of the output / its gradient
};

// A lane owns bands `lane` and `lane + 64` (M <= 80) and keeps their taps in registers: the filterbank is the same for every
// frame a wave transforms, and a per-tap table load in the band loop (one dependent L1 round trip per tap, 2 x span per
// frame) was most of the mel kernels' time.  Taps past the band's run, past `span` or past the last bin are 0 with a clamped
// bin index: `fma(0, X[k], acc)` leaves acc unchanged.
template <int CAP>
struct MelTaps {
    float w[2][CAP];
    int f0[2];
};

template <int CAP>
__device__ __forceinline__ void load_mel_taps(MelTaps<CAP> &t, int lane, const int32_t *__restrict__ fb_start,
                                              const float *__restrict__ fb_w, int span, int M) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int m = lane + 64 * pass;
        const bool live = m < M;
        const int f0 = live ? fb_start[m] : 0;
        t.f0[pass] = f0;
#pragma unroll
        for (int j = 0; j < CAP; ++j) t.w[pass][j] = (live && j < span && f0 + j < kBins) ? fb_w[m * span + j] : 0.0f;
    }
}

template <int CAP>
__device__ __forceinline__ void mel_bands(const Lds &L, int wave, int lane, const MelTaps<CAP> &t, int M, float2 *y) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int m = lane + 64 * pass;
        if (m < M) {
            float re = 0.0f, im = 0.0f;
#pragma unroll
            for (int j = 0; j < CAP; ++j) {
                const int k = t.f0[pass] + j < kBins ? t.f0[pass] + j : kBins - 1;
                const float2 z = L.xs_of(wave)[k];
                re = fmaf(t.w[pass][j], z.x, re);
                im = fmaf(t.w[pass][j], z.y, im);
            }
            y[m] = make_float2(re, im);
        }
    }
}

// grid (ceil(NF / 16), B)
template <int CAP>
__global__ __launch_bounds__(kThreads) void stft_mel_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const int32_t *__restrict__ fb_start,
                                                            const float *__restrict__ fb_w, int span,
                                                            float *__restrict__ out, int T, int NF, int hop, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LdsMel &S = *reinterpret_cast<LdsMel *>(raw);
    Lds &L = S.c;
    fill_twiddles(L);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = blockIdx.y;
    const int f_base = blockIdx.x * kFramesPerBlockFwd;
    MelTaps<CAP> taps;
    load_mel_taps(taps, lane, fb_start, fb_w, span, M);
    for (int r = 0; r < kFramesPerBlockFwd / kWavesPerBlock; ++r) {
        const int fl = r * kWavesPerBlock + wave, f = f_base + fl;
        if (f >= NF) break;                       // wave-uniform
        frame_spectrum(x + b * T, w, T, f, hop, L, wave, lane);
        mel_bands(L, wave, lane, taps, M, S.y[wave]);
        wave_lds_sync();
        for (int m = lane; m < M; m += 64) {
            const float2 v = S.y[wave][m];
            S.stage[0][m][fl] = sqrtf(v.x * v.x + v.y * v.y);
            S.stage[1][m][fl] = atan2f(v.y, v.x);
        }
        wave_lds_sync();
    }
    __syncthreads();
    // (plane, band) rows of up to 16 consecutive frames
    const int frames_here = (NF - f_base) < kFramesPerBlockFwd ? (NF - f_base) : kFramesPerBlockFwd;
    for (int i = threadIdx.x; i < 2 * M * kFramesPerBlockFwd; i += kThreads) {
        const int fl = i % kFramesPerBlockFwd, row = i / kFramesPerBlockFwd, plane = row / M, m = row - plane * M;
        if (fl < frames_here) out[((b * 2 + plane) * M + m) * NF + f_base + fl] = S.stage[plane][m][fl];
    }
}

struct LdsMelBwd {
    Lds c;
    float dframe[kFramesPerBlockBwd][kNfft];
    float2 y[kWavesPerBlock][kMelMax];
    float stage[2][kMelMax][kFramesPerBlockBwd + 1];
};

// grid (ceil(NF / 4), B).  dout (B, 2, M, NF) -> dx (B, T), dx zeroed by the host entry point.
// SPT >= span_t: a lane owns bins lane + 64 p (p < 5) and keeps their transposed-filterbank taps in registers as well, loaded
// before the transform so the table's latency hides behind it.
template <int CAP, int SPT>
__global__ __launch_bounds__(kThreads) void stft_mel_backward_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                     const float *__restrict__ dout,
                                                                     const int32_t *__restrict__ fb_start,
                                                                     const float *__restrict__ fb_w, int span,
                                                                     const int32_t *__restrict__ fbt_start,
                                                                     const float *__restrict__ fbt_w, int span_t,
                                                                     float *__restrict__ dx, int T, int NF, int hop, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LdsMelBwd &S = *reinterpret_cast<LdsMelBwd *>(raw);
    Lds &L = S.c;
    fill_twiddles(L);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = blockIdx.y;
    const int f_base = blockIdx.x * kFramesPerBlockBwd;
    const int frames_here = (NF - f_base) < kFramesPerBlockBwd ? (NF - f_base) : kFramesPerBlockBwd;
    for (int i = threadIdx.x; i < 2 * M * kFramesPerBlockBwd; i += kThreads) {
        const int fl = i % kFramesPerBlockBwd, row = i / kFramesPerBlockBwd, plane = row / M, m = row - plane * M;
        S.stage[plane][m][fl] = fl < frames_here ? dout[((b * 2 + plane) * M + m) * NF + f_base + fl] : 0.0f;
    }
    __syncthreads();
    const float *xb = x + b * T;
    MelTaps<CAP> taps;
    load_mel_taps(taps, lane, fb_start, fb_w, span, M);
    constexpr int kBinPasses = (kBins + 63) / 64;
    float tw_t[kBinPasses][SPT];
    int m0_t[kBinPasses];
#pragma unroll
    for (int p = 0; p < kBinPasses; ++p) {
        const int k = lane + 64 * p;
        const bool in = k < kBins;
        m0_t[p] = in ? fbt_start[k] : 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) tw_t[p][j] = (in && j < span_t && m0_t[p] + j < M) ? fbt_w[k * span_t + j] : 0.0f;
    }
    for (int r = 0; r < kFramesPerBlockBwd / kWavesPerBlock; ++r) {
        const int fl = r * kWavesPerBlock + wave, f = f_base + fl;
        const bool live = f < NF;
        frame_spectrum(xb, w, T, live ? f : NF - 1, hop, L, wave, lane);
        mel_bands(L, wave, lane, taps, M, S.y[wave]);
        wave_lds_sync();
        // d(|Y|, angle Y) -> (dL/dRe Y, dL/dIm Y): torch's abs / angle backward, 0 at Y = 0
        for (int m = lane; m < M; m += 64) {
            const float2 v = S.y[wave][m];
            const float gm = S.stage[0][m][fl], gp = S.stage[1][m][fl];
            const float n2 = v.x * v.x + v.y * v.y;
            float2 gy = make_float2(0.0f, 0.0f);
            if (n2 > 0.0f) {
                const float inv = 1.0f / sqrtf(n2), inv2 = 1.0f / n2;
                gy.x = gm * v.x * inv - gp * v.y * inv2;
                gy.y = gm * v.y * inv + gp * v.x * inv2;
            }
            S.y[wave][m] = gy;
        }
        wave_lds_sync();
        // dX[k] = sum_j fbt_w[k, j] dY[fbt_start[k] + j]
#pragma unroll
        for (int p = 0; p < kBinPasses; ++p) {
            const int k = lane + 64 * p;
            if (k < kBins) {
                float re = 0.0f, im = 0.0f;
#pragma unroll
                for (int j = 0; j < SPT; ++j) {
                    const int m = m0_t[p] + j < M ? m0_t[p] + j : M - 1;      // past the run: weight 0
                    const float2 gy = S.y[wave][m];
                    re = fmaf(tw_t[p][j], gy.x, re);
                    im = fmaf(tw_t[p][j], gy.y, im);
                }
                L.xs_of(wave)[k] = make_float2(re, im);
            }
        }
        wave_lds_sync();
        spectrum_grad_to_frame<false>(L, wave, lane, w, S.dframe[fl], live);
    }
    __syncthreads();
    overlap_add_block(S.dframe, dx + b * T, f_base, NF, hop, T);
}

// The same gradient WITHOUT recomputing the spectrum: Y = fb X is linear in X, so d x needs only d Y, and d Y needs only Y —
// which is the forward output itself, Y = |Y| e^{i phase}:   d Y = g_mag (cos, sin) + (g_phase / |Y|) (-sin, cos)   (0 at |Y| = 0).
// No framing, no forward FFT, no un-packing, no band projection: about 45 % of stft_mel_backward_kernel's instructions.
struct LdsMelBwdOut {
    Lds c;
    float dframe[kFramesPerBlockBwd][kNfft];
    float2 y[kWavesPerBlock][kMelMax];
    float stage[4][kMelMax][kFramesPerBlockBwd + 1];       // planes 0, 1: d out; planes 2, 3: out (magnitude, phase)
};

template <int SPT>
__global__ __launch_bounds__(kThreads) void stft_mel_backward_out_kernel(const float *__restrict__ w, const float *__restrict__ dout,
                                                                         const float *__restrict__ out,
                                                                         const int32_t *__restrict__ fbt_start,
                                                                         const float *__restrict__ fbt_w, int span_t,
                                                                         float *__restrict__ dx, int T, int NF, int hop, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LdsMelBwdOut &S = *reinterpret_cast<LdsMelBwdOut *>(raw);
    Lds &L = S.c;
    fill_twiddles(L);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = blockIdx.y;
    const int f_base = blockIdx.x * kFramesPerBlockBwd;
    const int frames_here = (NF - f_base) < kFramesPerBlockBwd ? (NF - f_base) : kFramesPerBlockBwd;
    for (int i = threadIdx.x; i < 4 * M * kFramesPerBlockBwd; i += kThreads) {
        const int fl = i % kFramesPerBlockBwd, row = i / kFramesPerBlockBwd, plane = row / M, m = row - plane * M;
        const float *src = plane < 2 ? dout : out;
        S.stage[plane][m][fl] = fl < frames_here ? src[((b * 2 + (plane & 1)) * M + m) * NF + f_base + fl] : 0.0f;
    }
    constexpr int kBinPasses = (kBins + 63) / 64;
    float tw_t[kBinPasses][SPT];
    int m0_t[kBinPasses];
#pragma unroll
    for (int p = 0; p < kBinPasses; ++p) {
        const int k = lane + 64 * p;
        const bool in = k < kBins;
        m0_t[p] = in ? fbt_start[k] : 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) tw_t[p][j] = (in && j < span_t && m0_t[p] + j < M) ? fbt_w[k * span_t + j] : 0.0f;
    }
    __syncthreads();
    for (int r = 0; r < kFramesPerBlockBwd / kWavesPerBlock; ++r) {
        const int fl = r * kWavesPerBlock + wave;
        const bool live = f_base + fl < NF;
        for (int m = lane; m < M; m += 64) {
            const float gm = S.stage[0][m][fl], gp = S.stage[1][m][fl], mag = S.stage[2][m][fl], ph = S.stage[3][m][fl];
            float2 gy = make_float2(0.0f, 0.0f);
            if (mag > 0.0f) {
                float sn, cs;
                sincosf(ph, &sn, &cs);
                const float q = gp / mag;
                gy.x = gm * cs - q * sn;
                gy.y = gm * sn + q * cs;
            }
            S.y[wave][m] = gy;
        }
        wave_lds_sync();
#pragma unroll
        for (int p = 0; p < kBinPasses; ++p) {
            const int k = lane + 64 * p;
            if (k < kBins) {
                float re = 0.0f, im = 0.0f;
#pragma unroll
                for (int j = 0; j < SPT; ++j) {
                    const int m = m0_t[p] + j < M ? m0_t[p] + j : M - 1;
                    const float2 gy = S.y[wave][m];
                    re = fmaf(tw_t[p][j], gy.x, re);
                    im = fmaf(tw_t[p][j], gy.y, im);
                }
                L.xs_of(wave)[k] = make_float2(re, im);
            }
        }
        wave_lds_sync();
        spectrum_grad_to_frame<false>(L, wave, lane, w, S.dframe[fl], live);
    }
    __syncthreads();
    overlap_add_block(S.dframe, dx + b * T, f_base, NF, hop, T);
}

constexpr int64_t kMaxGridY = 65535;

// The table is a pure function of nothing: writing it again is harmless, so a racy "done" flag per device is enough.
inline void ensure_twiddles(hipStream_t st) {
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !done[dev]) {
        hipLaunchKernelGGL(stft_twiddle_kernel, dim3(2), dim3(256), 0, st);
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
}

}  // namespace

#define STFT_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

size_t advstep_stft_bands_block_count(int64_t B, int64_t NF) {
    if (B <= 0 || NF <= 0) return 0;
    return (size_t)(B * ceil_div(NF, kFramesPerBlockFwd));
}

int advstep_stft_bands_supported(int64_t nfft, int64_t hop, int64_t T) {
    // hop >= 128: the frames of one backward workgroup must advance past the overlap with its neighbour (see below)
    return nfft == kNfft && hop >= 128 && hop <= kNfft && T > kN + 1;
}

int advstep_stft_bands_f32(const float *x, const float *window, const int32_t *fb_start, const float *fb_w, int64_t span,
                           float *band_db, float *block_max, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft,
                           int64_t M, advstep_stream_t stream) {
    STFT_REQUIRE(B >= 0 && NF >= 0 && M >= 0 && span >= 1);
    if (B == 0 || NF == 0 || M == 0) return ADVSTEP_OK;
    STFT_REQUIRE(x && window && fb_start && fb_w && band_db && block_max && B <= kMaxGridY);
    STFT_REQUIRE(advstep_stft_bands_supported(nfft, hop, T) && NF == 1 + T / hop);
    ensure_twiddles(as_stream(stream));
    const dim3 grid((unsigned)ceil_div(NF, kFramesPerBlockFwd), (unsigned)B);
    hipLaunchKernelGGL(stft_bands_kernel, grid, dim3(kThreads), 0, as_stream(stream), x, window, fb_start, fb_w, (int)span,
                       band_db, block_max, (int)T, (int)NF, (int)hop, (int)M);
    return status_after_launch();
}

int advstep_stft_bands_backward_f32(const float *x, const float *window, const float *dband, const int32_t *fbt_start,
                                    const float *fbt_w, int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF,
                                    int64_t hop, int64_t nfft, int64_t M, advstep_stream_t stream) {
    STFT_REQUIRE(B >= 0 && NF >= 0 && M >= 0 && span_t >= 1);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    STFT_REQUIRE(x && window && dband && fbt_start && fbt_w && dx && B <= kMaxGridY);
    STFT_REQUIRE(advstep_stft_bands_supported(nfft, hop, T) && NF == 1 + T / hop && span_t <= kMaxSpanT && M <= kMaxBands);
    // a sample may be reached by at most two workgroups (two-operand float atomics commute): a workgroup's frames must
    // advance by at least the overlap between neighbouring workgroups' ranges
    STFT_REQUIRE(kFramesPerBlockBwd * hop >= kNfft - hop);
    hipStream_t st = as_stream(stream);
    ensure_twiddles(st);
    if (hipMemsetAsync(dx, 0, (size_t)B * T * sizeof(float), st) != hipSuccess) return ADVSTEP_ELAUNCH;
    const dim3 grid((unsigned)ceil_div(NF, kFramesPerBlockBwd), (unsigned)B);
    auto go = [&](auto kernel, size_t lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, st, x, window, dband, fbt_start, fbt_w, (int)span_t, dx, (int)T,
                           (int)NF, (int)hop, (int)M);
    };
    if (span_t <= 2) go(stft_bands_backward_kernel<2>, sizeof(LdsBwd<2>));
    else go(stft_bands_backward_kernel<kMaxSpanT>, sizeof(LdsBwd<kMaxSpanT>));
    return status_after_launch();
}

int advstep_stft_mel_f32(const float *x, const float *window, const int32_t *fb_start, const float *fb_w, int64_t span,
                         float *out, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft, int64_t M,
                         advstep_stream_t stream) {
    STFT_REQUIRE(B >= 0 && NF >= 0 && M >= 0 && span >= 1);
    if (B == 0 || NF == 0 || M == 0) return ADVSTEP_OK;
    STFT_REQUIRE(x && window && fb_start && fb_w && out && B <= kMaxGridY && M <= kMelMax && span <= kMelMaxSpan);
    STFT_REQUIRE(advstep_stft_bands_supported(nfft, hop, T) && NF == 1 + T / hop);
    hipStream_t st = as_stream(stream);
    ensure_twiddles(st);
    const size_t lds = sizeof(LdsMel);
    auto go = [&](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(NF, kFramesPerBlockFwd), (unsigned)B), dim3(kThreads), lds, st, x,
                           window, fb_start, fb_w, (int)span, out, (int)T, (int)NF, (int)hop, (int)M);
    };
    if (span <= 16) go(stft_mel_kernel<16>);
    else go(stft_mel_kernel<kMelMaxSpan>);
    return status_after_launch();
}

int advstep_stft_mel_backward_f32(const float *x, const float *window, const float *dout, const int32_t *fb_start,
                                  const float *fb_w, int64_t span, const int32_t *fbt_start, const float *fbt_w,
                                  int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft,
                                  int64_t M, advstep_stream_t stream) {
    STFT_REQUIRE(B >= 0 && NF >= 0 && M >= 0 && span >= 1 && span_t >= 1);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    STFT_REQUIRE(x && window && dout && fb_start && fb_w && fbt_start && fbt_w && dx && B <= kMaxGridY && M <= kMelMax &&
                 span <= kMelMaxSpan && span_t <= kMaxSpanT);
    STFT_REQUIRE(advstep_stft_bands_supported(nfft, hop, T) && NF == 1 + T / hop);
    STFT_REQUIRE(kFramesPerBlockBwd * hop >= kNfft - hop);
    hipStream_t st = as_stream(stream);
    ensure_twiddles(st);
    if (hipMemsetAsync(dx, 0, (size_t)B * T * sizeof(float), st) != hipSuccess) return ADVSTEP_ELAUNCH;
    const size_t lds = sizeof(LdsMelBwd);
    auto go = [&](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(NF, kFramesPerBlockBwd), (unsigned)B), dim3(kThreads), lds, st, x,
                           window, dout, fb_start, fb_w, (int)span, fbt_start, fbt_w, (int)span_t, dx, (int)T, (int)NF, (int)hop,
                           (int)M);
    };
    if (span <= 16 && span_t <= 2) go(stft_mel_backward_kernel<16, 2>);
    else if (span_t <= 2) go(stft_mel_backward_kernel<kMelMaxSpan, 2>);
    else go(stft_mel_backward_kernel<kMelMaxSpan, kMaxSpanT>);
    return status_after_launch();
}

int advstep_stft_mel_backward_from_output_f32(const float *window, const float *dout, const float *out, const int32_t *fbt_start,
                                              const float *fbt_w, int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF,
                                              int64_t hop, int64_t nfft, int64_t M, advstep_stream_t stream) {
    STFT_REQUIRE(B >= 0 && NF >= 0 && M >= 0 && span_t >= 1);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    STFT_REQUIRE(window && dout && out && fbt_start && fbt_w && dx && B <= kMaxGridY && M <= kMelMax && span_t <= kMaxSpanT);
    STFT_REQUIRE(advstep_stft_bands_supported(nfft, hop, T) && NF == 1 + T / hop);
    STFT_REQUIRE(kFramesPerBlockBwd * hop >= kNfft - hop);
    hipStream_t st = as_stream(stream);
    ensure_twiddles(st);
    if (hipMemsetAsync(dx, 0, (size_t)B * T * sizeof(float), st) != hipSuccess) return ADVSTEP_ELAUNCH;
    const size_t lds = sizeof(LdsMelBwdOut);
    auto go = [&](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(NF, kFramesPerBlockBwd), (unsigned)B), dim3(kThreads), lds, st, window,
                           dout, out, fbt_start, fbt_w, (int)span_t, dx, (int)T, (int)NF, (int)hop, (int)M);
    };
    if (span_t <= 2) go(stft_mel_backward_out_kernel<2>);
    else go(stft_mel_backward_out_kernel<kMaxSpanT>);
    return status_after_launch();
}

}  // extern "C"
