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
#include <stdlib.h>

#include "advstep_frontend.h"

namespace {

constexpr int kN = 256;            // complex FFT length (n_fft = 512 real samples)
constexpr int kNfft = 2 * kN;
constexpr int kBins = kN + 1;      // one-sided spectrum
constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * 64;
constexpr int kFramesPerBlockBwd = 4;   // one per wave: 35 KB of LDS per workgroup = 4 workgroups (16 waves) per CU
constexpr float kAmin = 1e-10f;
// d/d band of 10 log10(clamp(band, amin)), band recovered from its dB value (the same function as in lfcc.hip)
__device__ __forceinline__ float dlog_of_db(float db) {
    return (db > -100.0f) ? 4.342944819032518f / expf(db * 0.23025850929940457f) : 0.0f;
}

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

// ---- round 4: the 256-point transform as 16 x 16 with the 16-point transforms in REGISTERS --------------------------------
// The radix-4 Stockham form below (still used by the mel-spec kernels) puts one frame on a wave, 4 points on a lane, and
// every one of its 4 stages through LDS; its kernels are VALU-issue-bound (profiles/r03_model_kernel_pmc.txt: 72 % of the
// duration is vector-ALU issue; 406 / 928 wave-instructions per frame forward / backward), and a third of those
// instructions is LDS addressing and swizzling.  Here a frame lives on 16 lanes (a wave carries 4 frames), a lane holds
// 16 complex points, and with n = 16 n1 + n2, k = k1 + 16 k2
//     X[k1 + 16 k2] = sum_n2 W16^(n2 k2) . W256^(n2 k1) . [ sum_n1 x[16 n1 + n2] W16^(n1 k1) ]
// lane n2 transforms its 16 points over n1 in registers, multiplies by its 15 lane-constant twiddles, the 16 x 16 block is
// transposed through LDS ONCE (16 ds_write_b64 + 16 ds_read_b64 per lane, pitch 17: conflict-free), and lane k1 transforms
// over n2: the spectrum ends up "lane k1, register k2".  The inverse runs the same two passes on that layout (first over
// k2, twiddle, transpose, then over k1) and ends with sample n = lane + 16 r in register r — the layout the frame was
// loaded in, so the window constants are shared.  ~95 wave-instructions per frame and transform instead of ~260.
#include "stft_tables.inc"

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {        // fused: 2 mul + 2 fma
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmulc(float2 a, float cr, float ci) { return cmulf(a, make_float2(cr, ci)); }

// 4-point transform of (a, b, c, d): forward y1 = (a - c) - i (b - d), inverse y1 = (a - c) + i (b - d)
template <bool INV>
__device__ __forceinline__ void fft4(float2 &a, float2 &b, float2 &c, float2 &d) {
    const float2 s02 = cadd(a, c), d02 = csub(a, c), s13 = cadd(b, d), d13 = csub(b, d);
    const float2 r = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
    a = cadd(s02, s13);
    b = cadd(d02, r);
    c = csub(s02, s13);
    d = csub(d02, r);
}

// 16-point transform in place, natural order in and out: x[4 n1 + n2] -> 4-point over n1 -> W16^(n2 k1) -> 4-point over n2.
template <bool INV>
__device__ __forceinline__ void fft16(float2 (&x)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) fft4<INV>(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);      // x[4 k1 + n2] = b[n2][k1]
    // W16^m = (cos, -sin)(2 pi m / 16); INV: conjugate
    const float sg = INV ? 1.0f : -1.0f;
    x[4 + 1] = cmulc(x[4 + 1], C1, sg * S1);       // n2 = 1, k1 = 1: W^1
    x[8 + 1] = cmulc(x[8 + 1], H, sg * H);         // n2 = 1, k1 = 2: W^2
    x[12 + 1] = cmulc(x[12 + 1], S1, sg * C1);     // n2 = 1, k1 = 3: W^3
    x[4 + 2] = cmulc(x[4 + 2], H, sg * H);         // n2 = 2, k1 = 1: W^2
    x[8 + 2] = INV ? make_float2(-x[8 + 2].y, x[8 + 2].x) : make_float2(x[8 + 2].y, -x[8 + 2].x);   // W^4 = -i (forward)
    x[12 + 2] = cmulc(x[12 + 2], -H, sg * H);      // n2 = 2, k1 = 3: W^6
    x[4 + 3] = cmulc(x[4 + 3], S1, sg * C1);       // n2 = 3, k1 = 1: W^3
    x[8 + 3] = cmulc(x[8 + 3], -H, sg * H);        // n2 = 3, k1 = 2: W^6
    x[12 + 3] = cmulc(x[12 + 3], -C1, -sg * S1);   // n2 = 3, k1 = 3: W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) fft4<INV>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);   // -> y[k1 + 4 k2] at x[4 k1 + k2]
    // un-transpose the 4 x 4 block: y[k1 + 4 k2] sits at x[4 k1 + k2]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const float2 t = x[4 * i + j];
            x[4 * i + j] = x[4 * j + i];
            x[4 * j + i] = t;
        }
}

// An opaque copy of a lane index: loads from the constant tables indexed with it cannot be issued before this point (they are
// invariant loads, which hipcc otherwise hoists to the top of the loop body and holds in registers across everything)
__device__ __forceinline__ int here(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// "These 16 values are complete HERE": an empty asm every one of them passes through, with a memory clobber.  The loads of the
// next phase cannot be issued above it and the arithmetic producing the values cannot sink below it — without it hipcc overlaps
// a phase's loads with the previous phase's arithmetic and the live set of the two (up to 180 registers) costs a wave of occupancy.
__device__ __forceinline__ void phase_done(float2 (&x)[16]) {
    asm volatile("" : "+v"(x[0].x), "+v"(x[0].y), "+v"(x[1].x), "+v"(x[1].y), "+v"(x[2].x), "+v"(x[2].y), "+v"(x[3].x), "+v"(x[3].y),
                      "+v"(x[4].x), "+v"(x[4].y), "+v"(x[5].x), "+v"(x[5].y), "+v"(x[6].x), "+v"(x[6].y), "+v"(x[7].x), "+v"(x[7].y)
                 :: "memory");
    asm volatile("" : "+v"(x[8].x), "+v"(x[8].y), "+v"(x[9].x), "+v"(x[9].y), "+v"(x[10].x), "+v"(x[10].y), "+v"(x[11].x), "+v"(x[11].y),
                      "+v"(x[12].x), "+v"(x[12].y), "+v"(x[13].x), "+v"(x[13].y), "+v"(x[14].x), "+v"(x[14].y), "+v"(x[15].x), "+v"(x[15].y)
                 :: "memory");
}

template <int BASE>
__device__ __forceinline__ void half_done(float2 (&x)[16]) {       // the same for x[BASE .. BASE + 7]
    asm volatile("" : "+v"(x[BASE].x), "+v"(x[BASE].y), "+v"(x[BASE + 1].x), "+v"(x[BASE + 1].y), "+v"(x[BASE + 2].x), "+v"(x[BASE + 2].y),
                      "+v"(x[BASE + 3].x), "+v"(x[BASE + 3].y), "+v"(x[BASE + 4].x), "+v"(x[BASE + 4].y), "+v"(x[BASE + 5].x),
                      "+v"(x[BASE + 5].y), "+v"(x[BASE + 6].x), "+v"(x[BASE + 6].y), "+v"(x[BASE + 7].x), "+v"(x[BASE + 7].y)
                 :: "memory");
}

constexpr int kPitch = 17;                       // float2 slots per row of a frame's 16 x 16 exchange block
constexpr int kFrameSlots = 16 * kPitch;         // 272 float2 per frame (also holds a frame's 512 floats / 257 bins)
constexpr int kGroup = 4;                        // frames per wave at a time (16 lanes each)

// LDS-only ordering inside a wave (its LDS instructions execute in order; only the compiler and the counter must agree)
__device__ __forceinline__ void lds_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 256-point transform of the wave's 4 frames.  Forward: in x[r] = z[16 r + l] (l = lane & 15), out x[r] = Z[l + 16 r].
// Inverse: in x[r] = Z[l + 16 r], out x[r] = z[l + 16 r] (unnormalised).  `blk`: the frame's kFrameSlots float2 of LDS.
template <bool INV>
__device__ __forceinline__ void fft256_reg(float2 (&x)[16], const float2 *twl, float2 *blk, int l) {
    fft16<INV>(x);
    phase_done(x);
    // W256^(l j), j = 1 .. 15: row l of the workgroup's 16 x 16 table in LDS (30 registers per lane if kept: they cost a wave
    // of occupancy; the reads ride under the first 16-point transform)
#pragma unroll
    for (int j = 1; j < 16; ++j) {
        const float2 t = twl[l * kPitch + j];
        x[j] = cmulf(x[j], INV ? conjf2(t) : t);
    }
    lds_wave_fence();                            // whatever the caller last read from `blk` has been delivered
#pragma unroll
    for (int j = 0; j < 16; ++j) blk[j * kPitch + l] = x[j];
    lds_wave_fence();
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = blk[l * kPitch + j];
    fft16<INV>(x);
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

// The radix-4 path's per-stage tables, from the compile-time constants (round 4: no device global, no first-launch fill):
// stage s = 1, 2, 3 (Ns = 4^s): exp(-2 pi i t k / (4 Ns)) = W256^(t k 64 / Ns)
__device__ __forceinline__ float2 stage_twiddle(int m) {
    const int stage = m < 12 ? 1 : (m < 60 ? 2 : 3), base = stage == 1 ? 0 : (stage == 2 ? 12 : 60), Ns = 1 << (2 * stage);
    const int t = (m - base) / Ns + 1, k = (m - base) % Ns;
    return kTw256[(t * k * (64 / Ns)) & 255];
}

__device__ __forceinline__ void fill_twiddles(Lds &L) {
    for (int m = threadIdx.x; m < kN; m += kThreads) {
        if (m < kStageTw) L.tw[m] = stage_twiddle(m);
        L.tw512[m] = kTw512[m];
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
template <int STRIDE>
__device__ __attribute__((noinline)) void overlap_add_frames_cold(const float *dframe, float *__restrict__ dxb, int f_base, int NF, int hop,
                                                                  int T, int tid, int nthreads);

template <int STRIDE>
__device__ __forceinline__ void overlap_add_frames(const float *dframe, float *__restrict__ dxb, int f_base, int NF, int hop, int T,
                                                   int tid, int nthreads) {
    const int p_lo = f_base * hop;                                          // first padded position of the first frame
    const int frames_here = (NF - f_base) < kFramesPerBlockBwd ? (NF - f_base) : kFramesPerBlockBwd;
    const int p_hi = p_lo + (frames_here - 1) * hop + kNfft;                // one past the last
    auto add_from = [&](int p, float &acc) {     // frames in ascending order: a fixed summation order
        if (p < p_lo || p >= p_hi) return;
#pragma unroll
        for (int fl = 0; fl < kFramesPerBlockBwd; ++fl) {
            const int n = p - p_lo - fl * hop;
            if (fl < frames_here && n >= 0 && n < kNfft) acc += dframe[fl * STRIDE + n];
        }
    };
    const bool near_left = p_lo < 2 * kN, near_right = p_hi > T;            // wave-uniform, almost always false
    const int t_first = p_lo - kN, span_all = p_hi - p_lo;
    for (int i = tid; i < span_all; i += nthreads) {
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
    for (int i = tid; i < span_all; i += nthreads) {
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

__device__ __forceinline__ void overlap_add_block(const float (*dframe)[kNfft], float *__restrict__ dxb, int f_base, int NF,
                                                  int hop, int T) {
    overlap_add_frames<kNfft>(&dframe[0][0], dxb, f_base, NF, hop, T, threadIdx.x, kThreads);
}

// the same as a real call: the first / last groups of an utterance only, kept out of the hot kernel's register allocation
template <int STRIDE>
__device__ __attribute__((noinline)) void overlap_add_frames_cold(const float *dframe, float *__restrict__ dxb, int f_base, int NF, int hop,
                                                                  int T, int tid, int nthreads) {
    overlap_add_frames<STRIDE>(dframe, dxb, f_base, NF, hop, T, tid, nthreads);
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
        const float2 t0 = stage_twiddle(m < kStageTw ? m : 0), t1 = kTw512[m];
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

// ---- round 4: the LFCC pair on the register-resident transform -----------------------------------------------------------
#ifndef STFT_FWD_GROUPS
#define STFT_FWD_GROUPS 2
#endif
constexpr int kGroupsPerWave = STFT_FWD_GROUPS;                                                   // 4-frame groups a wave works through
constexpr int kBandsFramesPerBlock = kWavesPerBlock * kGroupsPerWave * kGroup;     // 32 consecutive frames per workgroup

struct LdsReg {
    float2 blk[kWavesPerBlock][kGroup][kFrameSlots];   // per wave and frame: exchange block / 256-vector / 512 floats
    float2 twl[16 * kPitch];                           // [l][j] = W256^(l j), rows at pitch 17 (conflict-free)
    float red[kWavesPerBlock];
};
// (the window and exp(-2 pi i k / 512) are read from global memory where they are used: both are 2 KB, L1-resident, and
// keeping them out of LDS is what lets four workgroups share a CU)

__device__ __forceinline__ void fill_reg_tables(LdsReg &S) {
    for (int m = threadIdx.x; m < kN; m += kThreads) S.twl[(m >> 4) * kPitch + (m & 15)] = kTw256[((m >> 4) * (m & 15)) & 255];
}

// x[r] = (w[2n] x[q + 2n], w[2n + 1] x[q + 2n + 1]),  n = 16 r + l,  q = f hop - 256 (reflect padding at the ends)
__device__ __forceinline__ void load_frame_reg(const float *__restrict__ xb, const float2 *__restrict__ win2, int T, int f, int hop,
                                               float2 (&x)[16], int l) {
    const int q_first = f * hop - kN;
    if (q_first >= 0 && q_first + kNfft <= T && ((q_first & 1) == 0) && ((reinterpret_cast<uintptr_t>(xb) & 7u) == 0)) {
        const float2 *x2 = reinterpret_cast<const float2 *>(xb + q_first);
        float2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = x2[16 * r + l];          // 16 requests in flight
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 wv = win2[16 * r + l];
            x[r] = make_float2(wv.x * v[r].x, wv.y * v[r].y);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 16 * r + l;
        int q0 = q_first + 2 * n, q1 = q0 + 1;
        q0 = q0 < 0 ? -q0 : q0;
        q0 = q0 >= T ? 2 * (T - 1) - q0 : q0;
        q1 = q1 < 0 ? -q1 : q1;
        q1 = q1 >= T ? 2 * (T - 1) - q1 : q1;
        const float2 wv = win2[n];
        x[r] = make_float2(wv.x * xb[q0], wv.y * xb[q1]);
    }
}

// A frame's 256-vector in its block: element k = l + 16 r (lane l, register r) at slot 17 r + l; the elements of lane 0 are
// stored a second time in the spare 17th column of the row before (element 16 r at slot 17 (r - 1) + 16, element 0 at
// 17 . 15 + 16), so that EVERY lane finds its mirror element (N - k) mod N at slot 17 (15 - r) + (16 - l): one lane-constant
// base and compile-time offsets, no index arithmetic.
__device__ __forceinline__ void park_vector(float2 *blk, const float2 (&x)[16], int l) {
#pragma unroll
    for (int r = 0; r < 16; ++r) blk[r * kPitch + l] = x[r];
    if (l == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) blk[((r + 15) & 15) * kPitch + 16] = x[r];
    }
}
__device__ __forceinline__ float2 mirror_of(const float2 *blk, int r, int l) { return blk[(15 - r) * kPitch + (16 - l)]; }

// Z (the transform of the packed frame, lane l register r = Z[l + 16 r]) -> 2 X[l + 16 r] in place and 2 X[256] (real) in `nyq`
// (meaningful on lane l == 0):  2 X[k] = (Z[k] + Z*[N - k]) + W512^k . (-i) (Z[k] - Z*[N - k]).  The factor 2 of the real-input
// un-packing is left in: the forward folds 1/4 into its filterbank weights, the backward 1/2 into its bin coefficients.
__device__ __forceinline__ void unpack_real_reg(float2 (&x)[16], float &nyq, float2 *blk, int l) {
    lds_wave_fence();
    park_vector(blk, x, l);
    lds_wave_fence();
    nyq = 2.0f * (x[0].x - x[0].y);
    const int lt = here(l);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r == 8) half_done<0>(x);               // two batches of 8 mirror + 8 table reads, not one of 16 + 16
        const float2 zc = conjf2(mirror_of(blk, r, l));
        const float2 e = cadd(x[r], zc), d = csub(x[r], zc);
        x[r] = cadd(e, cmulf(kTw512[lt + 16 * r], make_float2(d.y, -d.x)));
    }
}

constexpr int kRegSpan = 4;                    // taps per band the register-FFT forward kernel takes (the linear bank: 3)

struct LdsRegFwd {
    LdsReg c;
    float fbw[kMaxBands * kRegSpan];           // [m][j], 0 beyond a band's run, beyond `span` and for m >= M
    int32_t fbs[kMaxBands];
};

// grid (ceil(NF / 32), B); span <= kRegSpan, M <= kMaxBands
__global__ __launch_bounds__(kThreads) void stft_bands_reg_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                  const int32_t *__restrict__ fb_start,
                                                                  const float *__restrict__ fb_w, int span,
                                                                  float *__restrict__ band_db, float *__restrict__ bmax, int T,
                                                                  int NF, int hop, int M, int bmax_tail) {
    __shared__ LdsRegFwd SF;
    LdsReg &S = SF.c;
    fill_reg_tables(S);
    for (int i = threadIdx.x; i < kMaxBands * kRegSpan; i += kThreads) {
        const int m = i / kRegSpan, j = i % kRegSpan;
        SF.fbw[i] = (m < M && j < span) ? 0.25f * fb_w[m * span + j] : 0.0f;      // |2 X|^2 / 4
    }
    for (int m = threadIdx.x; m < kMaxBands; m += kThreads) SF.fbs[m] = m < M ? fb_start[m] : 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fs = lane >> 4, l = lane & 15;
    __syncthreads();
    const int64_t b = blockIdx.y;
    float2 *blk = S.blk[wave][fs];
    float vmax = -INFINITY;
    for (int g = 0; g < kGroupsPerWave; ++g) {
        const int f0 = blockIdx.x * kBandsFramesPerBlock + (wave * kGroupsPerWave + g) * kGroup;
        if (f0 >= NF) break;                                   // wave-uniform
        const int f = f0 + fs;
        const bool live = f < NF;
        // (the window and exp(-2 pi i k / 512) values a lane reads are the same for every group: they are indexed through
        // here(), or hipcc hoists them out of this loop into ~64 registers per lane — a wave of occupancy)
        float2 xr[16];
        load_frame_reg(x + b * T, reinterpret_cast<const float2 *>(w), T, live ? f : NF - 1, hop, xr, here(l));
        fft256_reg<false>(xr, S.twl, blk, l);
        float nyq;
        unpack_real_reg(xr, nyq, blk, l);
        phase_done(xr);
        // power spectrum -> the frame's 257 floats (every lane's reads of the 256-vector were issued before these writes);
        // entries 257 .. 259 are zeroed: a band's padded taps may reach them (with weight 0)
        float *pw = reinterpret_cast<float *>(blk);
        lds_wave_fence();
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[l + 16 * r] = fmaf(xr[r].x, xr[r].x, xr[r].y * xr[r].y);
        if (l == 0) pw[kN] = nyq * nyq;
        if (l >= 1 && l <= kRegSpan) pw[kN + l] = 0.0f;
        lds_wave_fence();
        // bands l, l + 16, ..., l + 112 of this lane's frame: starts, then weights and bins, each batch of reads in flight at once
        float *row = band_db + (b * NF + (live ? f : NF - 1)) * M;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            int k0[4];
            float wt[4][kRegSpan], pv[4][kRegSpan];
#pragma unroll
            for (int i = 0; i < 4; ++i) k0[i] = SF.fbs[l + 16 * (4 * half + i)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < kRegSpan; ++j) {
                    wt[i][j] = SF.fbw[(l + 16 * (4 * half + i)) * kRegSpan + j];
                    pv[i][j] = pw[k0[i] + j];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = l + 16 * (4 * half + i);
                float band = 0.0f;
#pragma unroll
                for (int j = 0; j < kRegSpan; ++j) band = fmaf(wt[i][j], pv[i][j], band);
                // 10 log10(v) = 3.0103 log2(v): the hardware log2 (1 ulp) keeps the dB value within 1e-6 relative
                const float db = 3.010299956639812f * __log2f(band > kAmin ? band : (band != band ? band : kAmin));
                if (live && m < M) {
                    row[m] = db;
                    vmax = max_nan(vmax, db);
                }
            }
        }
        lds_wave_fence();                                      // pw is rewritten by the next group's exchange
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = max_nan(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) S.red[wave] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = S.red[0];
        for (int i = 1; i < kWavesPerBlock; ++i) r = max_nan(r, S.red[i]);
        bmax[b * gridDim.x + blockIdx.x] = r;
        // the rest of block_max (it is sized for the radix-4 kernel's 16-frame workgroups, and its consumer reduces over all of
        // it) must not hold garbage: shared out over the workgroups instead of a fill launch in front of this kernel
        const int total = gridDim.x * gridDim.y;
        for (int j = (int)b * gridDim.x + blockIdx.x; j < bmax_tail; j += total) bmax[total + j] = -INFINITY;
    }
}

// The overlap-add of a wave's 4 consecutive frames when no reflection is involved (all but the first and last groups of an
// utterance): sample i of the 3 hop + 512 = 992 the frames cover is the sum of dframe[fl][i - fl hop] over the frames that reach
// it, in frame order.  HOP = 160 is a compile-time constant so that, fully unrolled, most (64-sample slice, frame) pairs are
// decided statically: all 64 lanes read, or none do; only the slices a frame's ends cut need a per-lane test.
// Who else touches a sample: [0, 352) the group before, [640, 992) the group after (4 hops = 640 = ten slices), [352, 640)
// nobody.  A wave works through consecutive groups, so the last 352 sums of a group stay in registers (`carry`, same lane, slice
// j - 10) and are added to the next group's first 352, which are then complete: plain stores.  Only a wave's first and last
// group meet other waves' sums, with one float atomic per sample onto the zeroed dx (two operands: commutative, bit-reproducible).
// L2 float atomics were a quarter of this kernel (1 M lane-atomics per XCD and launch at one group per wave): 248 -> 44 per frame.
template <int STRIDE>
__device__ __forceinline__ void overlap_add_interior(const float *df, float *__restrict__ out, int lane, float (&carry)[6],
                                                     bool carry_in, bool carry_out) {
    constexpr int HOP = 160, kSpan = 3 * HOP + kNfft, kShared = kNfft - HOP;       // 992, 352
    static_assert(4 * HOP == 640 && kSpan <= 16 * 64, "slices");
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j % 4 == 0) lds_wave_fence();          // four slices' reads in flight at a time (16 would cost ~40 registers)
        float acc = 0.0f;
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            const int lo = HOP * fl - 64 * j, hi = lo + kNfft;         // lanes [lo, hi) of this slice lie inside frame fl
            if (hi <= 0 || lo >= 64) continue;
            if (lo <= 0 && hi >= 64) {
                acc += df[fl * STRIDE + lane - lo];
            } else {
                const int n = lane - lo;
                if ((unsigned)n < (unsigned)kNfft) acc += df[fl * STRIDE + n];
            }
        }
        const int i = lane + 64 * j;
        if (j < 6) {                               // slices 0 .. 5: [0, 352) shared with the group before, [352, 384) nobody's
            const bool shared = i < kShared;
            if (carry_in) {
                out[i] = shared ? acc + carry[j] : acc;
            } else if (j < 5) {
                atomicAdd(out + i, acc);
            } else {
                if (shared) atomicAdd(out + i, acc);
                else out[i] = acc;
            }
        } else if (j < 10) {                       // [384, 640): nobody else's
            out[i] = acc;
        } else {                                   // [640, 992): shared with the group after
            if (carry_out) carry[j - 10] = acc;
            else if (i < kSpan) atomicAdd(out + i, acc);
        }
    }
}

template <int SPAN_CAP>
struct LdsRegBwd {
    LdsReg c;
    float fbt_w[(kBins + 1) * SPAN_CAP];       // [k][j], 0 beyond a bin's run and beyond span_t
    int32_t fbt_start[kBins + 1];
};   // 40 104 B with SPAN_CAP = 2: four workgroups per CU (a fifth array — the band gradients had one — makes it three)

#ifndef STFT_BWD_GROUPS
#define STFT_BWD_GROUPS 4
#endif
constexpr int kBwdGroupsPerWave = STFT_BWD_GROUPS;                                                    // 16 consecutive frames per wave
constexpr int kBwdFramesPerBlock = kWavesPerBlock * kBwdGroupsPerWave * kGroup;        // 64 per workgroup

// grid (ceil(NF / 64), B).  A wave owns 4 CONSECUTIVE frames at a time and overlap-adds them itself: its 4 windowed frame
// gradients meet in its own LDS block, every sample they touch is summed in frame order, and the <= 2 waves that share a sample
// (4 hops = 640 samples of advance against 352 of overlap) combine with one float atomic each onto the zeroed dx — two
// operands, commutative, so bit-reproducible; no workgroup barrier after the tables are staged.
// Waves per SIMD the register allocator is asked to fit (and the dispatcher allowed to place): 4 = what the 40 KB of LDS per
// workgroup admit (four workgroups per CU).  Round 4 shipped this kernel pinned at 2 - a leftover of a build that needed more than
// 128 registers - although it allocates 118: 101.7 us pinned at 2, 98.2 at 3, 78.1 at 4 (151.7 at 1), round 5.
#ifndef STFT_BWD_WAVES
#define STFT_BWD_WAVES 4
#endif
template <int SPAN_CAP>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(STFT_BWD_WAVES, STFT_BWD_WAVES)))
void stft_bands_backward_reg_kernel(const float *__restrict__ x,
                                                                           const float *__restrict__ w,
                                                                           const float *__restrict__ dband,
                                                                           const int32_t *__restrict__ fbt_start,
                                                                           const float *__restrict__ fbt_w, int span_t,
                                                                           float *__restrict__ dx, int T, int NF, int hop,
                                                                           int M, const float *__restrict__ band_db,
                                                                           const float *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LdsRegBwd<SPAN_CAP> &S = *reinterpret_cast<LdsRegBwd<SPAN_CAP> *>(raw);
    // advstep_lfcc_floor_fixup_f32 folded in (stats != null): the gradient the dB floor swallowed (stats[2], summed by the
    // projection's backward) goes to the elements equal to the batch maximum, stats[2] / stats[1] each.  stats[2] == 0 - nothing
    // was floored, the usual case - costs three scalar loads; otherwise the frame's dB row is read next to its gradient row.
    float fix_share = 0.0f, fix_max = 0.0f;
    bool fix = false;
    if (stats != nullptr) {
        const float s2 = stats[2];
        if (s2 != 0.0f) {
            fix = true;
            fix_max = stats[0];
            fix_share = s2 / (stats[1] > 0.0f ? stats[1] : 1.0f);
        }
    }
    fill_reg_tables(S.c);
    for (int i = threadIdx.x; i < kBins * SPAN_CAP; i += kThreads) {
        const int k = i / SPAN_CAP, j = i % SPAN_CAP;
        S.fbt_w[i] = j < span_t ? fbt_w[k * span_t + j] : 0.0f;
    }
    for (int i = threadIdx.x; i < kBins; i += kThreads) S.fbt_start[i] = fbt_start[i];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fs = lane >> 4, l = lane & 15;
    __syncthreads();
    const int64_t b = blockIdx.y;
    const float *xb = x + b * T;
    float2 *blk = S.c.blk[wave][fs];
    float carry[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool carried = false;                                      // the previous group left its last 352 sums in `carry`
    // a group whose frames involve no reflection and all exist (wave-uniform)
    auto interior = [&](int g0) {
        return hop == 160 && g0 + kGroup <= NF && g0 * hop >= 2 * kN && g0 * hop + 3 * hop + kNfft <= T;
    };
    for (int g = 0; g < kBwdGroupsPerWave; ++g) {
        const int f0 = blockIdx.x * kBwdFramesPerBlock + (wave * kBwdGroupsPerWave + g) * kGroup;
        if (f0 >= NF) break;                                   // wave-uniform
        const int f = f0 + fs;
        const bool live = f < NF;
        const int fc = live ? f : NF - 1;
        // the frame's band gradients: requested now (8 registers: the latency hides behind the transform), parked after the
        // un-packing in the frame's own block, which is free between the two exchanges; rows shorter than kMaxBands and the
        // SPAN_CAP entries behind them read 0 (padded taps reach them with weight 0)
        float dreg[kMaxBands / 16];
        {
            const float *src = dband + (b * NF + fc) * M;
#pragma unroll
            for (int i = 0; i < kMaxBands / 16; ++i) {
                const int m = l + 16 * i;
                dreg[i] = m < M ? src[m] : 0.0f;
            }
            if (fix) {                                            // workgroup-uniform
                const float *dbrow = band_db + (b * NF + fc) * M;
#pragma unroll
                for (int i = 0; i < kMaxBands / 16; ++i) {
                    const int m = l + 16 * i;
                    const float db = m < M ? dbrow[m] : 0.0f;
                    if (m < M && db == fix_max) dreg[i] += fix_share * dlog_of_db(db);
                }
            }
        }
        float2 xr[16];
        load_frame_reg(xb, reinterpret_cast<const float2 *>(w), T, fc, hop, xr, here(l));
        phase_done(xr);
        fft256_reg<false>(xr, S.c.twl, blk, l);
        float nyq;
        unpack_real_reg(xr, nyq, blk, l);
        phase_done(xr);
        float *drow = reinterpret_cast<float *>(blk);
        lds_wave_fence();
#pragma unroll
        for (int i = 0; i < kMaxBands / 16; ++i) drow[l + 16 * i] = dreg[i];
        if (l < SPAN_CAP) drow[kMaxBands + l] = 0.0f;
        // G[k] = d L / d X[k] for the one-sided inverse: 2 X[k] dp[k], interior bins halved, DC / Nyquist real; xr holds 2 X.
        // dp[k] = sum_j fbt_w[k][j] dband[fbt_start[k] + j]: four bins at a time, their starts first, then every weight and
        // band gradient in flight at once
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lds_wave_fence();          // one chunk's 20 reads in flight, not all four's: 66 registers fewer
            int m0[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) m0[i] = S.fbt_start[l + 16 * (4 * q + i)];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * q + i;
                float dp = 0.0f;
#pragma unroll
                for (int j = 0; j < SPAN_CAP; ++j) dp = fmaf(S.fbt_w[(l + 16 * r) * SPAN_CAP + j], drow[m0[i] + j], dp);
                const bool dc = r == 0 && l == 0;
                const float c = dc ? dp : 0.5f * dp;
                xr[r] = make_float2(c * xr[r].x, dc ? 0.0f : c * xr[r].y);
            }
            // this chunk's products are complete before the next chunk's reads are requested (see phase_done)
            asm volatile("" : "+v"(xr[4 * q].x), "+v"(xr[4 * q].y), "+v"(xr[4 * q + 1].x), "+v"(xr[4 * q + 1].y), "+v"(xr[4 * q + 2].x),
                              "+v"(xr[4 * q + 2].y), "+v"(xr[4 * q + 3].x), "+v"(xr[4 * q + 3].y) :: "memory");
        }
        float g_nyq = 0.0f;
        if (l == 0) {
            const int m0n = S.fbt_start[kN];
            float dpn = 0.0f;
#pragma unroll
            for (int j = 0; j < SPAN_CAP; ++j) dpn = fmaf(S.fbt_w[kN * SPAN_CAP + j], drow[m0n + j], dpn);
            g_nyq = dpn * nyq;
        }
        // pack for the inverse: Z'[k] = (G[k] + G*[N - k]) + i W512^-k (G[k] - G*[N - k]);  N - 0 is the Nyquist bin
        phase_done(xr);
        lds_wave_fence();
        park_vector(blk, xr, l);
        lds_wave_fence();
        int lp = here(l);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r == 8) half_done<0>(xr);
            float2 gc = conjf2(mirror_of(blk, r, l));
            if (r == 0 && l == 0) gc = make_float2(g_nyq, 0.0f);
            const float2 e = cadd(xr[r], gc), d = csub(xr[r], gc);
            const float2 wd = cmulf(conjf2(kTw512[lp + 16 * r]), d);
            xr[r] = make_float2(e.x - wd.y, e.y + wd.x);
        }
        fft256_reg<true>(xr, S.c.twl, blk, l);
        // z'[n] = dframe[2n] + i dframe[2n + 1], n = l + 16 r: window, park the frame's 512 floats in its block
        lds_wave_fence();
        lp = here(l);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 wv = reinterpret_cast<const float2 *>(w)[lp + 16 * r];
            blk[l + 16 * r] = live ? make_float2(wv.x * xr[r].x, wv.y * xr[r].y) : make_float2(0.0f, 0.0f);
        }
        lds_wave_fence();
        const float *df = reinterpret_cast<const float *>(S.c.blk[wave][0]);
        if (interior(f0)) {
            const bool carry_out = g + 1 < kBwdGroupsPerWave && f0 + kGroup < NF && interior(f0 + kGroup);
            overlap_add_interior<2 * kFrameSlots>(df, dx + b * T + (f0 * hop - kN), lane, carry, carried, carry_out);
            carried = carry_out;
        } else {
            overlap_add_frames_cold<2 * kFrameSlots>(df, dx + b * T, f0, NF, hop, T, lane, 64);
            carried = false;
        }
        lds_wave_fence();
    }
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

// ---- round 4: the mel-spec pair on the register-resident transform -----------------------------------------------------------
constexpr int kMelRegSpan = 16;                 // taps per band the register-FFT forward kernel takes (80 HTK bands over 257 bins: 16)
constexpr int kMelBandsPerLane = (kMelMax + 15) / 16;      // a lane owns bands l, l + 16, ..., of its frame

struct LdsMelReg {
    LdsReg c;
    float fbw[kMelMax * kMelRegSpan];           // [m][j], 0 beyond a band's run, beyond `span` and for m >= M
    int32_t fbs[kMelMax];
};
static_assert(2 * kMelMax * (kBandsFramesPerBlock + 1) * sizeof(float) <= sizeof(LdsReg::blk), "the output tile aliases the exchange blocks");

// grid (ceil(NF / 32), B); span <= kMelRegSpan, M <= kMelMax.  A wave works through two groups of 4 frames; a lane keeps the
// magnitude / phase of its 5 bands of both groups in registers until every wave of the workgroup is done with its exchange
// blocks, which then hold the workgroup's (plane, band) x 32-frame output tile: rows of 128 bytes go out.
__global__ __launch_bounds__(kThreads) void stft_mel_reg_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                const int32_t *__restrict__ fb_start,
                                                                const float *__restrict__ fb_w, int span,
                                                                float *__restrict__ out, int T, int NF, int hop, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LdsMelReg &SF = *reinterpret_cast<LdsMelReg *>(raw);
    LdsReg &S = SF.c;
    fill_reg_tables(S);
    for (int i = threadIdx.x; i < kMelMax * kMelRegSpan; i += kThreads) {
        const int m = i / kMelRegSpan, j = i % kMelRegSpan;
        SF.fbw[i] = (m < M && j < span) ? 0.5f * fb_w[m * span + j] : 0.0f;       // the transform below delivers 2 X
    }
    for (int m = threadIdx.x; m < kMelMax; m += kThreads) {
        const int k0 = m < M ? fb_start[m] : 0;
        SF.fbs[m] = k0 + kMelRegSpan <= kBins + 15 ? k0 : kBins + 15 - kMelRegSpan;      // padded taps stay inside the frame's block
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fs = lane >> 4, l = lane & 15;
    __syncthreads();
    const int64_t b = blockIdx.y;
    const int f_base = blockIdx.x * kBandsFramesPerBlock;
    float2 *blk = S.blk[wave][fs];
    float mag[kGroupsPerWave][kMelBandsPerLane], ph[kGroupsPerWave][kMelBandsPerLane];
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) {
        const int f0 = f_base + (wave * kGroupsPerWave + g) * kGroup;
#pragma unroll
        for (int i = 0; i < kMelBandsPerLane; ++i) mag[g][i] = ph[g][i] = 0.0f;
        if (f0 >= NF) continue;                                 // wave-uniform
        const int f = f0 + fs;
        const bool live = f < NF;
        float2 xr[16];
        load_frame_reg(x + b * T, reinterpret_cast<const float2 *>(w), T, live ? f : NF - 1, hop, xr, here(l));
        phase_done(xr);
        fft256_reg<false>(xr, S.twl, blk, l);
        float nyq;
        unpack_real_reg(xr, nyq, blk, l);
        phase_done(xr);
        // 2 X[0 .. 256] as a plain 257-vector in the frame's block (the mirror reads above were issued before these writes);
        // slots 257 .. 271 read 0: padded taps reach them with weight 0
        lds_wave_fence();
#pragma unroll
        for (int r = 0; r < 16; ++r) blk[l + 16 * r] = xr[r];
        if (l == 0) blk[kN] = make_float2(nyq, 0.0f);
        if (l >= 1) blk[kN + l] = make_float2(0.0f, 0.0f);
        lds_wave_fence();
        // Y[m] = sum_j fb[m][j] X[k0(m) + j] for the lane's bands, one band's 16 weights + 16 bins in flight at a time
#pragma unroll
        for (int i = 0; i < kMelBandsPerLane; ++i) {
            const int m = l + 16 * i;
            if (m < kMelMax) {
                const int k0 = SF.fbs[m];
                float re = 0.0f, im = 0.0f;
#pragma unroll
                for (int j = 0; j < kMelRegSpan; ++j) {
                    const float wt = SF.fbw[m * kMelRegSpan + j];
                    const float2 z = blk[k0 + j];
                    re = fmaf(wt, z.x, re);
                    im = fmaf(wt, z.y, im);
                }
                mag[g][i] = sqrtf(re * re + im * im);
                ph[g][i] = atan2f(im, re);
                asm volatile("" : "+v"(mag[g][i]), "+v"(ph[g][i]) :: "memory");       // this band is complete before the next one's reads
            }
        }
        lds_wave_fence();
    }
    __syncthreads();                                            // every wave is done with its exchange blocks
    float *tile = reinterpret_cast<float *>(&S.blk[0][0][0]);   // [plane][band][33]
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) {
        const int fl = (wave * kGroupsPerWave + g) * kGroup + fs;
#pragma unroll
        for (int i = 0; i < kMelBandsPerLane; ++i) {
            const int m = l + 16 * i;
            if (m < M) {
                tile[(0 * kMelMax + m) * (kBandsFramesPerBlock + 1) + fl] = mag[g][i];
                tile[(1 * kMelMax + m) * (kBandsFramesPerBlock + 1) + fl] = ph[g][i];
            }
        }
    }
    __syncthreads();
    const int frames_here = (NF - f_base) < kBandsFramesPerBlock ? (NF - f_base) : kBandsFramesPerBlock;
    for (int i = threadIdx.x; i < 2 * M * kBandsFramesPerBlock; i += kThreads) {
        const int fl = i % kBandsFramesPerBlock, row = i / kBandsFramesPerBlock, plane = row / M, m = row - plane * M;
        if (fl < frames_here)
            out[((b * 2 + plane) * M + m) * NF + f_base + fl] = tile[(plane * kMelMax + m) * (kBandsFramesPerBlock + 1) + fl];
    }
}

template <int SPAN_CAP>
struct LdsMelRegBwd {
    LdsReg c;
    float fbt_w[(kBins + 1) * SPAN_CAP];
    int32_t fbt_start[kBins + 1];
};

// grid (ceil(NF / 64), B): d x from d out and out alone (see stft_mel_backward_out_kernel), a wave owning 16 consecutive frames
// in 4 groups, its overlap-add carried in registers from group to group (see stft_bands_backward_reg_kernel).  Per group the
// frame's (d magnitude, d phase, magnitude, phase) rows are staged in the frame's own exchange block (4 x 80 floats), turned
// into d Y (80 complex, behind them), projected back onto the bins, packed and inverse-transformed.
template <int SPAN_CAP>
__global__ __launch_bounds__(kThreads) void stft_mel_backward_out_reg_kernel(const float *__restrict__ w, const float *__restrict__ dout,
                                                                             const float *__restrict__ out,
                                                                             const int32_t *__restrict__ fbt_start,
                                                                             const float *__restrict__ fbt_w, int span_t,
                                                                             float *__restrict__ dx, int T, int NF, int hop, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    LdsMelRegBwd<SPAN_CAP> &S = *reinterpret_cast<LdsMelRegBwd<SPAN_CAP> *>(raw);
    fill_reg_tables(S.c);
    for (int i = threadIdx.x; i < kBins * SPAN_CAP; i += kThreads) {
        const int k = i / SPAN_CAP, j = i % SPAN_CAP;
        S.fbt_w[i] = j < span_t ? fbt_w[k * span_t + j] : 0.0f;
    }
    for (int i = threadIdx.x; i < kBins; i += kThreads) S.fbt_start[i] = fbt_start[i];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fs = lane >> 4, l = lane & 15;
    __syncthreads();
    const int64_t b = blockIdx.y;
    float2 *blk = S.c.blk[wave][fs];
    float carry[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool carried = false;
    auto interior = [&](int g0) {
        return hop == 160 && g0 + kGroup <= NF && g0 * hop >= 2 * kN && g0 * hop + 3 * hop + kNfft <= T;
    };
    constexpr int kRows = 4 * kMelMax;                          // staged floats per frame; d Y follows at float offset kRows
    static_assert((kRows + 2 * (kMelMax + 8)) * sizeof(float) <= kFrameSlots * sizeof(float2), "stage + d Y fit a frame's block");
    for (int g = 0; g < kBwdGroupsPerWave; ++g) {
        const int f0 = blockIdx.x * kBwdFramesPerBlock + (wave * kBwdGroupsPerWave + g) * kGroup;
        if (f0 >= NF) break;                                    // wave-uniform
        const int f = f0 + fs;
        const bool live = f < NF;
        // (plane, band, frame) -> the frame's block: plane 0, 1 = d out, planes 2, 3 = out; a wave-instruction reads 16 rows x 4
        // consecutive frames (16-byte runs: the tensor is frame-minor)
        float *stage = reinterpret_cast<float *>(blk);
        {
            float *wave_blocks = reinterpret_cast<float *>(S.c.blk[wave][0]);
            for (int i = lane; i < 4 * M * kGroup; i += 64) {
                const int fl = i & 3, row = i >> 2, plane = row / M, m = row - plane * M;
                const float *src = plane < 2 ? dout : out;
                const float v = f0 + fl < NF ? src[((b * 2 + (plane & 1)) * M + m) * NF + f0 + fl] : 0.0f;
                wave_blocks[fl * 2 * kFrameSlots + plane * kMelMax + m] = v;
            }
        }
        lds_wave_fence();
        // d (|Y|, angle Y) -> d Y from the forward OUTPUT: g_mag (cos, sin) + (g_phase / |Y|) (-sin, cos), 0 at |Y| = 0
        float2 *dy = reinterpret_cast<float2 *>(stage + kRows);
#pragma unroll
        for (int i = 0; i < kMelBandsPerLane; ++i) {
            const int m = l + 16 * i;
            if (m < kMelMax) {
                float2 gy = make_float2(0.0f, 0.0f);
                if (m < M) {
                    const float gm = stage[m], gp = stage[kMelMax + m], mg = stage[2 * kMelMax + m], phv = stage[3 * kMelMax + m];
                    if (mg > 0.0f) {
                        float sn, cs;
                        sincosf(phv, &sn, &cs);
                        const float q = gp / mg;
                        gy.x = gm * cs - q * sn;
                        gy.y = gm * sn + q * cs;
                    }
                }
                dy[m] = gy;
            }
        }
        if (l < 8) dy[kMelMax + l] = make_float2(0.0f, 0.0f);   // padded taps read these with weight 0
        lds_wave_fence();
        // d X[k] = sum_j fbt_w[k][j] d Y[fbt_start[k] + j], four bins at a time (see stft_bands_backward_reg_kernel)
        float2 xr[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lds_wave_fence();
            int m0[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) m0[i] = S.fbt_start[l + 16 * (4 * q + i)];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * q + i;
                float re = 0.0f, im = 0.0f;
#pragma unroll
                for (int j = 0; j < SPAN_CAP; ++j) {
                    const float wt = S.fbt_w[(l + 16 * r) * SPAN_CAP + j];
                    const float2 gy = dy[m0[i] + j];
                    re = fmaf(wt, gy.x, re);
                    im = fmaf(wt, gy.y, im);
                }
                // one-sided inverse: interior bins halved, the imaginary part of DC dropped
                const bool dc = r == 0 && l == 0;
                xr[r] = dc ? make_float2(re, 0.0f) : make_float2(0.5f * re, 0.5f * im);
            }
            asm volatile("" : "+v"(xr[4 * q].x), "+v"(xr[4 * q].y), "+v"(xr[4 * q + 1].x), "+v"(xr[4 * q + 1].y), "+v"(xr[4 * q + 2].x),
                              "+v"(xr[4 * q + 2].y), "+v"(xr[4 * q + 3].x), "+v"(xr[4 * q + 3].y) :: "memory");
        }
        float g_nyq = 0.0f;
        if (l == 0) {
            const int m0n = S.fbt_start[kN];
#pragma unroll
            for (int j = 0; j < SPAN_CAP; ++j) g_nyq = fmaf(S.fbt_w[kN * SPAN_CAP + j], dy[m0n + j].x, g_nyq);
        }
        phase_done(xr);
        lds_wave_fence();
        park_vector(blk, xr, l);
        lds_wave_fence();
        int lp = here(l);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r == 8) half_done<0>(xr);
            float2 gc = conjf2(mirror_of(blk, r, l));
            if (r == 0 && l == 0) gc = make_float2(g_nyq, 0.0f);
            const float2 e = cadd(xr[r], gc), d = csub(xr[r], gc);
            const float2 wd = cmulf(conjf2(kTw512[lp + 16 * r]), d);
            xr[r] = make_float2(e.x - wd.y, e.y + wd.x);
        }
        fft256_reg<true>(xr, S.c.twl, blk, l);
        lds_wave_fence();
        lp = here(l);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 wv = reinterpret_cast<const float2 *>(w)[lp + 16 * r];
            blk[l + 16 * r] = live ? make_float2(wv.x * xr[r].x, wv.y * xr[r].y) : make_float2(0.0f, 0.0f);
        }
        lds_wave_fence();
        const float *df = reinterpret_cast<const float *>(S.c.blk[wave][0]);
        if (interior(f0)) {
            const bool carry_out = g + 1 < kBwdGroupsPerWave && f0 + kGroup < NF && interior(f0 + kGroup);
            overlap_add_interior<2 * kFrameSlots>(df, dx + b * T + (f0 * hop - kN), lane, carry, carried, carry_out);
            carried = carry_out;
        } else {
            overlap_add_frames_cold<2 * kFrameSlots>(df, dx + b * T, f0, NF, hop, T, lane, 64);
            carried = false;
        }
        lds_wave_fence();
    }
}

constexpr int64_t kMaxGridY = 65535;

// ADVSTEP_STFT_REG=0 (read per call): the radix-4 in-LDS kernels of rounds 1 - 3 for the LFCC pair (A/B measurements)
inline bool reg_fft_enabled() {
    const char *e = getenv("ADVSTEP_STFT_REG");
    return !(e && e[0] == '0');
}

}  // namespace

#define STFT_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

namespace {
// band_db / stats: null, or the dB rows and the statistics the floor fix-up needs (see the kernel); dx_is_zero: the caller
// (advstep_lfcc_project_backward_zero_f32) has already zero-filled dx on this stream
int stft_bands_backward_launch(const float *x, const float *window, const float *dband, const int32_t *fbt_start, const float *fbt_w,
                               int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft, int64_t M,
                               const float *band_db, const float *stats, int dx_is_zero, advstep_stream_t stream) {
    STFT_REQUIRE(B >= 0 && NF >= 0 && M >= 0 && span_t >= 1);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    STFT_REQUIRE(x && window && dband && fbt_start && fbt_w && dx && B <= kMaxGridY);
    STFT_REQUIRE(advstep_stft_bands_supported(nfft, hop, T) && NF == 1 + T / hop && span_t <= kMaxSpanT && M <= kMaxBands);
    // a sample may be reached by at most two workgroups (two-operand float atomics commute): a workgroup's frames must
    // advance by at least the overlap between neighbouring workgroups' ranges
    STFT_REQUIRE(kFramesPerBlockBwd * hop >= kNfft - hop);
    hipStream_t st = as_stream(stream);
    if (!dx_is_zero && hipMemsetAsync(dx, 0, (size_t)B * T * sizeof(float), st) != hipSuccess) return ADVSTEP_ELAUNCH;
    const bool reg = reg_fft_enabled();
    const dim3 grid((unsigned)ceil_div(NF, reg ? kBwdFramesPerBlock : kFramesPerBlockBwd), (unsigned)B);
    if (reg) {
        auto go = [&](auto kernel, size_t lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, st, x, window, dband, fbt_start, fbt_w, (int)span_t, dx, (int)T,
                               (int)NF, (int)hop, (int)M, band_db, stats);
        };
        if (span_t <= 2) go(stft_bands_backward_reg_kernel<2>, sizeof(LdsRegBwd<2>));
        else go(stft_bands_backward_reg_kernel<kMaxSpanT>, sizeof(LdsRegBwd<kMaxSpanT>));
    } else {
        auto go = [&](auto kernel, size_t lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, st, x, window, dband, fbt_start, fbt_w, (int)span_t, dx, (int)T,
                               (int)NF, (int)hop, (int)M);
        };
        if (span_t <= 2) go(stft_bands_backward_kernel<2>, sizeof(LdsBwd<2>));
        else go(stft_bands_backward_kernel<kMaxSpanT>, sizeof(LdsBwd<kMaxSpanT>));
    }
    return status_after_launch();
}
}  // namespace

extern "C" {

size_t advstep_stft_bands_block_count(int64_t B, int64_t NF) {
    if (B <= 0 || NF <= 0) return 0;
    return (size_t)(B * ceil_div(NF, kFramesPerBlockFwd));      // the larger of the two kernels' workgroup counts
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
    if (reg_fft_enabled() && span <= kRegSpan && M <= kMaxBands) {
        // the unused tail of block_max (sized for the radix-4 kernel's 16-frame workgroups) must not hold garbage: the
        // reduction over it reads advstep_stft_bands_block_count entries
        const int64_t blocks32 = ceil_div(NF, kBandsFramesPerBlock), blocks16 = ceil_div(NF, kFramesPerBlockFwd);
        STFT_REQUIRE(B * blocks16 <= INT32_MAX);
        const dim3 grid((unsigned)blocks32, (unsigned)B);
        hipLaunchKernelGGL(stft_bands_reg_kernel, grid, dim3(kThreads), 0, as_stream(stream), x, window, fb_start, fb_w, (int)span,
                           band_db, block_max, (int)T, (int)NF, (int)hop, (int)M, (int)(B * (blocks16 - blocks32)));
        return status_after_launch();
    }
    const dim3 grid((unsigned)ceil_div(NF, kFramesPerBlockFwd), (unsigned)B);
    hipLaunchKernelGGL(stft_bands_kernel, grid, dim3(kThreads), 0, as_stream(stream), x, window, fb_start, fb_w, (int)span,
                       band_db, block_max, (int)T, (int)NF, (int)hop, (int)M);
    return status_after_launch();
}

int advstep_stft_bands_backward_f32(const float *x, const float *window, const float *dband, const int32_t *fbt_start,
                                    const float *fbt_w, int64_t span_t, float *dx, int64_t B, int64_t T, int64_t NF,
                                    int64_t hop, int64_t nfft, int64_t M, advstep_stream_t stream) {
    return stft_bands_backward_launch(x, window, dband, fbt_start, fbt_w, span_t, dx, B, T, NF, hop, nfft, M, nullptr, nullptr, 0,
                                      stream);
}

int advstep_stft_bands_backward_fixup_f32(const float *x, const float *window, float *dband, const float *band_db,
                                          const float *stats, const int32_t *fbt_start, const float *fbt_w, int64_t span_t,
                                          float *dx, int dx_is_zero, int64_t B, int64_t T, int64_t NF, int64_t hop,
                                          int64_t nfft, int64_t M, advstep_stream_t stream) {
    STFT_REQUIRE(band_db && stats);
    if (!reg_fft_enabled()) {                 // the radix-4 kernels have no folded fix-up: its own launch, on dband in place
        if (B > 0 && NF > 0 && M > 0) {
            const int st = advstep_lfcc_floor_fixup_f32(band_db, stats, dband, B * NF * M, stream);
            if (st != ADVSTEP_OK) return st;
        }
        return stft_bands_backward_launch(x, window, dband, fbt_start, fbt_w, span_t, dx, B, T, NF, hop, nfft, M, nullptr, nullptr,
                                          dx_is_zero, stream);
    }
    return stft_bands_backward_launch(x, window, dband, fbt_start, fbt_w, span_t, dx, B, T, NF, hop, nfft, M, band_db, stats,
                                      dx_is_zero, stream);
}

int advstep_stft_mel_f32(const float *x, const float *window, const int32_t *fb_start, const float *fb_w, int64_t span,
                         float *out, int64_t B, int64_t T, int64_t NF, int64_t hop, int64_t nfft, int64_t M,
                         advstep_stream_t stream) {
    STFT_REQUIRE(B >= 0 && NF >= 0 && M >= 0 && span >= 1);
    if (B == 0 || NF == 0 || M == 0) return ADVSTEP_OK;
    STFT_REQUIRE(x && window && fb_start && fb_w && out && B <= kMaxGridY && M <= kMelMax && span <= kMelMaxSpan);
    STFT_REQUIRE(advstep_stft_bands_supported(nfft, hop, T) && NF == 1 + T / hop);
    hipStream_t st = as_stream(stream);
    if (reg_fft_enabled() && span <= kMelRegSpan) {
        const size_t lds_reg = sizeof(LdsMelReg);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(stft_mel_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_reg);
        hipLaunchKernelGGL(stft_mel_reg_kernel, dim3((unsigned)ceil_div(NF, kBandsFramesPerBlock), (unsigned)B), dim3(kThreads),
                           lds_reg, st, x, window, fb_start, fb_w, (int)span, out, (int)T, (int)NF, (int)hop, (int)M);
        return status_after_launch();
    }
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
    if (hipMemsetAsync(dx, 0, (size_t)B * T * sizeof(float), st) != hipSuccess) return ADVSTEP_ELAUNCH;
    if (reg_fft_enabled()) {
        auto go_reg = [&](auto kernel, size_t lds_reg) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg);
            hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(NF, kBwdFramesPerBlock), (unsigned)B), dim3(kThreads), lds_reg, st, window,
                               dout, out, fbt_start, fbt_w, (int)span_t, dx, (int)T, (int)NF, (int)hop, (int)M);
        };
        if (span_t <= 2) go_reg(stft_mel_backward_out_reg_kernel<2>, sizeof(LdsMelRegBwd<2>));
        else go_reg(stft_mel_backward_out_reg_kernel<kMaxSpanT>, sizeof(LdsMelRegBwd<kMaxSpanT>));
        return status_after_launch();
    }
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
