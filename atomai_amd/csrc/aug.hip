// aug.hip — on-the-fly training-data augmentation on the resident batch.
//
// The reference augments every mini-batch on the CPU (numpy / skimage / scipy / cv2, float64) and re-uploads it:
//   datatransform.run                     atomai/transforms/imaug.py:302-358
//   apply_rotation / gauss / poisson / sp / blur / contrast / background    imaug.py:108-299
//   hook: BaseTrainer.dataloader          atomai/trainers/trainer.py:339-341
// At > 1 k images/s that path starves the GPU.  Here the batch never leaves HBM:
//   amx_aug_minmax   global (min, max) of a tensor, two deterministic stages          [(x - min) / ptp, imaug.py:316, 357]
//   amx_aug_point    ONE pass over the output pixels: flip / 90-degree rotation gather, initial normalisation,
//                    gaussian noise (+clip), poisson noise, salt & pepper, gamma contrast, 2-D gaussian background;
//                    per-image scalar parameters come from the host (they are a handful of np.random.randint draws in
//                    the reference's order), per-pixel randomness from a counter-based Philox4x32-10 generator keyed
//                    by (seed, image, pixel, operation) — or from caller-supplied noise fields (test hook: the
//                    arithmetic is then comparable to the reference element by element)
//   amx_aug_blur     separable gaussian filter, scipy.ndimage 'reflect' boundary, truncate 4 sigma   [imaug.py:170-180]
//   amx_aug_labels   the same flip / rotation applied to the integer class map + per-image class-presence bit mask
//                    (squeeze_channels drops image-label pairs in which a class is absent, imaug.py:361-395)
// All of it is HBM-bound elementwise work: 16 B / pixel read + written for the point pass.
#include "amx_device.h"
#include <cmath>

// ---------------------------------------------------------------------------------------------- Philox4x32-10
struct Philox { unsigned c[4]; };
static __device__ __forceinline__ unsigned amx_mulhi32(unsigned a, unsigned b) {
    return (unsigned)(((unsigned long long)a * b) >> 32);
}
static __device__ __forceinline__ Philox philox(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2,
                                                unsigned c3) {
    #pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned h0 = amx_mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = amx_mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox p; p.c[0] = c0; p.c[1] = c1; p.c[2] = c2; p.c[3] = c3;
    return p;
}
static __device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

// Poisson(lam) from a stream of uniforms: multiplication method below 10, Hoermann's PTRS above.
static __device__ float poisson_draw(float lam, unsigned k0, unsigned k1, unsigned c0, unsigned c1) {
    if (!(lam > 0.f)) return 0.f;
    unsigned ctr = 0;
    Philox r = philox(k0, k1, c0, c1, 2u, ctr);
    int used = 0;
    auto next = [&]() {
        if (used == 4) { r = philox(k0, k1, c0, c1, 2u, ++ctr); used = 0; }
        return u01(r.c[used++]);
    };
    if (lam < 10.f) {
        const float L = expf(-lam);
        float p = 1.f; int k = 0;
        do { ++k; p *= next(); } while (p > L && k < 200);
        return (float)(k - 1);
    }
    const float slam = sqrtf(lam), loglam = logf(lam);
    const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
    const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.f);
    for (int it = 0; it < 64; ++it) {
        const float U = next() - 0.5f, V = next();
        const float us = 0.5f - fabsf(U);
        const float k = floorf((2.f * a / us + b) * U + lam + 0.43f);
        if (us >= 0.07f && V <= vr) return k;
        if (k < 0.f || (us < 0.013f && V > us)) continue;
        if (logf(V) + logf(invalpha) - logf(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.f)) return k;
    }
    return floorf(lam + 0.5f);
}

// ---------------------------------------------------------------------------------------------- min / max
__global__ __launch_bounds__(256) void aug_minmax_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
    __shared__ float smn[256], smx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + s]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smn[0]; part[2 * blockIdx.x + 1] = smx[0]; }
}
__global__ __launch_bounds__(256) void aug_minmax_final(const float* __restrict__ part, int nb, float* __restrict__ out) {
    __shared__ float smn[256], smx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += 256) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + s]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = smn[0]; out[1] = smx[0]; }
}

// out[0] = min(x), out[1] = max(x); work: 2 * amx_aug_minmax_blocks(n) floats
extern "C" int amx_aug_minmax_blocks(long n) {
    long nb = (n + 256 * 16 - 1) / (256 * 16);
    return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}
extern "C" int amx_aug_minmax(const float* x, long n, float* work, float* out, void* stream) {
    if (!x || !work || !out || n <= 0) AMX_BADARG(1);
    const int nb = amx_aug_minmax_blocks(n);
    AMX_LAUNCH(aug_minmax_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, work);
    AMX_LAUNCH(aug_minmax_final, dim3(1), dim3(256), 0, (hipStream_t)stream, work, nb, out);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------- point pass
// per-image parameter record (floats), written by the host:
//   [0] flip code: -1 both axes, 0 vertical (rows reversed), 1 horizontal (columns reversed), 2 rot90 counter-clockwise,
//       3 rot90 clockwise, 4 none                               (cv2.flip / cv2.rotate codes of imaug.py:262-280)
//   [1] gaussian sigma (0 = off)         [2] poisson scale `vals` (0 = off)      [3] salt & pepper amount (0 = off)
//   [4] gamma (0 = off)                  [5..9] background x0, y0, a, b, fwhm    [10] background amplitude (0 = off)
#define AUG_NP 12

struct AugArgs {
    const float* x; float* y;          // [N][H][W] in / out
    const float* params;               // [N][AUG_NP]
    const float* mnmx;                 // device (min, max) of x for the initial normalisation, or nullptr
    const float* f_gauss;              // optional injected fields [N][H][W]: standard normals,
    const float* f_pois;               //   poisson draws (already sampled for lam = v * vals),
    const float* f_sp1; const float* f_sp2;   //   uniforms deciding "flipped" / "salted"
    const int* jitter;                 // [N][H] row shifts z: out[y][x] = noisy[y][(x - z) mod W] (np.roll per row,
                                       // applied AFTER the gaussian noise: imaug.py:123-135, 332-335), or nullptr
    int N, H, W;
    unsigned seed0, seed1;
};

static __device__ __forceinline__ void src_of(int code, int H, int W, int oy, int ox, int& sy, int& sx) {
    switch (code) {
        case -1: sy = H - 1 - oy; sx = W - 1 - ox; break;
        case 0: sy = H - 1 - oy; sx = ox; break;
        case 1: sy = oy; sx = W - 1 - ox; break;
        case 2: sy = ox; sx = W - 1 - oy; break;          // rotate 90 counter-clockwise: out[i][j] = in[j][W-1-i]
        case 3: sy = H - 1 - ox; sx = oy; break;          // rotate 90 clockwise:         out[i][j] = in[H-1-j][i]
        default: sy = oy; sx = ox;
    }
}

__global__ __launch_bounds__(256) void aug_point_kernel(AugArgs a) {
    const long hw = (long)a.H * a.W, total = hw * a.N;
    float mn = 0.f, inv = 1.f;
    if (a.mnmx) { mn = a.mnmx[0]; inv = a.mnmx[1] - a.mnmx[0]; }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / hw);
        const int r = (int)(i - (long)n * hw);
        const int oy = r / a.W, ox = r - oy * a.W;
        const float* P = a.params + (size_t)n * AUG_NP;
        // jitter rolls the rows of the rotated, noise-carrying image: this output pixel shows the pixel `jx` of its row
        int jx = ox;
        if (a.jitter) { jx = (ox - a.jitter[(long)n * a.H + oy]) % a.W; if (jx < 0) jx += a.W; }
        const int rj = oy * a.W + jx;
        int sy, sx;
        src_of((int)P[0], a.H, a.W, oy, jx, sy, sx);
        float v = a.x[(long)n * hw + (long)sy * a.W + sx];
        if (a.mnmx) v = (v - mn) / inv;
        if (P[1] > 0.f) {                                   // skimage random_noise(mode='gaussian', clip=True)
            float z;
            if (a.f_gauss) z = a.f_gauss[(long)n * hw + rj];
            else {
                const Philox q = philox(a.seed0, a.seed1, (unsigned)rj, (unsigned)n, 1u, 0u);
                z = sqrtf(-2.f * logf(u01(q.c[0]))) * cosf(6.28318530718f * u01(q.c[1]));
            }
            v = fminf(fmaxf(v + P[1] * z, 0.f), 1.f);
        }
        if (P[2] > 0.f) {                                   // np.random.poisson(image * vals) / vals
            const float k = a.f_pois ? a.f_pois[i] : poisson_draw(v * P[2], a.seed0, a.seed1, (unsigned)r, (unsigned)n);
            v = k / P[2];
        }
        if (P[3] > 0.f) {                                   // skimage random_noise(mode='s&p', salt_vs_pepper=0.5)
            float u1, u2;
            if (a.f_sp1) { u1 = a.f_sp1[i]; u2 = a.f_sp2[i]; }
            else { const Philox q = philox(a.seed0, a.seed1, (unsigned)r, (unsigned)n, 3u, 0u); u1 = u01(q.c[0]); u2 = u01(q.c[1]); }
            if (u1 < P[3]) v = u2 < 0.5f ? 1.f : 0.f;
        }
        if (P[4] > 0.f) v = powf(fmaxf(v, 0.f), P[4]);      // skimage exposure.adjust_gamma (gain 1, float image)
        if (P[10] != 0.f) {                                 // asymmetric 2-D gaussian, imaug.py:236-254
            // np.linspace(0, h, h): coordinate of row y is y * h / (h - 1)
            const float cy = a.H > 1 ? (float)oy * a.H / (a.H - 1) : 0.f, cx = a.W > 1 ? (float)ox * a.W / (a.W - 1) : 0.f;
            const float dy = cy - P[5], dx = cx - P[6];
            v += P[10] * expf(-0.69314718056f * (P[7] * dy * dy + P[8] * dx * dx) / (P[9] * P[9]));
        }
        a.y[i] = v;
    }
}

extern "C" int amx_aug_point(const float* x, float* y, const float* params, const float* mnmx, const float* f_gauss,
                             const float* f_pois, const float* f_sp1, const float* f_sp2, const int* jitter,
                             int N, int H, int W, long seed, void* stream) {
    if (!x || !y || !params || x == y) AMX_BADARG(1);
    if (N <= 0 || H <= 0 || W <= 0) AMX_BADARG(2);
    if ((f_sp1 == nullptr) != (f_sp2 == nullptr)) AMX_BADARG(3);
    AugArgs a;
    a.x = x; a.y = y; a.params = params; a.mnmx = mnmx; a.f_gauss = f_gauss; a.f_pois = f_pois; a.f_sp1 = f_sp1;
    a.f_sp2 = f_sp2; a.jitter = jitter; a.N = N; a.H = H; a.W = W;
    a.seed0 = (unsigned)(seed & 0xffffffffL); a.seed1 = (unsigned)((unsigned long long)seed >> 32) ^ 0x5bd1e995u;
    long nb = ((long)N * H * W + 255) / 256;
    if (nb > 8192) nb = 8192;
    AMX_LAUNCH(aug_point_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
    AMX_CHECK_LAUNCH();
    return 0;
}

// (x - mn) / (mx - mn) in place with (mn, mx) read from the device
__global__ void aug_renorm_kernel(float* __restrict__ x, long n, const float* __restrict__ mnmx) {
    const float mn = mnmx[0], ptp = mnmx[1] - mnmx[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = (x[i] - mn) / ptp;
}
extern "C" int amx_aug_renorm(float* x, long n, const float* mnmx, void* stream) {
    if (!x || !mnmx || n <= 0) AMX_BADARG(1);
    long nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    AMX_LAUNCH(aug_renorm_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, n, mnmx);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------- gaussian blur
// One axis of scipy.ndimage.gaussian_filter (mode='reflect': (d c b a | a b c d | d c b a), truncate = 4):
// weights w[j] = exp(-j^2 / (2 sigma^2)) / sum, radius = int(4 sigma + 0.5).  sigma[n] <= 0 copies the image.
__global__ void aug_blur_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ sigma,
                                int N, int H, int W, int axis) {
    const long hw = (long)H * W, total = hw * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / hw);
        const int r = (int)(i - (long)n * hw);
        const int oy = r / W, ox = r - oy * W;
        const float sg = sigma[n];
        const int rad = sg > 0.f ? (int)(4.f * sg + 0.5f) : 0;
        if (rad == 0) { y[i] = x[i]; continue; }
        const int L = axis == 0 ? H : W, c = axis == 0 ? oy : ox;
        const float* base = x + (long)n * hw;
        float acc = 0.f, wsum = 0.f;
        for (int j = -rad; j <= rad; ++j) {
            const float w = expf(-0.5f * (float)(j * j) / (sg * sg));
            int q = c + j;
            // reflect about the edges (period 2L)
            const int P2 = 2 * L;
            q = ((q % P2) + P2) % P2;
            if (q >= L) q = P2 - 1 - q;
            acc += w * (axis == 0 ? base[(long)q * W + ox] : base[(long)oy * W + q]);
            wsum += w;
        }
        y[i] = acc / wsum;
    }
}
extern "C" int amx_aug_blur(const float* x, float* y, const float* sigma, int N, int H, int W, int axis, void* stream) {
    if (!x || !y || !sigma || x == y) AMX_BADARG(1);
    if (N <= 0 || H <= 0 || W <= 0 || (axis != 0 && axis != 1)) AMX_BADARG(2);
    long nb = ((long)N * H * W + 255) / 256;
    if (nb > 8192) nb = 8192;
    AMX_LAUNCH(aug_blur_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, sigma, N, H, W, axis);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------- labels
// Integer class maps [N][H][W] (int64) under the same flip / rotation codes; present[n] = OR of (1 << class) over the
// image (integer atomicOr: exact, order independent).  `present` must be zeroed by the caller.
__global__ void aug_labels_kernel(const long long* __restrict__ t, long long* __restrict__ out,
                                  const float* __restrict__ params, int* __restrict__ present, int N, int H, int W) {
    const long hw = (long)H * W, total = hw * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / hw);
        const int r = (int)(i - (long)n * hw);
        const int oy = r / W, ox = r - oy * W;
        int sy, sx;
        src_of((int)params[(size_t)n * AUG_NP], H, W, oy, ox, sy, sx);
        const long long v = t[(long)n * hw + (long)sy * W + sx];
        out[i] = v;
        if (present && v >= 0 && v < 31) atomicOr(present + n, 1 << (int)v);
    }
}
extern "C" int amx_aug_labels(const long long* t, long long* out, const float* params, int* present, int N, int H, int W,
                              void* stream) {
    if (!t || !out || !params || t == out) AMX_BADARG(1);
    if (N <= 0 || H <= 0 || W <= 0) AMX_BADARG(2);
    long nb = ((long)N * H * W + 255) / 256;
    if (nb > 8192) nb = 8192;
    AMX_LAUNCH(aug_labels_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, t, out, params, present, N, H, W);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------- zoom / resize
// apply_zoom (imaug.py:195-227): centred zv x zv crop -> cv2.resize(..., (S, S), interpolation=cv2.INTER_CUBIC), image
// clipped to [0, 1], masks rounded.  apply_imresize (imaug.py:276-300): cv2.resize(img, (w', h'), rs_method) — the third
// POSITIONAL parameter of cv2.resize is `dst`, so rs_method never reaches `interpolation` and the call runs with the
// default INTER_LINEAR; that is what mode 0 provides (mode 1 = INTER_CUBIC).  OpenCV's arithmetic, restated from its
// documentation / imgproc sources (cv2 is absent in this image: UNPINNED against cv2 itself; the numpy restatement in
// oracle/aug_oracle.py is checked against torch's F.interpolate(align_corners=False), which documents the same
// conventions: half-pixel centres, a = -0.75, clamped tap indices):
//   source coordinate  f = (d + 0.5) * (src / dst) - 0.5,  s = floor(f),  t = f - s
//   linear:  s < 0 -> (s, t) = (0, 0);  s >= src - 1 -> (src - 1, 0);  value = S[s] * (1 - t) + S[s + 1] * t
//   cubic:   taps s - 1 .. s + 2 with indices clamped to [0, src - 1];  weights (A = -0.75, formed in float):
//            w0 = ((A (t + 1) - 5A)(t + 1) + 8A)(t + 1) - 4A,  w1 = ((A + 2) t - (A + 3)) t^2 + 1,
//            w2 = ((A + 2)(1 - t) - (A + 3))(1 - t)^2 + 1,     w3 = 1 - w0 - w1 - w2
//   area (mode 2, cv2.INTER_AREA when the image is ENLARGED along an axis — what utils/img.py:cv_resize selects for
//            `img.shape[0] < target`, predictors/predictor.py:203-204): OpenCV runs its linear kernel with "area-mode"
//            coefficients:  s = floor(d * src / dst),  t = (d + 1) - (s + 1) * dst / src,  t <= 0 -> 0 else t - floor(t);
//            same border rule as linear.  (True area averaging — both axes shrunk under INTER_AREA — is a different
//            kernel and is not provided: cv_resize only reaches it for non-square frames.)
//   horizontal pass first, then vertical (separable; evaluated per output pixel here).
// win: per image (y0, x0, h, w) of the source window inside the [Hs][Ws] frame (zoom: centred crop; resize: whole frame).
struct ResampleTaps { int idx[4]; float w[4]; int n; };

static __device__ __forceinline__ ResampleTaps resample_taps(int d, int src, int dst, int mode) {
    ResampleTaps r;
    const double scale = (double)src / (double)dst;          // OpenCV: the coordinate is formed in double, then cast
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    float t = f - (float)s;
    if (mode == 2) {
        const double inv_scale = (double)dst / (double)src;
        s = (int)floor((double)d * scale);
        t = (float)((double)(d + 1) - (double)(s + 1) * inv_scale);
        t = t <= 0.f ? 0.f : t - floorf(t);
    }
    if (mode == 0 || mode == 2) {
        if (s < 0) { s = 0; t = 0.f; }
        if (s >= src - 1) { s = src - 1; t = 0.f; }
        r.n = 2;
        r.idx[0] = s; r.idx[1] = s + 1 < src ? s + 1 : src - 1;
        r.w[0] = 1.f - t; r.w[1] = t;
        r.idx[2] = r.idx[3] = 0; r.w[2] = r.w[3] = 0.f;
    } else {
        const float A = -0.75f;
        r.n = 4;
        r.w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
        r.w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
        r.w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
        r.w[3] = 1.f - r.w[0] - r.w[1] - r.w[2];
        #pragma unroll
        for (int j = 0; j < 4; ++j) { int q = s - 1 + j; q = q < 0 ? 0 : (q > src - 1 ? src - 1 : q); r.idx[j] = q; }
    }
    return r;
}

// images: x [N][Hs][Ws] -> y [N][Hd][Wd];  clip01: np.clip(img, 0, 1) of apply_zoom;  round_out: np.around (binary masks)
__global__ void aug_resample_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ win,
                                    int N, int Hs, int Ws, int Hd, int Wd, int mode, int clip01, int round_out) {
    const long total = (long)N * Hd * Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dx = (int)(i % Wd); const long r_ = i / Wd; const int dy = (int)(r_ % Hd); const int n = (int)(r_ / Hd);
        const int y0 = win[n * 4 + 0], x0 = win[n * 4 + 1], wh = win[n * 4 + 2], ww = win[n * 4 + 3];
        const ResampleTaps ty = resample_taps(dy, wh, Hd, mode), tx = resample_taps(dx, ww, Wd, mode);
        const float* img = x + (size_t)n * Hs * Ws;
        float acc = 0.f;
        for (int a = 0; a < ty.n; ++a) {
            const float* row = img + (size_t)(y0 + ty.idx[a]) * Ws + x0;
            float h = 0.f;
            for (int b = 0; b < tx.n; ++b) h = fmaf(row[tx.idx[b]], tx.w[b], h);       // horizontal pass of this row
            acc = fmaf(h, ty.w[a], acc);
        }
        if (clip01) acc = fminf(fmaxf(acc, 0.f), 1.f);
        if (round_out) acc = rintf(acc);                                                // np.around: half to even
        y[i] = acc;
    }
}

extern "C" int amx_aug_resample(const float* x, float* y, const int* win, int N, int Hs, int Ws, int Hd, int Wd, int mode,
                                int clip01, int round_out, void* stream) {
    if (!x || !y || !win || x == y) AMX_BADARG(1);
    if (N <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || mode < 0 || mode > 2) AMX_BADARG(2);
    long nb = ((long)N * Hd * Wd + 255) / 256;
    if (nb > 16384) nb = 16384;
    AMX_LAUNCH(aug_resample_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, win, N, Hs, Ws, Hd, Wd,
               mode, clip01, round_out);
    AMX_CHECK_LAUNCH();
    return 0;
}

// class maps: the reference resamples the K one-hot masks separately (zoom, then resize, each followed by np.around) and
// squeezes them back at the end with label = sum_c c * mask_c (squeeze_channels, imaug.py:361-393) — a pixel whose rounded
// masks are all 0 becomes class 0, one with two masks set becomes the SUM of their indices.  So the class map is expanded
// to K float planes (amx_aug_onehot), the planes go through amx_aug_resample (round_out = 1) like images, and
// amx_aug_squeeze forms the sum; values[n]: bit v set if the value v occurs in image n (the reference keeps the pair iff
// exactly K distinct values occur).
__global__ void aug_onehot_kernel(const long long* __restrict__ t, float* __restrict__ m, int N, int K, long HW) {
    const long total = (long)N * K * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW; const long r_ = i / HW; const int c = (int)(r_ % K); const long n = r_ / K;
        m[i] = t[n * HW + hw] == c ? 1.f : 0.f;
    }
}
extern "C" int amx_aug_onehot(const long long* t, float* masks, int N, int K, long HW, void* stream) {
    if (!t || !masks || N <= 0 || K < 2 || K > 32 || HW <= 0) AMX_BADARG(1);   // (the class-presence word of amx_aug_squeeze has 32 bits)
    long nb = ((long)N * K * HW + 255) / 256;
    if (nb > 16384) nb = 16384;
    AMX_LAUNCH(aug_onehot_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, t, masks, N, K, HW);
    AMX_CHECK_LAUNCH();
    return 0;
}

__global__ void aug_squeeze_kernel(const float* __restrict__ m, long long* __restrict__ out, int* __restrict__ values,
                                   int N, int K, long HW) {
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW; const long n = i / HW;
        float v = 0.f;
        for (int c = 0; c < K; ++c) v = fmaf(m[((size_t)n * K + c) * HW + hw], (float)c, v);     // exact: small integers
        const int iv = (int)v;
        out[i] = iv;
        if (values && iv >= 0 && iv < 32) atomicOr(values + n, 1 << iv);
    }
}
extern "C" int amx_aug_squeeze(const float* masks, long long* out, int* values, int N, int K, long HW, void* stream) {
    if (!masks || !out || N <= 0 || K < 2 || K > 32 || HW <= 0) AMX_BADARG(1);   // (the class-presence word of amx_aug_squeeze has 32 bits)
    long nb = ((long)N * HW + 255) / 256;
    if (nb > 16384) nb = 16384;
    AMX_LAUNCH(aug_squeeze_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, masks, out, values, N, K, HW);
    AMX_CHECK_LAUNCH();
    return 0;
}
