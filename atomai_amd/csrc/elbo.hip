// elbo.hip — the VAE / rVAE evidence lower bound terms, forward and backward (HBM-bound, tiny).
//
//   reconstruction_loss('mse'): 0.5 * sum_pixels (x_rec - x)^2 per sample   atomai/losses_metrics/vi_losses.py:23-26
//   reconstruction_loss('ce'):  sum of binary_cross_entropy_with_logits(x_rec, x) = max(v,0) - v t + log1p(exp(-|v|)),
//       per sample for a 2-D in_dim; for a 3-D in_dim the reference sums over the channels only and its .mean() then
//       runs over samples x pixels — i.e. the per-sample sum times 1 / (H W): `recon_scale`     vi_losses.py:27-34
//   kld_normal: sum_d (-logsd + 0.5 sd^2 + 0.5 mu^2 - 0.5)                  vi_losses.py:40-57
//   kld_rot:    -logsd_phi + log(phi_prior) + sd_phi^2 / (2 phi_prior^2) - 0.5   vi_losses.py:77-84
//   rvae_loss: the rotation latent (index 0) gets kld_rot, ALL remaining latents (translation + content)
//   enter kld_normal (vi_losses.py:129-133); vae_loss: every latent enters kld_normal (vi_losses.py:105).
// One workgroup per sample; the three per-sample terms are written as [B] vectors (their means and the
// optional capacity term |KL - C| are B-element plumbing on the host side).
#include "amx_device.h"

// recon_kind: 0 = 'mse', 1 = 'ce' (logits in xrec).  Term and derivative of one element:
static __device__ __forceinline__ float elbo_recon_term(int kind, float v, float t) {
    if (kind == 0) { const float d = v - t; return d * d; }                       // (0.5 applied once to the sum)
    return fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
}
static __device__ __forceinline__ float elbo_recon_grad(int kind, float v, float t) {
    if (kind == 0) return v - t;
    const float e = expf(-fabsf(v));
    return (v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e)) - t;                      // sigmoid(v) - t
}

__global__ __launch_bounds__(256) void elbo_terms_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ xrec,
                                                             const float* __restrict__ zmean,
                                                             const float* __restrict__ zlogsd, int n, int Z,
                                                             int rot, float phi_prior, int recon_kind,
                                                             float recon_scale, float* __restrict__ recon,
                                                             float* __restrict__ klz, float* __restrict__ klrot) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    if (recon_kind == 0) {
        for (int i = tid; i < n; i += 256) {
            const float d = xrec[(size_t)b * n + i] - x[(size_t)b * n + i];
            acc = fmaf(d, d, acc);
        }
    } else {
        for (int i = tid; i < n; i += 256) acc += elbo_recon_term(1, xrec[(size_t)b * n + i], x[(size_t)b * n + i]);
    }
    red[tid] = acc; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
        recon[b] = recon_kind == 0 ? 0.5f * red[0] : recon_scale * red[0];
        float kz = 0.f;
        for (int d = rot ? 1 : 0; d < Z; ++d) {
            const float ls = zlogsd[(size_t)b * Z + d], mu = zmean[(size_t)b * Z + d];
            const float sd = expf(ls);
            kz += -ls + 0.5f * sd * sd + 0.5f * mu * mu - 0.5f;
        }
        klz[b] = kz;
        if (rot) {
            const float ls = zlogsd[(size_t)b * Z];
            const float sd = expf(ls);
            klrot[b] = -ls + logf(phi_prior) + sd * sd / (2.f * phi_prior * phi_prior) - 0.5f;
        } else if (klrot) klrot[b] = 0.f;
    }
}

extern "C" int amx_elbo_terms_fwd(const float* x, const float* xrec, const float* zmean, const float* zlogsd,
                                  int B, int n, int Z, int rot, float phi_prior, int recon_kind, float recon_scale,
                                  float* recon, float* klz, float* klrot, void* stream) {
    if (!x || !xrec || !zmean || !zlogsd || !recon || !klz || (rot && !klrot)) AMX_BADARG(1);
    if (B <= 0 || n <= 0 || Z <= 0 || (rot && phi_prior <= 0.f)) AMX_BADARG(2);
    if (recon_kind < 0 || recon_kind > 1 || !(recon_scale > 0.f)) AMX_BADARG(3);
    AMX_LAUNCH(elbo_terms_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, xrec, zmean, zlogsd, n, Z,
               rot, phi_prior, recon_kind, recon_scale, recon, klz, klrot);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Backward: given d loss / d recon[b], d/d klz[b], d/d klrot[b]:
//   dxrec[b][i] = g_recon[b] * (xrec - x);  dmean[b][d] = g_klz[b] * mu (d in KL_z);
//   dlogsd[b][d] = g_klz[b] * (-1 + sd^2);  dlogsd[b][0] (rot) = g_klrot[b] * (-1 + sd^2 / phi_prior^2)
__global__ __launch_bounds__(256) void elbo_terms_bwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ xrec,
                                                             const float* __restrict__ zmean,
                                                             const float* __restrict__ zlogsd,
                                                             const float* __restrict__ g_recon,
                                                             const float* __restrict__ g_klz,
                                                             const float* __restrict__ g_klrot, int n, int Z,
                                                             int rot, float phi_prior, int recon_kind,
                                                             float recon_scale, float* __restrict__ dxrec,
                                                             float* __restrict__ dmean, float* __restrict__ dlogsd) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float gr = recon_kind == 0 ? g_recon[b] : g_recon[b] * recon_scale;
    if (recon_kind == 0) {
        for (int i = tid; i < n; i += 256)
            dxrec[(size_t)b * n + i] = gr * (xrec[(size_t)b * n + i] - x[(size_t)b * n + i]);
    } else {
        for (int i = tid; i < n; i += 256)
            dxrec[(size_t)b * n + i] = gr * elbo_recon_grad(1, xrec[(size_t)b * n + i], x[(size_t)b * n + i]);
    }
    if (tid < Z) {
        const float ls = zlogsd[(size_t)b * Z + tid], mu = zmean[(size_t)b * Z + tid];
        const float sd2 = expf(2.f * ls);
        float dm, dl;
        if (rot && tid == 0) { dm = 0.f; dl = g_klrot[b] * (-1.f + sd2 / (phi_prior * phi_prior)); }
        else { dm = g_klz[b] * mu; dl = g_klz[b] * (-1.f + sd2); }
        dmean[(size_t)b * Z + tid] = dm;
        dlogsd[(size_t)b * Z + tid] = dl;
    }
}

extern "C" int amx_elbo_terms_bwd(const float* x, const float* xrec, const float* zmean, const float* zlogsd,
                                  const float* g_recon, const float* g_klz, const float* g_klrot, int B, int n,
                                  int Z, int rot, float phi_prior, int recon_kind, float recon_scale, float* dxrec,
                                  float* dmean, float* dlogsd, void* stream) {
    if (!x || !xrec || !zmean || !zlogsd || !g_recon || !g_klz || !dxrec || !dmean || !dlogsd) AMX_BADARG(1);
    if (rot && !g_klrot) AMX_BADARG(2);
    if (B <= 0 || n <= 0 || Z <= 0 || Z > 256) AMX_BADARG(3);
    if (recon_kind < 0 || recon_kind > 1 || !(recon_scale > 0.f)) AMX_BADARG(4);
    AMX_LAUNCH(elbo_terms_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, xrec, zmean, zlogsd,
               g_recon, g_klz, g_klrot, n, Z, rot, phi_prior, recon_kind, recon_scale, dxrec, dmean, dlogsd);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ fused scalar ELBO (no capacity term)
// vae_loss / rvae_loss without `capacity` reduce the three per-sample terms to ONE number:
//   ELBO = -mean(recon) - mean(klz) - mean(klrot)                                   vi_losses.py:105-108, 129-137
// and its gradient w.r.t. every per-sample term is the same scalar -g / B.  Formed by ATen that was 3 mean kernels + 3
// elementwise kernels forward and ~9 more in backward, each ~5 us, on a 5.7 ms step.  amx_elbo_combine: one block,
// fixed-order fp64 sum.  amx_elbo_bwd_scalar: the backward kernel above with g_recon = g_klz = g_klrot = coef * g[0],
// g read from device memory (the upstream gradient is a device scalar; no host round trip).
__global__ __launch_bounds__(256) void elbo_combine_kernel(const float* __restrict__ recon, const float* __restrict__ klz,
                                                           const float* __restrict__ klrot, int B, float* __restrict__ out) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double a = 0.0;
    for (int b = tid; b < B; b += 256) a += (double)recon[b] + (double)klz[b] + (klrot ? (double)klrot[b] : 0.0);
    red[tid] = a; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) out[0] = (float)(-red[0] / (double)B);
}

extern "C" int amx_elbo_combine(const float* recon, const float* klz, const float* klrot, int B, float* out, void* stream) {
    if (!recon || !klz || !out || B <= 0) AMX_BADARG(1);
    AMX_LAUNCH(elbo_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, recon, klz, klrot, B, out);
    AMX_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void elbo_bwd_scalar_kernel(const float* __restrict__ x, const float* __restrict__ xrec,
                                                              const float* __restrict__ zmean,
                                                              const float* __restrict__ zlogsd,
                                                              const float* __restrict__ gscalar, float coef, int n, int Z,
                                                              int rot, float phi_prior, int recon_kind, float recon_scale,
                                                              float* __restrict__ dxrec,
                                                              float* __restrict__ dmean, float* __restrict__ dlogsd) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float gr = gscalar[0] * coef;
    if (recon_kind == 0) {
        for (int i = tid; i < n; i += 256)
            dxrec[(size_t)b * n + i] = gr * (xrec[(size_t)b * n + i] - x[(size_t)b * n + i]);
    } else {
        const float gc = gr * recon_scale;
        for (int i = tid; i < n; i += 256)
            dxrec[(size_t)b * n + i] = gc * elbo_recon_grad(1, xrec[(size_t)b * n + i], x[(size_t)b * n + i]);
    }
    if (tid < Z) {
        const float ls = zlogsd[(size_t)b * Z + tid], mu = zmean[(size_t)b * Z + tid];
        const float sd2 = expf(2.f * ls);
        float dm, dl;
        if (rot && tid == 0) { dm = 0.f; dl = gr * (-1.f + sd2 / (phi_prior * phi_prior)); }
        else { dm = gr * mu; dl = gr * (-1.f + sd2); }
        dmean[(size_t)b * Z + tid] = dm;
        dlogsd[(size_t)b * Z + tid] = dl;
    }
}

extern "C" int amx_elbo_bwd_scalar(const float* x, const float* xrec, const float* zmean, const float* zlogsd,
                                   const float* gscalar, float coef, int B, int n, int Z, int rot, float phi_prior,
                                   int recon_kind, float recon_scale, float* dxrec, float* dmean, float* dlogsd,
                                   void* stream) {
    if (!x || !xrec || !zmean || !zlogsd || !gscalar || !dxrec || !dmean || !dlogsd) AMX_BADARG(1);
    if (B <= 0 || n <= 0 || Z <= 0 || Z > 256 || (rot && phi_prior <= 0.f)) AMX_BADARG(2);
    if (recon_kind < 0 || recon_kind > 1 || !(recon_scale > 0.f)) AMX_BADARG(3);
    AMX_LAUNCH(elbo_bwd_scalar_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, xrec, zmean, zlogsd, gscalar, coef,
               n, Z, rot, phi_prior, recon_kind, recon_scale, dxrec, dmean, dlogsd);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ rVAE latent plumbing in one pass
// rvae.py:118-137: z = mean + exp(logsd) * eps;  phi = z[:, 0];  dx = z[:, 1:3] * dx_prior (translation);  content = the rest.
// Outputs theta [B][3] = (phi, dx, dy) (zeros without translation) and zc [B][Z - skip] (skip = 3 or 1).  ATen forms this
// with ~6 small kernels forward and, through the slice / cat backward nodes, ~10 more in backward.
__global__ void rvae_latent_fwd_kernel(const float* __restrict__ zmean, const float* __restrict__ zlogsd,
                                       const float* __restrict__ eps, int B, int Z, int translation, float dx_prior,
                                       float* __restrict__ theta, float* __restrict__ zc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Z) return;
    const int b = i / Z, d = i - b * Z;
    const float z = fmaf(expf(zlogsd[i]), eps[i], zmean[i]);
    const int skip = translation ? 3 : 1;
    if (d == 0) theta[b * 3] = z;
    else if (translation && d < 3) theta[b * 3 + d] = z * dx_prior;
    else zc[b * (Z - skip) + d - skip] = z;
    if (!translation && d == 0) { theta[b * 3 + 1] = 0.f; theta[b * 3 + 2] = 0.f; }
}

extern "C" int amx_rvae_latent_fwd(const float* zmean, const float* zlogsd, const float* eps, int B, int Z,
                                   int translation, float dx_prior, float* theta, float* zc, void* stream) {
    if (!zmean || !zlogsd || !eps || !theta || B <= 0 || Z < (translation ? 3 : 1)) AMX_BADARG(1);
    if (Z > (translation ? 3 : 1) && !zc) AMX_BADARG(2);
    AMX_LAUNCH(rvae_latent_fwd_kernel, dim3((B * Z + 255) / 256), dim3(256), 0, (hipStream_t)stream, zmean, zlogsd, eps,
               B, Z, translation, dx_prior, theta, zc);
    AMX_CHECK_LAUNCH();
    return 0;
}

// dz[b][0] = dtheta[b][0]; dz[b][1:3] = dtheta[b][1:3] * dx_prior; dz[b][skip:] = dzc;  dmean = dz; dlogsd = dz * eps * exp(logsd)
__global__ void rvae_latent_bwd_kernel(const float* __restrict__ zlogsd, const float* __restrict__ eps,
                                       const float* __restrict__ dtheta, const float* __restrict__ dzc, int B, int Z,
                                       int translation, float dx_prior, float* __restrict__ dmean,
                                       float* __restrict__ dlogsd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Z) return;
    const int b = i / Z, d = i - b * Z;
    const int skip = translation ? 3 : 1;
    float dz;
    if (d == 0) dz = dtheta ? dtheta[b * 3] : 0.f;
    else if (translation && d < 3) dz = dtheta ? dtheta[b * 3 + d] * dx_prior : 0.f;
    else dz = dzc ? dzc[b * (Z - skip) + d - skip] : 0.f;
    dmean[i] = dz;
    dlogsd[i] = dz * eps[i] * expf(zlogsd[i]);
}

extern "C" int amx_rvae_latent_bwd(const float* zlogsd, const float* eps, const float* dtheta, const float* dzc, int B,
                                   int Z, int translation, float dx_prior, float* dmean, float* dlogsd, void* stream) {
    if (!zlogsd || !eps || !dmean || !dlogsd || B <= 0 || Z < (translation ? 3 : 1)) AMX_BADARG(1);
    AMX_LAUNCH(rvae_latent_bwd_kernel, dim3((B * Z + 255) / 256), dim3(256), 0, (hipStream_t)stream, zlogsd, eps, dtheta,
               dzc, B, Z, translation, dx_prior, dmean, dlogsd);
    AMX_CHECK_LAUNCH();
    return 0;
}
