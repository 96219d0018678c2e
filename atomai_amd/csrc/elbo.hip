// elbo.hip — the VAE / rVAE evidence lower bound terms, forward and backward (HBM-bound, tiny).
//
//   reconstruction_loss('mse'): 0.5 * sum_pixels (x_rec - x)^2 per sample   atomai/losses_metrics/vi_losses.py:23-26
//   kld_normal: sum_d (-logsd + 0.5 sd^2 + 0.5 mu^2 - 0.5)                  vi_losses.py:40-57
//   kld_rot:    -logsd_phi + log(phi_prior) + sd_phi^2 / (2 phi_prior^2) - 0.5   vi_losses.py:77-84
//   rvae_loss: the rotation latent (index 0) gets kld_rot, ALL remaining latents (translation + content)
//   enter kld_normal (vi_losses.py:129-133); vae_loss: every latent enters kld_normal (vi_losses.py:105).
// One workgroup per sample; the three per-sample terms are written as [B] vectors (their means and the
// optional capacity term |KL - C| are B-element plumbing on the host side).
#include "amx_device.h"

__global__ __launch_bounds__(256) void elbo_terms_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ xrec,
                                                             const float* __restrict__ zmean,
                                                             const float* __restrict__ zlogsd, int n, int Z,
                                                             int rot, float phi_prior, float* __restrict__ recon,
                                                             float* __restrict__ klz, float* __restrict__ klrot) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float d = xrec[(size_t)b * n + i] - x[(size_t)b * n + i];
        acc = fmaf(d, d, acc);
    }
    red[tid] = acc; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
        recon[b] = 0.5f * red[0];
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
                                  int B, int n, int Z, int rot, float phi_prior, float* recon, float* klz,
                                  float* klrot, void* stream) {
    if (!x || !xrec || !zmean || !zlogsd || !recon || !klz || (rot && !klrot)) AMX_BADARG(1);
    if (B <= 0 || n <= 0 || Z <= 0 || (rot && phi_prior <= 0.f)) AMX_BADARG(2);
    AMX_LAUNCH(elbo_terms_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, xrec, zmean, zlogsd, n, Z,
               rot, phi_prior, recon, klz, klrot);
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
                                                             int rot, float phi_prior, float* __restrict__ dxrec,
                                                             float* __restrict__ dmean, float* __restrict__ dlogsd) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float gr = g_recon[b];
    for (int i = tid; i < n; i += 256)
        dxrec[(size_t)b * n + i] = gr * (xrec[(size_t)b * n + i] - x[(size_t)b * n + i]);
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
                                  int Z, int rot, float phi_prior, float* dxrec, float* dmean, float* dlogsd,
                                  void* stream) {
    if (!x || !xrec || !zmean || !zlogsd || !g_recon || !g_klz || !dxrec || !dmean || !dlogsd) AMX_BADARG(1);
    if (rot && !g_klrot) AMX_BADARG(2);
    if (B <= 0 || n <= 0 || Z <= 0 || Z > 256) AMX_BADARG(3);
    AMX_LAUNCH(elbo_terms_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, xrec, zmean, zlogsd,
               g_recon, g_klz, g_klrot, n, Z, rot, phi_prior, dxrec, dmean, dlogsd);
    AMX_CHECK_LAUNCH();
    return 0;
}
