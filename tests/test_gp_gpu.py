"""`gpu` tier for the DKL covariance path."""
import numpy as np
import pytest
import torch

import _gp_checks as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind", ["rbf", "matern"])
def test_kernel_matrix_and_matvec(dtype, kind):
    G.check_kernel_matrix("cuda", dtype, kind, N=700, M=450, D=3)
    G.check_kernel_matrix("cuda", dtype, kind, N=333, M=1260, D=8)
    G.check_kernel_matrix("cuda", dtype, kind, N=257, M=515, D=13)


@pytest.mark.parametrize("kind", ["rbf", "matern"])
def test_kernel_backward_and_mll(kind):
    G.check_mll_and_grads("cuda", kind, N=600, D=2)


def test_dklgpr_api():
    G.check_dklgpr_api()


def test_feature_extractor_vs_reference_golden():
    G.check_extractor_golden("cuda")


def test_config5_covariance_properties():
    """BASELINE.json configs[4] size (N = 16384 embedded points, RBF): size-independent properties of the tiled
    builder — symmetry, unit-scaled diagonal, agreement of K @ v with the matrix-free mat-vec, row checksums vs
    the fp64 build."""
    from atomai_amd.nets.gp import kernel_matrix, kernel_matvec
    N, D = 16384, 2
    rs = np.random.RandomState(0)
    Z = torch.from_numpy(rs.uniform(-1, 1, (N, D)).astype(np.float32)).cuda()
    ls = torch.full((D,), float(np.log(2.0)), device="cuda")            # softplus(0)
    s2 = float(np.log(2.0))
    K = kernel_matrix(Z, Z, ls, s2, 0)
    assert K.shape == (N, N)
    assert float((K - K.T).abs().max()) == 0.0
    assert float((torch.diagonal(K) - s2).abs().max()) < 1e-6
    v = torch.from_numpy(rs.randn(N, 2).astype(np.float32)).cuda()
    y1, y2 = K @ v, kernel_matvec(Z, Z, ls, s2, v, 0)
    assert float((y1 - y2).abs().max() / y1.abs().max()) < 1e-4
    K64 = kernel_matrix(Z.double(), Z.double(), ls.double(), s2, 0)
    assert float((K.double().sum(1) - K64.sum(1)).abs().max() / K64.sum(1).abs().max()) < 1e-5


def test_gpytorch_known_answer_vectors():
    G.check_gpytorch_known_answers_kernel("cuda")
    G.check_scale_to_bounds_module("cuda")


def test_posterior_is_factorised_once_per_model_state():
    G.check_posterior_cache("cuda")


def test_conv_feature_extractor_vs_stock_torch():
    G.check_conv_feature_extractor("cuda")
    G.check_conv_feature_extractor("cuda", N=300, p=16, nf=16)


def test_config5_dklgpr_conv_extractor_n16384():
    """BASELINE.json configs[4] end to end: dklGPR with the conv feature extractor on N = 16384 patches of 16x16,
    RBF kernel, the reference's KISS-GP layer (grid_size 50: 2500 grid nodes), fp32."""
    m = G.check_dklgpr_conv_extractor("cuda", N=16384, p=16, cycles=3, precision="single")
    assert m.gp_model.train_inputs[0].shape == (16384, 256)
    assert m.gp_model.gp == "kissgp" and m.gp_model.grid.m == 2500


def test_config5_exact_gp_n16384():
    """The same with gp='exact': the dense tiled covariance of all 16384 points + Cholesky (fp32)."""
    m = G.check_dklgpr_conv_extractor("cuda", N=16384, p=16, cycles=2, precision="single", gp="exact")
    assert m.gp_model.gp == "exact"


@pytest.mark.parametrize("kind", ["rbf", "matern"])
@pytest.mark.parametrize("D,gs", [(1, 30), (2, 16)])
def test_kiss_gp_mll_and_gradients_equal_the_dense_evaluation(kind, D, gs):
    G.check_ski_mll_and_grads("cuda", kind, N=500, D=D, G=gs)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_kiss_gp_gram_kernels(dtype):
    G.check_ski_gram_is_deterministic_and_ragged("cuda", dtype)


@pytest.mark.parametrize("precision", ["double", "single"])
def test_kiss_gp_posterior_vs_oracle(precision):
    G.check_ski_posterior("cuda", precision)


def test_kiss_gp_fallbacks():
    G.check_ski_fallbacks()


def test_kiss_gp_kronecker_core_equals_lu_core():
    r = G.check_ski_kron_core_equals_lu_core("cuda", N=2000, G=50)
    assert r < 1000


def test_kiss_gp_matches_exact_gp_at_config5_scale():
    """N = 16384 points, grid 50 x 50, fp64: the KISS-GP marginal log likelihood (m x m algebra) against the exact dense GP's
    at the same hyper-parameters — the two models differ by the interpolation error only (oracle: < 1e-4 of the output scale
    at this lengthscale), and two evaluations are bit-identical."""
    from atomai_amd.nets.gp import SkiGrid, _ExactMLLFn, _SkiMLLFn
    N = 16384
    rs = np.random.RandomState(0)
    Z = torch.from_numpy(rs.uniform(-0.95, 0.95, (N, 2))).cuda()
    y = torch.from_numpy(np.sin(3 * Z[:, 0].cpu().numpy()) + 0.1 * rs.randn(N)).cuda()
    ls = torch.full((1, 1, 2), 0.6931, dtype=torch.float64, device="cuda")
    s2 = torch.tensor([0.6931], dtype=torch.float64, device="cuda")
    nz = torch.tensor([0.05], dtype=torch.float64, device="cuda")
    mu = torch.tensor([0.0], dtype=torch.float64, device="cuda")
    grid = SkiGrid(2, 50)
    grid.update(Z)
    a = _SkiMLLFn.apply(Z, y[None], ls, s2, nz, mu, 0, grid)
    b = _SkiMLLFn.apply(Z, y[None], ls, s2, nz, mu, 0, grid)
    e = _ExactMLLFn.apply(Z, y, ls[0], s2[0], nz[0], mu[0], 0)
    assert a.item() == b.item()
    assert abs(a.item() - e.item()) < 1e-3 * abs(e.item()), (a.item(), e.item())
