"""Child process of tests/test_nccl_gpu.py: a ONE-rank `nccl` (= RCCL) process group on the leased MI355X.

Everything the N-GPU run does except talk to a peer: init_process_group("nccl"), the broadcast of the flat parameter
buffer and the BatchNorm buffers, the sum all-reduce of the flat CUDA gradient bucket on RCCL's own stream after a
backward whose weight gradients were written on the tape's side stream, 1/world folded into the fused Adam, rank-0 save.
With one rank the all-reduce is the identity, so the run must be BIT-identical to the same run without the wrapper.
Prints one JSON line."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import atomai_amd as aoi
    from atomai_amd import _lib
    from atomai_amd.engine import Tape
    from atomai_amd.parallel import DataParallelGrads, init_distributed
    import torch.distributed as dist
    _lib.load()
    rank, world, local = init_distributed(force=True)
    out = {"backend": dist.get_backend(), "world": world, "side_stream": bool(Tape.use_side_stream)}
    assert dist.is_initialized() and dist.get_backend() == "nccl"

    # ---- default U-Net (nb_filters 16, nb_classes 3), 3 steps, with / without the wrapper
    rs = np.random.RandomState(0)
    X = rs.rand(8, 128, 128).astype(np.float32)
    y = rs.randint(0, 3, (8, 128, 128))

    def unet_run(use_dp):
        m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
        m.compile_trainer((X, y, X[:4], y[:4]), training_cycles=3, batch_size=4, plot_training_history=False)
        if use_dp:
            m.dp = DataParallelGrads(m.optimizer, m.net)
        losses = [m.train_step(m.X_train[i % 2], m.y_train[i % 2])[0] for i in range(3)]
        torch.cuda.synchronize()
        return losses, {k: v.detach().cpu().numpy().copy() for k, v in m.net.state_dict().items()}

    l0, s0 = unet_run(False)
    l1, s1 = unet_run(True)
    out["unet_losses"] = [l0, l1]
    out["unet_bit_identical"] = bool(l0 == l1 and all(np.array_equal(s0[k], s1[k]) for k in s0))

    # ---- rVAE, 3 steps (eps from the seeded device generator: identical in both runs)
    xv = rs.rand(64, 32, 32).astype(np.float32)

    def rvae_run(use_dp):
        m = aoi.models.rVAE((32, 32), latent_dim=2, seed=0)
        m.dx_prior, m.kdict_["phi_prior"] = 0.1, 0.1
        m.compile_trainer((xv, None), None, batch_size=64)
        if use_dp:
            m.dp = DataParallelGrads(m.optim)
        torch.manual_seed(7)
        xt = torch.from_numpy(xv).cuda()
        elbos = []
        for _ in range(3):
            m.encoder_net.train(), m.decoder_net.train()
            m.optim.zero_grad()
            elbo = m.forward_compute_elbo(xt)
            (-elbo).backward()
            if m.dp is not None:
                m.dp.allreduce_grads()
            m.optim.step()
            elbos.append(elbo.item())
        sd = {"e" + k: v.detach().cpu().numpy().copy() for k, v in m.encoder_net.state_dict().items()}
        sd.update({"d" + k: v.detach().cpu().numpy().copy() for k, v in m.decoder_net.state_dict().items()})
        return elbos, sd

    e0, v0 = rvae_run(False)
    e1, v1 = rvae_run(True)
    out["rvae_elbos"] = [e0, e1]
    out["rvae_bit_identical"] = bool(e0 == e1 and all(np.array_equal(v0[k], v1[k]) for k in v0))

    # ---- the entry point: fit(..., distributed=True) on the 1-rank group, against plain fit()
    with tempfile.TemporaryDirectory() as tmp:
        def fit_run(distributed):
            m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
            m.fit(X, y, X[:4], y[:4], training_cycles=4, batch_size=4, plot_training_history=False,
                  distributed=distributed, filename=os.path.join(tmp, f"m{int(distributed)}"))
            return m
        a, b = fit_run(False), fit_run(True)
        out["fit_dp_attached"] = b.dp is not None and a.dp is None
        out["fit_losses"] = [a.loss_acc["train_loss"], b.loss_acc["train_loss"]]
        out["fit_bit_identical"] = bool(a.loss_acc["train_loss"] == b.loss_acc["train_loss"] and all(
            torch.equal(p, q) for p, q in zip(a.net.state_dict().values(), b.net.state_dict().values())))
        out["fit_saved"] = os.path.exists(os.path.join(tmp, "m1_metadict_final.tar"))
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
