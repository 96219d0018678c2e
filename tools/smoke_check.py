"""smoke(): one tiny forward + backward + Adam step of the Segmentor hot path on cuda:0, checked against
the CPU oracle (oracle/seg_oracle.py is the checker here, never the thing executed as the product).
Lives outside the package on purpose: nothing under atomai_amd/ imports the oracle."""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch


def run() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs an MI355X (cuda:0); there is no CPU fallback")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tools/ -> repo root
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import seg_oracle as so
    import atomai_amd as aoi
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    net, _ = aoi.nets.init_fcnn_model("Unet", 3, nb_filters=8)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    for seed in range(20):      # inputs that stay clear of the LeakyReLU kink (oracle.KINK_PROBE)
        rs = np.random.RandomState(seed)
        x = torch.from_numpy(rs.rand(2, 1, 64, 64).astype(np.float32))
        y = torch.from_numpy(rs.randint(0, 3, (2, 64, 64)))
        if so.min_abs_preactivation("Unet", sd, x.double()) > 2e-5:
            break
    net.to(dev).train()
    opt = aoi.FusedAdam(net.parameters(), lr=1e-3)
    opt.prepare()
    crit = aoi.losses_metrics.select_loss("ce", 3)
    logits = net(x.to(dev))
    loss = crit(logits, y.to(dev))
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    ref_loss, ref_logits, ref_grads = so.loss_and_grads("Unet", so.cast(sd, torch.float64), x.double(), y, 3)
    e_log = float((logits.detach().cpu().double() - ref_logits).abs().max() / ref_logits.abs().max())
    e_loss = abs(loss.item() - float(ref_loss)) / abs(float(ref_loss))
    gmax = max(float(g.abs().max()) for g in ref_grads.values())
    e_g = max(float((p.grad.cpu().double() - ref_grads[k]).abs().max()) / gmax
              for k, p in net.named_parameters())
    print(f"smoke: logits rel err {e_log:.2e}, loss rel err {e_loss:.2e}, grad err/gmax {e_g:.2e}")
    assert e_log < 1e-4 and e_loss < 1e-5 and e_g < 1e-4, "HIP path disagrees with the oracle"
