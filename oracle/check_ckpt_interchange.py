"""Dev-container check (needs /root/reference; never runs on the GPU box): a checkpoint written by
atomai_amd's trainers is loaded by the REFERENCE's own load_model (atomai/models/loaders.py:25-64) and the
reference's prediction from it is compared with atomai_amd's.  Test infrastructure only.

    python oracle/check_ckpt_interchange.py
"""
import os
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    import ref_harness
    ref = ref_harness.import_reference()
    if not torch.cuda.is_available():
        import emu_backend
        emu_backend.use_emulator()
    import atomai_amd as amd
    warnings.simplefilter("ignore")
    rs = np.random.RandomState(7)
    X, Xt = rs.rand(6, 16, 16).astype(np.float32), rs.rand(4, 16, 16).astype(np.float32)
    y, yt = rs.randint(0, 3, (6, 16, 16)), rs.randint(0, 3, (4, 16, 16))
    Xp = rs.rand(2, 16, 16).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        seg = amd.models.Segmentor("Unet", nb_classes=3, nb_filters=4)
        seg.fit(X, y, Xt, yt, training_cycles=3, batch_size=2, filename=os.path.join(d, "m"),
                plot_training_history=False)
        mine = seg.predict(Xp, compute_coords=False)
        theirs_model = ref.models.load_model(os.path.join(d, "m_metadict_final.tar"))
        theirs = theirs_model.predict(Xp, compute_coords=False)
        err = np.abs(mine - theirs).max()
        print("Segmentor: reference.load_model(atomai_amd checkpoint) max |dprob| =", err)
        assert err < 1e-5
        v = amd.models.rVAE((16, 16), latent_dim=2, seed=0, numhidden_encoder=32, numhidden_decoder=32)
        Xv = rs.rand(8, 16, 16).astype(np.float32)
        v.fit(Xv, training_cycles=2, batch_size=4, filename=os.path.join(d, "v"))
        lv = ref.models.load_model(os.path.join(d, "v.tar"))
        z1, _ = v.encode(Xv)
        z2, _ = lv.encode(Xv)
        d1 = v.decode(np.array([[0.3, -0.2]], dtype=np.float32))
        d2 = lv.decode(np.array([[0.3, -0.2]], dtype=np.float32))
        print("rVAE: max |dz| =", np.abs(z1 - z2).max(), " max |ddecode| =", np.abs(d1 - d2).max())
        assert np.abs(z1 - z2).max() < 1e-5 and np.abs(d1 - d2).max() < 1e-5
    print("checkpoint interchange OK (atomai_amd -> reference)")


if __name__ == "__main__":
    main()
