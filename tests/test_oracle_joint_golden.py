"""CPU: oracle/joint_ref.py (joint 3-D + 2-D/512 pre-training iteration, per-frame loss feedback, schedules) against the epoch
the REAL reference loop ran (oracle/gen_golden_joint.py -> tests/golden/joint_small.npz)."""
import json
import os

import numpy as np
import torch

from oracle import joint_ref as J
from oracle import mae3d_ref as O


def load_joint(golden_dir):
    z = np.load(os.path.join(golden_dir, "joint_small.npz"))
    cfg = O.MAEConfig(**json.loads(str(z["cfg"])))
    P0 = O.init_params(cfg, seed=int(z["param_seed"]), bias_std=0.02)
    assert abs(sum(float(v.double().sum()) for v in P0.values()) - float(z["param_checksum"])) < 1e-9
    gd = torch.Generator().manual_seed(int(z["data_seed"]))
    vols = torch.rand(4, 2, 1, 6, 256, 256, generator=gd)
    imgs2d = torch.rand(4, 2, 1, 3, 512, 512, generator=gd)
    assert abs(float(vols.double().sum()) - float(z["vol_checksum"])) < 1e-5 and abs(float(imgs2d.double().sum()) - float(z["img2d_checksum"])) < 1e-5
    torch.manual_seed(int(z["noise_seed"]))                     # the reference drew rand(N, L) per forward in this order
    n3, n2 = [], []
    for _ in range(4):
        n3.append(torch.rand(2, 512)); n2.append(torch.rand(2, 1024))
    frames = [[[f"v{it}_{j}_f{nf}" for j in range(2)] for nf in range(6)] for it in range(4)]
    return z, cfg, P0, vols, imgs2d, n3, n2, frames


def test_joint_epoch_matches_reference(golden_dir):
    z, cfg, P0, vols, imgs2d, n3, n2, frames = load_joint(golden_dir)
    out = J.joint_epoch(P0, cfg, vols, imgs2d, n3, n2, frames, epoch=1, lr=1e-3, min_lr=1e-6, warmup_epochs=1, epochs=4, accum_iter=2,
                        mask_ratio=0.75, mask_ratio_2d=0.8, clip_grad=1.0)
    stats = json.loads(str(z["stats"]))
    assert abs(np.mean(out["loss"]) - stats["loss"]) <= 2e-5 * stats["loss"]
    assert abs(np.mean(out["loss_2d"]) - stats["loss_2d"]) <= 2e-5 * stats["loss_2d"]
    np.testing.assert_allclose(out["norms"], z["norms"], rtol=2e-4)
    ref_tab = json.loads(str(z["frame_dict"]))
    assert set(out["frame_table"]) == set(ref_tab)
    for k, e in ref_tab.items():
        assert abs(out["frame_table"][k]["mse_loss"] - e["mse_loss"]) <= 2e-5 * abs(e["mse_loss"]) + 1e-9, k
        assert out["frame_table"][k]["hardness"] == out["frame_table"][k]["mse_loss"]
    for k in z.files:
        if k.startswith("final/"):
            n = k[len("final/"):]
            mine = out["params"][n]
            mine = mine if mine.numel() <= 8192 else mine.flatten()[::7]
            assert float((mine.reshape(z[k].shape) - torch.from_numpy(z[k])).abs().max()) <= 2e-5, n


def test_schedules():
    from octcubem_amd import engine_pretrain as E
    for f_ref, f in ((J.K_scheduler, E.K_scheduler), (J.mask_ratio_2d_scheduler, E.mask_ratio_2d_scheduler)):
        for ep in (0, 5, 10, 11, 55, 100):
            for kw in ({}, {"all_epoch": 50, "warmup_epochs": 5}, {"epoch_offset": 3}):
                assert f(ep, **kw) == f_ref(ep, **kw)
    # hand-checked anchor points of main_pretrain_oph_joint_2d512_flash_attn.py:53-67
    assert E.K_scheduler(10) == 0.7 and abs(E.K_scheduler(100) - 0.3) < 1e-12 and abs(E.K_scheduler(55) - 0.5) < 1e-12
    assert E.mask_ratio_2d_scheduler(0) == 0.75 and abs(E.mask_ratio_2d_scheduler(100) - 0.85) < 1e-12


def test_record_frame_losses_host_logic():
    from octcubem_amd import engine_pretrain as E
    fl = torch.tensor([[0.1, 0.2], [0.3, 0.4]])
    frames = [[f"a{nf}", f"b{nf}"] for nf in range(6)]
    tab = {n: {} for row in frames for n in row}
    E.record_frame_losses(fl, {"frames": frames}, tab)
    ref = {}
    J.record_frame_losses(fl, frames, ref)
    assert {k: v for k, v in tab.items()} == ref
    assert abs(tab["a0"]["mse_loss"] - 0.1) < 1e-7 and abs(tab["a5"]["hardness"] - 0.2) < 1e-7 and abs(tab["b3"]["mse_loss"] - 0.4) < 1e-7
