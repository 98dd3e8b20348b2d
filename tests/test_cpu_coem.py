"""CPU: octcubem_amd.coem.ClipLoss / gather_features against the reference's own ClipLoss (oracle/gen_golden_coem.py ->
tests/golden/coem_loss.npz): world size 1 (plain and corrected labels) and world size 2 over gloo for every
(local_loss, gather_with_grad) combination -- loss, feature gradients and the temperature gradient of each rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def feats(seed, n, d=32, dup=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    b = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    if dup:
        b[2] = b[0]
    return a, b


def run(loss_mod, a, b):
    a = a.clone().requires_grad_(True); b = b.clone().requires_grad_(True)
    ls = torch.tensor(np.log(1 / 0.07), dtype=torch.float32, requires_grad=True)
    loss = loss_mod(a, b, ls.exp())
    loss.backward()
    return loss.detach().numpy(), a.grad.numpy(), b.grad.numpy(), ls.grad.numpy()


def check(z, tag, got):
    for k, v in zip(("loss", "ga", "gb", "gls"), got):
        np.testing.assert_allclose(v, z[f"{tag}/{k}"], rtol=2e-5, atol=1e-7, err_msg=f"{tag}/{k}")


def test_clip_loss_world1(golden_dir):
    from octcubem_amd.coem import ClipLoss
    z = np.load(os.path.join(golden_dir, "coem_loss.npz"))
    check(z, "w1/plain", run(ClipLoss(), *feats(7, 6)))
    check(z, "w1/corrected", run(ClipLoss(correct_label=1), *feats(7, 6, dup=True)))
    cached = ClipLoss(cache_labels=True)
    check(z, "w1/plain", run(cached, *feats(7, 6)))
    check(z, "w1/plain", run(cached, *feats(7, 6)))             # second call takes the cached labels


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octcubem_amd.coem import ClipLoss
    out = {}
    for ll in (False, True):
        for gg in (False, True):
            got = run(ClipLoss(local_loss=ll, gather_with_grad=gg, rank=rank, world_size=world), *feats(100 + rank, 3))
            for k, v in zip(("loss", "ga", "gb", "gls"), got):
                out[f"w2/ll{int(ll)}_gg{int(gg)}/r{rank}/{k}"] = np.array(v)
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_clip_loss_gloo_world2(golden_dir):
    z = np.load(os.path.join(golden_dir, "coem_loss.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        got.update(q.get(timeout=150))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == 32
    for k, v in got.items():
        np.testing.assert_allclose(v, z[k], rtol=2e-5, atol=1e-7, err_msg=k)


# ---------------------------------------------------------------------------------------------- three modalities
def feats3(seed, n, d=32):
    g = torch.Generator().manual_seed(seed)
    f = [torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1) for _ in range(3)]
    w1 = (torch.rand(n, generator=g) > 0.3).float(); w2 = (torch.rand(n, generator=g) > 0.4).float()
    return f[0], f[1], f[2], w1, w2


def run3(loss_mod, seed, n, w_override=None):
    a, b, c, w1, w2 = feats3(seed, n)
    if w_override is not None:
        w1, w2 = w_override(w1, w2)
    for t in (a, b, c):
        t.requires_grad_(True)
    ls = [torch.tensor(np.log(1 / t_), dtype=torch.float32, requires_grad=True) for t_ in (0.07, 0.05, 0.1)]
    loss = loss_mod(a, b, c, ls[0].exp(), ls[1].exp(), ls[2].exp(), w1, w2)
    loss.backward()
    z = lambda t: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
    return {"loss": loss.detach().numpy(), "ga": z(a), "gb": z(b), "gc": z(c), "gls": np.array([float(z(l)) for l in ls])}


def test_three_modality_clip_loss_world1(golden_dir):
    """ThreeModalityClipLoss (retinal-COEM/src/open_clip/loss.py:230-385) against the reference's own: every sample present,
    some missing, one modality missing entirely (its four terms are 0)."""
    from octcubem_amd.coem import ThreeModalityClipLoss
    z = np.load(os.path.join(golden_dir, "coem_loss.npz"))
    cases = {"all": lambda w1, w2: (torch.ones_like(w1), torch.ones_like(w2)), "some": None,
             "none2": lambda w1, w2: (w1, torch.zeros_like(w2))}
    for name, ov in cases.items():
        got = run3(ThreeModalityClipLoss(), 55, 7, ov)
        for k, v in got.items():
            np.testing.assert_allclose(v, z[f"w1m3/{name}/{k}"], rtol=2e-5, atol=1e-7, err_msg=f"{name}/{k}")


def _worker3(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octcubem_amd.coem import ThreeModalityClipLoss
    out = {}
    for ll, gg in ((False, False), (False, True)):
        got = run3(ThreeModalityClipLoss(local_loss=ll, gather_with_grad=gg, rank=rank, world_size=world), 300 + rank, 4)
        for k, v in got.items():
            out[f"w2m3/ll{int(ll)}_gg{int(gg)}/r{rank}/{k}"] = np.array(v)
    try:                                        # as in the reference: labels from the GLOBAL count do not fit local logits
        run3(ThreeModalityClipLoss(local_loss=True, gather_with_grad=True, rank=rank, world_size=world), 300 + rank, 4)
        out["local_loss_raises"] = np.array(0)
    except ValueError:
        out["local_loss_raises"] = np.array(1)
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_three_modality_clip_loss_gloo_world2(golden_dir):
    z = np.load(os.path.join(golden_dir, "coem_loss.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        got.update(q.get(timeout=150))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert int(got.pop("local_loss_raises")) == 1
    assert len(got) == 20
    for k, v in got.items():
        np.testing.assert_allclose(v, z[k], rtol=2e-5, atol=1e-7, err_msg=k)


def test_towers_from_the_shipped_model_config():
    """coem.create_model_from_config on the (width/depth-reduced) structure of the reference's
    model_configs/vit_large_patch16_retFound-vit_large_patch16_OCTCube.json: both towers project to embed_dim, the OCT tower is
    the no-dropout ST ViT with the flash path's semantics, parameter names are the reference's."""
    import json
    from octcubem_amd import coem
    cfg = {"embed_dim": 32,
           "vision_cfg": {"image_size": 64, "layers": 2, "width": 64, "patch_size": 16, "num_heads": 2, "t_patch_size": 3, "in_chans": 1,
                          "mlp_ratio": 4, "norm_layer_eps": 1e-6, "drop_path_rate": 0.2, "use_flash_attn": True, "attn_drop_rate": 0.0,
                          "drop_rate": 0.0, "global_pool": True, "model_name": "ViT_ST_nodrop", "model_ckpt": "no/such/OCTCube.pth",
                          "num_frames": 12},
           "text_cfg": {"image_size": 64, "layers": 2, "width": 64, "patch_size": 16, "num_heads": 2, "in_chans": 3, "mlp_ratio": 4,
                        "norm_layer_eps": 1e-6, "drop_path_rate": 0.2, "use_flash_attn": True, "dropout": 0.5, "global_pool": True,
                        "vit_model_name": "ViT_flash_attn", "model_ckpt": ""}}
    m = coem.create_model_from_config(json.loads(json.dumps(cfg)))
    assert isinstance(m, coem.CustomTextCLIP)
    assert m.visual.head.weight.shape == (32, 64) and m.text.head.weight.shape == (32, 64)
    assert m.visual.flash_compat and m.visual.global_pool and m.visual.patch_embed.t_grid_size == 4
    assert abs(float(m.logit_scale) - np.log(1 / 0.07)) < 1e-6
    keys = set(m.state_dict())
    assert {"logit_scale", "visual.pos_embed_spatial", "visual.blocks.1.attn.q.weight", "text.blocks.0.attn.qkv.weight",
            "text.fc_norm.weight"} <= keys
    with pytest.raises(NotImplementedError):
        coem.build_towers_from_config({**cfg, "vision_cfg": {**cfg["vision_cfg"], "model_name": "longnet_x"}})
