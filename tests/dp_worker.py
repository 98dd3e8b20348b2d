"""2-rank data-parallel worker (run under torch.distributed.run by tests/test_gpu_comm.py):
rank r trains the small 3-D MAE on ITS volumes through FlatGradReducer, then every rank recomputes both ranks' local gradients
alone and checks  exchanged == mean(local_0, local_1).  Rank 0 writes result.json.
  OCTMAE_DP_BACKEND=rccl (default; needs >= 2 GPUs): one GPU per rank, the native RCCL communicator (octmae_comm_*).
  OCTMAE_DP_BACKEND=gloo: BOTH ranks on GPU 0, the exchange through torch.distributed's gloo group on the device tensors --
  the only world-size-2 run of the GPU training path (HIP kernels, readiness-ordered chunks launched from backward, learning
  step, cold chunks) a one-GPU box allows; RCCL does not accept two ranks on one device."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    gloo = os.environ.get("OCTMAE_DP_BACKEND", "rccl") == "gloo"
    if gloo:              # (RCCL with both ranks on device 0 was tried on the box: ncclCommInitRank fails, as NCCL's does)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend="gloo" if gloo else "nccl", rank=rank, world_size=world)
    from functools import partial
    from octcubem_amd import comm as ocomm, models_mae, misc, optim as foptim
    from octcubem_amd.parallel import FlatGradReducer
    from oracle import mae3d_ref as O
    comm = None if gloo else ocomm.NativeComm.from_store(dist.distributed_c10d._get_default_store(), rank, world, local)

    def bcast(t):
        if comm is not None:
            comm.broadcast_async(t, 0); comm.wait()
        else:
            dist.broadcast(t, src=0)

    def allreduce_scalar(v, op):
        if comm is not None:
            return comm.all_reduce_scalar(v, ocomm.SUM if op == "sum" else ocomm.MAX)
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
        return float(t.item())
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
                      decoder_num_heads=2, num_frames=12, t_patch_size=3, pred_t_dim=12, high_res_input_size=128)
    P = O.init_params(cfg, seed=7 + rank, bias_std=0.02)          # DIFFERENT weights per rank: the broadcast must fix that
    m = models_mae.MaskedAutoencoderViT(
        input_size=64, patch_size=16, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
        decoder_num_heads=2, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=12, t_patch_size=3, sep_pos_embed=True,
        cls_embed=True, pred_t_dim=12, high_res_input_size=128)
    m.load_state_dict(P, strict=True)
    m = m.to(dev).train()
    m.prepare()
    red = FlatGradReducer(m, n_chunks=4, comm=comm)
    red.broadcast_parameters(0)
    m.arena.refresh_lp()
    w = m.arena.flat.clone()
    w0 = w.clone()
    bcast(w0)
    torch.cuda.synchronize()
    params_equal = bool(torch.equal(w, w0))

    def data(r):
        imgs = torch.rand(4, 1, 12, 64, 64, generator=torch.Generator().manual_seed(100 + r)).to(dev)
        noise = torch.rand(4, cfg.num_patches, generator=torch.Generator().manual_seed(200 + r)).to(dev)
        return imgs, noise

    scaler = misc.NativeScalerWithGradNormCount(fp32=True, reducer=red)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=0.0, betas=(0.9, 0.95))
    worst = 0.0
    for step in range(3):           # step 0 learns which parameters are cold; 1 and 2 use the readiness-aware layout
        opt.zero_grad()
        imgs, noise = data(rank)
        loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        scaler(loss, opt, parameters=list(m.parameters()))
        torch.cuda.synchronize()
        exchanged = m.arena.grad.clone()
        ref = torch.zeros_like(exchanged)
        for r in range(world):
            m.arena.zero_grad()
            imgs, noise = data(r)
            loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
            loss.backward()
            torch.cuda.synchronize()
            ref += m.arena.grad / world
        worst = max(worst, float((exchanged - ref).abs().max() / ref.abs().max()))
    other = exchanged.clone()
    bcast(other)
    torch.cuda.synchronize()
    agree = bool(torch.equal(other, exchanged))
    agree_all = allreduce_scalar(1.0 if agree else 0.0, "sum") == float(world)
    worst = allreduce_scalar(worst, "max")
    if rank == 0:
        with open(os.path.join(os.environ["OCTMAE_DP_OUT"], "result.json"), "w") as f:
            json.dump({"world": world, "backend": "gloo on device tensors" if gloo else "octmae_comm", "params_equal_after_broadcast": params_equal,
                       "max_rel_err": worst, "ranks_agree": bool(agree_all), "reducer": red.stats,
                       "cold": sorted(n for n, p in m.named_parameters() if id(p) in red._cold)}, f)
    if comm is not None:
        comm.barrier()
        comm.destroy()
    else:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
