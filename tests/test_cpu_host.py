"""CPU-side tests: the C-ABI library loads and exports every symbol include/octmae.h declares, host logic of the
drop-in modules (constructor contract, state_dict keys, schedules, grouping), loud failure without a GPU."""
import ctypes
import math
import os
import re
import sys

import pytest
import torch

from oracle import mae3d_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "octmae.h")).read()
    return sorted(set(re.findall(r"\bint\s+(octmae_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from octcubem_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
        assert s in _lib.SIGNATURES, f"{s} declared in octmae.h but not bound in _lib.SIGNATURES"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.octmae_abi_version() == _lib.expected_abi_version() and lib.octmae_mt_chunk_elems() == 65536


def test_half_build_exports_the_same_abi_and_reports_its_operand_type():
    """liboctmae_f16.so (make F16=1; the verification build of tests/test_gpu_f16_parity.py) is the SAME source: every declared
    symbol, the same ABI number, octmae_lp_dtype() == 1 where the product library says 0; and the host side maps that to the torch
    dtype it allocates 16-bit buffers with (in a child process: a process binds ONE library)."""
    import subprocess
    from octcubem_amd import _lib
    path = os.path.join(ROOT, "octcubem_amd", "liboctmae_f16.so")
    assert os.path.exists(path), "make -C octcubem_amd/csrc both"
    lib = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert lib.octmae_abi_version() == _lib.expected_abi_version()
    assert lib.octmae_lp_dtype() == 1 and _lib.load().octmae_lp_dtype() == 0
    code = "from octcubem_amd import ops; print(ops.BF16, ops.LP_IS_F16, ops.ATTN_OPTIMISTIC)"
    for lp, want in ((path, "torch.float16 True False"), (_lib.LIB_PATH, "torch.bfloat16 False True")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, OCTMAE_LIB=lp))
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == want, (r.stdout, r.stderr)


def test_argument_errors_are_reported_without_a_gpu():
    from octcubem_amd import _lib
    lib = _lib.load()
    # NULL pointers / bad sizes are rejected before any launch
    assert lib.octmae_gemm_bf16(None, None, None, None, None, None, 8, 8, 8, 8, 8, 8, 0, 0, 0, 0, 1, None) == -1
    assert lib.octmae_attn_fwd(None, None, None, None, 1, 1, 1, 64, 0.125, None) == -1
    assert lib.octmae_layernorm_fwd(None, None, None, None, None, None, 1, 64, 1e-6, None) == -1
    assert lib.octmae_random_masking_ids(None, None, None, None, None, 1, 8, 2, None) == -1
    with pytest.raises(_lib.OctmaeError):
        _lib.call("octmae_cast_f32_bf16", None, None, 8, None)


def test_weight_gradient_split_plan_tiles_the_rows_exactly_once():
    """octmae_wgrad_split_plan (csrc/gemm.hip: the host-side copy of the arithmetic the weight-gradient kernels run, split_range_of +
    wgrad_stagger_for): whatever the row count, the requested split, the tile count and the stagger option, the k slices must
    cover [0, ceil(M / 64)) contiguously, none empty; staggered slices rise linearly and the shortest keeps at least half the mean
    length and 8 k-tiles; the rule applies only to >= 8 slices of <= 96 k-tiles; option 0 switches it off."""
    from octcubem_amd import _lib
    lib = _lib.load()
    import random
    rng = random.Random(4)
    prev = lib.octmae_set_option(b"wgrad_stagger", 29)
    try:
        seen_staggered = 0
        for v in (29, 0, 200, 100000):
            assert lib.octmae_set_option(b"wgrad_stagger", v) >= 0
            for _ in range(400):
                M = rng.choice([1, 63, 64, 65, 700, 40992, 163968, 655488, rng.randrange(1, 700000)])
                splitk = rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 21, 64, rng.randrange(1, 300)])
                tiles = rng.choice([1, 4, 12, 16, 48, 64, 128])
                slices = ctypes.c_int(0)
                bounds = (ctypes.c_int * (splitk + 2))()
                d = lib.octmae_wgrad_split_plan(M, splitk, tiles, ctypes.byref(slices), bounds)
                S, ktiles = slices.value, (M + 63) // 64
                b = list(bounds[:S + 1])
                assert 1 <= S <= max(1, min(splitk, ktiles)), (M, splitk, S)
                assert b[0] == 0 and b[-1] == ktiles, (M, splitk, tiles, v, b)
                lens = [b[i + 1] - b[i] for i in range(S)]
                assert min(lens) >= 1, (M, splitk, tiles, v, lens)
                if d == 0:                                   # equal slices: all but the last have the same length
                    assert len(set(lens[:-1])) <= 1 and lens[-1] <= lens[0]
                else:
                    seen_staggered += 1
                    assert v > 0 and S >= 8 and ktiles <= 96 * S
                    mean = ktiles / S
                    assert min(lens) >= min(mean / 2, mean - 8) - 1.01 and min(lens) >= 7, (lens, mean)
                    # rising by d / 256 k-tiles per slice; each bound carries two floors (error in (-1, 1)), a length two bounds,
                    # a difference of lengths three
                    step = d / 256.0
                    assert all(abs((lens[i + 1] - lens[i]) - step) < 4.0 for i in range(S - 1)), (lens, step)
                    assert abs((lens[-1] - lens[0]) - step * (S - 1)) < 4.0, (lens, step)
                if v == 0:
                    assert d == 0
        assert seen_staggered > 50
        assert lib.octmae_wgrad_split_plan(0, 4, 16, None, None) == -1
    finally:
        lib.octmae_set_option(b"wgrad_stagger", prev)


def test_model_contract_and_state_dict_keys():
    from octcubem_amd import models_mae
    from functools import partial
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
                      decoder_num_heads=2, num_frames=12, t_patch_size=3, pred_t_dim=12, high_res_input_size=128)
    m = models_mae.MaskedAutoencoderViT(input_size=64, patch_size=16, in_chans=1, embed_dim=128, depth=2, num_heads=2,
                                        decoder_embed_dim=64, decoder_depth=2, decoder_num_heads=2,
                                        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=12, t_patch_size=3,
                                        sep_pos_embed=True, cls_embed=True, pred_t_dim=12, high_res_input_size=128)
    shapes = O.param_shapes(cfg)
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    # attributes downstream reference code reads (video_vit.py:49-67, models_mae…:83-84,503-507)
    pe = m.patch_embed
    assert (pe.num_patches, pe.input_size, pe.patch_size, pe.grid_size, pe.t_grid_size, pe.frames, pe.t_patch_size) == \
        (64, (4, 4, 4), (16, 16), 4, 4, 12, 3)
    assert m.t_pred_patch_size == 3 and m.high_res_input_size == (4, 8, 8)
    # the flash-layout checkpoint keys are accepted (reverse of the reference's remap, models_mae…:693-724)
    P = O.init_params(cfg, seed=1)
    flash = {}
    for k, v in P.items():
        mm = re.match(r"(.*blocks\.\d+)\.attn\.(q|k|v)\.(weight|bias)$", k)
        if mm:
            continue
        flash[k.replace(".attn.proj.", ".mixer.out_proj.")] = v
    for pre in [f"blocks.{i}" for i in range(2)] + [f"decoder_blocks.{i}" for i in range(2)]:
        for kind in ("weight", "bias"):
            flash[f"{pre}.mixer.Wqkv.{kind}"] = torch.cat([P[f"{pre}.attn.{n}.{kind}"] for n in "qkv"], 0)
    res = m.load_state_dict_to_backbone(flash, strict=True)
    for k in P:
        assert torch.equal(m.state_dict()[k], P[k]), k
    # patchify / unpatchify round trip (visualiser helpers)
    x = torch.rand(2, 1, 12, 64, 64)
    assert torch.equal(m.unpatchify(m.patchify(x)), x)
    assert torch.allclose(m.patchify(x), O.patchify(x, cfg))


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from octcubem_amd import models_mae
    from functools import partial
    m = models_mae.MaskedAutoencoderViT(input_size=64, patch_size=16, in_chans=1, embed_dim=128, depth=1, num_heads=2,
                                        decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2,
                                        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=6, t_patch_size=3,
                                        sep_pos_embed=True, cls_embed=True, pred_t_dim=6, high_res_input_size=128)
    with pytest.raises(RuntimeError, match="GPU only|no CPU fallback|cuda"):
        m(torch.rand(1, 1, 6, 64, 64))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "octcubem_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", "").replace("oracle's", ""), fn


def test_lr_schedule_and_weight_decay_groups():
    from octcubem_amd import lr_sched, misc, models_mae

    class A: pass
    a = A(); a.lr = 1.6e-3; a.min_lr = 1e-6; a.warmup_epochs = 5; a.epochs = 50

    class Opt:
        def __init__(self): self.param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]
    for e in (0.0, 0.37, 4.999, 5.0, 17.3, 49.99):
        o = Opt()
        lr = lr_sched.adjust_learning_rate(o, e, a)
        assert abs(lr - O.cosine_lr(e, 1.6e-3, 1e-6, 5, 50)) < 1e-15
        assert o.param_groups[0]["lr"] == lr and o.param_groups[1]["lr"] == lr * 0.5
    from functools import partial
    m = models_mae.MaskedAutoencoderViT(input_size=64, patch_size=16, in_chans=1, embed_dim=128, depth=1, num_heads=2,
                                        decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2,
                                        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=6, t_patch_size=3,
                                        sep_pos_embed=True, cls_embed=True, pred_t_dim=6, high_res_input_size=128)
    groups = misc.add_weight_decay(m, 0.05)
    names = {id(p): n for n, p in m.named_parameters()}
    nd, d = O.weight_decay_groups([(n, tuple(p.shape)) for n, p in m.named_parameters()], 0.05)
    assert [names[id(p)] for p in groups[0]["params"]] == nd and [names[id(p)] for p in groups[1]["params"]] == d
    assert groups[0]["weight_decay"] == 0.0 and groups[1]["weight_decay"] == 0.05


def test_arena_ordering_makes_qkv_adjacent():
    from octcubem_amd.arena import _ordered
    names = ["a.norm1.weight", "a.attn.q.weight", "a.attn.q.bias", "a.attn.k.weight", "a.attn.k.bias", "a.attn.v.weight",
             "a.attn.v.bias", "a.attn.proj.weight", "a.attn.proj.bias"]
    out = [n for n, _ in _ordered([(n, None) for n in names])]
    assert out == ["a.norm1.weight", "a.attn.q.weight", "a.attn.k.weight", "a.attn.v.weight", "a.attn.q.bias", "a.attn.k.bias",
                   "a.attn.v.bias", "a.attn.proj.weight", "a.attn.proj.bias"]


def test_lr_decay_groups_match_reference_golden(golden_dir):
    """octcubem_amd.lr_decay on a parameter-only stand-in for the ST ViT (no GPU) vs the groups the reference built."""
    import json
    import numpy as np
    import torch
    from octcubem_amd import lr_decay
    from oracle import vit_ref as V
    z = np.load(os.path.join(golden_dir, "finetune_small.npz"))
    cfg = V.ViTSTConfig(**json.loads(str(z["cfg"])))
    shapes = V.vit_st_param_shapes(cfg)

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = [None] * cfg.depth
            self._names = {}
            for n, s in shapes.items():
                key = n.replace(".", "__")
                self.register_parameter(key, torch.nn.Parameter(torch.zeros(s)))
                self._names[key] = n

        def named_parameters(self, *a, **k):
            for key, p in super().named_parameters(*a, **k):
                yield self._names[key], p
    m = Stub()
    groups = lr_decay.param_groups_lrd(m, 0.05, no_weight_decay_list=("cls_token", "pos_embed", "pos_embed_spatial",
                                                                    "pos_embed_temporal", "pos_embed_class"), layer_decay=0.75)
    id2name = {id(p): n for n, p in m.named_parameters()}
    ref = json.loads(str(z["groups"]))
    key = lambda g: (g["lr_scale"], g["weight_decay"])
    assert len(groups) == len(ref)
    for a, b in zip(sorted(groups, key=key), sorted(ref, key=key)):
        assert sorted(id2name[id(p)] for p in a["params"]) == sorted(b["params"])
        assert a["weight_decay"] == b["weight_decay"] and abs(a["lr_scale"] - b["lr_scale"]) < 1e-15
    for n, lid in json.loads(str(z["layer_ids"])).items():
        assert lr_decay.get_layer_id_for_vit(n, cfg.depth + 1) == lid


def test_graft_entry_build_compiles_and_agrees_on_the_abi_number():
    """__graft_entry__.build() must run clean from this tree: make (a no-op when up to date), import, and the ABI number
    the library reports == OCTMAE_ABI_VERSION in include/octmae.h (the one place it is written)."""
    import __graft_entry__ as g
    from octcubem_amd import _lib
    g.build()
    hdr = open(os.path.join(ROOT, "include", "octmae.h")).read()
    assert int(re.search(r"#define\s+OCTMAE_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.load().octmae_abi_version()
    src = open(os.path.join(ROOT, "octcubem_amd", "csrc", "probe.hip")).read()
    assert "return OCTMAE_ABI_VERSION" in src


def test_generated_attention_bodies_are_current():
    """csrc/attn_bwd1w_body.inc and attn_bwd1w_body_hd64.inc (the placed tile bodies of the one-wave-per-SIMD attention backward
    kernels) are committed so that a build needs no Python; they must be what tools/gen_attn_bwd1w.py produces."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_bwd1w.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_native_comm_bootstrap_over_a_store(monkeypatch):
    """comm.NativeComm.from_store: rank 0 publishes the 128-byte RCCL unique id under one key of the launcher's store and every
    rank -- rank 0 included -- constructs its communicator from the bytes it reads back (the communicator itself needs GPUs:
    its constructor and the id source are replaced; octmae_comm_* argument errors are checked in
    test_argument_errors_are_reported_without_a_gpu)."""
    import threading
    import torch.distributed as dist
    from octcubem_amd import comm as ocomm
    made = {}
    the_id = bytes(range(128))
    monkeypatch.setattr(ocomm.NativeComm, "unique_id", staticmethod(lambda: the_id))
    monkeypatch.setattr(ocomm.NativeComm, "__init__", lambda self, idb, rank, world, device: made.__setitem__(rank, (bytes(idb), world, device)))
    monkeypatch.setattr(ocomm.NativeComm, "__del__", lambda self: None, raising=False)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    store = dist.HashStore()
    ths = [threading.Thread(target=ocomm.NativeComm.from_store, args=(store, r, 3, r)) for r in (2, 1, 0)]   # rank 0 arrives last
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=30)
    assert made == {r: (the_id, 3, r) for r in range(3)}
    assert len(the_id) == ocomm.ID_BYTES == 128


def test_bench_board_sampler_without_hwmon_reports_nothing(tmp_path, monkeypatch):
    """bench.BoardSampler reads the amdgpu hwmon files by plain file reads (no child process after the GPU is initialised); on a
    host without them -- this container -- it must start, stop and report None instead of failing the benchmark."""
    import glob
    import bench
    monkeypatch.setattr(glob, "glob", lambda pat, **kw: [])
    b = bench.BoardSampler(0, period=0.01)
    assert b.cards == {}
    b.start()
    assert b.stop() is None


def test_bench_board_sampler_reads_power_and_clock(tmp_path, monkeypatch):
    import glob
    import time
    import bench
    h = tmp_path / "card0" / "device" / "hwmon" / "hwmon3"
    h.mkdir(parents=True)
    (h / "power1_input").write_text("1350000000\n"); (h / "freq1_input").write_text("1950000000\n"); (h / "power1_cap").write_text("1400000000\n")
    monkeypatch.setattr(glob, "glob", lambda pat, **kw: [str(h)])
    b = bench.BoardSampler(0, period=0.01)
    b.start(); time.sleep(0.1)
    st = b.stop()
    assert st is not None and st["samples"] >= 2
    assert abs(st["power_w_avg"] - 1350.0) < 1e-6 and abs(st["sclk_mhz_avg"] - 1950.0) < 1e-6 and st["power_cap_w"] == 1400.0


def test_device_prefetcher_is_a_pass_through_on_the_cpu():
    from octcubem_amd import misc
    batches = [(torch.full((2, 3), float(i)), {"id": i, "t": torch.tensor([i])}) for i in range(5)]
    assert misc.prefetched(batches, "cpu") is batches                      # no GPU: the loader itself
    pf = misc.DevicePrefetcher(batches, "cpu", only=(0,))
    assert len(pf) == 5
    out = list(pf)
    assert all(a is b for a, b in zip(out, batches))
    assert list(misc.DevicePrefetcher([], "cpu")) == []


def test_autocast_invariant_decorator_switches_the_context_off_for_forward_methods_only():
    """octcubem_amd/_autocast.py on the CPU (the GPU side is tests/test_gpu_autocast.py): a decorated module's forward / forward_* /
    encode_* run with autocast off -- an fp32 matmul stays fp32 inside `torch.autocast("cpu", dtype=torch.bfloat16)` -- other methods
    and undecorated modules are untouched, and outside autocast the wrapper adds nothing."""
    from octcubem_amd._autocast import autocast_invariant

    class Plain(torch.nn.Module):
        def forward(self, a, b):
            return a @ b

        def forward_features(self, a, b):
            return a @ b

        def encode_image(self, a, b):
            return a @ b

        def helper(self, a, b):
            return a @ b

    Inv = autocast_invariant(type("Inv", (Plain,), dict(Plain.__dict__)))
    a, b = torch.randn(8, 8), torch.randn(8, 8)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        assert Plain()(a, b).dtype == torch.bfloat16                      # what autocast does to an ATen island
        m = Inv()
        for out in (m(a, b), m.forward_features(a, b), m.encode_image(a, b)):
            assert out.dtype == torch.float32
        assert m.helper(a, b).dtype == torch.bfloat16                     # not a forward-like method: untouched
        assert torch.is_autocast_enabled("cpu")                           # the caller's context is restored
    exact = a @ b
    with torch.autocast("cpu", dtype=torch.bfloat16):
        assert torch.equal(Inv()(a, b), exact)
    assert torch.equal(Inv()(a, b), exact)
    assert getattr(Inv.forward, "_octmae_no_autocast", False) and not hasattr(Inv.helper, "_octmae_no_autocast")


def test_small_launch_plan():
    """The cost model that sends a forward / dgrad GEMM to the small-launch kernel (csrc/gemm.hip plan128, reached without a GPU through
    octmae_gemm_small_plan).  Properties the kernel relies on: 1 <= slices <= 4, no EMPTY slice (its workgroup would publish a zero
    tile -- harmless -- but a tile whose counter never reaches `slices` would never be written: (S - 1) * ceil(ktiles / S) < ktiles),
    a split only with the workspace lent and within its 1024 slots, a ring of 4 or 2 stages.  Policy: the Linear shapes of ONE
    volume (the reference's shipped recipe) take it and the long reductions are split; the shapes of the headline step (32 / 64 / 128
    volumes per micro-batch) never do."""
    from octcubem_amd import _lib
    lib = _lib.load()
    S, st = ctypes.c_int(0), ctypes.c_int(0)

    def plan(NA, NB, K, cus=256, ws=1, big=1):
        use = lib.octmae_gemm_small_plan(NA, NB, K, cus, ws, big, ctypes.byref(S), ctypes.byref(st))
        assert use in (0, 1)
        return use, S.value, st.value

    for cus in (256, 304, 64):
        for NA in (512, 1024, 1536, 2048, 3072, 4096):
            for NB in (1, 64, 1281, 2562, 5121, 4 * 1281, 8 * 5121, 32 * 1281, 128 * 5121):
                for K in (64, 512, 768, 1024, 2048, 3072, 4096):
                    for ws in (0, 1):
                        use, s_, n_ = plan(NA, NB, K, cus, ws, int(NA >= 256 and NB >= 256))
                        kt = K // 64
                        assert 1 <= s_ <= 4 and n_ in (2, 4)
                        if not use:
                            assert s_ == 1
                            continue
                        assert ws or s_ == 1
                        if s_ > 1:
                            assert (s_ - 1) * -(-kt // s_) < kt and kt // s_ >= 8
                            assert -(-NA // 128) * -(-NB // 128) * s_ <= 1024
    # one volume per step: every forward / dgrad Linear of the encoder (1281 rows, D 1024) and decoder (5121 rows, D 512)
    for NA, NB, K in ((3072, 1281, 1024), (1024, 1281, 1024), (4096, 1281, 1024), (1024, 1281, 4096), (1024, 1281, 3072),
                      (1536, 5121, 512), (512, 5121, 512), (512, 5121, 2048)):
        assert plan(NA, NB, K)[0] == 1, (NA, NB, K)
    assert plan(1024, 1281, 4096)[1] > 1 and plan(1024, 1281, 4096, ws=0)[1] == 1       # the long reductions are split when they can be
    # the headline step never takes it
    for vols in (32, 64, 128):
        for NA, rows, K in ((3072, 1281, 1024), (1024, 1281, 1024), (4096, 1281, 1024), (1024, 1281, 4096), (1024, 1281, 3072),
                            (1536, 5121, 512), (512, 5121, 512), (2048, 5121, 512), (512, 5121, 2048), (512, 5121, 1536)):
            assert plan(NA, vols * rows, K)[0] == 0, (vols, NA, rows, K)
    prev = lib.octmae_set_option(b"gemm_small", 0)
    try:
        assert plan(1024, 1281, 4096)[0] == 0                                              # switched off: never
    finally:
        lib.octmae_set_option(b"gemm_small", prev)


def test_weight_gradient_split_rule():
    """ops._splitk_for (host arithmetic): as many k slices as keep tiles x slices within one round of 256 workgroups with >= 8 k-tiles
    each -- unless the fp32-atomic epilogues of the slices (0.22 us per tile and slice) cost more than the main-loop time they save
    (1.4 us per k-tile): one or two volumes per step stay unsplit where round 5 split two ways; from 8 volumes on nothing changes
    (the headline's choices are those of rounds 2-5)."""
    from octcubem_amd import ops
    enc_fc, enc_qkv, dec_fc, dec_qkv = 128, 64, 32, 16          # 256 x 256 output tiles of the four weight-gradient pairs of ViT-L
    kt = lambda rows: (rows + 63) // 64
    for vols in (8, 16, 32, 64, 128):                           # unchanged from the old rule
        assert ops._splitk_for(enc_fc, kt(vols * 1281), 256) == 2
        assert ops._splitk_for(enc_qkv, kt(vols * 1281), 256) == 4
        assert ops._splitk_for(dec_fc, kt(vols * 5121), 256) == 8
        assert ops._splitk_for(dec_qkv, kt(vols * 5121), 256) in (15, 16)
    assert ops._splitk_for(enc_fc, kt(1281), 256) == 1 and ops._splitk_for(enc_qkv, kt(1281), 256) == 1       # one volume: no atomics
    assert ops._splitk_for(enc_fc, kt(2 * 1281), 256) == 1 and ops._splitk_for(enc_qkv, kt(2 * 1281), 256) == 2
    assert 3 <= ops._splitk_for(dec_fc, kt(5121), 256) <= 5 and 5 <= ops._splitk_for(dec_qkv, kt(5121), 256) <= 7
    for tiles in (1, 7, 16, 64, 128, 300):                      # always a legal split
        for ktiles in (1, 5, 8, 21, 81, 2562):
            s_ = ops._splitk_for(tiles, ktiles, 256)
            assert 1 <= s_ <= max(1, 256 // tiles) and (s_ == 1 or ktiles // s_ >= 8)
            assert ops._splitk_for(tiles, ktiles, 1024) >= 1   # the 128-tile register-staged kernel's target: the old rule


def test_bench_parity_compliant_child_failure_is_recorded_not_raised():
    """bench.py's `parity_compliant` record comes from a child process on the half-operand build; whatever goes wrong there (here: no GPU
    in this container, so the child exits with an error) must end up IN the record -- the headline run goes on."""
    import argparse
    import bench
    rec = bench.run_parity_compliant_child(argparse.Namespace(global_batch=256, micro_batch=0), steps=1, warmup=0)
    assert rec["dtype"] == "f16" and rec["lib"] == "liboctmae_f16.so" and rec["value"] is None
    assert "error" in rec and rec["error"]
