"""The small-launch kernel of the forward / dgrad GEMMs (csrc/gemm.hip: gemm128d_kernel, plan128; octmae_gemm_bf16_ws and the workspace
arguments of the fused entry points).  The reference's shipped recipe runs ONE volume per GPU
(Pre-training/scripts/run_chunks_pretraining_vitl_oph_joint_flash_attn.sh:25-30): nn.Linear forward / backward over 1281 and 5121 token
rows (video_vit.py:114-135, timm Mlp), where 256 x 256 tiles leave most of the 256 CUs idle.

Checked here: the AUTOMATIC choice takes the kernel at those shapes (and splits the long reductions), its results equal the 256-tile
kernels' up to the order of the fp32 additions of a split -- every fused epilogue of both kinds --, it is deterministic, its
arrival-counter hand-off is race-free under memory pressure, and the 128-volume shapes of the headline step never take it.
(tests/test_gpu_kernels.py runs every GEMM test through the forced forms of the kernel as well.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import ops
    BF16 = ops.BF16

DEV = "cuda"


def bf(t):
    return t.to(BF16)


def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def launches():
    return ops.set_option("gemm_small_launches", 0), ops.set_option("gemm_small_split_launches", 0)


class never_small:
    """Context: the same calls on the kernels of rounds 1-5 (256-tile / register-staged)."""

    def __enter__(self):
        self.prev, ops.FORCE_SMALL_LAUNCH = ops.FORCE_SMALL_LAUNCH, -1

    def __exit__(self, *a):
        ops.FORCE_SMALL_LAUNCH = self.prev


# (M, N, K, taken by the automatic choice?, split?) -- the Linear shapes of ONE volume (encoder: 1281 rows, D 1024; decoder: 5121, D 512)
FWD_SHAPES = [
    (1281, 1024, 4096, True, True),      # fc2 forward: 24 tiles of 256 x 256, 64 k-tiles
    (1281, 1024, 1024, True, False),     # proj forward: 16 k-tiles, nothing to split
    (1281, 3072, 1024, True, False),     # qkv forward: 264 tiles of 128 x 128 -- the 2-stage ring, two workgroups per CU
    (5121, 512, 2048, True, None),       # decoder fc2
    (5121, 512, 512, True, False),       # decoder proj
    (4 * 1281, 1024, 4096, None, None),  # four volumes: whatever the model says, the result must hold
    (16 * 1281, 1024, 1024, True, False),  # 16 volumes' rows: 324 tiles of 256 = 1.27 rounds of CUs against 2.5 rounds of 512 slots: taken (2-stage ring)
    (32 * 1281, 1024, 1024, False, False), # 32 volumes' rows (one rank of 8): 2.5 rounds against 5.02 -- a full-size launch keeps the 256-tile kernel
]


@pytest.mark.parametrize("M,N,K,taken,split", FWD_SHAPES)
def test_small_launch_forward_epilogues_equal_the_256_tile_kernels(M, N, K, taken, split):
    g = torch.Generator().manual_seed(M + N + K)
    x = bf(torch.randn(M, K, generator=g)).to(DEV)
    w = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    sc = torch.tensor([0.0, 1.25, 2.0], device=DEV)[torch.randint(0, 3, (M,), generator=g).to(DEV)]

    def run():
        return {"f32": ops.linear_fwd(x, w, b, "f32"), "bf16": ops.linear_fwd(x, w, b, "bf16"), "gelu": ops.linear_fwd(x, w, b, "gelu"),
                "resid": ops.linear_fwd(x, w, b, "resid", res=res),
                "rowscale": ops.linear_fwd(x, w, b, "resid", res=res, rowscale=sc, rows_per_scale=1)}

    n0 = launches()
    got = run()
    n1 = launches()
    again = run()
    with never_small():
        n2 = launches()
        ref = run()
        assert launches() == n2
    if taken is not None:
        assert n1[0] - n0[0] == (5 if taken else 0), (n0, n1)
    if split is not None:
        assert n1[1] - n0[1] == (5 if split else 0), (n0, n1)
    for k in got:                                                              # deterministic, split or not
        for a_, b_ in zip(got[k] if isinstance(got[k], tuple) else (got[k],), again[k] if isinstance(again[k], tuple) else (again[k],)):
            assert torch.equal(a_, b_), k
    exact = x.double() @ w.double().t() + b.double()
    assert rel(got["f32"], exact) < 2e-6 and rel(got["f32"], ref["f32"]) < 1e-6
    assert rel(got["resid"], ref["resid"]) < 1e-6 and rel(got["rowscale"], ref["rowscale"]) < 1e-6
    assert torch.equal(got["rowscale"][sc == 0], res[sc == 0])                 # a dropped sample passes the residual through bit-exactly
    for a_, r_ in ((got["bf16"], ref["bf16"]), (got["gelu"][0], ref["gelu"][0]), (got["gelu"][1], ref["gelu"][1])):
        assert rel(a_, r_) < 2e-4                                              # a few 1-ulp flips of the 16-bit rounding
        assert float((a_ != r_).float().mean()) < 2e-3
    assert rel(got["bf16"], exact) < 3e-3


# dx[M, N] = dy[M, K] @ w[K, N]: (M, N = output columns, K = reduction)
DGRAD_SHAPES = [(1281, 1024, 4096, True), (1281, 1024, 3072, True), (1281, 4096, 1024, True), (5121, 512, 2048, True), (2562, 1024, 1024, None)]


@pytest.mark.parametrize("M,N,K,taken", DGRAD_SHAPES)
def test_small_launch_dgrad_epilogues_equal_the_256_tile_kernels(M, N, K, taken):
    """dgrad (weight k-strided): plain, x GELU' with the bias-gradient column sums, and with the attention delta."""
    g = torch.Generator().manual_seed(M * 3 + N)
    dy = bf(torch.randn(M, K, generator=g)).to(DEV)
    w = bf(torch.randn(K, N, generator=g) * K ** -0.5).to(DEV)
    pre = bf(torch.randn(M, N, generator=g)).to(DEV)
    o = bf(torch.randn(M, N, generator=g)).to(DEV)
    H, HD = N // 64, 64

    def run():
        cs = torch.zeros(N, device=DEV)
        out = [ops.linear_dgrad(dy, w), ops.linear_dgrad(dy, w, pre=pre, colsum=cs), cs]
        dx, delta = ops.linear_dgrad_delta(dy, w, o, H, HD)
        return out + [dx, delta]

    n0 = launches()
    got = run()
    n1 = launches()
    got2 = run()
    with never_small():
        ref = run()
    if taken is not None:
        assert n1[0] - n0[0] == (3 if taken else 0), (n0, n1)
    for a_, b_ in zip(got, got2):
        assert a_ is None or torch.equal(a_, b_)
    exact = dy.double() @ w.double()
    assert rel(got[0], exact) < 3e-3 and rel(got[0], ref[0]) < 2e-4
    assert rel(got[1], ref[1]) < 2e-4 and rel(got[2], ref[2]) < 1e-4
    assert rel(got[3], ref[3]) < 2e-4 and got[4] is not None and ref[4] is not None and rel(got[4], ref[4]) < 1e-4
    # delta against its definition on the 16-bit-rounded dO the attention backward reads
    dd = -(got[3].double() * o.double()).view(M, H, HD).sum(-1)
    assert rel(got[4], dd) < 1e-5


def test_small_launch_delta_at_head_dim_32():
    """The decoder's heads (hd 32: four lanes of the epilogue's 8-lane rows per head)."""
    M, N, K = 5121, 512, 512
    g = torch.Generator().manual_seed(3)
    dy = bf(torch.randn(M, K, generator=g)).to(DEV)
    w = bf(torch.randn(K, N, generator=g) * K ** -0.5).to(DEV)
    o = bf(torch.randn(M, N, generator=g)).to(DEV)
    n0 = launches()
    dx, delta = ops.linear_dgrad_delta(dy, w, o, 16, 32)
    assert launches()[0] - n0[0] == 1 and delta is not None
    assert rel(dx, dy.double() @ w.double()) < 3e-3
    assert rel(delta, -(dx.double() * o.double()).view(M, 16, 32).sum(-1)) < 1e-5


def test_split_hand_off_is_race_free_under_memory_pressure():
    """Back-to-back launches of three split shapes (the workspace and its counters are re-used by every launch) while a second stream
    saturates HBM: every result must be bit-identical to the first (a partial read before it was complete, a counter seen before its
    payload, or a counter not back at zero would show as a different sum or a missing tile)."""
    g = torch.Generator().manual_seed(5)
    cases = []
    for M, N, K in ((1281, 1024, 4096), (1281, 1024, 3072), (2562, 512, 2048)):
        x = bf(torch.randn(M, K, generator=g)).to(DEV)
        w = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        ops.FORCE_SMALL_LAUNCH, ops.FORCE_SPLITK = 4, 2 + len(cases)          # 2-, 3- and 4-way splits, forced
        cases.append((x, w, b, ops.FORCE_SPLITK, ops.linear_fwd(x, w, b, "f32").clone()))
    big = torch.empty(1 << 28, dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    stop = 0
    n0 = launches()
    try:
        for it in range(150):
            if it % 4 == 0:
                with torch.cuda.stream(side):
                    big.add_(1.0)
            x, w, b, S, first = cases[it % 3]
            ops.FORCE_SPLITK = S
            y = ops.linear_fwd(x, w, b, "f32")
            if not torch.equal(y, first):
                stop += 1
        torch.cuda.synchronize()
    finally:
        ops.FORCE_SMALL_LAUNCH, ops.FORCE_SPLITK = 0, 1
    n1 = launches()
    assert n1[0] - n0[0] == 150 and n1[1] - n0[1] == 150
    assert stop == 0, f"{stop} of 150 launches differed from the first"
    x, w, b, _, first = cases[0]
    assert rel(first, x.double() @ w.double().t() + b.double()) < 2e-6


def test_without_the_workspace_nothing_is_split():
    M, N, K = 1281, 1024, 4096
    g = torch.Generator().manual_seed(9)
    x = bf(torch.randn(M, K, generator=g)).to(DEV)
    w = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    prev, ops.SPLIT_WS = ops.SPLIT_WS, False
    try:
        n0 = launches()
        y = ops.linear_fwd(x, w, None, "f32")
        n1 = launches()
    finally:
        ops.SPLIT_WS = prev
    assert n1[0] - n0[0] == 1 and n1[1] == n0[1]
    assert rel(y, x.double() @ w.double().t()) < 2e-6
    prev = ops.set_option("gemm_small", 0)
    try:
        n0 = launches()
        y2 = ops.linear_fwd(x, w, None, "f32")
        assert launches() == n0
    finally:
        ops.set_option("gemm_small", prev)
    assert rel(y2, y) < 1e-6


# (M, D): the weight-gradient pairs of ONE and TWO volumes -- fc2 + fc1 and proj + qkv of an encoder Block (D 1024) and a decoder Block (D 512)
@pytest.mark.parametrize("M,D,taken", [(1281, 1024, True), (2562, 1024, True), (5121, 512, True), (32 * 1281, 1024, False)])
def test_small_launch_weight_gradient_pairs(M, D, taken):
    """octmae_wgrad_accum_pair on 128 x 128 tiles (gemm128d_wgrad_kernel) where the cost model takes it -- short reductions on few
    tiles --: both gradients of both pairs against an fp64 product on top of a non-zero accumulator, the qkv bias gradient (a launch
    of its own on this path), twice in a row (accumulation), and the same numbers as the 256-tile pair kernel to fp32 rounding."""
    g = torch.Generator().manual_seed(M + D)
    y1 = bf(torch.randn(M, D, generator=g)).to(DEV); act = bf(torch.randn(M, 4 * D, generator=g)).to(DEV)
    dpre = bf(torch.randn(M, 4 * D, generator=g)).to(DEV); o = bf(torch.randn(M, D, generator=g)).to(DEV)
    dqkv = bf(torch.randn(M, 3 * D, generator=g)).to(DEV); d3 = bf(torch.randn(M, D, generator=g)).to(DEV)

    def run():
        gw2 = torch.ones(D, 4 * D, device=DEV); gw1 = torch.ones(4 * D, D, device=DEV)
        gwp = torch.ones(D, D, device=DEV); gwq = torch.ones(3 * D, D, device=DEV); gbq = torch.ones(3 * D, device=DEV)
        for _ in range(2):
            ops.linear_wgrad_accum_pair((d3, act, gw2, None), (dpre, y1, gw1, None))
            ops.linear_wgrad_accum_pair((d3, o, gwp, None), (dqkv, y1, gwq, gbq))
        return gw2, gw1, gwp, gwq, gbq

    n0 = ops.set_option("gemm_small_wgrad_launches", 0)
    got = run()
    n1 = ops.set_option("gemm_small_wgrad_launches", 0)
    assert n1 - n0 == (4 if taken else 0), (n0, n1)
    prev = ops.set_option("gemm_small", 0)
    try:
        ref = run()
        assert ops.set_option("gemm_small_wgrad_launches", 0) == n1
    finally:
        ops.set_option("gemm_small", prev)
    exact = (1 + 2 * d3.double().t() @ act.double(), 1 + 2 * dpre.double().t() @ y1.double(), 1 + 2 * d3.double().t() @ o.double(),
             1 + 2 * dqkv.double().t() @ y1.double(), 1 + 2 * dqkv.double().sum(0))
    for a_, r_, e_ in zip(got, ref, exact):
        assert rel(a_, e_) < 1e-5 and rel(a_, r_) < 2e-6
