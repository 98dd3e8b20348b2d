"""Stream-K over the partial last round of the 256-tile forward / dgrad GEMMs (csrc/gemm.hip: gemm256p_sk_kernel; octmae_gemm_bf16_ws
and the workspace arguments of the fused entry points): same results as the plain launch up to the order of the fp32 additions of the
split tiles, every fused epilogue, deterministic, and free of hand-off races under memory pressure.

Shapes: fewer tiles than CUs (small batches: every tile split), a partial last round inside the planner's window (the per-rank shape
of an 8-GPU step in miniature), and a nearly empty last round taken only with gemm_streamk = 2."""
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
    return ops.set_option("gemm_streamk_launches", 0)


class plain:
    """Context: the same calls without the stream-K workspace (the plain one-tile-per-workgroup launch)."""

    def __enter__(self):
        self.prev, ops.STREAMK = ops.STREAMK, False

    def __exit__(self, *a):
        ops.STREAMK = self.prev


@pytest.fixture(autouse=True)
def _lend_the_workspace():
    """The workspace is not lent by default (stream-K measured as no gain in the step: ops.STREAMK); these tests lend it."""
    prev, ops.STREAMK = ops.STREAMK, True
    yield
    ops.STREAMK = prev


# (M, N, K, streamk option)
SHAPES = [
    (1281, 1024, 4096, 1),            # 6 x 4 = 24 tiles on 256 CUs, 64 k-tiles: every tile split four ways (fc2 forward of one volume)
    (2562, 512, 3072, 1),             # 11 x 2 = 22 tiles, 48 k-tiles each: four workgroups per tile
    (1281, 1024, 1024, 1),            # 16 k-tiles: too short to split ... plain launch
    (256 * 80 + 100, 1024, 512, 1),   # 324 tiles = 1.27 rounds: the last 68 tiles shared by all CUs
    (256 * 70 + 9, 1024, 1024, 1),    # 284 tiles = 1.11 rounds: below the planner's window at 256 CUs ... plain launch
    (256 * 64 + 17, 1024, 2048, 2),   # 260 tiles, 32 k-tiles: the 4 left-over tiles shared by 32 workgroups, only when forced
    (256 * 64 + 17, 1024, 256, 2),    # ... and not even then with 4 k-tiles each (every workgroup gets >= 4 k-tiles)
]


def planned(nt, ktiles, ncu, opt):
    """The planner of csrc/gemm.hip (sk_plan) restated: is the launch taken by the stream-K kernel?"""
    if opt == 0 or ktiles < 4:
        return False
    if nt >= ncu:
        rem = nt % ncu
        g = min(ncu, rem * ktiles // 4)
        if g < 2 * rem or rem == 0:
            return False
        return True if opt == 2 else (rem * 8 >= ncu and rem * 10 <= ncu * 8)
    return 4 * nt <= ncu and ktiles // 4 >= 12


@pytest.mark.parametrize("M,N,K,opt", SHAPES)
def test_streamk_forward_epilogues_equal_the_plain_launch(M, N, K, opt):
    g = torch.Generator().manual_seed(M + N + K)
    x = bf(torch.randn(M, K, generator=g)).to(DEV)
    w = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    sc = torch.tensor([0.0, 1.25, 2.0], device=DEV)[torch.randint(0, 3, (M,), generator=g).to(DEV)]
    prev = ops.set_option("gemm_streamk", opt)
    try:
        n0 = launches()
        got = {"f32": ops.linear_fwd(x, w, b, "f32"), "bf16": ops.linear_fwd(x, w, b, "bf16"), "gelu": ops.linear_fwd(x, w, b, "gelu"),
               "resid": ops.linear_fwd(x, w, b, "resid", res=res),
               "rowscale": ops.linear_fwd(x, w, b, "resid", res=res, rowscale=sc, rows_per_scale=1)}
        again = ops.linear_fwd(x, w, b, "resid", res=res)
        taken = launches() - n0
        with plain():
            ref = {"f32": ops.linear_fwd(x, w, b, "f32"), "bf16": ops.linear_fwd(x, w, b, "bf16"), "gelu": ops.linear_fwd(x, w, b, "gelu"),
                   "resid": ops.linear_fwd(x, w, b, "resid", res=res),
                   "rowscale": ops.linear_fwd(x, w, b, "resid", res=res, rowscale=sc, rows_per_scale=1)}
            assert launches() - n0 == taken
    finally:
        ops.set_option("gemm_streamk", prev)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nt = -(-M // 256) * (N // 256)
    assert taken == (6 if planned(nt, K // 64, ncu, opt) else 0), (taken, nt, ncu)
    assert torch.equal(again, got["resid"])                                   # deterministic
    exact = x.double() @ w.double().t() + b.double()
    assert rel(got["f32"], exact) < 2e-6 and rel(got["f32"], ref["f32"]) < 1e-6
    assert rel(got["resid"], ref["resid"]) < 1e-6 and rel(got["rowscale"], ref["rowscale"]) < 1e-6
    assert torch.equal(got["rowscale"][sc == 0], res[sc == 0])
    for a_, r_ in ((got["bf16"], ref["bf16"]), (got["gelu"][0], ref["gelu"][0]), (got["gelu"][1], ref["gelu"][1])):
        assert rel(a_, r_) < 2e-4                                             # a few 1-ulp flips of the 16-bit rounding
        assert float((a_ != r_).float().mean()) < 2e-3
    assert rel(got["bf16"], exact) < 3e-3


@pytest.mark.parametrize("M,N,K,opt", SHAPES[:2] + SHAPES[3:4])
def test_streamk_dgrad_epilogues_equal_the_plain_launch(M, N, K, opt):
    """dgrad (weight k-strided): plain, x GELU' with the bias-gradient column sums, and with the attention delta."""
    g = torch.Generator().manual_seed(M * 3 + N)
    dy = bf(torch.randn(M, K, generator=g)).to(DEV)                # [M, "N_out" = K here]: dx[M, N] = dy[M, K] @ w[K, N]
    w = bf(torch.randn(K, N, generator=g) * K ** -0.5).to(DEV)
    pre = bf(torch.randn(M, N, generator=g)).to(DEV)
    o = bf(torch.randn(M, N, generator=g)).to(DEV)
    H, HD = N // 64, 64

    def run():
        cs = torch.zeros(N, device=DEV)
        out = [ops.linear_dgrad(dy, w), ops.linear_dgrad(dy, w, pre=pre, colsum=cs), cs]
        dx, delta = ops.linear_dgrad_delta(dy, w, o, H, HD)
        return out + [dx, delta]

    prev = ops.set_option("gemm_streamk", opt)
    try:
        n0 = launches()
        got = run()
        taken = launches() - n0
        got2 = run()
        with plain():
            ref = run()
    finally:
        ops.set_option("gemm_streamk", prev)
    assert taken == 3, taken
    for a_, b_ in zip(got, got2):
        assert a_ is None or torch.equal(a_, b_)
    exact = dy.double() @ w.double()
    assert rel(got[0], exact) < 3e-3 and rel(got[0], ref[0]) < 2e-4
    assert rel(got[1], ref[1]) < 2e-4 and rel(got[2], ref[2]) < 1e-4
    assert rel(got[3], ref[3]) < 2e-4 and got[4] is not None and rel(got[4], ref[4]) < 1e-4


def test_streamk_hand_off_is_race_free_under_memory_pressure():
    """Back-to-back launches of three split shapes (the workspace and its flags are re-used with a new generation each time) while a
    second stream saturates HBM: every result must be bit-identical to the first (a partial read before it was complete, or a flag
    seen before its payload, would show as a different sum)."""
    g = torch.Generator().manual_seed(5)
    cases = []
    for M, N, K in ((1281, 1024, 4096), (256 * 80 + 100, 1024, 512), (2562, 512, 3072)):
        x = bf(torch.randn(M, K, generator=g)).to(DEV)
        w = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        cases.append((x, w, b, ops.linear_fwd(x, w, b, "f32").clone()))
    big = torch.empty(1 << 28, dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    stop = 0
    prev = ops.set_option("gemm_streamk", 1)
    n0 = launches()
    for it in range(120):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                big.add_(1.0)
        x, w, b, first = cases[it % 3]
        y = ops.linear_fwd(x, w, b, "f32")
        if not torch.equal(y, first):
            stop += 1
    torch.cuda.synchronize()
    ops.set_option("gemm_streamk", prev)
    assert launches() - n0 == 120
    assert stop == 0, f"{stop} of 120 launches differed from the first"
