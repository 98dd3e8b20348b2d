import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the sampled guard behind the arena's skipped operand cast checks at EVERY forward in the test suites (octcubem_amd/arena.py;
# must be set before the package is imported)
os.environ.setdefault("OCTMAE_CHECK_LP", "1")


def pytest_configure(config):
    # The CPU oracle runs inside many GPU tests, beside child processes that run it too (tests/test_gpu_f16_parity.py, the multi-rank
    # children of tests/test_gpu_comm.py).  On the 256-thread hosts of the GPU pool torch's default -- one thread per host thread, in
    # every process -- is oversubscribed: the oracle's ViT-L forward takes 106 s on 256 threads against 8.8 s on 32
    # (profiles/r05_cpu_threads.txt), and one full-size test of this suite took 220 s waiting for the CPU.
    try:
        import torch
        torch.set_num_threads(min(32, os.cpu_count() or 1))
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "small_batch_rule: runs with ops.attn_bwd_use_fused's fill rule active (see conftest.py)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fused_attention_backward_at_test_sizes(request):
    """The GPU tests run models of 2-3 samples and 2-3 heads.  ops.attn_bwd_use_fused would hand such a backward to the dQ + dK/dV pair
    (B * H workgroups do not fill the chip) and the production kernels of the benchmark -- the fused single-pass backward with the
    delta from the proj dgrad's epilogue -- would never run inside a model test: the fill rule is switched off for the tests (the
    fused form wherever ATTN_BWD_FUSED says so, as before the rule existed) except in those that mark themselves
    ``small_batch_rule`` and test the rule itself."""
    if "gpu" not in request.keywords or "small_batch_rule" in request.keywords:
        yield
        return
    from octcubem_amd import ops
    old = ops.ATTN_BWD_FUSED_MIN_FILL
    ops.ATTN_BWD_FUSED_MIN_FILL = 0.0
    yield
    ops.ATTN_BWD_FUSED_MIN_FILL = old


# ---- measured-error ledger: parity(label, value, bound) asserts value <= bound AND records the measurement; at session end the
# ledger is printed (-s) and written to gpurun_out/parity_measured.json so that bounds can be kept at "measured x 1.5".
_LEDGER = []


def parity(label, value, bound):
    value = float(value)
    _LEDGER.append((label, value, float(bound)))
    assert value <= bound, f"{label}: measured {value:.3e} > bound {bound:.3e}"


@pytest.fixture(scope="session", autouse=True)
def _parity_ledger():
    yield
    if not _LEDGER:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.json"), "w") as f:
            json.dump([{"label": l, "measured": v, "bound": b} for l, v, b in _LEDGER], f, indent=1)
    except OSError:
        pass
    print("\n[parity ledger] " + "; ".join(f"{l} {v:.2e} (<= {b:.1e})" for l, v, b in _LEDGER))


def pytest_collection_finish(session):
    """Modules that run helper processes beside the session (tests/test_gpu_f16_parity.py: one ledger per build of the library) start
    them here -- once the selection is known, and only when at least one of their tests is in it (`-m "not gpu"`, `-k ...` and
    `--collect-only` start nothing)."""
    if session.config.option.collectonly:
        return
    seen = set()
    for item in session.items:
        mod = getattr(item, "module", None)
        hook = getattr(mod, "start_children", None)
        if hook is not None and id(mod) not in seen:
            seen.add(id(mod))
            hook()
