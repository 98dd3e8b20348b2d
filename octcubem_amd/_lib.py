"""ctypes binding of liboctmae.so (the C ABI declared in include/octmae.h).

There is NO fallback: if the shared library is missing or a symbol is absent the import of any
compute entry point raises.  PyTorch is used only for device memory, streams and autograd plumbing.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCTMAE_LIB", os.path.join(_HERE, "liboctmae.so"))   # OCTMAE_LIB: A/B a variant build

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argtypes ; every function returns int (0 ok, <0 bad argument, >0 hipError_t)
SIGNATURES = {
    "octmae_abi_version": [],
    "octmae_lp_dtype": [],
    "octmae_set_option": [C.c_char_p, _i],
    "octmae_gemm_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "octmae_gemm_bf16_ws": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp],
    "octmae_gemm_split_ws_kib": [],
    "octmae_gemm_small_plan": [_i, _i, _i, _i, _i, _i, _vp, _vp],
    "octmae_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "octmae_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "octmae_layernorm_bwd_ws_floats": [_i, _i],
    "octmae_attn_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "octmae_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "octmae_attn_bwd_fused_ws_kib": [_i, _i, _i, _i],
    "octmae_attn_bwd_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "octmae_attn_bwd_fused_delta": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "octmae_attn_bwd_rowconst": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "octmae_attn_bwd_dq_rowconst": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "octmae_attn_bwd_dq": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "octmae_attn_bwd_dkv": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "octmae_random_masking_ids": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "octmae_cast_f32_bf16": [_vp, _vp, _ll, _vp],
    "octmae_cast_rowscale_f32_bf16": [_vp, _vp, _vp, _ll, _i, _i, _vp],
    "octmae_linear_resid_rowscale": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp],
    "octmae_colsum_accum": [_vp, _i, _vp, _i, _i, _i, _vp],
    "octmae_dgelu_colsum_ws_rows": [_i],
    "octmae_linear_dgrad_dgelu": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp],
    "octmae_linear_dgrad_delta": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp],
    "octmae_wgrad_split_plan": [_i, _i, _i, _vp, _vp],
    "octmae_wgrad_accum_pair": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "octmae_patch_gather": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "octmae_enc_assemble": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "octmae_dec_assemble": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "octmae_gather_rows_cast": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "octmae_scatter_add_rows": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "octmae_dec_assemble_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "octmae_mse_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "octmae_mse_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "octmae_mt_chunk_elems": [],
    "octmae_mt_sumsq": [_vp, _vp, _vp, _i, _vp, _vp],
    "octmae_mt_finish_norm": [_vp, _i, _f, _vp, _vp, _vp],
    "octmae_mt_adamw": [_vp, _vp, _vp, _i, _vp, _f, _f, _f, _f, _f, _i, _vp],
    "octmae_mt_adamw_fused": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _vp],
    "octmae_comm_available": [],
    "octmae_comm_unique_id": [_vp],
    "octmae_comm_init": [_vp, _vp, _i, _i, _i],
    "octmae_comm_destroy": [_vp],
    "octmae_comm_rank": [_vp],
    "octmae_comm_world": [_vp],
    "octmae_comm_allreduce_async": [_vp, _vp, _ll, _i, _i, _vp],
    "octmae_comm_broadcast_async": [_vp, _vp, _ll, _i, _i, _vp],
    "octmae_comm_allgather_async": [_vp, _vp, _vp, _ll, _i, _vp],
    "octmae_comm_reduce_scatter_async": [_vp, _vp, _vp, _ll, _i, _i, _vp],
    "octmae_comm_wait": [_vp, _vp],
    "octmae_comm_stream": [_vp, _vp],
    "octmae_probe_mfma32": [_vp, _vp, _vp, _vp],
    "octmae_probe_trread": [_vp, _vp, _vp],
}

_lib = None
_on_load = []       # callbacks run once, right after the library has been loaded and bound (ops: operand-type check)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "octmae.h")


def expected_abi_version() -> int:
    """OCTMAE_ABI_VERSION as include/octmae.h defines it -- the single place the number is written."""
    import re
    m = re.search(r"^#define\s+OCTMAE_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read(), re.M)
    if m is None:
        raise OctmaeError(f"{HEADER_PATH} does not define OCTMAE_ABI_VERSION")
    return int(m.group(1))


class OctmaeError(RuntimeError):
    pass


def load():
    """Load liboctmae.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OctmaeError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C octcubem_amd/csrc).  There is no CPU or eager fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:        # e.g. a copied .so on a box without the HIP runtime it links against
        raise OctmaeError(f"{LIB_PATH} could not be loaded: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise OctmaeError(f"liboctmae.so does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if os.path.exists(HEADER_PATH) and lib.octmae_abi_version() != expected_abi_version():
        raise OctmaeError(f"{LIB_PATH} reports ABI {lib.octmae_abi_version()}, include/octmae.h declares "
                          f"{expected_abi_version()}: rebuild (make -C octcubem_amd/csrc)")
    _lib = lib
    for cb in list(_on_load):
        cb(lib)
    return lib


_FN = {}


def call(name, *args):
    """Invoke a C-ABI entry point and turn its status code into an exception."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        kind = ("bad argument" if rc == -1 else "unsupported combination" if rc == -2 else "librccl not found" if rc == -3
                else f"ncclResult_t {rc - 10000}" if rc >= 10000 else f"hipError_t {rc}")
        raise OctmaeError(f"{name} failed: {kind} (rc={rc})")
    return rc
