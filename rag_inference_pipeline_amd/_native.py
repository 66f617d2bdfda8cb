"""Loader and builder for librag_amd.so (the C ABI declared in include/rag_amd.h).

The library is built IN-TREE (rag_inference_pipeline_amd/csrc/librag_amd.so) with hipcc for
gfx950 and bound with ctypes — plain pointers and sizes, no torch types.  There is no fallback:
if the library is missing or was not built, importing the product components raises.
"""

from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_PATH = os.environ.get("RAG_AMD_LIB") or os.path.join(_CSRC, "librag_amd.so")  # override: experiments only
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "rag_amd.h")
_SOURCES = ["rag_amd.hip", "rag_bert.hip", "rag_lz4.cpp", "rag_text.cpp"]
_DEPS = ["rag_amd.hip", "flat_kernels.hip.h", "rag_bert.hip", "bert_kernels.hip.h", "gemm_wl.hip.h", "gemm_wt.hip.h", "bert_tiled.hip.h", "rag_common.h",
         "rag_lz4.cpp", "rag_text.cpp"]

RAG_OK = 0
RAG_ERR_INVALID_ARG = 1
RAG_ERR_NO_DEVICE = 2
RAG_ERR_HIP = 3
RAG_ERR_OOM = 4
RAG_ERR_UNSUPPORTED = 5
RAG_ERR_STATE = 6

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1

SEARCH_DEFAULT, SEARCH_EXACT_ONE_PASS, SEARCH_DEFER_FALLBACK = 0, 1, 2  # include/rag_amd.h RAG_SEARCH_*

# rag_bert_config (include/rag_amd.h)
ACT_GELU, ACT_GELU_TANH, ACT_RELU = 1, 2, 3
HEAD_NONE, HEAD_BERT, HEAD_ROBERTA = 0, 1, 2
BERT_OUT_MEAN, BERT_OUT_CLS, BERT_OUT_LOGITS, BERT_OUT_PROBS, BERT_OUT_HIDDEN = 0, 1, 2, 3, 4


class BertConfigStruct(C.Structure):
    _fields_ = [(name, C.c_int32) for name in (
        "vocab_size", "hidden", "n_layers", "n_heads", "intermediate", "max_positions", "type_vocab",
        "pos_offset", "act", "head", "n_labels")] + [("ln_eps", C.c_float), ("gemm_mode", C.c_int32)]


_lib = None
_lock = threading.Lock()


class NativeLibraryError(RuntimeError):
    """librag_amd.so is missing, stale or failed to load."""


class RagAmdError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code: int, message: str) -> None:
        super().__init__(f"rag_amd status {code}: {message}")
        self.code = code


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise NativeLibraryError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, f) for f in _DEPS] + [HEADER_PATH]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into csrc/librag_amd.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [
        _hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
        "-Wall", "-Wno-unused-function",
        "-o", LIB_PATH + ".tmp",
    ] + [os.path.join(_CSRC, s) for s in _SOURCES]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise NativeLibraryError(f"hipcc failed:\n{' '.join(cmd)}\n{proc.stderr}")
    if verbose and proc.stderr:
        print(proc.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def _declare(lib: C.CDLL) -> None:
    f32p, i64p, vp = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_void_p
    i32p = C.POINTER(C.c_int32)
    sig = {
        "rag_abi_version": (C.c_int, []),
        "rag_device_count": (C.c_int, []),
        "rag_last_error": (C.c_char_p, []),
        "rag_index_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]),
        "rag_index_destroy": (C.c_int, [vp]),
        "rag_index_reserve": (C.c_int, [vp, C.c_int64]),
        "rag_index_add": (C.c_int, [vp, f32p, C.c_int64]),
        "rag_index_add_device": (C.c_int, [vp, vp, C.c_int64, vp]),
        "rag_index_add_synthetic": (C.c_int, [vp, C.c_int64, C.c_uint64, C.c_int64]),
        "rag_index_ntotal": (C.c_int64, [vp]),
        "rag_index_dim": (C.c_int32, [vp]),
        "rag_index_metric": (C.c_int32, [vp]),
        "rag_index_set_id_offset": (C.c_int, [vp, C.c_int64]),
        "rag_index_search": (C.c_int, [vp, f32p, C.c_int32, C.c_int32, f32p, i64p]),
        "rag_index_search_device": (C.c_int, [vp, vp, C.c_int32, C.c_int32, vp, vp, vp]),
        "rag_index_search_device_ex": (C.c_int, [vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp]),
        "rag_index_search_device_host_out": (C.c_int, [vp, vp, C.c_int32, C.c_int32, f32p, i64p, vp]),
        "rag_index_get_rows": (C.c_int, [vp, C.c_int64, C.c_int64, f32p]),
        "rag_index_profile_enable": (C.c_int, [vp, C.c_int32]),
        "rag_index_profile": (C.c_int, [vp, C.POINTER(C.c_double), i64p, C.c_int32]),
        "rag_index_max_k": (C.c_int32, [C.c_int32, C.c_int32]),
        "rag_lz4_compress_bound": (C.c_int64, [C.c_int64]),
        "rag_lz4_block_compress": (C.c_int64, [vp, C.c_int64, vp, C.c_int64]),
        "rag_xxh32": (C.c_uint32, [vp, C.c_int64, C.c_uint32]),
        "rag_hash_encode_pairs": (C.c_int64, [vp, i64p, vp, i64p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int64]),
        "rag_index_set_screening": (C.c_int, [vp, C.c_int32]),
        "rag_index_screening": (C.c_int32, [vp]),
        "rag_index_screen_stats": (C.c_int, [vp, i64p, i64p, C.POINTER(C.c_double), C.c_int32]),
        "rag_merge_topk_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            vp, vp, vp, vp, vp]),
        "rag_merge_topk_packed_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                   vp, C.c_int64, C.c_int64, vp, vp, vp]),
        "rag_merge_topk_packed_flagged_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                           vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, vp]),
        "rag_bert_weight_count": (C.c_int32, [C.POINTER(BertConfigStruct)]),
        "rag_bert_create": (C.c_int, [C.POINTER(BertConfigStruct), C.POINTER(vp), C.c_int32, C.c_int32,
                                      C.POINTER(vp)]),
        "rag_bert_destroy": (C.c_int, [vp]),
        "rag_bert_forward": (C.c_int, [vp, i32p, i32p, i32p, C.c_int32, C.c_int32, C.c_int32, f32p]),
        "rag_bert_range_events": (C.c_int, [vp, i64p, i32p]),
        "rag_bert_set_background": (C.c_int, [vp, C.c_int32]),
        "rag_bert_set_cu_budget": (C.c_int, [vp, C.c_int32]),
        "rag_bert_set_stream": (C.c_int, [vp, C.c_void_p]),
        "rag_stream_wait": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p]),
        "rag_device_cu_count": (C.c_int, [C.c_int32, C.POINTER(C.c_int32)]),
        "rag_index_set_cu_budget": (C.c_int, [vp, C.c_int32]),
        "rag_stream_create_masked": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
        "rag_stream_destroy": (C.c_int, [C.c_int32, C.c_void_p]),
        "rag_bert_forward_to_device": (C.c_int, [vp, i32p, i32p, i32p, C.c_int32, C.c_int32, C.c_int32, vp, vp,
                                                 C.POINTER(vp)]),
        "rag_bert_forward_device": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args


def lib() -> C.CDLL:
    """The loaded library.  Raises NativeLibraryError if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeLibraryError(
                    f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(there is no CPU fallback)")
            # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  Two runtimes in one process do
            # not share a device ("No HIP GPUs are available" from whichever initialises second), so
            # torch's copy is mapped first and librag_amd.so binds to it by SONAME.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            try:
                handle = C.CDLL(LIB_PATH)
            except OSError as e:  # e.g. libamdhip64 missing
                raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
            _declare(handle)
            if handle.rag_abi_version() != 1:
                raise NativeLibraryError("librag_amd.so ABI version mismatch; rebuild")
            _lib = handle
        return _lib


def check(status: int) -> None:
    if status != RAG_OK:
        msg = lib().rag_last_error()
        raise RagAmdError(status, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    return int(lib().rag_device_count())
