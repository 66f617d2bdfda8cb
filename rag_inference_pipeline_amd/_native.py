"""Loader and builder for librag_amd.so (the C ABI declared in include/rag_amd.h).

The library is built IN-TREE (rag_inference_pipeline_amd/csrc/librag_amd.so) with hipcc for
gfx950 and bound with ctypes — plain pointers and sizes, no torch types.  There is no fallback:
if the library is missing or was not built, importing the product components raises.
"""

from __future__ import annotations

import ctypes as C
import fcntl
import hashlib
import os
import re
import shutil
import subprocess
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_PATH = os.environ.get("RAG_AMD_LIB") or os.path.join(_CSRC, "librag_amd.so")  # override: experiments only
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "rag_amd.h")
_SOURCES = ["rag_amd.hip", "rag_bert.hip", "rag_comm.hip", "rag_digest.cpp", "rag_lz4.cpp", "rag_text.cpp"]
_SOURCE_SUFFIXES = (".hip", ".h", ".cpp")

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


def header_abi_version() -> int:
    """RAG_AMD_ABI_VERSION as the header beside this package declares it (the number the loaded library must report)."""
    with open(HEADER_PATH, "r", encoding="utf-8") as f:
        m = re.search(r"^#define\s+RAG_AMD_ABI_VERSION\s+(\d+)\s*$", f.read(), re.M)
    if not m:
        raise NativeLibraryError(f"{HEADER_PATH} does not define RAG_AMD_ABI_VERSION")
    return int(m.group(1))


def source_digest() -> str:
    """Hex SHA-256 over include/rag_amd.h and every source file of csrc/ (name order, names included).  The build passes
    it to the compiler; a library reports it through rag_source_digest().  Content, not mtime: a working tree that was
    copied (gpurun snapshots, rsync, a checkout) keeps no useful timestamps."""
    h = hashlib.sha256()
    files = [HEADER_PATH] + sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(_SOURCE_SUFFIXES))
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def built_digest(path: str | None = None) -> str | None:
    """The digest a built library carries, read from the file WITHOUT loading it (a stale library must not be mapped:
    dlopen of a second build of the same path returns the first).  None: no library, or one that predates the digest."""
    path = path or LIB_PATH
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(rb"rag-amd-source-digest:([0-9a-f]{64})", blob)
    return m.group(1).decode() if m else None


def needs_build() -> bool:
    return built_digest() != source_digest()


def _tu_digest(src: str, flags: list[str]) -> str:
    """Digest of one translation unit: its text, the text of every file it includes (transitively, "quoted" includes
    only) and the compiler flags — the key of its cached object file."""
    seen: dict[str, bytes] = {}

    def visit(path: str) -> None:
        path = os.path.normpath(path)
        if path in seen or not os.path.exists(path):
            return
        with open(path, "rb") as f:
            text = f.read()
        seen[path] = text
        for inc in re.findall(rb'^\s*#\s*include\s+"([^"]+)"', text, re.M):
            visit(os.path.join(os.path.dirname(path), inc.decode()))

    visit(src)
    h = hashlib.sha256(" ".join(flags).encode())
    for path in sorted(seen):
        h.update(os.path.basename(path).encode() + b"\0" + seen[path] + b"\0")
    return h.hexdigest()[:20]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into csrc/librag_amd.so (cross-compiles without a GPU).  Translation units
    are compiled side by side and their objects cached by content (csrc/build/), so an edit recompiles only what
    includes it.  Safe to call from several processes at once (one rank per GPU): a file lock lets one of them
    compile, the others find it done."""
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor

    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():   # another process built it while this one waited
                return LIB_PATH
            digest = source_digest()
            hipcc = _hipcc()
            flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
            objdir = os.path.join(_CSRC, "build")
            os.makedirs(objdir, exist_ok=True)

            def compile_one(name: str) -> str:
                src = os.path.join(_CSRC, name)
                # only rag_digest.cpp carries the whole-library digest
                extra = [f'-DRAG_AMD_SOURCE_DIGEST="rag-amd-source-digest:{digest}"'] if name == "rag_digest.cpp" else []
                obj = os.path.join(objdir, f"{name}.{_tu_digest(src, flags + extra)}.o")
                if force or not os.path.exists(obj):
                    cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj + f".tmp.{os.getpid()}"]
                    proc = subprocess.run(cmd, capture_output=True, text=True)
                    if proc.returncode != 0:
                        raise NativeLibraryError(f"hipcc failed:\n{' '.join(cmd)}\n{proc.stderr}")
                    if verbose and proc.stderr:
                        print(proc.stderr)
                    os.replace(obj + f".tmp.{os.getpid()}", obj)
                return obj

            with ThreadPoolExecutor(max_workers=len(_SOURCES)) as pool:
                objs = list(pool.map(compile_one, _SOURCES))
            tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"]
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise NativeLibraryError(f"link failed:\n{' '.join(cmd)}\n{proc.stderr}")
            os.replace(tmp, LIB_PATH)
            keep = set(objs)
            for f in os.listdir(objdir):   # objects of older source states
                path = os.path.join(objdir, f)
                if path not in keep and f.endswith(".o"):
                    os.remove(path)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def signatures() -> dict:
    """name -> (restype, argtypes) for every entry point of include/rag_amd.h.  tests/test_native_abi.py parses the
    header's prototypes and holds this table to them (argument count and kind), so the two cannot drift apart."""
    f32p, i64p, vp = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_void_p
    i32p = C.POINTER(C.c_int32)
    sig = {
        "rag_abi_version": (C.c_int, []),
        "rag_source_digest": (C.c_char_p, []),
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
        "rag_ivf_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]),
        "rag_ivf_destroy": (C.c_int, [vp]),
        "rag_ivf_set_lists": (C.c_int, [vp, f32p, C.c_int64, f32p, i64p, i64p]),
        "rag_ivf_ntotal": (C.c_int64, [vp]),
        "rag_ivf_nlist": (C.c_int64, [vp]),
        "rag_ivf_two_stage": (C.c_int32, [vp]),
        "rag_ivf_screen_stats": (C.c_int, [vp, i64p, i64p, C.POINTER(C.c_double), C.c_int32]),
        "rag_ivf_search": (C.c_int, [vp, f32p, C.c_int32, C.c_int32, C.c_int32, f32p, i64p]),
        "rag_ivf_search_device": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]),
        "rag_ivf_search_device_host_out": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, f32p, i64p, vp]),
        "rag_comm_runtime": (C.c_int, [C.c_char_p, i32p]),
        "rag_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
        "rag_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]),
        "rag_comm_destroy": (C.c_int, [vp]),
        "rag_comm_rank": (C.c_int32, [vp]),
        "rag_comm_world": (C.c_int32, [vp]),
        "rag_comm_all_gather_device": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
        "rag_comm_broadcast_device": (C.c_int, [vp, vp, C.c_int64, C.c_int32, vp]),
        "rag_comm_request_device": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32, vp, C.c_uint64, vp]),
        "rag_comm_wait_head": (C.c_int, [vp, C.c_uint64, C.c_int64, i64p]),
        "rag_pack_layout": (C.c_int, [C.c_int32, C.c_int32, i64p, i64p, i64p]),
        "rag_index_search_gather_device": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp,
                                                     vp, vp]),
        "rag_ivf_search_gather_device": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp,
                                                   vp, vp]),
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
    return sig


def _declare(lib: C.CDLL) -> None:
    for name, (res, args) in signatures().items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args


def _ensure_current() -> None:
    """The library on disk must have been compiled from the sources beside it.  A stale one (the sources moved on and
    nobody rebuilt: *.so files travel with a working tree but are not tracked) is rebuilt when hipcc is at hand,
    refused otherwise — never loaded: its entry points would be called through a newer argument table."""
    if os.environ.get("RAG_AMD_LIB"):
        return  # an experiment build named explicitly: the caller answers for it
    if os.path.exists(LIB_PATH) and not needs_build():
        return
    state = "missing" if not os.path.exists(LIB_PATH) else "stale (built from other sources than the ones beside it)"
    if os.environ.get("RAG_AMD_NO_REBUILD") == "1":
        raise NativeLibraryError(f"{LIB_PATH} is {state} and RAG_AMD_NO_REBUILD=1: run "
                                 "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
    try:
        build()
    except NativeLibraryError as e:
        raise NativeLibraryError(f"{LIB_PATH} is {state} and could not be rebuilt: {e} (there is no CPU fallback)") from e


def lib() -> C.CDLL:
    """The loaded library.  Raises NativeLibraryError if it is missing or stale and cannot be rebuilt."""
    global _lib
    with _lock:
        if _lib is None:
            _ensure_current()
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
            want, have = header_abi_version(), handle.rag_abi_version()
            if have != want:
                raise NativeLibraryError(f"librag_amd.so reports ABI version {have}, include/rag_amd.h declares {want}; rebuild")
            _lib = handle
        return _lib


def rccl_library_path() -> str | None:
    """The librccl a PyTorch-ROCm process has already mapped (torch/lib/librccl.so), for rag_comm_runtime: one RCCL and
    one HIP runtime per process.  None: let the library search by itself."""
    try:
        import torch
    except ImportError:
        return None
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else None


def check(status: int) -> None:
    if status != RAG_OK:
        msg = lib().rag_last_error()
        raise RagAmdError(status, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    return int(lib().rag_device_count())
