"""CPU: librag_amd.so loads, exports every symbol include/rag_amd.h declares, and refuses to
compute without a GPU (no compute calls are made here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from rag_inference_pipeline_amd import _native, index_io

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rag_amd.h")


@pytest.fixture(scope="module")
def lib():
    _native.build()
    return _native.lib()


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rag_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in rag_amd.h but not exported"


def test_abi_version_and_error_string(lib):
    assert lib.rag_abi_version() == _native.header_abi_version() >= 4
    assert isinstance(lib.rag_last_error(), bytes)


# ---- the ctypes table against the header's prototypes ---------------------------------------------------

def _kind_of_c(decl: str) -> str:
    """'ptr' | 'i32' | 'u32' | 'i64' | 'u64' | 'int' | 'f32' | 'f64' | 'void' for one C parameter or return type."""
    d = re.sub(r"\bconst\b", " ", decl).strip()
    if "*" in d:
        return "ptr"
    base = d.split()[0] if d.split() else ""
    return {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "int": "int", "float": "f32",
            "double": "f64", "void": "void"}[base]


def _kind_of_ctypes(t) -> str:
    if t is None:
        return "void"
    if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or issubclass(t, ctypes._Pointer):
        return "ptr"
    return {ctypes.c_int32: "i32", ctypes.c_uint32: "u32", ctypes.c_int64: "i64", ctypes.c_uint64: "u64",
            ctypes.c_int: "int", ctypes.c_float: "f32", ctypes.c_double: "f64"}[t]


def _header_prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(rag_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        args = [] if params in ("", "void") else [_kind_of_c(a) for a in params.split(",")]
        protos[name] = (_kind_of_c(ret), args)
    return protos


def test_ctypes_table_matches_every_prototype_of_the_header():
    """Argument COUNT and pointer / integer-width KIND of every entry point, header against _native.signatures().
    (c_int and int32_t are both 'a 32-bit int' on this ABI; the test keeps them apart anyway: the header is the
    contract.)  A prototype that gains an argument without the table following — or the other way round — fails here,
    on the CPU, instead of passing a stream handle where a flag pointer is expected."""
    protos = _header_prototypes()
    table = _native.signatures()
    assert set(protos) == set(table), (sorted(set(protos) - set(table)), sorted(set(table) - set(protos)))
    same_width = {"int": "i32"}
    for name, (ret, args) in protos.items():
        res, argtypes = table[name]
        got_ret, got_args = _kind_of_ctypes(res), [_kind_of_ctypes(t) for t in argtypes]
        norm = lambda k: same_width.get(k, k)  # noqa: E731
        assert norm(got_ret) == norm(ret), f"{name}: return {got_ret} in the table, {ret} in the header"
        assert len(got_args) == len(args), f"{name}: {len(got_args)} arguments in the table, {len(args)} in the header"
        for i, (g, w) in enumerate(zip(got_args, args)):
            assert norm(g) == norm(w), f"{name}: argument {i} is {g} in the table, {w} in the header"


def test_bert_config_struct_matches_the_header():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct rag_bert_config \{(.*?)\} rag_bert_config;", text, flags=re.S).group(1)
    fields = [(t, n) for t, n in re.findall(r"\b(int32_t|float|uint32_t|int64_t)\s+([a-z_0-9]+)\s*;", body)]
    want = [(n, {"int32_t": ctypes.c_int32, "float": ctypes.c_float}[t]) for t, n in fields]
    assert want == list(_native.BertConfigStruct._fields_)


def test_stale_library_is_detected_by_content_not_by_mtime(lib, tmp_path, monkeypatch):
    """The library carries the digest of the sources it was built from; the loader compares it with the sources beside
    it.  A copy of the tree with one source byte changed is stale whatever the timestamps say, and with no compiler
    at hand the loader refuses it instead of calling it through a newer argument table."""
    import shutil
    assert _native.built_digest() == _native.source_digest()
    assert lib.rag_source_digest().decode() == "rag-amd-source-digest:" + _native.source_digest()
    csrc = tmp_path / "csrc"
    shutil.copytree(_native._CSRC, csrc, ignore=shutil.ignore_patterns("build", "*.lock", "*.tmp.*"))
    monkeypatch.setattr(_native, "_CSRC", str(csrc))
    monkeypatch.setattr(_native, "LIB_PATH", str(csrc / "librag_amd.so"))
    assert not _native.needs_build()
    with open(csrc / "rag_common.h", "a") as f:
        f.write("// edited\n")
    os.utime(csrc / "rag_common.h", (0, 0))          # older than the library: mtime would say "fresh"
    assert _native.needs_build()
    monkeypatch.setenv("RAG_AMD_NO_REBUILD", "1")
    with pytest.raises(_native.NativeLibraryError, match="stale"):
        _native._ensure_current()
    monkeypatch.delenv("RAG_AMD_NO_REBUILD")
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    monkeypatch.setattr(_native.shutil, "which", lambda name: None)
    monkeypatch.setattr(_native.os.path, "exists", lambda p, _e=os.path.exists: False if p == "/opt/rocm/bin/hipcc" else _e(p))
    with pytest.raises(_native.NativeLibraryError, match="could not be rebuilt"):
        _native._ensure_current()


def test_pack_layout_is_one_definition(lib):
    from rag_inference_pipeline_amd.sharded import pack_layout
    for nq, k in ((32, 10), (1, 1), (7, 100), (32, 100), (3, 5)):
        s, f, b = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        assert lib.rag_pack_layout(nq, k, ctypes.byref(s), ctypes.byref(f), ctypes.byref(b)) == 0
        assert (s.value, f.value, b.value) == pack_layout(nq, k)
    assert lib.rag_pack_layout(0, 10, None, None, None) == _native.RAG_ERR_INVALID_ARG


def test_rccl_binds_at_run_time_and_argument_errors_are_status_codes(lib):
    """C1's entry points without a GPU: RCCL is found and bound (the copy torch has mapped, else ROCm's), every bad
    argument is a status code.  No communicator is created here (that needs a device)."""
    ver = ctypes.c_int32(0)
    rc = lib.rag_comm_runtime(None, ctypes.byref(ver))
    if rc != 0:
        pytest.skip("no librccl on this machine: " + lib.rag_last_error().decode())
    assert ver.value >= 20000
    assert lib.rag_comm_create(None, 0, 1, 0, ctypes.byref(ctypes.c_void_p())) == _native.RAG_ERR_INVALID_ARG
    ident = (ctypes.c_uint8 * 128)()
    assert lib.rag_comm_create(ident, 2, 2, 0, ctypes.byref(ctypes.c_void_p())) == _native.RAG_ERR_INVALID_ARG
    assert lib.rag_comm_destroy(None) == 0 and lib.rag_comm_world(None) == 0 and lib.rag_comm_rank(None) == -1
    assert lib.rag_comm_all_gather_device(None, None, None, 8, None) == _native.RAG_ERR_INVALID_ARG
    assert lib.rag_comm_request_device(None, None, None, 64, 0, None, 1, None) == _native.RAG_ERR_INVALID_ARG
    # the head mirror protocol is plain host memory: a posted head is seen, a missing one times out
    mirror = (ctypes.c_uint64 * 5)(1, 32, 10, 768, 0)
    head = (ctypes.c_int64 * 4)()
    assert lib.rag_comm_wait_head(mirror, 7, 2000, head) == _native.RAG_ERR_STATE
    mirror[4] = 7
    assert lib.rag_comm_wait_head(mirror, 7, 2000, head) == 0 and list(head) == [1, 32, 10, 768]


def test_invalid_arguments_are_status_codes_not_crashes(lib):
    h = ctypes.c_void_p()
    assert lib.rag_index_create(0, 0, 0, ctypes.byref(h)) == _native.RAG_ERR_INVALID_ARG
    assert lib.rag_index_create(64, 7, 0, ctypes.byref(h)) == _native.RAG_ERR_INVALID_ARG
    assert lib.rag_index_create(64, 0, 0, None) == _native.RAG_ERR_INVALID_ARG
    assert lib.rag_index_ntotal(None) == 0
    assert lib.rag_index_destroy(None) == _native.RAG_OK


def test_no_device_means_loud_failure_not_fallback(lib):
    if lib.rag_device_count() > 0:
        pytest.skip("a GPU is present")
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    with pytest.raises(_native.RagAmdError) as exc:
        FlatIndex(64)
    assert exc.value.code == _native.RAG_ERR_NO_DEVICE and "no CPU path" in str(exc.value)


def test_max_k_reports_lds_budget(lib):
    assert lib.rag_index_max_k(384, 32) == 240
    assert lib.rag_index_max_k(768, 32) == 112
    assert lib.rag_index_max_k(1024, 32) == 48
    assert lib.rag_index_max_k(4096, 32) == 112  # d > 1024 is scanned in 768-column chunks


def test_index_file_formats_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((37, 24), dtype=np.float32)
    for metric in (0, 1):
        p = tmp_path / f"m{metric}.faiss"
        index_io.write_flat_index(p, X, metric)
        rows, m = index_io.read_index_file(p)
        assert m == metric
        np.testing.assert_array_equal(rows, X)
        rows_mm, _ = index_io.read_index_file(p, mmap=True)
        np.testing.assert_array_equal(np.asarray(rows_mm), X)
    np.save(tmp_path / "x.npy", X)
    rows, m = index_io.read_index_file(tmp_path / "x.npy", default_metric=1)
    assert m == 1 and rows.shape == (37, 24)
    raw = tmp_path / "x.f32"
    X.tofile(raw)
    (tmp_path / "x.f32.json").write_text('{"d": 24, "ntotal": 37, "metric": "l2"}')
    rows, m = index_io.read_index_file(raw)
    assert m == 1
    np.testing.assert_array_equal(rows, X)
    with pytest.raises(FileNotFoundError, match="FAISS index not found"):
        index_io.read_index_file(tmp_path / "missing.bin")
    bad = tmp_path / "ivfpq.bin"
    bad.write_bytes(b"IwPQ" + b"\0" * 64)
    with pytest.raises(ValueError, match="compressed IVF index"):
        index_io.read_index_file(bad)
    # IndexIVFFlat (what the reference's generator writes): unpacked back into row order
    for sparse in (False, True):
        ivf = tmp_path / f"ivf{int(sparse)}.bin"
        index_io.write_ivfflat_index(ivf, X, nlist=5 if not sparse else 200, metric=1, sparse=sparse)
        rows, m = index_io.read_index_file(ivf)
        assert m == 1
        np.testing.assert_array_equal(rows, X)
    broken = tmp_path / "ivf_broken.bin"
    data = bytearray((tmp_path / "ivf0.bin").read_bytes())
    data[-8:] = (10 ** 6).to_bytes(8, "little")  # last id out of range
    broken.write_bytes(bytes(data))
    with pytest.raises(ValueError, match="permutation"):
        index_io.read_index_file(broken)
    trunc = tmp_path / "trunc.bin"
    trunc.write_bytes((tmp_path / "m0.faiss").read_bytes()[:-8])
    with pytest.raises(ValueError, match="truncated"):
        index_io.read_index_file(trunc)


def test_faiss_store_error_contract_without_gpu(tmp_path):
    from rag_inference_pipeline_amd.components.faiss_store import FAISSStore
    from rag_inference_pipeline_amd.config import PipelineSettings
    store = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=str(tmp_path / "nope.bin")))
    assert store.is_loaded is False and store.index_size == 0
    with pytest.raises(FileNotFoundError, match="FAISS index not found"):
        store.load()
    with pytest.raises(RuntimeError, match="not loaded"):
        store.search(np.zeros((1, 768), np.float32), 1)
    store.unload()  # no-op when not loaded


def test_product_package_never_touches_the_oracle_or_the_reference():
    """oracle/ is test infrastructure: nothing under rag_inference_pipeline_amd/ may import it, and
    nothing there may read the reference checkout at run time."""
    import pathlib
    pkg = pathlib.Path(__file__).resolve().parent.parent / "rag_inference_pipeline_amd"
    offenders = []
    for path in pkg.rglob("*.py"):
        text = path.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "/root/reference" in text:
            offenders.append(str(path))
    for path in list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        if "/root/reference" in path.read_text():
            offenders.append(str(path))
    assert not offenders, offenders


def test_lz4_block_compressor_under_address_and_ub_sanitizers(tmp_path):
    """The host-side C++ of the ABI, built for the CPU with -fsanitize=address,undefined and fuzzed
    (tests/native/lz4_fuzz.cpp: 20 000 buffers, exact-size destinations, too-small destinations refused).
    GPU sanitizers are not available; this is the part of the library a CPU sanitizer can see."""
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "lz4_fuzz")
    build = subprocess.run([gxx, "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17",
                            "-I", os.path.join(root, "include"), os.path.join(root, "tests", "native", "lz4_fuzz.cpp"),
                            os.path.join(root, "rag_inference_pipeline_amd", "csrc", "rag_lz4.cpp"), "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower() + build.stdout.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "fuzz ok" in run.stdout and "runtime error" not in run.stderr, run.stdout + run.stderr
