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
    assert lib.rag_abi_version() == 1
    assert isinstance(lib.rag_last_error(), bytes)


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
