"""Builds experiment variants of librag_amd.so next to the product library (never the product build)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rag_inference_pipeline_amd", "csrc")


def build(tag: str, defines: list[str]) -> str:
    out_dir = os.path.join(CSRC, "exp")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"librag_amd_{tag}.so")
    srcs = [os.path.join(CSRC, "rag_amd.hip"), os.path.join(CSRC, "rag_bert.hip"), os.path.join(CSRC, "rag_lz4.cpp")]
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC"] + [f"-D{d}" for d in defines] + ["-o", out] + srcs
        subprocess.run(cmd, check=True)
    return out
