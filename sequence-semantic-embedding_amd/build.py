"""Builds libsse_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python sequence-semantic-embedding_amd/build.py [--force]

The .so is kept in-tree next to this file so that it travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, os.environ.get("SSE_LIB_NAME", "libsse_hip.so"))   # SSE_LIB_NAME + SSE_HIPCC_EXTRA: measurement builds
# the host-only entry points of the same C ABI (targetEncodingIndex.tsv text I/O, CRC-32C of TF checkpoints) once more as a
# library WITHOUT the HIP runtime: reading a checkpoint or parsing an index file must not initialise a GPU (or import torch)
HOST_LIB = os.path.join(HERE, "libsse_host.so")
HOST_SOURCES = ["index_io.cpp"]
SOURCES = ["sse_api.hip", "lstm_fwd.hip", "lstm_fwd_gs.hip", "lstm_small.hip", "lstm_persist.hip", "lstm_cluster.hip", "lstm_fwd_x3.hip", "cnn_fwd.hip", "cnn_fwd_bf16.hip", "score_topk.hip", "pack.hip", "train.hip", "lstm_bwd2.hip", "cnn_bwd.hip", "cnn_bwd_mfma.hip", "lstm_generic.hip", "index_io.cpp"]
EXTRA = os.environ.get("SSE_HIPCC_EXTRA", "").split()   # e.g. -DSSE_SCORE_MEASURE for the measurement builds of tools/
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "sse_hip.h"))
    return hdrs


def build(force=False, verbose=False):
    """Compile every HIP source and link libsse_hip.so; returns its path."""
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", os.environ["SSE_LIB_NAME"] + ".o.d") if os.environ.get("SSE_LIB_NAME") else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_mtime = max(os.path.getmtime(p) for p in _deps())
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_mtime):
            flags = FLAGS if src.endswith(".hip") else [f for f in FLAGS if not f.startswith("--offload-arch")] + ["-pthread"]
            jobs.append([hipcc] + flags + EXTRA + ["-c", s, "-o", o])
    if jobs:  # independent translation units: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    rebuilt = bool(jobs)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    host_objs = [os.path.join(objdir, os.path.splitext(src)[0] + ".o") for src in HOST_SOURCES]
    if not os.environ.get("SSE_LIB_NAME") and (rebuilt or not os.path.exists(HOST_LIB) or
                                               os.path.getmtime(HOST_LIB) < max(os.path.getmtime(o) for o in host_objs)):
        cmd = [os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-pthread", "-o", HOST_LIB] + host_objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
