"""Build libuic_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m unpaired_image_captioning_amd.build

hipcc cross-compiles without a GPU, so this also is the "does it build" check.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["api.hip", "gemm.hip", "gemm_pp.hip", "gemm_tn.hip", "gemm_tn_pp.hip", "attention.hip", "rnn_persist.hip", "rnn_bwd_persist.hip", "bptt_fused.hip", "pointwise.hip", "batchnorm.hip", "beam.hip", "topdown.hip", "fcmodel.hip", "nmt.hip", "nmt_persist.hip", "cider.hip", "loader.hip", "loader_io.hip", "comm.hip", "discriminator.hip", "gcn.hip"]
LIB = os.path.join(HERE, "libuic_hip.so")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "uic_common.h"), os.path.join(CSRC, "uic_host.h"), os.path.join(CSRC, "rnn_persist_common.h"), os.path.join(HERE, "..", "include", "uic_hip.h")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + os.environ.get("UIC_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
