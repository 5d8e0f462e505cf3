"""The boundary really is a C ABI: tests/c_abi_client.c -- plain C, no Python, no torch -- is compiled with gcc against
include/uic_hip.h, linked with libuic_hip.so and run on the GPU (size queries, error reporting, uic_linear,
uic_attention_fwd and uic_adam_step against loops written in C)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client_links_and_runs(tmp_path):
    gcc = shutil.which("gcc")
    assert gcc, "gcc not found"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    libdir = os.path.join(ROOT, "unpaired_image_captioning_amd")
    exe = str(tmp_path / "c_abi_client")
    # a C compiler, not hipcc: the header must be plain C (the HIP runtime is only needed for hipMalloc / streams)
    cmd = [gcc, "-std=c11", "-O1", "-Wall", "-Werror=implicit-function-declaration", os.path.join(ROOT, "tests", "c_abi_client.c"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__",
           "-L" + libdir, "-luic_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C ABI OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
