"""The C++ successor header (sea-current_amd/sea_current.hpp) compiles without Eigen/toppra, links
against the C-ABI library, fails loudly without a GPU, and (gpu) reproduces the reference examples'
inputs end to end (tests/cpp/test_header.cpp)."""
import os
import subprocess

import pytest

import sea_current_amd as sc

SRC = os.path.join(sc.REPO_ROOT, "tests", "cpp", "test_header.cpp")


def _build(tmp_path, std):
    sc.build()
    exe = str(tmp_path / f"test_header_{std}")
    subprocess.check_call(["g++", f"-std={std}", "-O1", "-Wall", "-Werror=return-type", "-o", exe, SRC,
                           "-L", sc.NATIVE_DIR, "-lsea_current_hip", f"-Wl,-rpath,{sc.NATIVE_DIR}"])
    return exe


@pytest.mark.parametrize("std", ["c++17", "c++20"])
def test_header_compiles_and_fails_loudly_without_gpu(tmp_path, std):
    import torch
    exe = _build(tmp_path, std)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


def test_header_with_debug_asserts_compiles(tmp_path):
    sc.build()
    subprocess.check_call(["g++", "-std=c++20", "-DDEBUG", "-fsyntax-only", SRC])


@pytest.mark.gpu
def test_header_end_to_end_on_gpu(tmp_path):
    exe = _build(tmp_path, "c++20")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
