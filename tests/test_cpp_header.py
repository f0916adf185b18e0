"""The C++ successor header (sea-current_amd/sea_current.hpp) compiles without Eigen/toppra, links
against the C-ABI library, fails loudly without a GPU, and (gpu) reproduces the reference examples'
inputs end to end (tests/cpp/test_header.cpp)."""
import os
import subprocess

import pytest

import sea_current_amd as sc

SRC = os.path.join(sc.REPO_ROOT, "tests", "cpp", "test_header.cpp")
SRC_CALLSITES = os.path.join(sc.REPO_ROOT, "tests", "cpp", "test_callsites.cpp")


def _build(tmp_path, std, src=SRC):
    sc.build()
    exe = str(tmp_path / (os.path.splitext(os.path.basename(src))[0] + "_" + std))
    subprocess.check_call(["g++", f"-std={std}", "-O1", "-Wall", "-Werror=return-type", "-o", exe, src,
                           "-L", sc.NATIVE_DIR, "-lsea_current_hip", f"-Wl,-rpath,{sc.NATIVE_DIR}"])
    return exe


def _eigen_include():
    """Directory that holds Eigen/Dense, if this machine has one (the build container does not)."""
    for d in ("/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/conda/include/eigen3", "/opt/rocm/include/eigen3"):
        if os.path.exists(os.path.join(d, "Eigen", "Dense")):
            return d
    return None


@pytest.mark.parametrize("std", ["c++17", "c++20"])
def test_header_compiles_and_fails_loudly_without_gpu(tmp_path, std):
    import torch
    exe = _build(tmp_path, std)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


@pytest.mark.parametrize("std", ["c++17", "c++20"])
def test_reference_call_sites_compile(tmp_path, std):
    """tests/cpp/test_callsites.cpp spells the reference's callers' names (toppra::Vector{1} with (0,0) access in a
    [](toppra::value_type) lambda, Eigen::Vector<value_type,1>{x}, bezier_curve / join_splines / hodograph / chebfit /
    chebeval, pts(i, c), pts.col(c)): it must compile against the successor header as it stands, without toppra and without
    Eigen -- and with Eigen where there is one."""
    sc.build()
    subprocess.check_call(["g++", f"-std={std}", "-Wall", "-fsyntax-only", SRC_CALLSITES])
    inc = _eigen_include()
    if inc:
        subprocess.check_call(["g++", f"-std={std}", "-Wall", "-fsyntax-only", "-I", inc, SRC_CALLSITES])
        subprocess.check_call(["g++", f"-std={std}", "-Wall", "-fsyntax-only", "-I", inc, SRC])


def test_header_with_debug_asserts_compiles(tmp_path):
    sc.build()
    subprocess.check_call(["g++", "-std=c++20", "-DDEBUG", "-fsyntax-only", SRC])


@pytest.mark.gpu
def test_header_end_to_end_on_gpu(tmp_path):
    exe = _build(tmp_path, "c++20")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


@pytest.mark.gpu
def test_reference_call_sites_run_on_gpu(tmp_path, golden_dir):
    """The call-site translation unit end to end: its own checks (closed forms of the hand-built curves, limits of the
    position-dependent profile), and the service reply it builds the reference's way against the recorded run."""
    import json
    import numpy as np
    exe = _build(tmp_path, "c++20", SRC_CALLSITES)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "call sites OK" in r.stdout
    r = subprocess.run([exe, "--serve"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    states = json.loads(r.stdout)
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    assert len(states) == 4328
    col = lambda f: np.array([f(s) for s in states])
    assert np.abs(col(lambda s: s["time"]) - fx["time"]).max() < 2e-5
    assert np.abs(col(lambda s: s["velocity"]) - fx["vel"]).max() < 2e-6
    assert np.abs(col(lambda s: s["angularVelocity"]) - fx["ang_vel"]).max() < 2e-6
    assert np.abs(col(lambda s: s["pose"]["translation"]["x"]) - fx["pos_x"]).max() < 5e-5
    assert np.abs(col(lambda s: s["pose"]["translation"]["y"]) - fx["pos_y"]).max() < 5e-5


@pytest.mark.gpu
def test_service_reply_matches_recorded_output(tmp_path, golden_dir):
    """The example service's request -> reply (examples/zmq_test.cpp:25-101) through the successor header, against the
    reference's recorded run (examples/output.json; it predates the per-state reply schema, so values are compared)."""
    import json
    import numpy as np
    exe = _build(tmp_path, "c++20")
    r = subprocess.run([exe, "--serve-recorded-request"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    states = json.loads(r.stdout)
    fx = np.load(os.path.join(golden_dir, "toppra_1dof_output.npz"))
    assert len(states) == fx["time"].shape[0] == 4328
    assert set(states[0]) == {"time", "velocity", "acceleration", "angularVelocity", "pose", "holonomicRotation", "holonomicAngularVelocity"}
    col = lambda f: np.array([f(s) for s in states])
    assert np.abs(col(lambda s: s["time"]) - fx["time"]).max() < 2e-5
    assert np.abs(col(lambda s: s["velocity"]) - fx["vel"]).max() < 2e-6
    assert np.abs(col(lambda s: s["acceleration"]) - fx["acc"]).max() < 2e-6
    assert np.abs(col(lambda s: s["angularVelocity"]) - fx["ang_vel"]).max() < 2e-6
    assert np.abs(col(lambda s: s["pose"]["translation"]["x"]) - fx["pos_x"]).max() < 5e-5
    assert np.abs(col(lambda s: s["pose"]["translation"]["y"]) - fx["pos_y"]).max() < 5e-5
