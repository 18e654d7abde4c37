"""The C++ host mirror (include/hadi_host.hpp: the reference's launcher names over std::vector) compiles against the C
ABI with plain g++, and -- on a GPU -- reproduces the recorded reference prices / Jacobian row from C++."""
import os
import subprocess

import pytest

import __graft_entry__ as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pde_based_heston_solver_gpu_accelerated_amd")
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")


def _build():
    G.build_libhadi()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-o", EXE,
                           "-L", PKG, "-lhadi", "-Wl,-rpath," + PKG])
    return EXE


def test_cpp_host_mirror_compiles_and_links_against_the_c_abi():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_host_mirror_reproduces_reference_outputs():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "all C++ host-mirror checks passed" in out.stdout
