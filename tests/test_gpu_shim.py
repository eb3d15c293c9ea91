"""-m gpu: builds tests/cpp/shim_test.cpp (the reference's known-answer tests written against the C++17 shim classes
with the reference's signatures) against libmygram_shim.so (the built host layer) + libmygram_gpu.so and runs it."""
import os
import subprocess

import pytest

from pkg import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_shim_known_answers(tmp_path):
    exe = str(tmp_path / "shim_test")
    lib_dir = os.path.join(ROOT, "mygram-db_amd")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"),
                    "-L" + lib_dir, "-lmygram_shim", "-lmygram_gpu", "-Wl,-rpath," + lib_dir, "-pthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
