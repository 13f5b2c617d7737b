"""The C++ host API (tfhe_amd/host/tfhe.hpp) on a GPU: its parity program
compares every gate with the CPU oracle bit for bit; the example mirrors the
reference's examples/homomorphic_xor.rs."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tfhe_amd", "host", "bin")


def _run(name):
    exe = os.path.join(BIN, name)
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    return subprocess.run([exe], capture_output=True, text=True, timeout=600)


def test_host_api_matches_oracle():
    r = _run("host_api_test")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_api_test: ok" in r.stdout


def test_homomorphic_xor_example():
    r = _run("homomorphic_xor")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "not(enc(false)) decrypts to true" in r.stdout
