"""Sanitizer runs of the CPU side (`-fsanitize=address,undefined` where it is
available -- GPU AddressSanitizer is not, on this pool).

  * host side of the product: every host source of lightdock-rust_amd/csrc built by g++ with
    ASan + UBSan against tests/asan/hip_stub.cpp (device memory = host memory, kernels = no-ops) and
    driven through the C ABI by tests/asan/host_check.cpp: PDB / setup.json / npy / DCparams readers,
    model builders, tile layout, LUT builders, scorer and GSO bookkeeping, gso_*.out writers, the
    CLI's usage-error and panic paths;
  * the oracle CLI, same flags, a few GSO steps of the 1azp example.
"""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def clean(text):
    return "AddressSanitizer" not in text and "LeakSanitizer" not in text and "runtime error" not in text


@pytest.mark.timeout(900)
def test_host_side_under_asan_ubsan(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lightdock-rust_amd"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = os.path.join(ROOT, "lightdock-rust_amd", "build", "asan", "host_check")
    r = subprocess.run([exe, GOLDEN, str(tmp_path)], capture_output=True, text=True, env=ENV)
    out = r.stdout + r.stderr
    assert clean(out), out[-4000:]
    assert r.returncode == 0 and "host_check: 0 failures" in out, out[-3000:]
    assert os.path.exists(tmp_path / "swarm_b" / "gso_13.out")


@pytest.mark.timeout(600)
def test_oracle_cli_under_asan_ubsan(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = os.path.join(ROOT, "oracle", "ld_oracle_cli_asan")
    src = os.path.join(GOLDEN, "1azp")
    for f in ("rec_nm.npy", "lig_nm.npy"):
        shutil.copy(os.path.join(src, f), tmp_path)
    r = subprocess.run([exe, os.path.join(src, "setup.json"), os.path.join(src, "initial_positions_0.dat"), "3", "dna"],
                       cwd=tmp_path, capture_output=True, text=True, env=ENV)
    out = r.stdout + r.stderr
    assert clean(out), out[-4000:]
    assert r.returncode == 0, out[-2000:]
    assert os.path.exists(tmp_path / "swarm_0" / "gso_1.out")
    # usage errors and a panic path
    for args in ([], ["x", "y", "z", "dna"], [os.path.join(src, "setup.json"), os.path.join(src, "initial_positions_0.dat"), "3", "nomethod"]):
        r = subprocess.run([exe] + args, cwd=tmp_path, capture_output=True, text=True, env=ENV)
        assert clean(r.stdout + r.stderr), (args, (r.stdout + r.stderr)[-3000:])
