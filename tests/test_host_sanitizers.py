"""Host-side C++ of the product (bit packing, bincode .sketch container, FASTA / gzip readers) under
AddressSanitizer + UBSan on the CPU build: round trips, truncated files, CRLF / empty / header-only inputs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_formats_under_asan_ubsan(tmp_path):
    exe = tmp_path / "driver"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "native", "formats_sanitizer_driver.cpp"),
                           os.path.join(ROOT, "hyper-gen_amd", "csrc", "hg_formats.cpp"), "-I", os.path.join(ROOT, "include"),
                           "-lz", "-o", str(exe)])
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "asan driver ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_call_pool_and_piece_packer_under_tsan(tmp_path):
    """the host threads of a host-fed call (hg_sketch_batch packs its sub-batches with them) under ThreadSanitizer"""
    exe = tmp_path / "pool_driver"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "native", "pool_tsan_driver.cpp"),
                           os.path.join(ROOT, "hyper-gen_amd", "csrc", "hg_formats.cpp"), "-I", os.path.join(ROOT, "include"),
                           "-lz", "-lpthread", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in out.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert out.returncode == 0 and "tsan driver ok" in out.stdout, out.stdout + out.stderr
