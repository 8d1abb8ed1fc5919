"""memory / undefined-behaviour check of this repository's host code.  The CPU restatement and the
node shell are compiled with -fsanitize=address,undefined and driven through their edge cases by small C / C++ programs
(tests/aux_c/); any report aborts the program (-fno-sanitize-recover)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def _probe(tmp_path, cc):
    src = tmp_path / "probe.c"
    src.write_text("int main(void){return 0;}\n")
    return subprocess.run([cc] + SAN + [str(src), "-o", str(tmp_path / "probe")], capture_output=True).returncode == 0


def test_oracle_under_asan_ubsan(tmp_path):
    if not shutil.which("gcc") or not _probe(tmp_path, "gcc"):
        pytest.skip("no sanitizer runtime for gcc")
    exe = str(tmp_path / "oracle_san")
    subprocess.check_call(["gcc", "-std=gnu99", "-ffp-contract=off"] + SAN +
                          [os.path.join(ROOT, "tests", "aux_c", "oracle_sanitize_driver.c"), os.path.join(ROOT, "oracle", "apriltag_oracle.c"),
                           "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert r.stdout.strip().endswith("ok") and "tag frame: 1" in r.stdout


def test_node_shell_under_asan_ubsan(tmp_path, built):
    from isaac_ros_apriltag_amd import capi
    if not shutil.which("g++") or not _probe(tmp_path, "gcc"):
        pytest.skip("no sanitizer runtime for g++")
    if not os.path.exists(capi.LIB_PATH):
        pytest.skip("libapriltag_amd.so not built")
    exe = str(tmp_path / "node_san")
    subprocess.check_call(["g++", "-std=c++17"] + SAN +
                          [os.path.join(ROOT, "tests", "aux_c", "node_shell_sanitize_driver.cpp"),
                           os.path.join(ROOT, "isaac_ros_apriltag_amd", "csrc", "node_shell.cpp"), "-ldl", "-o", exe])
    os.symlink(capi.LIB_PATH, str(tmp_path / "libapriltag_amd.so"))   # the shell looks for the library next to itself
    # (leak detection off here: the dlopen'ed HIP runtime keeps process-lifetime allocations of its own)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert r.stdout.strip().endswith("ok")
