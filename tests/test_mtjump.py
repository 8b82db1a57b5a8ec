"""CPU: the MT19937-64 jump-ahead (host side of the multi-GPU bootstrap seek) against libstdc++'s
std::mt19937_64::discard -- the stream the reference draws its replicates from
(src/BootstrapSample.cpp:60-73)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jump_ahead_matches_libstdcxx_discard(tmp_path):
    exe = str(tmp_path / "mtjump_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "msweep_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "mtjump_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(" ok") == 10 and "MISMATCH" not in out.stdout
