"""CPU: pine_amd/csrc/pine_libm.h (the device's sinf/cosf) against the host libm, the one the reference
links.  The exhaustive run (2.2e9 floats, stride 1) is `tools/check_libm.cpp`; here a 1/61 subsample."""
import os
import subprocess

from conftest import ROOT


def test_sincos_match_host_libm(tmp_path):
    exe = tmp_path / "check_libm"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-pthread",
                           os.path.join(ROOT, "tools", "check_libm.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "61"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert '"sin_mismatch": 0' in out.stdout and '"cos_mismatch": 0' in out.stdout


def test_powf_logf_match_host_libm(tmp_path):
    """powf over every float in [0, 1] x {5, 2.5, 0.5} (Schlick), logf over all 2^31 positive floats, special
    values, and a few million random pairs (tools/check_libm_pow.cpp runs 10^9 of them)."""
    exe = tmp_path / "check_libm_pow"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-pthread",
                           os.path.join(ROOT, "tools", "check_libm_pow.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "20000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    for k in ("schlick_mismatch", "logf_mismatch", "pair_mismatch", "special_mismatch"):
        assert f'"{k}": 0' in out.stdout


def test_atanf_atan2f_acosf_match_host_libm(tmp_path):
    """atanf and acosf over ALL 2^32 arguments, atan2f over its special cases, every exponent difference and random
    pairs (tools/check_libm_atan.cpp; its default run checks 4 x 10^9 pairs)."""
    exe = tmp_path / "check_libm_atan"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread",
                           os.path.join(ROOT, "tools", "check_libm_atan.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "5000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    for k in ("atanf_mismatch", "acosf_mismatch", "atan2f_mismatch"):
        assert f'"{k}": 0' in out.stdout
