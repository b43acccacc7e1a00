"""CPU-side sanitizer runs (the reference has none; the GPU pool offers no device sanitizer, so the C that
runs on the host is what can be checked): `make -C oracle asan` builds the CPU restatement, the HDF5 plugin's C source and the
tests' HDF5 driver with -fsanitize=address,undefined; here the golden-vector suite runs against the instrumented restatement,
the filter callback's failure paths run under ASan + LeakSanitizer, and an H5Dwrite through the instrumented plugin must fail
cleanly (there is no GPU in the CPU container: the callback returns 0 and HDF5 unwinds)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def san():
    if _libasan() is None:
        pytest.skip("no libasan in this toolchain")
    if not os.path.exists(os.path.join(ROOT, "deltarice_amd", "libdeltarice_hip.so")):
        pytest.skip("run `make` first (the plugin links the codec library)")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return SAN


def test_oracle_golden_suite_under_asan_ubsan(san):
    env = dict(os.environ, DRO_ORACLE_SO=os.path.join(san, "libdeltarice_oracle_asan.so"), LD_PRELOAD=_libasan(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    # (the one deselected case drives the REFERENCE's compiled filter, oracle/_ref: the preloaded runtime intercepts its
    # memcpy as well and stops at the overlapping compaction copy of /root/reference/src/deltaRice.c:429-432 -- SURVEY
    # Appendix B7, not this repo's code)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-k", "not compiled_reference",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "passed" in r.stdout, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]


def test_filter_callback_failure_paths_under_asan_lsan(san):
    # leaks are checked too: whatever the HIP runtime itself keeps after a failed initialisation is not ours
    supp = os.path.join(san, "lsan.supp")
    with open(supp, "w") as f:
        f.write("leak:libamdhip64\nleak:libhsa-runtime64\nleak:libdeltarice_hip\nleak:libdrm\nleak:libnuma\nleak:libstdc++\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", LSAN_OPTIONS=f"suppressions={supp}:print_suppressions=0",
               UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(san, "filter_failpath_asan")], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "failpath ok" in r.stdout, out[-4000:]
    assert "AddressSanitizer" not in out and "LeakSanitizer" not in out and "runtime error:" not in out, out[-4000:]


def test_hdf5_write_through_instrumented_plugin_fails_cleanly(san, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the failure path is what this test drives")
    raw = tmp_path / "raw.bin"
    np.random.default_rng(3).normal(0, 10, 40 * 1000).astype(np.int16).tofile(raw)
    env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(san, "plugin"), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(san, "h5_roundtrip_asan"), "write", str(tmp_path / "f.h5"), str(raw), "40", "1000", "20", "8", "1000"],
                       env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0, "the write cannot succeed without a GPU"
    assert "deltarice" in out, out[-3000:]  # the callback said why
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
