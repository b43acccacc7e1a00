"""The Python binding `deltaRice.h5` (Cython, deltaRice/h5.pyx) and the plugin install step.

Counterparts in the reference: src/h5.pyx (module attributes, import-time registration),
setup.py:133-161 (artefact names), setup.py:186-227 (`install --h5plugin --h5plugin-dir`), and
tests/test.py (six h5py round trips; they run here only where h5py is installed).

CPU tests: the extension builds, exports PyInit_h5, and -- driven in a child process with a TEST DOUBLE for
h5py whose only content is seven module objects whose __file__ is the real libhdf5 of this image -- binds
H5Zregister through the first library that has it and registers filter 32025 with that libhdf5.  The double
stands in for h5py's extension modules only as "a shared object that links libhdf5", which is all h5.pyx
uses of them (src/h5.pyx:36-42)."""
import ctypes as C
import glob
import os
import subprocess
import sys
import sysconfig
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDF5_DIR = os.environ.get("HDF5_DIR", "/opt/conda")
EXT = os.path.join(ROOT, "deltaRice", "h5" + sysconfig.get_config_var("EXT_SUFFIX"))


def test_cython_extension_builds_and_exports_module_init():
    subprocess.run(["make", "-C", ROOT, "pyext"], check=True, capture_output=True)
    assert os.path.exists(EXT), "make pyext must produce deltaRice/h5*.so"
    assert not os.path.exists(os.path.join(ROOT, "deltaRice", "h5.py")), "the binding is the compiled module"
    lib = C.CDLL(EXT)  # resolves libh5deltarice.so (and through it the HIP codec library) by run path
    assert hasattr(lib, "PyInit_h5")
    # it calls into the plugin library, it does not carry a codec of its own
    out = subprocess.run(["nm", "-D", "--undefined-only", EXT], capture_output=True, text=True, check=True).stdout
    assert "deltarice_register_h5filter" in out and "init_filter" in out


def test_header_declares_reference_names():
    # src/deltaRice.h:7-15
    txt = open(os.path.join(ROOT, "include", "deltarice_h5filter.h")).read()
    for name in ("H5Z_FILTER_DELTARICE 32025", "typedef unsigned long long int superint;", "H5Z_DELTARICE[1]",
                 "H5Z_filter_deltarice(", "deltarice_register_h5filter(void)"):
        assert name in txt, name
    # and it compiles as C with those names usable
    src = '#include "deltarice_h5filter.h"\nint main(void){ superint s = H5Z_FILTER_DELTARICE; return (int)(s != 32025) + (H5Z_DELTARICE[0].id != 32025); }\n'
    r = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-", f"-I{ROOT}/include", f"-I{HDF5_DIR}/include"], input=src, text=True,
                       capture_output=True)
    assert r.returncode == 0, r.stderr


def _libhdf5():
    c = sorted(glob.glob(os.path.join(HDF5_DIR, "lib", "libhdf5.so*")))
    return c[0] if c else None


@pytest.mark.skipif(_libhdf5() is None, reason="no libhdf5 in this image")
def test_import_registers_with_the_hdf5_h5py_uses(tmp_path):
    """Import-time behaviour of deltaRice.h5 against the real libhdf5 (no GPU needed: registering is not filtering)."""
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed: test_reference_cases_through_h5py covers the real thing")
    except ImportError:
        pass
    subprocess.run(["make", "-C", ROOT, "pyext"], check=True, capture_output=True)
    fake = tmp_path / "h5py"
    fake.mkdir()
    lib = _libhdf5()
    # the double: h5d / h5fd lack HDF5 (a libc), so the probe must move on to h5s, which links it
    (fake / "__init__.py").write_text("")
    for name in ("h5d", "h5fd"):
        (fake / f"{name}.py").write_text("__file__ = '/lib/x86_64-linux-gnu/libm.so.6'\n")
    for name in ("h5s", "h5t", "h5p", "h5z", "defs"):
        (fake / f"{name}.py").write_text(f"__file__ = {lib!r}\n")
    code = textwrap.dedent(f"""
        import ctypes as C, sys
        sys.path.insert(0, {str(tmp_path)!r}); sys.path.insert(0, {ROOT!r})
        h5 = C.CDLL({lib!r}, mode=C.RTLD_GLOBAL)
        h5.H5Zfilter_avail.argtypes = [C.c_int]
        h5.H5open()
        assert h5.H5Zfilter_avail(32025) <= 0, "not registered before the import"
        import deltaRice.h5 as m
        assert m.H5FILTER == 32025 and callable(m.register_h5_filter)
        assert m.HDF5_LIBRARY == {lib!r}, m.HDF5_LIBRARY      # libm was probed first and rejected
        assert h5.H5Zfilter_avail(32025) > 0, "import must register the filter"
        m.register_h5_filter()                                    # registering again is fine (HDF5 replaces the entry)
        print("ok")
    """)
    env = dict(os.environ, HDF5_PLUGIN_PATH="/nonexistent")  # availability must come from the registration
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


def test_import_without_h5py_fails_like_the_reference():
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, "-c", "import deltaRice.h5"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "h5py" in r.stderr


def test_install_plugin_target(tmp_path):
    """`make install-plugin PLUGIN_DIR=...` = the reference's `setup.py install --h5plugin --h5plugin-dir=...`
    (setup.py:186-227): the plugin lands in that directory under the reference's artefact name and loads from there."""
    dest = tmp_path / "hdf5" / "lib" / "plugin"
    r = subprocess.run(["make", "-C", ROOT, "install-plugin", f"PLUGIN_DIR={dest}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"Installed HDF5 filter plugins to {dest}" in r.stdout
    so = dest / "libh5deltarice.so"
    assert so.exists() and (dest.parent / "libdeltarice_hip.so").exists()
    p = C.CDLL(str(so))  # finds the codec library through its $ORIGIN/.. run path
    p.H5PLget_plugin_info.restype = C.c_void_p
    assert p.H5PLget_plugin_info() == C.addressof(C.c_void_p.in_dll(p, "H5Z_DELTARICE"))


# ---- the reference's own six tests (tests/test.py:8-83), where h5py exists -------------------------------------------
REF_CASES = [
    ("worst_case", "uniform", None),
    ("different_m", "uniform", (16,)),
    ("different_m_different_segment_length", "uniform", (8, 1024)),
    ("different_filter", "uniform", (8, 1024, 1, 1)),
    ("all_signed_values", "arange_i16", (8, 1024, 1, 1)),
    ("all_unsigned_values", "arange_u16", (8, 1024, 1, 1)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,opts", REF_CASES)
def test_reference_cases_through_h5py(tmp_path, name, kind, opts):
    h5py = pytest.importorskip("h5py")
    import deltaRice.h5
    if kind == "uniform":
        dset = np.random.default_rng(len(name)).uniform(-32768, 32768, size=2 ** 16).astype(np.int16)
    elif kind == "arange_i16":
        dset = np.arange(-32768, 32768).astype(np.int16)
    else:
        dset = np.arange(0, 65536).astype(np.uint16)
    path = tmp_path / "testFile.h5"
    kw = {} if opts is None else {"compression_opts": opts}
    with h5py.File(path, "w") as f:
        f.create_dataset("test", data=dset, compression=deltaRice.h5.H5FILTER, **kw)
    with h5py.File(path, "r") as f:
        assert np.array_equal(f["test"][()], dset), f"Failed {name}"
