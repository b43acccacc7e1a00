"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; the HDF5
plugin exports the reference's surface.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_hip_library_exports_declared_abi():
    from deltarice_amd import _lib
    lib = _lib.load()  # raises if the library is missing: there is no fallback
    names = declared("deltarice_hip.h", "drx_")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/deltarice_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "python binding and header disagree"
    assert b"gfx950" in lib.drx_version()


def test_parse_cd_values_matches_reference_rules():
    # src/deltaRice.c:248-291 (defaults) and :114-136 (M must be a power of two)
    from deltarice_amd.codec import parse_opts
    from deltarice_amd import DeltaRiceError
    o = parse_opts(())
    assert (o.rice_k, o.wave_len, o.n_taps, o.taps[0], o.taps[1]) == (3, -1, 2, 1, -1)
    o = parse_opts((16,))
    assert (o.rice_k, o.wave_len) == (4, -1)
    o = parse_opts((8, 7000))
    assert (o.rice_k, o.wave_len) == (3, 7000)
    o = parse_opts((8, 0xFFFFFFFF))
    assert o.wave_len == -1
    o = parse_opts((8, 1024, 4, 1, 0xFFFFFFFF, 1, 0xFFFFFFFF))
    assert o.n_taps == 4 and list(o.taps[:4]) == [1, -1, 1, -1]
    for bad in [(0,), (3,), (6,), (65536,), (8, 0), (8, 1024, 0), (8, 1024, 3, 1, 1), (8, 1024, 2, 0, 1)]:
        with pytest.raises(DeltaRiceError):
            parse_opts(bad)


def test_plugin_exports_reference_surface():
    # names of /root/reference/src/deltaRice.h:10-15, src/deltaRice_h5plugin.c:4-5, src/hdf5_dl.c:194
    from deltarice_amd import PLUGIN_PATH
    assert os.path.exists(PLUGIN_PATH), "run `make`"
    p = C.CDLL(PLUGIN_PATH)
    for n in ("H5Z_filter_deltarice", "deltarice_register_h5filter", "H5PLget_plugin_type",
              "H5PLget_plugin_info", "init_filter", "H5Z_DELTARICE"):
        assert hasattr(p, n), n
    p.H5PLget_plugin_type.restype = C.c_int
    assert p.H5PLget_plugin_type() == 0  # H5PL_TYPE_FILTER
    p.H5PLget_plugin_info.restype = C.c_void_p
    info = p.H5PLget_plugin_info()
    cls = C.c_void_p.in_dll(p, "H5Z_DELTARICE")
    assert info == C.addressof(cls), "plugin info must be the class record (the reference returns 32025)"

    class H5ZClass2(C.Structure):  # H5Z_class2_t, H5Zpublic.h
        _fields_ = [("version", C.c_int), ("id", C.c_int), ("encoder_present", C.c_uint),
                    ("decoder_present", C.c_uint), ("name", C.c_char_p), ("can_apply", C.c_void_p),
                    ("set_local", C.c_void_p), ("filter", C.c_void_p)]
    rec = H5ZClass2.from_address(info)
    assert (rec.version, rec.id, rec.encoder_present, rec.decoder_present) == (1, 32025, 1, 1)
    assert rec.name == b"deltarice" and not rec.can_apply and not rec.set_local
    assert rec.filter == C.cast(p.H5Z_filter_deltarice, C.c_void_p).value


def test_python_h5_module_surface():
    # src/h5.pyx:27,55-61: H5FILTER constant and register_h5_filter()
    from deltarice_amd import h5
    assert h5.H5FILTER == 32025
    assert callable(h5.register_h5_filter)


def test_filter_callback_fails_loudly_without_gpu(capfd):
    """No GPU in this container: the H5Z callback must report failure the HDF5 way (return 0, buffers
    untouched) and say why -- it must not fall back to any CPU codec."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from deltarice_amd import PLUGIN_PATH
    p = C.CDLL(PLUGIN_PATH)
    p.H5Z_filter_deltarice.restype = C.c_size_t
    p.H5Z_filter_deltarice.argtypes = [C.c_uint, C.c_size_t, C.POINTER(C.c_uint), C.c_size_t,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    buf = C.c_void_p(libc.malloc(4096))
    C.memset(buf, 0, 4096)
    size = C.c_size_t(4096)
    cd = (C.c_uint * 2)(8, 1024)
    before = buf.value
    ret = p.H5Z_filter_deltarice(0, 2, cd, 2048, C.byref(size), C.byref(buf))
    assert ret == 0 and buf.value == before and size.value == 4096
    assert "deltarice" in capfd.readouterr().err
    libc.free(buf)
    from deltarice_amd import codec, DeltaRiceError
    with pytest.raises(DeltaRiceError):
        codec.Context(0)


def test_reference_import_path_alias():
    """`import deltaRice.h5` (the reference's module path) resolves to this codec; it needs h5py,
    which this image lacks, so only the failure mode can be checked here."""
    import importlib
    import deltaRice  # noqa: F401
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            importlib.import_module("deltaRice.h5")
    else:
        m = importlib.import_module("deltaRice.h5")
        assert m.H5FILTER == 32025 and callable(m.register_h5_filter)
