"""GPU: filter 32025 through the real HDF5 library (C API, HDF5 1.10.6 of this image) with the
plugin loaded dynamically from HDF5_PLUGIN_PATH -- the path on which the reference's own shim
crashes (src/deltaRice_h5plugin.c:5 returns 32025 instead of the class record).

Counterpart of the reference's tests/test.py (write -> close -> reopen -> read -> equal) and of
README.md:64-92 (BASELINE config #1), plus byte-level checks the reference never had:
the chunks stored in the file are the oracle's / the reference's bytes, and a file holding
chunks encoded by the CPU filter reads back through the GPU filter."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDF5_DIR = os.environ.get("HDF5_DIR", "/opt/conda")


@pytest.fixture(scope="module")
def h5tool(tmp_path_factory):
    if not os.path.exists(os.path.join(HDF5_DIR, "include", "hdf5.h")):
        pytest.skip("no HDF5 C library in this image")
    d = tmp_path_factory.mktemp("h5tool")
    exe = str(d / "h5_roundtrip")
    subprocess.run(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tests", "h5_roundtrip.c"),
                    f"-I{HDF5_DIR}/include", f"-L{HDF5_DIR}/lib", "-lhdf5", f"-Wl,-rpath,{HDF5_DIR}/lib"], check=True)
    env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(ROOT, "deltarice_amd", "plugin"))

    def run(*args, extra_env=None):
        return subprocess.run([exe, *map(str, args)], env=dict(env, **(extra_env or {})), check=True, capture_output=True, text=True)
    return run


CASES = [
    # rows, cols, chunk_rows, M, L, data
    (100, 7000, 20, 8, 7000, "gauss"),      # README.md:75-82 = BASELINE config #1
    (64, 1024, 16, 8, 1024, "uniform"),      # tests/test.py:33-44 shape
    (32, 2048, 8, 16, 2048, "gauss"),
    (16, 4096, 16, 8, 512, "arange"),        # several waveforms per row
]


@pytest.mark.parametrize("rows,cols,crows,M,L,kind", CASES)
def test_hdf5_roundtrip_and_stored_bytes(h5tool, tmp_path, rows, cols, crows, M, L, kind):
    from oracle import oracle as O
    rng = np.random.default_rng(rows * cols)
    if kind == "gauss":
        x = rng.normal(0, 10, (rows, cols)).astype(np.int16)
    elif kind == "uniform":
        x = rng.integers(-32768, 32768, (rows, cols)).astype(np.int16)
    else:
        x = np.arange(rows * cols, dtype=np.int64).astype(np.uint16).view(np.int16).reshape(rows, cols)
    raw, h5, back = tmp_path / "raw.bin", tmp_path / "t.h5", tmp_path / "back.bin"
    x.tofile(raw)
    h5tool("write", h5, raw, rows, cols, crows, M, L)           # encode on the GPU, chunk by chunk
    h5tool("read", h5, back)                                     # decode on the GPU
    assert np.array_equal(np.fromfile(back, np.int16).reshape(rows, cols), x)
    # the bytes HDF5 stored are exactly the CPU filter's
    n = int(h5tool("chunks", h5, tmp_path / "chunk").stdout)
    assert n == rows // crows
    for c in range(n):
        stored = np.fromfile(f"{tmp_path}/chunk.{c}", np.uint32)
        ref = O.encode_chunk(x[c * crows:(c + 1) * crows], (M, L))
        assert np.array_equal(stored, ref), f"chunk {c} differs from the oracle"
    # a file whose chunks were encoded on the CPU reads back through the GPU filter
    for c in range(n):
        O.encode_chunk(x[c * crows:(c + 1) * crows], (M, L)).tofile(f"{tmp_path}/cpu.{c}")
    h5cpu, back2 = tmp_path / "cpu.h5", tmp_path / "back2.bin"
    h5tool("writeraw", h5cpu, rows, cols, crows, M, L, tmp_path / "cpu")
    h5tool("read", h5cpu, back2)
    assert np.array_equal(np.fromfile(back2, np.int16).reshape(rows, cols), x)


# the six cases of the reference's tests/test.py (write -> close -> reopen -> read -> array_equal)
REF_TESTS = [
    ("worst_case", "uniform", ()),                      # tests/test.py:8-18
    ("different_m", "uniform", (16,)),                  # :20-31
    ("m_and_segment_length", "uniform", (8, 1024)),     # :33-44
    ("different_filter", "uniform", (8, 1024, 1, 1)),   # :46-57
    ("all_signed", "arange_i16", (8, 1024, 1, 1)),      # :59-70
    ("all_unsigned", "arange_u16", (8, 1024, 1, 1)),    # :72-83
]


@pytest.mark.parametrize("name,kind,opts", REF_TESTS)
def test_reference_test_suite_cases(h5tool, tmp_path, name, kind, opts):
    from oracle import oracle as O
    if kind == "uniform":
        x = np.random.default_rng(len(opts)).uniform(-32768, 32768, size=2 ** 16).astype(np.int16)
    elif kind == "arange_i16":
        x = np.arange(-32768, 32768).astype(np.int16)
    else:
        x = np.arange(0, 65536).astype(np.uint16).view(np.int16)
    rows, cols, crows = 8, 8192, 2  # the filter sees four 2 x 8192 sub-chunks, like h5py's auto-chunking
    raw, h5, back = tmp_path / "raw.bin", tmp_path / "t.h5", tmp_path / "back.bin"
    x.tofile(raw)
    h5tool("write", h5, raw, rows, cols, crows, *opts)
    h5tool("read", h5, back)
    assert np.array_equal(np.fromfile(back, np.int16), x), f"Failed {name}"
    n = int(h5tool("chunks", h5, tmp_path / "chunk").stdout)
    for c in range(n):
        stored = np.fromfile(f"{tmp_path}/chunk.{c}", np.uint32)
        ref = O.encode_chunk(x.reshape(rows, cols)[c * crows:(c + 1) * crows], opts)
        assert np.array_equal(stored, ref), f"{name}: chunk {c} differs from the oracle"


def test_reference_example_program_shape(h5tool, tmp_path):
    """examples/testCode.c of the reference: every short value in 10 columns (65536 x 10), chunks of 32768 x 5,
    cd_values = {8, 32768} (:15-18,32,51-53): four chunks, each the row-major bytes of a 32768 x 5 tile cut into five
    32768-sample waveforms.  Round trip through HDF5 and stored bytes against the oracle."""
    from oracle import oracle as O
    rows, cols, crows, ccols = 65536, 10, 32768, 5
    x = np.repeat(np.arange(rows, dtype=np.int64).astype(np.uint16).view(np.int16)[:, None], cols, axis=1)  # wdata[i][j] = i
    raw, h5, back = tmp_path / "raw.bin", tmp_path / "t.h5", tmp_path / "back.bin"
    np.ascontiguousarray(x).tofile(raw)
    h5tool("write", h5, raw, rows, cols, crows, 8, 32768, extra_env={"H5RT_CHUNK_COLS": str(ccols)})
    h5tool("read", h5, back)
    assert np.array_equal(np.fromfile(back, np.int16).reshape(rows, cols), x), "Do they all match? no"
    n = int(h5tool("chunks", h5, tmp_path / "chunk").stdout)
    assert n == (rows // crows) * (cols // ccols)
    idx = 0
    for r in range(0, rows, crows):
        for c in range(0, cols, ccols):
            tile = np.ascontiguousarray(x[r:r + crows, c:c + ccols])
            stored = np.fromfile(f"{tmp_path}/chunk.{idx}", np.uint32)
            assert np.array_equal(stored, O.encode_chunk(tile, (8, 32768))), f"chunk {idx} differs from the oracle"
            idx += 1


def test_h5dump_sees_the_filter(h5tool, tmp_path):
    h5dump = os.path.join(HDF5_DIR, "bin", "h5dump")
    if not os.path.exists(h5dump):
        pytest.skip("no h5dump")
    x = np.random.default_rng(0).normal(0, 10, (40, 7000)).astype(np.int16)
    raw, h5 = tmp_path / "raw.bin", tmp_path / "t.h5"
    x.tofile(raw)
    h5tool("write", h5, raw, 40, 7000, 20, 8, 7000)
    env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(ROOT, "deltarice_amd", "plugin"))
    out = subprocess.run([h5dump, "-pH", str(h5)], env=env, capture_output=True, text=True, check=True).stdout
    assert "FILTER_ID 32025" in out and "PARAMS { 8 7000 }" in out


def test_h5repack_and_h5diff_through_the_plugin(h5tool, tmp_path):
    """The HDF5 tools as callers: h5repack re-filters an existing file with `UD=32025,...` through the dynamically
    loaded plugin, and back to no filter; h5diff (which reads through the plugin) finds no difference; the chunks h5repack
    stored are the reference's bytes."""
    from oracle import oracle as O
    h5repack, h5diff = (os.path.join(HDF5_DIR, "bin", t) for t in ("h5repack", "h5diff"))
    if not (os.path.exists(h5repack) and os.path.exists(h5diff)):
        pytest.skip("no h5repack / h5diff")
    x = np.random.default_rng(3).normal(0, 10, (60, 1024)).astype(np.int16)
    raw, plain, packed, back = (tmp_path / n for n in ("raw.bin", "plain.h5", "packed.h5", "back.h5"))
    x.tofile(raw)
    env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(ROOT, "deltarice_amd", "plugin"))
    # an unfiltered, chunked file (filter 32025 with the tool's write mode would already go through the plugin)
    h5tool("write", plain, raw, 60, 1024, 20, 8, 1024)
    subprocess.run([h5repack, "-f", "NONE", str(plain), str(back)], env=env, check=True, capture_output=True)
    r = subprocess.run([h5repack, "-f", "test:UD=32025,0,2,16,1024", str(back), str(packed)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([os.path.join(HDF5_DIR, "bin", "h5dump"), "-pH", str(packed)], env=env, capture_output=True, text=True,
                         check=True).stdout
    assert "FILTER_ID 32025" in out and "PARAMS { 16 1024 }" in out
    d = subprocess.run([h5diff, str(back), str(packed)], env=env, capture_output=True, text=True)
    assert d.returncode == 0, d.stdout[-2000:] + d.stderr[-2000:]
    n = int(h5tool("chunks", packed, tmp_path / "chunk").stdout)
    assert n == 3
    for i in range(n):
        stored = np.fromfile(f"{tmp_path}/chunk.{i}", np.uint32)
        assert np.array_equal(stored, O.encode_chunk(x[20 * i:20 * i + 20], (16, 1024))), f"chunk {i}"


def test_explicit_registration_through_python_module(tmp_path):
    # counterpart of `import deltaRice.h5` (src/h5.pyx:55-61): explicit H5Zregister in a given libhdf5
    import ctypes as C
    lib = os.path.join(HDF5_DIR, "lib", "libhdf5.so")
    if not os.path.exists(lib):
        pytest.skip("no libhdf5")
    h5 = C.CDLL(lib, mode=C.RTLD_GLOBAL)
    from deltarice_amd import h5 as drh5
    drh5.register_h5_filter(lib)
    h5.H5Zfilter_avail.argtypes = [C.c_int]
    assert h5.H5Zfilter_avail(32025) > 0
