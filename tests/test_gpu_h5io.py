"""GPU: the direct-chunk file <-> VRAM path against the filter path and the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDF5_DIR = os.environ.get("HDF5_DIR", "/opt/conda")


@pytest.fixture(scope="module")
def env():
    import deltarice_amd as dr
    from deltarice_amd import h5io
    if not os.path.exists(h5io.H5IO_PATH):
        pytest.skip("libdeltarice_h5io.so not built (no HDF5 C library?)")
    c = dr.Context(0)
    yield c, h5io
    c.close()


@pytest.fixture(scope="module")
def h5tool(tmp_path_factory):
    d = tmp_path_factory.mktemp("h5tool")
    exe = str(d / "h5_roundtrip")
    subprocess.run(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tests", "h5_roundtrip.c"),
                    f"-I{HDF5_DIR}/include", f"-L{HDF5_DIR}/lib", "-lhdf5", f"-Wl,-rpath,{HDF5_DIR}/lib"], check=True)
    e = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(ROOT, "deltarice_amd", "plugin"))
    return lambda *a: subprocess.run([exe, *map(str, a)], env=e, check=True, capture_output=True, text=True)


@pytest.mark.parametrize("rows,cols,crows,M,L", [(100, 7000, 20, 8, 7000), (64, 4096, 8, 16, 1024), (6, 1000, 2, 8, 1000),
                                                  (103, 7000, 20, 8, 7000), (23, 1000, 8, 8, 1000)])  # rows that do not divide
def test_direct_path_is_file_compatible_both_ways(env, h5tool, tmp_path, rows, cols, crows, M, L):
    from oracle import oracle as O
    ctx, h5io = env
    x = np.random.default_rng(rows + cols).normal(0, 10, (rows, cols)).astype(np.int16)
    xd = torch.from_numpy(x.reshape(-1)).to(ctx.device)
    f1 = tmp_path / "direct.h5"
    st = h5io.write(ctx, str(f1), "test", xd, rows, cols, crows, M, L)
    n_chunks = -(-rows // crows)
    assert st["n_chunks"] == n_chunks and st["raw_bytes"] == x.nbytes
    xp = np.zeros((n_chunks * crows, cols), np.int16)  # what HDF5 hands the filter: the last chunk padded with the fill value
    xp[:rows] = x
    # (a) the ordinary HDF5 read (filter callback, dynamic plugin) sees the same data
    back = tmp_path / "back.bin"
    h5tool("read", f1, back)
    assert np.array_equal(np.fromfile(back, np.int16).reshape(rows, cols), x)
    # (b) the stored chunks are the CPU filter's bytes
    n = int(h5tool("chunks", f1, tmp_path / "chunk").stdout)
    tot = 0
    for c in range(n):
        stored = np.fromfile(f"{tmp_path}/chunk.{c}", np.uint32)
        assert np.array_equal(stored, O.encode_chunk(xp[c * crows:(c + 1) * crows], (M, L)))
        tot += stored.nbytes
    assert tot == st["stored_bytes"]
    # (c) direct read of a direct-written file: by preads at the chunk addresses HDF5 reports (the default for a plain
    # read-only file), and by H5Dread_chunk (any other file; DRX_H5_NO_RAW=1)
    y = torch.empty_like(xd)
    h5io.read(ctx, str(f1), "test", y)
    assert torch.equal(y, xd)
    os.environ["DRX_H5_NO_RAW"] = "1"
    try:
        y.zero_()
        h5io.read(ctx, str(f1), "test", y)
        assert torch.equal(y, xd)
    finally:
        del os.environ["DRX_H5_NO_RAW"]
    # (d) direct read of a file written chunk by chunk through the filter callback ...
    raw, f2 = tmp_path / "raw.bin", tmp_path / "filter.h5"
    x.tofile(raw)
    h5tool("write", f2, raw, rows, cols, crows, M, L)
    y.zero_()
    h5io.read(ctx, str(f2), "test", y)
    assert torch.equal(y, xd)
    # ... and of a file whose chunks were encoded by the CPU oracle
    for c in range(n):
        O.encode_chunk(xp[c * crows:(c + 1) * crows], (M, L)).tofile(f"{tmp_path}/cpu.{c}")
    f3 = tmp_path / "cpu.h5"
    h5tool("writeraw", f3, rows, cols, crows, M, L, tmp_path / "cpu")
    y.zero_()
    h5io.read(ctx, str(f3), "test", y)
    assert torch.equal(y, xd)


def test_direct_read_rejects_what_it_does_not_handle(env, h5tool, tmp_path):
    import deltarice_amd as dr
    ctx, h5io = env
    x = np.random.default_rng(3).normal(0, 20, (8, 1024)).astype(np.int16)
    raw, f = tmp_path / "raw.bin", tmp_path / "fir.h5"
    x.tofile(raw)
    h5tool("write", f, raw, 8, 1024, 2, 8, 1024, 4, 1, 0xFFFFFFFF, 1, 0xFFFFFFFF)  # general prediction filter in cd_values
    y = torch.empty(8 * 1024, dtype=torch.int16, device=ctx.device)
    h5io.read(ctx, str(f), "test", y)  # files written with a general filter are read too
    assert np.array_equal(y.cpu().numpy(), x.reshape(-1))
    # ... and can be written directly: same stored bytes as through the filter callback
    f2 = tmp_path / "fir_direct.h5"
    h5io.write(ctx, str(f2), "test", torch.from_numpy(x.reshape(-1)).to(ctx.device), 8, 1024, 2, 8, 1024, taps=(1, -1, 1, -1))
    n = int(h5tool("chunks", f, tmp_path / "a").stdout)
    assert n == int(h5tool("chunks", f2, tmp_path / "b").stdout) == 4
    for c in range(n):
        assert open(f"{tmp_path}/a.{c}", "rb").read() == open(f"{tmp_path}/b.{c}", "rb").read()
    h5tool("read", f2, tmp_path / "back.bin")  # and the filter callback reads the directly written file
    assert np.array_equal(np.fromfile(tmp_path / "back.bin", np.int16), x.reshape(-1))
    with pytest.raises(dr.DeltaRiceError):
        h5io.read(ctx, str(tmp_path / "missing.h5"), "test", y)
    with pytest.raises(dr.DeltaRiceError) as e:
        h5io.read(ctx, str(f), "test", y[:100])
    assert e.value.status in (3, 5)
