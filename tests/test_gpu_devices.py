"""Every entry point runs on the CONTEXT's device and leaves the calling thread's current device as it found it
(include/deltarice_hip.h).  With two visible devices a context on device 1 is driven while device 0 is current:
plan tables, scratch and the direct-chunk path's buffers must land on device 1 (they are dereferenced by kernels that
run there).  One-device boxes check what can be checked: the accessor and that the current device is not disturbed."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_ctx_device_accessor_and_current_device_kept():
    import deltarice_amd as dr
    before = torch.cuda.current_device()
    c = dr.Context(0)
    try:
        assert c.lib.drx_ctx_device(c._h) == 0
        assert c.lib.drx_ctx_device(None) == -1
        plan = c.plan([3 * 1000, 2 * 4096 + 17], [1000, 4096])  # ragged: every table upload of drx_plan_create
        plan.close()
        assert torch.cuda.current_device() == before
    finally:
        c.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible devices")
def test_plan_on_second_device_while_first_is_current(tmp_path):
    import deltarice_amd as dr
    from oracle import oracle as O
    torch.cuda.set_device(0)
    c = dr.Context(1)
    try:
        rng = np.random.default_rng(77)
        lens = [512, 7000, 20000, 300]
        samples = [40 * 512, 9 * 7000 + 5, 3 * 20000, 64 * 300]
        xs = [rng.normal(0, 12, n).astype(np.int16) for n in samples]
        plan = c.plan(samples, lens)  # created while device 0 is current
        assert torch.cuda.current_device() == 0
        xd = torch.from_numpy(np.concatenate(xs)).to(c.device)
        enc = plan.encode(xd)
        w, off = enc.to_numpy()
        ref = [O.encode_chunk(x, (8, L)) for x, L in zip(xs, lens)]
        assert np.array_equal(w, np.concatenate(ref))
        y = plan.decode(enc).cpu().numpy()
        assert np.array_equal(y, np.concatenate(xs))
        # uniform plans (scratch of the parallel walks / block decoder / pieces encoder) and the host path
        up = c.plan_uniform(2, 32 * 9000, (8, 9000))
        x2 = rng.normal(0, 30, 2 * 32 * 9000).astype(np.int16)
        e2 = up.encode(torch.from_numpy(x2).to(c.device))
        assert np.array_equal(up.decode(e2).cpu().numpy(), x2)
        b = c.filter_chunk(xs[1], (8, 7000))
        assert b == ref[1].tobytes()
        assert c.filter_chunk(b, (8, 7000), reverse=True) == xs[1].tobytes()
        assert torch.cuda.current_device() == 0
        # the direct-chunk file <-> VRAM path allocates with raw hipMalloc: same rule
        from deltarice_amd import h5io
        rows, cols = 50, 1000
        x3 = rng.normal(0, 9, rows * cols).astype(np.int16)
        path = str(tmp_path / "dev1.h5")
        h5io.write(c, path, "d", torch.from_numpy(x3).to(c.device), rows, cols, 20, 8, cols)
        y3 = torch.empty(rows * cols, dtype=torch.int16, device=c.device)
        h5io.read(c, path, "d", y3)
        assert np.array_equal(y3.cpu().numpy(), x3)
        assert torch.cuda.current_device() == 0
    finally:
        c.close()
