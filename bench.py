#!/usr/bin/env python3
"""Headline benchmark: encode+decode GB/s (int16 in) and compression ratio.

One *step* = one pass of the hot path over one batch: encode the whole batch of
chunks, then decode it, inputs and outputs resident in HBM.  Default workload is
BASELINE.json configs[1]: 1M x 7000 int16 Gaussian(sigma=10), m=8, framed as 500
HDF5 chunks of 2000 x 7000 (the reference's Nab chunk shape, docs/Performance.md:16).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: weak scaling -- every rank encodes/decodes its own 1M-waveform shard on its own
GPU; the only collective is the all-gather of encoded sizes (deltarice_amd/dist.py).
Without a launcher (WORLD_SIZE unset) `--gpus N` starts the N ranks itself: the parent
touches no GPU, checks that N devices exist, spawns one child per GPU (RANK/LOCAL_RANK/
WORLD_SIZE/MASTER_* in the environment, rendezvous on 127.0.0.1) and exits with their code.
Every rank verifies the group's world size against --gpus; the JSON line carries
`ranks_seen`, each rank's device and each rank's encoded size.

Rank 0 prints ONE JSON line.  `roofline` is for the decode kernel (north_star's
target): algorithmic bytes = 2*(1+ratio) per sample over the kernel's mean duration
measured with HIP events on the codec's stream.  `cpu_baseline` times the reference's
own filter (oracle/_ref, built from src/deltaRice.c) -- or this repo's port of it when
that library is absent -- on a bounded sample of the same workload on the host cores.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (the driver's round-end command is --steps 20 --warmup 5; the kernels of the first three or four steps of a process run
    # 5-15 % slower than the rest -- profiles/r04_notes.md section 15 -- so the defaults are the same)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--waves", type=int, default=1_000_000, help="waveforms per GPU")
    ap.add_argument("--wave-len", type=int, default=7000)
    ap.add_argument("--chunk-waves", type=int, default=2000, help="waveforms per HDF5 chunk")
    ap.add_argument("--m", type=int, default=8, help="RiceParameter")
    ap.add_argument("--dist", choices=["gauss", "ar1"], default="gauss")
    ap.add_argument("--decode-impl", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0: skip)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; "
                    "gloo only to rehearse the multi-rank flow on one GPU)")
    ap.add_argument("--debug-flags", type=int, default=0, help="kernel ablation switches (profiling only; results invalid)")
    ap.add_argument("--collect-traffic", dest="collect_traffic", action="store_true", default=None,
                    help="measure roofline.traffic in THIS run: two rocprofv3 --pmc passes of this command as child processes "
                         "before this process touches the GPU (default: on for --gpus 1 with the default workload)")
    ap.add_argument("--no-collect", dest="collect_traffic", action="store_false", help="do not run the PMC passes")
    return ap.parse_args()


def _traffic_tools():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_traffic_json", os.path.join(ROOT, "profiles", "make_traffic_json.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def collect_traffic(a):
    """HBM bytes per launch of the headline kernels, measured by this run: the two PMC passes MI355X_MICROARCH.md
    prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass), each a fresh child process with the program directly
    behind `rocprofv3 ... --` (this process has not touched the GPU yet and never execs).  Returns (kernels, source) or
    (None, why)."""
    import shutil
    # Under a profiler already (its preloaded library has initialised the GPU in this very process): starting the passes
    # from here would be an exec behind GPU initialisation, which this pool refuses -- and a profile of a profile anyway.
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    tools = _traffic_tools()
    fwd = ["--waves", str(a.waves), "--wave-len", str(a.wave_len), "--chunk-waves", str(a.chunk_waves), "--m", str(a.m),
           "--dist", a.dist, "--decode-impl", str(a.decode_impl), "--seed", str(a.seed)]
    passes = (("rd", ["FETCH_SIZE", "TCC_EA0_RDREQ_sum"]), ("wr", ["WRITE_SIZE", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]))
    csvs = []
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        env = dict(os.environ, TMPDIR="/tmp")
        for tag, counters in passes:
            out = os.path.join(td, tag)
            cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["-d", out, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0",
                   "--no-collect"] + fwd
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                return None, f"PMC pass {tag} timed out"
            found = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not found:
                return None, f"PMC pass {tag} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            csvs += found
        kernels = tools.summarise(csvs)
    return kernels, ("this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 3 --warmup 1 --cpu-seconds 0` as child "
                     "processes (FETCH_SIZE x 2 per MI355X_MICROARCH.md), mean per launch")


def synth(device, n_waves, L, kind, seed):
    """Synthetic batch generated in HBM (never crosses PCIe)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty(n_waves * L, dtype=torch.int16, device=device)
    slab = max(1, min(n_waves, (1 << 28) // L))  # ~1 GiB of float32 per slab
    for w0 in range(0, n_waves, slab):
        w1 = min(n_waves, w0 + slab)
        if kind == "gauss":
            # README.md:82: normal(0, 10).astype(int16) (truncation toward zero)
            v = torch.randn((w1 - w0) * L, device=device, generator=g).mul_(10.0).to(torch.int16)
            out[w0 * L:w1 * L] = v
        else:
            # AR(1) rho=0.95, marginal sigma=32, stationary start, round to nearest
            rho, sigma = 0.95, 32.0
            e = torch.randn((w1 - w0, L), device=device, generator=g).mul_(sigma * (1 - rho * rho) ** 0.5)
            x = torch.randn(w1 - w0, device=device, generator=g).mul_(sigma)
            e[:, 0] = x
            for t in range(1, L):
                e[:, t].add_(e[:, t - 1], alpha=rho)
            out[w0 * L:w1 * L] = torch.round(e).to(torch.int16).reshape(-1)
    return out


def cpu_baseline(x_host_chunks, opts, budget_s):
    """The CPU filter on a bounded sample of the workload, in child processes (oracle/cpu_time.py) so that
    OMP_NUM_THREADS takes effect: all host threads, then one thread (the reference publishes both,
    docs/Performance.md:24-25).  Only the filter call is timed (buffers are filled / read outside the timer)."""
    n_threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    arr = np.stack([np.ascontiguousarray(c).reshape(-1) for c in x_host_chunks])
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fd, path = tempfile.mkstemp(suffix=".npy", dir=shm)
    os.close(fd)
    try:
        np.save(path, arr)
        legs = {}
        # "all host cores": the box advertises every core of the host, a GPU slot's share of them is far smaller, and the
        # reference's OpenMP loop over 2000 waveforms collapses when over-subscribed (256 threads: slower than one).
        # Probe the team sizes the reference publishes numbers for (32) and a slot's share (16), keep the faster one.
        for name, th, share in (("t16", min(16, n_threads), 0.25), ("t32", min(32, n_threads), 0.25), ("one", 1, 0.5)):
            env = dict(os.environ, OMP_NUM_THREADS=str(th), OMP_PROC_BIND="false")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_time.py"), path, str(opts[0]), str(opts[1]),
                                str(budget_s * share)], env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError("cpu baseline leg failed: " + r.stderr[-2000:])
            legs[name] = json.loads(r.stdout.strip().splitlines()[-1])
        legs["all"] = max(legs["t16"], legs["t32"], key=lambda d: d["value"])
    finally:
        os.unlink(path)
    a, o = legs["all"], legs["one"]
    what = ("H5Z_filter_deltarice of src/deltaRice.c (oracle/_ref, OpenMP build)" if a["kind"] == "reference"
            else "oracle/deltarice_oracle.c (OpenMP)")
    return {
        "value": a["value"], "unit": "GB/s", "cores": a["threads"], "kind": a["kind"], "cpu": a["cpu"],
        "encode_GBps": a["encode_GBps"], "decode_GBps": a["decode_GBps"],
        "one_thread": {"value": o["value"], "encode_GBps": o["encode_GBps"], "decode_GBps": o["decode_GBps"],
                       "cores": o["threads"], "chunks": o["chunks"]},
        "threads_tried": {str(legs[k]["threads"]): legs[k]["value"] for k in ("t16", "t32")}, "host_cpus": n_threads,
        "sample": f"{a['chunks']} chunks ({a['raw_bytes'] / 1e6:.0f} MB raw) of the bench workload with {a['threads']} threads, "
                  f"{o['chunks']} chunks with 1 thread; {what}; only the filter call is timed",
    }


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """GPUs this process could use, counted WITHOUT touching HIP: KFD topology nodes that have SIMDs, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  (torch.cuda.device_count() falls through to hipGetDeviceCount on builds
    without amdsmi, which starts the runtime in this process.)  None where the topology cannot be read."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(a):
    """`--gpus N` without a launcher: start the N ranks here.  This process may initialise the HIP runtime (only as a last
    resort, to count devices); it therefore only ever STARTS child processes and never execs -- the ranks are ordinary
    fresh processes."""
    have = visible_gpus()
    if have is None:
        have = torch.cuda.device_count()
    need = 1 if a.backend != "nccl" else a.gpus
    if have < need:
        raise SystemExit(f"bench.py: --gpus {a.gpus} needs {need} visible GPUs, this host shows {have}")
    port = free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:  # a rank died: the others would wait in the rendezvous for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


def main():
    a = parse_args()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)  # does not return
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            if torch.cuda.device_count() <= local:
                raise SystemExit(f"bench.py: rank {rank} wants GPU {local}, this host shows {torch.cuda.device_count()}")
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            local = 0  # rehearsal: every rank on the one visible GPU
            dist.init_process_group(a.backend)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {a.gpus}")
    # in-run HBM traffic (child processes; nothing in this process has touched the GPU yet)
    traffic_kernels = traffic_note = None
    default_workload = (a.waves, a.wave_len, a.chunk_waves, a.m, a.dist) == (1_000_000, 7000, 2000, 8, "gauss")
    want = a.collect_traffic if a.collect_traffic is not None else (world == 1 and default_workload and not a.debug_flags)
    if want and world == 1:
        traffic_kernels, traffic_note = collect_traffic(a)
    import deltarice_amd as dr
    from deltarice_amd import dist as drdist

    ctx = dr.Context(local)
    ctx.set_option("decode_impl", a.decode_impl)
    ctx.set_option("profile", 1)
    if a.debug_flags:
        ctx.set_option("debug_flags", a.debug_flags)
    dev = ctx.device
    L, W = a.wave_len, a.chunk_waves
    n_waves = (a.waves // W) * W
    n_chunks = n_waves // W
    opts = (a.m, L)
    x = synth(dev, n_waves, L, a.dist, a.seed + rank)
    torch.cuda.synchronize(dev)
    plan = ctx.plan_uniform(n_chunks, W * L, opts)
    words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=dev)
    off = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
    y = torch.empty_like(x)
    raw_bytes = x.numel() * 2

    def step(collect=None):
        plan.encode_async(x, words, off)
        if collect is not None:
            collect["enc"].append(plan.last_timings())
        if world > 1:
            if a.backend == "nccl":
                with torch.cuda.stream(ctx.stream):
                    drdist.gather_encoded_sizes(off[-1])
            else:
                ctx.stream.synchronize()
                drdist.gather_encoded_sizes(off[-1].cpu())
        plan.decode_async(words, off, y)
        if collect is not None:
            collect["dec"].append(plan.last_timings())

    for _ in range(a.warmup):
        step()
    plan.finish()
    total_words = int(off[-1].item())
    assert a.debug_flags or torch.equal(x, y), "round trip failed"  # gating, not timed
    ratio = total_words * 4 / raw_bytes

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # --- timed region: EXACTLY K steps, launch only, no host sync inside -------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.stream.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    plan.finish()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # --- per-kernel durations (HIP events on the codec's stream), outside the timed region
    coll = {"enc": [], "dec": []}
    for _ in range(max(3, min(a.steps, 10))):
        step(coll)
        ctx.stream.synchronize()
    enc_ms = np.median(np.array(coll["enc"]), axis=0)
    dec_ms = np.median(np.array(coll["dec"]), axis=0)
    dec_kernel_ms = float(np.mean(np.array(coll["dec"])[:, 1]))
    algo_bytes = 2.0 * (1.0 + ratio) * x.numel()  # read 2*ratio + write 2 (decode), per sample
    achieved = algo_bytes / (dec_kernel_ms * 1e-3) / 1e9
    pack_ms = float(np.mean(np.array(coll["enc"])[:, 2]))

    # HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, profiles/make_traffic_json.py): measured by THIS run's own PMC
    # passes (collect_traffic(), before the GPU was touched) -- or, where rocprofv3 is missing or a pass failed, read from the
    # newest committed passes of the same command, flagged stale when the kernel sources have changed since
    traffic = traffic_enc = traffic_source = None
    # the encoder that RAN in the steps above (the dispatch goes by what the plan's last encode measured)
    enc_kernel = {1: "k_encode_pack", 2: "k_seg_pack", 3: "k_encode_fused", 4: "k_encode_pieces", 5: "k_encode_stream",
                  6: "k_encode_stream_segs"}.get(plan.last_encode_path(), "?")
    if traffic_kernels is not None:
        traffic = traffic_kernels.get("k_decode_lanes", {}).get("hbm_bytes")
        traffic_enc = (traffic_kernels.get(enc_kernel) or traffic_kernels.get("k_encode_fused", {})).get("hbm_bytes")
        traffic_source = traffic_note
    else:
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
        if tfiles and n_waves == 1_000_000 and L == 7000 and a.m == 8 and a.dist == "gauss":
            with open(tfiles[-1]) as f:
                tj = json.load(f)
            stale = tj.get("kernel_sources_sha16") != _traffic_tools().kernel_sources_sha()
            if not stale:
                traffic = tj["kernels"].get("k_decode_lanes", {}).get("hbm_bytes")
                traffic_enc = (tj["kernels"].get(enc_kernel) or tj["kernels"].get("k_encode_fused", {})).get("hbm_bytes")
            traffic_source = (f"profiles/{os.path.basename(tfiles[-1])} (commit {tj.get('git_head')}): rocprofv3 --pmc passes of this "
                              "command on an earlier run, NOT collected by this run"
                              + ("; dropped: the kernel sources have changed since" if stale else "")
                              + (f"; in-run collection: {traffic_note}" if traffic_note else ""))

    # every rank reports its device, encoded size and kernel times (microseconds): rank 0 prints what it SAW, not what it was told
    rank_info = [[rank, local, total_words * 4, int(round(pack_ms * 1e3)), int(round(dec_kernel_ms * 1e3))]]
    if world > 1:
        t = torch.tensor(rank_info[0], dtype=torch.int64, device=dev if a.backend == "nccl" else "cpu")
        allr = torch.empty(world * 5, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(allr, t)
        rank_info = allr.cpu().reshape(world, 5).tolist()
        if sorted(r[0] for r in rank_info) != list(range(world)):
            raise SystemExit(f"bench.py: ranks seen {rank_info}, expected 0..{world - 1}")

    if rank == 0:
        res = {
            "metric": "encode+decode GB/s (int16 in)",
            "value": world * raw_bytes * a.steps / dt / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "ranks_seen": len(rank_info), "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": f"{n_waves}x{L} int16 {'Gaussian(sigma=10)' if a.dist == 'gauss' else 'AR(1) rho=0.95 sigma=32'}, "
                                   f"m={a.m}, {n_chunks} chunks of {W}x{L} per GPU, encode then decode, HBM resident",
                       "waveforms_per_gpu": n_waves, "wave_len": L, "rice_m": a.m, "chunks_per_gpu": n_chunks,
                       "decode_impl": a.decode_impl},
            "compression_ratio": ratio, "decode_path": plan.last_decode_path(), "encode_path": plan.last_encode_path(),
            "rank_devices": [r[1] for r in rank_info], "rank_encoded_bytes": [r[2] for r in rank_info],
            # per-GPU kernel times next to the wall time: what every rank measured with HIP events on its own stream
            "rank_kernel_ms": {"encode": [r[3] / 1e3 for r in rank_info], "decode": [r[4] / 1e3 for r in rank_info]},
            "backend": (a.backend if world > 1 else None),
            "encode_GBps": raw_bytes / (enc_ms[3] * 1e-3) / 1e9,
            "decode_GBps": raw_bytes / (dec_ms[3] * 1e-3) / 1e9,
            # HIP events on the codec's stream.  encode: [state memset | - | k_encode_stream];
            # decode: [the header-chain walk, k_walk_sparse: 64 chains per chunk | k_decode_lanes]  (decode_path says which
            # decoders ran: DRX_PATH_LANES = behind a separate walk, DRX_PATH_LANES_FUSED = the walk inside the launch)
            "kernel_ms": {"encode_prepare": float(enc_ms[0] + enc_ms[1]), "encode_kernel": float(enc_ms[2]),
                          "decode_prepare": float(dec_ms[0]), "decode_kernel": float(dec_ms[1])},
            "roofline": {"bound": "hbm", "kernel": "k_decode_lanes", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         # true: the two rocprofv3 --pmc child passes ran in this session BEFORE the timed region (A/B scripts
                         # pass --no-collect; profiles/r04_notes.md has both forms measured on one box)
                         "traffic_collected_in_run": traffic_kernels is not None,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": dec_kernel_ms,
                         # the same with the walk kernel in front of it counted in (it moves ~0.1 % of the bytes)
                         "walk_ms": float(np.mean(np.array(coll["dec"])[:, 0])),
                         "frac_with_walk": algo_bytes / ((dec_kernel_ms + float(np.mean(np.array(coll["dec"])[:, 0]))) * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "roofline_encode": {"bound": "hbm", "kernel": enc_kernel, "achieved": algo_bytes / (pack_ms * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": algo_bytes / (pack_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic_enc, "traffic_source": traffic_source,
                                "kernel_ms": pack_ms},
        }
        if world == 1 and a.cpu_seconds > 0:
            nsamp = min(n_chunks, 64)
            chunks = [x[c * W * L:(c + 1) * W * L].cpu().numpy() for c in range(nsamp)]
            res["cpu_baseline"] = cpu_baseline(chunks, opts, a.cpu_seconds)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
