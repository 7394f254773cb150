#!/usr/bin/env python3
"""bench.py -- GraphPOPE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one synthetic Flickr-shaped input already resident in HBM
(BASELINE.json configs[1]: Flickr geodesic-stochastic, 256 anchors, 3-layer GraphSAGE on 1 MI355X):

    edge_index -> CSR -> multi-source BFS for all anchors -> hop planes -> [N, F+K] float32 features

`value` is POPE embeddings/s = N * K / step time (the first half of BASELINE.json's metric); the second half,
SAGE nodes/s (seed nodes through fwd + bwd + Adam on pre-sampled [25, 10] fan-out batches over those features),
is reported in the same line under "sage".  For N > 1 GPUs the POPE pass is WEAK scaled: every rank owns 256
anchors (K_total = 256 * N), planes are all-gathered (RCCL) and every rank materialises the full
[N, F + K_total] matrix, as every DDP rank of the reference needs it.

Rank 0 prints ONE JSON line.  At N = 1 it also carries
  roofline               the dominant kernel (k_bfs_level), duration measured live with HIP events on the launch stream
  cpu_baseline           Baseline A: the reference's CPU path (NetworkX per-pair loop under multiprocessing.Pool(6),
                         utils.py:92-107; main.py:39 default) on a bounded node sample; *_all_cores: the same on the box's CPU
                         share (16 workers);
                         cpu_baseline_bfs: one C BFS per anchor (the honest CPU algorithm)
  boundary_host_to_host  SURVEY.md §8(d)'s primary metric: utils.Graphpope() from CPU tensors to the returned CPU tensor,
                         with the PCIe roofline of the bytes that cross the link
  pagerank               biased anchor selection on the GPU against nx.pagerank (scores bit-identical)
  kmeans                 K-means anchors of the node2vec branch on the GPU against scikit-learn
  pairwise               configs[2]: node2vec-euclidean, 256 anchors, with the f32-MFMA roofline and the sklearn baseline
  config3 / config4      configs[3] (1 024 anchors) and configs[4] (R-MAT scale 22, 512 anchors) on this one GPU
  sage                   SAGE nodes/s + per-kernel view of layer 0
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher on the command line (WORLD_SIZE unset): start the N ranks as CHILD
    processes -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`,
    exactly the command the driver's contract names -- wait, and exit with their code; rank 0's JSON line goes to the inherited
    stdout.  This runs before torch is imported: the parent never touches the GPU and nothing is exec-replaced."""
    if "WORLD_SIZE" in os.environ:
        return
    gpus = 1
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            gpus = int(argv[i + 1])
        elif a.startswith("--gpus="):
            gpus = int(a.split("=", 1)[1])
    if gpus <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sock:                           # a free port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    print("[bench] WORLD_SIZE unset with --gpus %d: starting the ranks as child processes: %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from graphpope_amd import _lib, engine, synth  # noqa: E402
from graphpope_amd import distributed as pdist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3     # same guide: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PCIE_GBS = 63.0              # same guide: PCIe Gen5 x16 host link (spec)
F = 500                      # /root/reference/main.py:77-79
K_PER_GPU = 256
BATCH = 1550                 # main.py:44
HIDDEN = 256
NUM_WORKERS = 6              # main.py:39 --num_workers default: the reference's Pool size


def _event():
    return torch.cuda.Event(enable_timing=True)


def pope_step(x, ei, n, anchors, world):
    """One full geodesic GraphPOPE pass on the device; returns the [N, F+K] tensor."""
    if world == 1:
        return engine.geodesic_run(x if x.shape[1] else None, ei, n, anchors, reuse_workspace=True)[0]
    csr = engine.build_csr(ei, n, defer_check=True)
    return pdist.sharded_geodesic_features(x, n, anchors, None,
                                           bfs_fn=lambda a: engine.bfs(csr, a),
                                           finalize_fn=engine.finalize, finalize_all_fn=engine.finalize_shards,
                                           begin_fn=lambda a: engine.PendingBfs(csr, a), copy_x_fn=engine.copy_features)


def pope_phases(x, ei, n, anchors, timers):
    """The same pass through the separate entry points, bracketed by HIP events on the launch stream."""
    k = len(anchors)
    ev = [_event() for _ in range(4)]
    ev[0].record()
    csr = engine.build_csr(ei, n, defer_check=True)
    ev[1].record()
    hp = engine.bfs(csr, anchors)
    ev[2].record()
    out = torch.empty((n, F + k), dtype=torch.float32, device=x.device)
    engine.finalize(hp.planes, hp.n_hop_bits, n, k, x, F, out, 0)
    ev[3].record()
    torch.cuda.synchronize()
    timers["csr"].append(ev[0].elapsed_time(ev[1]))
    timers["bfs"].append(ev[1].elapsed_time(ev[2]))
    timers["finalize"].append(ev[2].elapsed_time(ev[3]))
    timers["max_hop"] = hp.max_hop
    timers["n_hop_bits"] = hp.n_hop_bits
    return out


def finalize_kernel_ms(x, ei, n, anchors, reps=20):
    """Average duration of the finalise launch: `reps` launches back to back on the launch stream between two HIP events.
    (Its 463 MB exceed L2 + Infinity Cache, so a repeat does not find its inputs cached.)"""
    k = len(anchors)
    hp = engine.bfs(engine.build_csr(ei, n, defer_check=True), anchors)
    out = torch.empty((n, F + k), dtype=torch.float32, device=x.device)
    for _ in range(3):
        engine.finalize(hp.planes, hp.n_hop_bits, n, k, x, F, out, 0)
    e0, e1 = _event(), _event()
    e0.record()
    for _ in range(reps):
        engine.finalize(hp.planes, hp.n_hop_bits, n, k, x, F, out, 0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def finalize_kernel_name(n, k, f, has_x, shards):
    """The finalise kernel the library launches for this shape (pope_finalize_kernel_name: the choice finalize_enqueue makes)."""
    buf = ctypes.create_string_buffer(64)
    _lib.check(_lib.load().pope_finalize_kernel_name(n, k, f, 1 if has_x else 0, shards, buf, 64))
    return buf.value.decode()


def level_kernel_name(n, k):
    """The level kernel the library launches for a BFS over n nodes from k anchors (pope_level_kernel_name)."""
    buf = ctypes.create_string_buffer(64)
    _lib.check(_lib.load().pope_level_kernel_name(n, k, buf, 64))
    return buf.value.decode()


def level_kernel_times(ei, n, anchors, reps):
    """Average duration of a k_bfs_level launch ON THE HOT PATH (pope_geodesic_run, the call the timed steps make): HIP
    events recorded by the library on the launch stream around each enqueued run of level launches (no events between
    the kernels, so this is what rocprofv3 --stats averages too: every launch of the kernel, trailing early-exit ones
    included -- the call sizes its run by the depth the previous call found, so after the first call there are none).
    Returns (ms per launch, launches per BFS, levels that did work, HopPlanes)."""
    lib = _lib.load()
    for _ in range(2):
        engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
    torch.cuda.synchronize()
    lib.pope_profile_levels(2)
    for _ in range(reps):
        hp = engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)[1]
    torch.cuda.synchronize()
    cap = 4096
    lv = (ctypes.c_int32 * cap)()
    ex = (ctypes.c_float * cap)()
    cnt = lib.pope_profile_read(lv, ex, cap)
    lib.pope_profile_levels(0)
    launches = sum(-lv[i] for i in range(cnt))
    total_ms = sum(ex[i] for i in range(cnt))
    active = hp.max_hop + 1                     # levels 1 .. max_hop reach something, level max_hop + 1 proves the end
    return total_ms / launches, launches // reps, active, hp


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only; run BEFORE anything touches the GPU: Baseline A forks a multiprocessing.Pool)
# ------------------------------------------------------------------------------------------------
def cpu_baselines_before_gpu(ei, n, anchors, sample_nodes):
    """Baseline A (the >= 50x contract): the reference's per-pair NetworkX loop under
    multiprocessing.Pool(W), W = 6 (CLI default) and W = every core, on a bounded random node sample; Baseline B: one C
    BFS per anchor (oracle), all anchors, one core."""
    from oracle import oracle
    res = {}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)             # a one-GPU box's CPU share (the host shows all 256 hardware threads to every tenant)
    nodes = np.random.RandomState(0).choice(n, sample_nodes, replace=False)
    want = None
    try:
        import networkx  # noqa: F401
        for key, w in (("cpu_baseline", NUM_WORKERS), ("cpu_baseline_all_cores", cores)):
            if key == "cpu_baseline_all_cores" and w == NUM_WORKERS:
                continue
            emb, tm = oracle.geodesic_pairs_networkx_pool(ei, n, anchors, nodes, w)
            pairs = emb.shape[0] * emb.shape[1]
            per_pair = tm["pool_s"] / pairs
            res[key] = {
                "value": pairs / tm["pool_s"], "unit": "embeddings/s", "cores": w, "kind": "port",
                "sample": f"{emb.shape[0]} random nodes x {len(anchors)} anchors = {pairs} pairs of the same graph, nx.shortest_path per pair "
                          f"under multiprocessing.Pool({w}) (utils.py:64-107), {tm['pool_s']:.1f} s (+ DiGraph build {tm['graph_build_s']:.1f} s, not counted)",
                "graph_build_s": tm["graph_build_s"], "pool_s": tm["pool_s"],
                "extrapolated_full_graph_s": tm["graph_build_s"] + per_pair * n * len(anchors),
            }
            want = (nodes[: emb.shape[0]], emb)
    except ImportError:
        pass
    t0 = time.perf_counter()
    hops = oracle.geodesic_hops(ei, n, anchors)
    emb_all = oracle.hops_to_embedding(hops)
    dt = time.perf_counter() - t0
    res["cpu_baseline_bfs"] = {
        "value": n * len(anchors) / dt, "unit": "embeddings/s", "cores": 1, "kind": "port",
        "sample": f"full N x K = {n} x {len(anchors)}, oracle/pope_oracle.c one BFS per anchor over the reversed CSR, {dt:.2f} s",
    }
    if want is not None:
        res["cpu_baseline"]["matches_c_oracle"] = bool(np.array_equal(want[1], emb_all[want[0]]))
    res.setdefault("cpu_baseline", res["cpu_baseline_bfs"])
    return res, hops


# ------------------------------------------------------------------------------------------------
# SURVEY.md §8(d) primary metric: the host -> host boundary call
# ------------------------------------------------------------------------------------------------
def boundary_check(out, x_cpu, want_hops):
    from oracle import oracle
    return bool(np.array_equal(out.numpy()[:, F:].view(np.uint32), oracle.hops_to_embedding(want_hops).view(np.uint32))
                and torch.equal(out[:, :F], x_cpu))


def boundary_leg(ei_np, n, x_cpu, want_hops, reps=5):
    """utils.Graphpope(data, ...) exactly as Flickr.setup calls it (main.py:94-98): CPU tensors in, CPU tensor out."""
    from graphpope_amd import utils as gp

    class Data:
        pass
    d = Data()
    d.x, d.edge_index, d.num_nodes = x_cpu, torch.as_tensor(ei_np), n
    import contextlib
    times = []
    out = None
    with contextlib.redirect_stdout(sys.stderr):           # the reference's banners (utils.py:141-146) stay off the JSON line
        for _ in range(reps + 1):
            gp.clear_cache()
            out = None                                   # the previous result is released OUTSIDE the timed call (a caller keeps its result)
            np.random.seed(42)
            t0 = time.perf_counter()
            out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", K_PER_GPU, None, NUM_WORKERS)
            times.append(time.perf_counter() - t0)
    gp.clear_cache()
    trace = (ctypes.c_double * 8)()
    _lib.load().pope_debug_boundary_trace(trace)      # phases of the LAST call's result assembly (csrc/host.cc)
    phases = dict(zip(("madvise", "wait_for_host_copy", "register_pages", "enqueue_dma", "join_threads", "stream_sync", "unregister", "total"),
                      (round(float(v), 3) for v in trace)))
    e, k = ei_np.shape[1], K_PER_GPU
    crossed = 16.0 * e + 1.0 * n * k + 1024.0          # edge_index int64 up, [N, K] hop codes (one byte each) + their 256 floats down; x stays on the host
    warm = float(np.median(times[1:]))
    ok = boundary_check(out, x_cpu, want_hops) if want_hops is not None else None      # None: the caller checks later (boundary_check)
    return {
        "_out": out if want_hops is None else None,
        "ms": warm * 1e3, "first_call_ms": times[0] * 1e3, "embeddings_per_s": n * k / warm, "all_calls_ms": [t * 1e3 for t in times],
        "last_call_assembly_phases_ms": phases,
        "what": "utils.Graphpope(data, 'flickr', 'geodesic', 'stochastic', 256) from CPU tensors (x [N, 500] f32, edge_index "
                "[2, E] int64) to the returned PAGEABLE CPU [N, 756] f32 tensor: anchor draw, H2D of edge_index straight from the "
                "caller's pages, CSR + BFS on the GPU, the [N, 256] block brought down as one byte per element (0 = no path, hops + 1 "
                "otherwise, plus the GPU's table of the 256 floats the bytes stand for) through a 3 x 8 MB pinned ring that is allocated "
                f"once per process, data.x copied host to host and the bytes looked up into floats by {engine.host_threads()} threads; the "
                "result's pages come from a one-entry pool, so calls after the first take no page faults "
                f"(ms: median of {reps} calls after the first; first_call_ms: the call a process actually makes -- the reference "
                "memoises, utils.py:195-208 -- here behind the timed geodesic steps; host-side time, 4.9-8.9 ms between runs of the same code)",
        "pcie": {"bound": "pcie", "bytes_crossed": crossed, "peak": PCIE_GBS, "unit": "GB/s",
                 "floor_ms": crossed / (PCIE_GBS * 1e9) * 1e3, "achieved": crossed / warm / 1e9,
                 "frac": crossed / warm / 1e9 / PCIE_GBS},
        "host_copy_bytes": 2.0 * 4.0 * n * F + 5.0 * n * k,       # x read + written, codes read from the ring + floats written
        "bound_note": "with a quarter of the embedding bytes on PCIe the call is bound by the host threads' copy (host_copy_bytes), not by the link",
        "bit_exact_vs_cpu": ok,
    }


# ------------------------------------------------------------------------------------------------
# configs[2]: node2vec-euclidean pairwise, 256 anchors
# ------------------------------------------------------------------------------------------------
def pairwise_leg(n, anchors, x, dev, steps):
    lib = _lib.load()
    d = 128
    table_cpu = torch.randn((n, d), generator=torch.Generator().manual_seed(0))      # generate_node2vec_embedding.py:23-28: an untrained N(0,1) table
    table = table_cpu.to(dev)
    k = len(anchors)
    anchors_dev = torch.as_tensor(np.asarray(anchors, dtype=np.int64), device=dev)    # inputs resident in HBM before the timed region
    for _ in range(3):
        out = engine.pairwise_features(x, table, anchors_dev, "euclidean")
    ev = [_event() for _ in range(2)]
    ev[0].record()
    for _ in range(steps):
        out = engine.pairwise_features(x, table, anchors_dev, "euclidean")
    ev[1].record()
    torch.cuda.synchronize()
    call_ms = ev[0].elapsed_time(ev[1]) / steps
    # the MFMA kernel on its own: the embedding-only entry point (no feature copy), events around the library call
    a = table.index_select(0, torch.as_tensor(np.asarray(anchors, dtype=np.int64), device=dev)).contiguous()
    emb = torch.empty((n, k), dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.pope_pairwise_scratch_bytes(n, k, d), dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def emb_only():
        _lib.check(lib.pope_pairwise_minmax(_lib.ptr(table), n, d, _lib.ptr(a), k, _lib.METRIC["euclidean"], _lib.ptr(emb), k, 0,
                                            _lib.ptr(scratch), scratch.numel(), stream))
    for _ in range(3):
        emb_only()
    ev[0].record()
    for _ in range(steps):
        emb_only()
    ev[1].record()
    torch.cuda.synchronize()
    emb_ms = ev[0].elapsed_time(ev[1]) / steps
    flops = 2.0 * n * k * d
    tile_traffic, tile_traffic_src = None, None
    for pmc_name in ("r04_pairwise_pmc.json", "r02_pairwise_pmc.json"):     # HBM bytes of the tile kernel from the committed counter passes, newest first
        try:
            with open(os.path.join(ROOT, "profiles", pmc_name)) as fh:
                kk = json.load(fh)["kernels"]["k_pairwise_persistent"]
            tile_traffic = kk["hbm_read_bytes"] + kk["hbm_write_bytes"]
            tile_traffic_src = f"profiles/{pmc_name} (rocprofv3 --pmc passes of tools/pairwise_time.py, a separate run: k_pairwise_persistent only)"
            break
        except (OSError, ValueError, KeyError):
            continue
    res = {
        "workload": f"configs[2]: node2vec-euclidean, X [{n}, {d}] f32 N(0,1) (torch seed 0), {k} anchors (np seed 42), "
                    "features [N, 500] resident in HBM -> [N, 756] f32 in HBM (feature copy + pairwise + column min-max)",
        "ms_per_call": call_ms, "embeddings_per_s": n * k / (call_ms * 1e-3),
        "embedding_only_ms": emb_ms,
        "hbm_floor": {"bytes": 4.0 * n * (d + 2 * F + 2 * k), "ms_at_6.29TBs": 4.0 * n * (d + 2 * F + 2 * k) / 6.29e12 * 1e3,
                      "note": "table in, features in and out, embedding out, each once; the call also re-reads and re-writes the embedding "
                              "in the min-max pass (+8NK bytes): with the feature copy running beside the MFMA kernel the whole call "
                              "is HBM-bound, the embedding alone is MFMA-bound"},
        "roofline": {"kernel": "k_pairwise_persistent + min-max fold + scaling pass", "bound": "mfma",
                     "achieved": flops / (emb_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / (emb_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, "traffic": tile_traffic, "traffic_source": tile_traffic_src,
                     "algorithmic_flops": flops,
                     "note": "2*N*K*D flops of the distance matrix over the time of the whole embedding call without the feature "
                             "copy (norms + tile kernel + min-max fold + scaling pass), HIP events on the launch stream; the tile "
                             "kernel alone is 80-95 us of it by rocprofv3 (profiles/r02_pairwise_kernel_stats.csv: min 79.9, avg 95.1)"},
        "whole_call_frac_of_mfma_peak": flops / (call_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
    }
    # parity on a row sample + CPU baseline (sklearn, the calls of utils.py:174-176, all cores)
    try:
        from sklearn.metrics.pairwise import euclidean_distances
        from sklearn.preprocessing import MinMaxScaler
        X = table_cpu.numpy()
        A = X[np.asarray(anchors, dtype=np.int64)]
        t0 = time.perf_counter()
        e = euclidean_distances(X, A)
        scaler = MinMaxScaler()
        scaler.fit(e)
        want = scaler.transform(e)
        dt = time.perf_counter() - t0
        got = out[:, F:].cpu().numpy()
        res["max_abs_err_vs_sklearn"] = float(np.abs(got - want).max())
        res["cpu_baseline"] = {"value": n * k / dt, "unit": "embeddings/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"full [{n}, {k}]: sklearn euclidean_distances + MinMaxScaler fit/transform (utils.py:174-176), {dt:.2f} s"}
    except Exception as exc:                                   # never take the bench down
        res["cpu_baseline"] = {"error": repr(exc)}
    return res


# ------------------------------------------------------------------------------------------------
# configs[3] and configs[4] on this one GPU
# ------------------------------------------------------------------------------------------------
def config3_leg(x, ei, n, steps):
    anchors = synth.seeded_anchors(n, 1024, 42)
    for _ in range(2):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del out
    # the level kernel of this shape (16 words per node as two 8-word tiles, a wave each), timed like the headline's; the byte model
    # E (8 + 8 W) + N 16 W per level that works.  The 11.4 MB frontier does not fit one XCD's L2: rows come across the fabric.
    e = int(ei.shape[1])
    lvl_ms, launches, active, hp = level_kernel_times(ei, n, anchors, reps=10)
    wp = hp.planes.shape[2]
    lvl_bytes = (e * (8.0 + 8.0 * wp) + n * 16.0 * wp) * active / launches
    f = 0 if x is None else x.shape[1]
    fin_bytes = 4.0 * n * f + 4.0 * n * (f + 1024) + 8.0 * n * wp * (1 + hp.n_hop_bits)
    pmc, pmc_src = config4_pmc(3)
    pmc = pmc or {}
    return {"workload": "configs[3] on ONE GPU: flickr-shaped graph, 1024 anchors (np seed 42, 1020 distinct), x resident -> [N, 1524] f32",
            "ms_per_step": dt * 1e3, "embeddings_per_s": n * 1024 / dt,
            "roofline": {"kernel": level_kernel_name(n, 1024), "bound": "hbm", "achieved": lvl_bytes / (lvl_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": lvl_bytes / (lvl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc.get("k_bfs_level_hbm_bytes_per_launch"),
                         "traffic_source": pmc_src, "algorithmic_bytes_per_launch": lvl_bytes, "avg_launch_ms": lvl_ms, "launches_per_step": launches},
            "finalize": {"kernel": finalize_kernel_name(n, 1024, f, f > 0, 1), "algorithmic_bytes_per_launch": fin_bytes,
                         "traffic": pmc.get("k_finalize_hbm_bytes_per_launch")},
            "note": "the 8-GPU form shards 128 anchors per rank (tests/test_configs_gpu.py runs it sharded over 2 ranks)"}


def config4_pmc(cfg=4):
    """HBM-side bytes per launch of configs[4]'s (or configs[3]'s) kernels from the committed rocprofv3 --pmc passes
    (profiles/r05_config4_pmc.json / r05_config3_pmc.json, separate runs of tools/r05_config{4,3}_profile.sh: not measured live)."""
    name = "r05_config%d_pmc.json" % cfg
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh), "profiles/" + name
    except (OSError, ValueError):
        return None, None


def config4_leg(dev, steps=3):
    from oracle import oracle
    lib = _lib.load()
    t0 = time.perf_counter()
    ei_np, n = synth.rmat(22, edge_factor=8, seed=1)
    gen = time.perf_counter() - t0
    k = 512
    anchors = synth.seeded_anchors(n, k, 42)
    ei = torch.as_tensor(ei_np, device=dev)
    for _ in range(2):
        out, hp = engine.geodesic_run(None, ei, n, anchors, reuse_workspace=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out, hp = engine.geodesic_run(None, ei, n, anchors, reuse_workspace=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    cols = np.random.RandomState(0).choice(k, 8, replace=False)
    t0 = time.perf_counter()
    want = oracle.geodesic_hops(ei_np, n, anchors[cols])
    cpu = time.perf_counter() - t0
    got = engine.hop_matrix(hp)[:, torch.as_tensor(cols, device=dev)].cpu().numpy()
    e = ei_np.shape[1]
    wp = hp.planes.shape[2]
    bits = hp.n_hop_bits
    # the level kernel on this graph: HIP events around every level launch of the hot path (pope_profile_levels(1)); the average over
    # EVERY launch is what rocprofv3 --stats averages, the densest level is the one the HBM roof applies to
    lib.pope_profile_levels(1)
    for _ in range(steps):
        engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
    torch.cuda.synchronize()
    cap = 4096
    lv = (ctypes.c_int32 * cap)()
    ex = (ctypes.c_float * cap)()
    cnt = lib.pope_profile_read(lv, ex, cap)
    lib.pope_profile_levels(0)
    per = {}
    for i in range(cnt):
        per.setdefault(lv[i], []).append(ex[i])
    level_ms = {int(l): float(np.median(v)) for l, v in sorted(per.items())}
    launches = len(level_ms)
    avg_ms = sum(level_ms.values()) / launches
    dense_bytes = e * (8.0 + 8.0 * wp) + n * 16.0 * wp                # DESIGN.md section 5 model: per slot index + neighbour row, per node seen read + frontier write
    dense_level = max(level_ms, key=level_ms.get)
    pmc, pmc_src = config4_pmc()
    pmc = pmc or {}
    traffic_avg = pmc.get("k_bfs_level_hbm_bytes_per_launch")
    # the finalise kernel of this shape: launches queued back to back between two HIP events
    fin_out = torch.empty((n, k), dtype=torch.float32, device=dev)
    planes = hp.valid().contiguous()
    for _ in range(2):
        engine.finalize(planes, bits, n, k, None, 0, fin_out, 0)
    ev = [_event(), _event()]
    ev[0].record()
    for _ in range(5):
        engine.finalize(planes, bits, n, k, None, 0, fin_out, 0)
    ev[1].record()
    torch.cuda.synchronize()
    fin_ms = ev[0].elapsed_time(ev[1]) / 5
    fin_bytes = 4.0 * n * k + 8.0 * n * wp * (1 + bits)
    del fin_out
    res = {"workload": f"configs[4] on ONE GPU: R-MAT scale 22 (a,b,c,d)=(.57,.19,.19,.05), N={n}, E={e} CSR slots, 512 anchors "
                       "(np seed 42), F=0: edge_index resident -> [N, 512] f32 (graph generation outside the timed region)",
           "ms_per_step": dt * 1e3, "embeddings_per_s": n * k / dt, "max_hop": hp.max_hop, "graph_generation_s": gen,
           "sampled_columns_bit_exact": bool(np.array_equal(got, want)),
           # the dominant kernel's roofline entry is its DENSEST launch (every slot gathers: the byte model holds); the model overstates the
           # sparse launches (the live table skips their gathers), so the average over all launches is given as measured traffic only
           "roofline": {"kernel": level_kernel_name(n, k), "bound": "hbm", "achieved": dense_bytes / (level_ms[dense_level] * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dense_bytes / (level_ms[dense_level] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": pmc.get("k_bfs_level_hbm_bytes_densest_launch"), "traffic_source": pmc_src,
                        "level": dense_level, "launch_ms": level_ms[dense_level],
                        "algorithmic_bytes_per_launch": dense_bytes, "gather_roof_ms": e / 56.0e9 * 1e3,
                        "level_ms": {str(l): round(v, 3) for l, v in level_ms.items()},
                        "every_launch": {"avg_launch_ms": avg_ms, "launches_per_step": launches, "traffic": traffic_avg,
                                         "traffic_rate": None if not traffic_avg else traffic_avg / (avg_ms * 1e-3) / 1e9},
                        "note": "W = 8 words per node, the whole row in one gather; every neighbour row is a random 64-byte gather that "
                                "costs a 128-byte line from the fabric (tools/micro/gather_rows.hip: 56 G lines/s chip-wide whatever the row size, "
                                "profiles/r05_micro_gather_rows.txt), so the densest level cannot take less than E / 56 G/s (gather_roof_ms); HIP events "
                                "between the launches, one level per launch"},
           "roofline_finalize": {"kernel": finalize_kernel_name(n, k, 0, False, 1), "bound": "hbm", "achieved": fin_bytes / (fin_ms * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fin_bytes / (fin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": pmc.get("k_finalize_hbm_bytes_per_launch"), "algorithmic_bytes_per_launch": fin_bytes, "avg_launch_ms": fin_ms,
                                 "store_roof_ms": 1.49,
                                 "note": "8.59 GB of columns written + 1.34 GB of planes read; a kernel that only stores the columns (no loads, no "
                                         "arithmetic: tools/micro/column_fill.hip) takes 1.49 ms = 5.8 TB/s on this part (store_roof_ms)"},
           "cpu_baseline": {"value": n * 8 / cpu, "unit": "embeddings/s", "cores": 1, "kind": "port",
                            "sample": f"8 of the 512 anchor columns (RandomState(0)), full N, oracle/pope_oracle.c one BFS per anchor, {cpu:.1f} s"}}
    del out, hp, ei, planes
    engine._WORKSPACE.clear()
    torch.cuda.empty_cache()
    return res


def pagerank_leg(ei_np, n, ei):
    """Biased anchor selection: sampling_method='pagerank' on the GPU against nx.pagerank."""
    t0 = time.perf_counter()
    got = engine.pagerank(ei, n)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = engine.pagerank(ei, n)
    torch.cuda.synchronize()
    gpu = time.perf_counter() - t0
    res = {"gpu_ms": gpu * 1e3, "gpu_first_call_ms": first * 1e3,
           "what": "engine.pagerank: two canonical CSR builds + SpMV power iteration (float64, SciPy's accumulation order), "
                   "edge_index resident; utils.py:26-30"}
    try:
        import networkx as nx
        t0 = time.perf_counter()
        g = nx.DiGraph()
        g.add_nodes_from(range(n))
        g.add_edges_from(zip(ei_np[0].tolist(), ei_np[1].tolist()))
        want = nx.pagerank(g)
        cpu = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": cpu, "unit": "s", "cores": 1, "kind": "port",
                               "sample": "the reference's call on the full graph: DiGraph build + nx.pagerank (SciPy power iteration)"}
        res["scores_bit_identical_to_networkx"] = bool(np.array_equal(got, np.array([want[v] for v in range(n)])))
    except Exception as exc:
        res["cpu_baseline"] = {"error": repr(exc)}
    return res


def kmeans_leg(n, dev):
    """K-means anchors of the node2vec branch (utils.py:168-170) at configs[2]'s size: GPU against the reference's scikit-learn call."""
    table_cpu = torch.randn((n, 128), generator=torch.Generator().manual_seed(0))
    table = table_cpu.to(dev)
    np.random.seed(9)
    engine.kmeans_centers(table, K_PER_GPU)
    torch.cuda.synchronize()
    np.random.seed(9)
    t0 = time.perf_counter()
    c = engine.kmeans_centers(table, K_PER_GPU)
    torch.cuda.synchronize()
    gpu = time.perf_counter() - t0
    res = {"gpu_ms": gpu * 1e3, "what": f"engine.kmeans_centers (the opt-in GPU clustering, GRAPHPOPE_KMEANS=gpu; by default the anchors come from the reference's own scikit-learn call): X [{n}, 128] f32 N(0,1) resident, K = {K_PER_GPU}: k-means++ seeding "
                                         "(7 local trials) + Lloyd iterations (MFMA tile assignment) until scikit-learn's stopping rule"}
    try:
        from sklearn.cluster import KMeans
        m = 16384                                              # bounded sample of the rows: the full table takes minutes on the host
        np.random.seed(9)
        t0 = time.perf_counter()
        km = KMeans(n_clusters=K_PER_GPU).fit(table_cpu[:m].numpy())
        cpu = time.perf_counter() - t0
        np.random.seed(9)
        t0 = time.perf_counter()
        engine.kmeans_centers(table[:m].contiguous(), K_PER_GPU)
        torch.cuda.synchronize()
        gpu_m = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": cpu, "unit": "s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"the reference's call KMeans(n_clusters={K_PER_GPU}).fit(X[:{m}]) (scikit-learn, {km.n_iter_} Lloyd "
                                         f"iterations); the GPU path on the same {m} rows: {gpu_m * 1e3:.1f} ms"}
    except Exception as exc:
        res["cpu_baseline"] = {"error": repr(exc)}
    return res


def _torch_sage_step_baseline(feats, batches, c_in, dev, steps):
    """SURVEY.md 8(d) "SAGE baseline ... on the same GPU": the step in stock torch ops (index_select + index_add_ segment
    mean + F.linear, nn.BatchNorm1d, relu_, F.dropout, F.cross_entropy, torch.optim.Adam), the same pre-sampled batches,
    the same model shape (main.py:182-222).  What the reference's stack does minus PyG's fused SpMM."""
    import torch.nn.functional as Fn

    class Conv(torch.nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.lin_l, self.lin_r = torch.nn.Linear(ci, co), torch.nn.Linear(ci, co, bias=False)

        def forward(self, x_src, seg, col64, n_dst, inv_deg):
            agg = torch.zeros(n_dst, x_src.shape[1], dtype=x_src.dtype, device=x_src.device)
            agg.index_add_(0, seg, x_src.index_select(0, col64))
            return self.lin_l(agg * inv_deg) + self.lin_r(x_src[:n_dst])

    torch.manual_seed(0)
    convs = torch.nn.ModuleList([Conv(c_in, HIDDEN), Conv(HIDDEN, HIDDEN)]).to(dev)
    bn = torch.nn.BatchNorm1d(HIDDEN).to(dev)
    params = list(convs.parameters()) + list(bn.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    prepared = []
    for n_id, adjs, y in batches:                                   # index lists a stock pipeline would hold per batch
        blocks = []
        for a in adjs:
            deg = (a.rowptr[1:] - a.rowptr[:-1]).to(torch.int64)
            blocks.append((torch.repeat_interleave(torch.arange(a.n_dst, device=dev), deg), a.col.to(torch.int64), a.n_dst,
                           (1.0 / deg.clamp(min=1).to(torch.float32))[:, None]))
        prepared.append((n_id, blocks, y))

    def step(i):
        n_id, blocks, y = prepared[i % len(prepared)]
        x = feats.index_select(0, n_id)                             # convert_batch: x = data.x[n_id] (main.py:118-123)
        x = convs[0](x, *blocks[0])
        x = Fn.dropout(bn(x).relu_(), 0.5, True)
        x = convs[1](x, *blocks[1])
        loss = Fn.cross_entropy(x, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": BATCH / dt, "unit": "seed nodes/s", "ms_per_step": dt * 1e3, "kind": "port",
            "what": "the same step in stock torch ops on the same GPU: index_select + index_add_ segment mean + F.linear (hipBLASLt), "
                    "nn.BatchNorm1d, relu_, F.dropout, F.cross_entropy, torch.optim.Adam; same pre-sampled batches, x = data.x[n_id] gathered per step"}


def sage_leg(feats, ei_np, n, dev, steps, warmup):
    """SAGE nodes/s: fwd + bwd + Adam on pre-sampled Flickr-shaped batches over the features + POPE matrix."""
    from graphpope_amd.sage import SAGE, IndexedFeatures, cross_entropy, sample_batch
    from graphpope_amd.optim import Adam
    from graphpope_amd.sampler import DeviceBatch
    from graphpope_amd.train import SageTrainStep
    from oracle import oracle
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei_np[0], minlength=n))])
    col = ei_np[1]                                               # synthetic edge list is sorted by source
    rng = np.random.default_rng(0)
    batches = []
    for b in range(8):
        seeds = rng.choice(n, BATCH, replace=False)
        n_id, adjs = sample_batch(rowptr, col, seeds, sizes=(25, 10), rng=rng)
        batches.append((torch.as_tensor(n_id, device=dev), [a.to(dev) for a in adjs],
                        torch.randint(0, 7, (BATCH,), device=dev, generator=torch.Generator(device=dev).manual_seed(b))))
    c_in = feats.shape[1]
    torch.manual_seed(0)
    model = SAGE(c_in, 7, HIDDEN, 3).to(dev)                     # --num_layers 3: two convs execute, logits 256 wide
    opt = Adam(model.parameters(), lr=1e-3)   # torch.optim.Adam rule, one launch per step (the step is launch-bound)
    params = list(model.parameters())
    one = torch.ones((), device=dev)

    # (1) the step as the product runs it (graphpope_amd.train.SageTrainStep): captured once into a HIP graph and replayed;
    #     the pre-sampled batch is loaded into the graph's fixed buffers by one launch, its sizes stay in device words
    pool = []
    for n_id, adjs, y in batches:
        db = DeviceBatch(BATCH, (25, 10), dev)
        db.load(n_id, adjs)
        pool.append((db, y))
    trainer = SageTrainStep(model, opt, feats, BATCH, (25, 10), sampler=None, graph=True)

    def gstep(i):
        db, y = pool[i % len(pool)]
        trainer.load_batch(db, y)
        return trainer.run()

    for i in range(max(warmup, 4)):                              # two eager calls, the capture, one replay
        gstep(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        gstep(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps

    # (2) the same step enqueued eagerly through autograd on host-sized batches (rounds 1-2's figure)
    model.dropout_seed_dev = None
    opt.use_device_step(None)

    def step(i):
        n_id, adjs, y = batches[i % len(batches)]
        x = IndexedFeatures(feats, n_id)                         # main.py:118-123 convert_batch without the copy: layer 0 reads feats[n_id[j]]
        for p in params:                                         # opt.zero_grad(set_to_none=True) without its bookkeeping
            p.grad = None
        loss = cross_entropy(model(x, adjs), y, unit_upstream=True, loss_in=opt)   # main.py:216 F.cross_entropy: one launch, gradient pre-scaled; the scalar is finished in Adam's launch
        loss.backward(gradient=one)                          # seeded with 1 (the promise unit_upstream makes): no backward launch for the loss
        opt.step()
        return loss

    # The step is launch-bound on the host: run the backward pass in the calling thread (no hand-off to autograd's
    # device thread: -35 % host time per step, tools/host_overhead.py).  The whole leg runs under this setting.
    single_thread = torch.autograd.set_multithreading_enabled(False)
    single_thread.__enter__()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    eager_dt = (time.perf_counter() - t0) / steps
    shapes = [(a.n_dst, a.n_src, int(a.col.numel())) for a in batches[0][1]]
    try:
        torch_gpu = _torch_sage_step_baseline(feats, batches, c_in, dev, steps)
    except Exception as exc:                                     # never take the bench down
        torch_gpu = {"error": repr(exc)}

    # per-kernel view of layer 0 forward: gather (HBM/L2 bound) and projection (f32 MFMA bound)
    n_id, adjs, _ = batches[0]
    x = feats.index_select(0, n_id)
    conv = model.convs[0]
    ev = [_event() for _ in range(2)]
    with torch.no_grad():
        for _ in range(3):
            conv((x, x[:adjs[0].n_dst]), adjs[0])
        ev[0].record()
        for _ in range(10):
            conv((x, x[:adjs[0].n_dst]), adjs[0])
        ev[1].record()
    torch.cuda.synchronize()
    l0_ms = ev[0].elapsed_time(ev[1]) / 10
    # the same layer with the gather and the projection one after the other (POPE_KNOB_SAGE_FORWARD_OVERLAP = 0): the projection
    # kernel's own time (one launch, both products) is this minus the gather
    with torch.no_grad():
        _lib.load().pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 0)
        try:
            for _ in range(3):
                conv((x, x[:adjs[0].n_dst]), adjs[0])
            ev3 = [_event() for _ in range(2)]
            ev3[0].record()
            for _ in range(10):
                conv((x, x[:adjs[0].n_dst]), adjs[0])
            ev3[1].record()
            torch.cuda.synchronize()
            l0_seq_ms = ev3[0].elapsed_time(ev3[1]) / 10
        finally:
            _lib.load().pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
    n_dst, _, nnz = shapes[0]
    # the neighbour gather + mean on its own (memory-bound half of the layer)
    lib = _lib.load()
    a0 = adjs[0]
    agg = torch.empty((a0.n_dst, c_in), dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def gather():
        _lib.check(lib.sage_gather_mean(_lib.ptr(a0.rowptr), _lib.ptr(a0.col), a0.n_src, a0.n_dst, a0.col.numel(), _lib.ptr(x), c_in,
                                        _lib.ptr(agg), stream))
    for _ in range(3):
        gather()
    ev[0].record()
    for _ in range(20):
        gather()
    ev[1].record()
    torch.cuda.synchronize()
    g_ms = ev[0].elapsed_time(ev[1]) / 20
    g_bytes = 4.0 * (nnz * c_in + n_dst * c_in) + 4.0 * nnz + 4.0 * (n_dst + 1)
    l0_flops = 2.0 * 2 * n_dst * c_in * HIDDEN
    l0_bytes = 4.0 * (nnz * c_in + 2 * n_dst * c_in + n_dst * HIDDEN) + 4.0 * nnz + 4.0 * (n_dst + 1)
    # the vendor library on the same product (torch.addmm -> hipBLASLt), as a yardstick for the hand-written projection
    xcat = torch.cat([agg, x[:n_dst]], 1)
    wcat = torch.cat([conv.lin_l.weight, conv.lin_r.weight], 1).detach()
    with torch.no_grad():
        for _ in range(3):
            torch.addmm(conv.lin_l.bias, xcat, wcat.t())
        ev[0].record()
        for _ in range(10):
            torch.addmm(conv.lin_l.bias, xcat, wcat.t())
        ev[1].record()
    torch.cuda.synchronize()
    lib_ms = ev[0].elapsed_time(ev[1]) / 10
    del xcat, wcat

    # the same step with the batch sampled ON THE DEVICE each step (graphpope_amd.sampler: SURVEY 8f rank 1) instead of
    # taken from the pre-sampled pool: what an epoch actually costs when nothing is prepared on the host.  The sampler runs
    # inside the replayed graph (device extents: no size ever comes back to the host).
    sampled = {}
    try:
        from graphpope_amd.sampler import NeighborSampler
        csr = engine.build_csr(torch.as_tensor(ei_np, device=dev), n)
        sampler = NeighborSampler(csr.rowptr, csr.col, n, (25, 10))
        perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        labels = torch.randint(0, 7, (n,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        strainer = SageTrainStep(model, opt, feats, BATCH, sampler=sampler, graph=True)

        # epoch mode: the loader's part is on the device too -- the step reads its seeds from the epoch's shuffled order through
        # a device cursor and gathers their labels itself (main.py:100-123: NeighborSampler(..., shuffle=True) + y = data.y[n_id[:B]])
        strainer.set_epoch(perm, labels)

        def sstep(i):
            return strainer.step_epoch()

        for i in range(max(warmup, 4)):
            sstep(i)
        assert strainer.batches_left() >= steps, "the epoch is shorter than the timed loop"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            sstep(i)
        torch.cuda.synchronize()
        sdt = (time.perf_counter() - t0) / steps
        sampled = {"nodes_per_s": BATCH / sdt, "ms_per_step": sdt * 1e3,
                   "note": "fan-out [25, 10] sampled on the GPU inside the replayed step (device-extent sampler, no host synchronisation), "
                           "seeds taken from the epoch's shuffled order through a device cursor and their labels gathered by the sampler's first "
                           "kernel (SageTrainStep.set_epoch / step_epoch), features gathered from the HBM-resident matrix"}
        # the same step (epoch mode, device extents, no host synchronisation) with eager launches instead of a replayed graph: on a host
        # that keeps the queue full the ~23 launches pipeline better than the graph's nodes do
        etrainer = SageTrainStep(model, opt, feats, BATCH, sampler=sampler, graph=False)
        etrainer.set_epoch(perm, labels)
        for i in range(max(warmup, 4)):
            etrainer.step_epoch()
        assert etrainer.batches_left() >= steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            etrainer.step_epoch()
        torch.cuda.synchronize()
        sampled["eager_launches_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
        sampled["graph_replay_ms_per_step"] = sampled["ms_per_step"]
        if sampled["eager_launches_ms_per_step"] < sampled["ms_per_step"]:
            sampled["ms_per_step"] = sampled["eager_launches_ms_per_step"]
            sampled["nodes_per_s"] = BATCH / (sampled["ms_per_step"] * 1e-3)
        sampled["mode"] = "graph replay" if sampled["ms_per_step"] == sampled["graph_replay_ms_per_step"] else "eager launches"
        # and as rounds 1-2 ran it: eager launches, sizes read back by the host every hop
        model.dropout_seed_dev = None
        opt.use_device_step(None)

        def estep(i):
            lo = (i * BATCH) % (n - BATCH)
            seeds = perm[lo: lo + BATCH]
            n_id_s, adjs_s = sampler.sample(seeds, seed=i)
            for p in params:
                p.grad = None
            loss = cross_entropy(model(IndexedFeatures(feats, n_id_s), adjs_s), labels[seeds])
            loss.backward()
            opt.step()

        for i in range(warmup):
            estep(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            estep(i)
        torch.cuda.synchronize()
        sampled["eager_host_sized_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
    except Exception as exc:
        sampled = {"error": repr(exc)}

    # CPU baseline: the torch restatement, fwd + bwd of the same model shape on the host cores (1 warm-up, median of 3)
    try:
        xc = x.cpu()
        adj_cpu = [(a.rowptr.cpu(), a.col.cpu()) for a in adjs]
        w = [(c.lin_l.weight.detach().cpu().requires_grad_(True), c.lin_l.bias.detach().cpu().requires_grad_(True),
              c.lin_r.weight.detach().cpu().requires_grad_(True)) for c in model.convs[:2]]
        cts = []
        for _ in range(4):
            t0 = time.perf_counter()
            h = xc
            for li, (rp, cl) in enumerate(adj_cpu):
                h = oracle.sage_conv_torch(h, rp, cl, *w[li])
                if li == 0:
                    h = torch.relu(h)
            h.sum().backward()
            cts.append(time.perf_counter() - t0)
        cdt = float(np.median(cts[1:]))
        cpu = {"value": BATCH / cdt, "unit": "seed nodes/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"1 batch fwd + bwd, torch CPU restatement (oracle.sage_conv_torch), median of 3 passes after 1 warm-up, {cdt:.3f} s"}
    except Exception as exc:                                    # the CPU leg must never take the bench down
        cpu = {"error": repr(exc)}
    single_thread.__exit__(None, None, None)
    proj_ms = l0_seq_ms - g_ms
    return {
        "nodes_per_s": BATCH / eager_dt, "ms_per_step": eager_dt * 1e3, "steps": steps, "batch_seed_nodes": BATCH,
        "how": "the step (fwd + cross-entropy + bwd + Adam) enqueued through autograd on host-sized pre-sampled batches, as in rounds 1-2 "
               "(GPU-bound: the kernels of a step add up to its wall time)",
        "graph_replay": {"ms_per_step": dt * 1e3, "nodes_per_s": BATCH / dt,
                         "how": "graphpope_amd.train.SageTrainStep: the same step on device-extent batches (sizes in device words, tensors at "
                                "capacity), captured once into a HIP graph and replayed; each pre-sampled batch is loaded into the graph's fixed "
                                "buffers by one launch.  No host work per step, but the step is GPU-bound either way and the capacity-shaped "
                                "launches cost ~5 % more GPU time than the host-sized ones"},
        "torch_gpu_baseline": torch_gpu,
        "model": f"SAGE {c_in}->{HIDDEN}->{HIDDEN} (num_layers 3, 2 executed), fan-out [25, 10], fp32, fused BN+ReLU+dropout epilogue, one-launch Adam",
        "block_shapes_n_dst_n_src_nnz": shapes,
        "layer0_forward_ms": l0_ms,
        "layer0_forward_sequential_ms": l0_seq_ms,
        "layer0_forward_note": "layer0_forward_ms: the layer as the step runs it -- launch 1 = the neighbour gather in the blocks beside the "
                               "x_dst W_r^T half of the projection, launch 2 = the agg W_l^T half added in; _sequential_ms: gather, then "
                               "the whole projection in one launch (round 2's order, POPE_KNOB_SAGE_FORWARD_OVERLAP = 0)",
        "layer0_forward_tflops": l0_flops / (l0_ms * 1e-3) / 1e12,
        "layer0_forward_gbs": l0_bytes / (l0_ms * 1e-3) / 1e9,
        "layer0_gather_mean": {"ms": g_ms, "algorithmic_bytes": g_bytes, "achieved_gbs": g_bytes / (g_ms * 1e-3) / 1e9,
                               "frac_of_hbm_peak": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "k_gather_mean alone, back to back on the same batch: source rows (115 MB) are served "
                                       "by the 256 MB Infinity Cache, so the rate can exceed the HBM peak"},
        "layer0_projection": {"ms": proj_ms, "tflops": l0_flops / (proj_ms * 1e-3) / 1e12,
                              "roofline": {"bound": "mfma", "achieved": l0_flops / (proj_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF,
                                           "unit": "TFLOP/s", "frac": l0_flops / (proj_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF},
                              "vendor_library_ms": lib_ms,
                              "note": "sequential layer-0 forward minus the gather (the one-launch projection kernel), HIP events; vendor_library_ms = torch.addmm (hipBLASLt) on "
                                      "the concatenated operands of the same product, timed in the same run"},
        "with_gpu_sampling": sampled,
        "cpu_baseline": cpu,
    }


def pmc_traffic():
    """HBM bytes per launch of the dominant kernels from the committed rocprofv3 --pmc summary of the same command
    (separate profiling run, newest round first); None if no summary is committed."""
    for name in ("r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as fh:
                return json.load(fh), "profiles/" + name
        except (OSError, ValueError):
            continue
    return None, None


PROSE_KEYS = ("what", "note", "how", "method", "value_note", "bound_note", "layer0_forward_note", "traffic_source", "model", "scope", "all_calls_ms",
              "last_call_assembly_phases_ms")


def compact(obj, depth=0):
    """The JSON line without its explanatory strings (--verbose keeps them; DESIGN.md says what every figure is): the driver's
    record keeps only the tail of stdout, so the line has to stay well under 8 KB."""
    if isinstance(obj, dict):
        out = {k: compact(v, depth + 1) for k, v in obj.items() if k not in PROSE_KEYS}
        # the side legs (everything but the headline's own cpu_baseline and config): what was timed, in brief
        for key, cap, from_depth in (("sample", 72, 2), ("workload", 120, 1)):
            if depth >= from_depth and "baseline_config" not in out and isinstance(out.get(key), str) and len(out[key]) > cap:
                out[key] = out[key][:cap - 3] + "..."
        if depth == 0:
            for leg in ("cpu_baseline_all_cores", "cpu_baseline_bfs"):
                if isinstance(out.get(leg), dict) and isinstance(out[leg].get("sample"), str) and len(out[leg]["sample"]) > 72:
                    out[leg]["sample"] = out[leg]["sample"][:69] + "..."
        return out
    if isinstance(obj, list):
        return [compact(v, depth + 1) for v in obj]
    if isinstance(obj, float):
        return float("%.6g" % obj)
    if isinstance(obj, str) and len(obj) > 260:
        return obj[:257] + "..."
    return obj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sage", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the boundary / pairwise / config3 / config4 legs")
    ap.add_argument("--cpu-sample-nodes", type=int, default=2400, help="nodes of the Baseline A sample (x 256 anchors)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 3, 4),
                    help="BASELINE.json configs[i]: 1 = Flickr-shaped, 256 anchors PER GPU (weak scaling, the default and the only one "
                         "with the extra legs); 3 = Flickr-shaped, 1 024 anchors in all, sharded over the ranks (128 per GPU at 8; strong "
                         "scaling); 4 = R-MAT scale 22, 512 anchors in all (64 per GPU at 8), no features (strong scaling)")
    ap.add_argument("--verbose", action="store_true", help="keep the explanatory strings (what / note / how) in the JSON line")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus"

    cfg = args.config
    if cfg == 4:
        ei_np, n = synth.rmat(22, edge_factor=8, seed=1)
        k_total, feat = 512, 0
    else:
        ei_np, n = synth.flickr_like(seed=1)
        k_total, feat = (K_PER_GPU * world, F) if cfg == 1 else (1024, F)
    k_rank = -(-k_total // world)                                # anchors of one rank's shard (distributed.shard_size)
    anchors = synth.seeded_anchors(n, k_total, 42)
    e = ei_np.shape[1]

    # Baseline A forks a multiprocessing.Pool like the reference does: run it before this process touches the GPU
    base, want_hops = {}, None
    if world == 1 and cfg == 1 and not args.no_cpu_baseline:
        base, want_hops = cpu_baselines_before_gpu(ei_np, n, anchors, args.cpu_sample_nodes)

    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank %= max(torch.cuda.device_count(), 1)          # rehearsals put several ranks on one card
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("GRAPHPOPE_BENCH_BACKEND", "nccl")          # "gloo": rehearsal on a one-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = engine.require_gpu()

    x = torch.rand((n, feat), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    ei = torch.as_tensor(ei_np, device=dev)
    if cfg == 4:
        args.steps, args.warmup = min(args.steps, 10), min(args.warmup, 2)       # a step is tens of milliseconds and 8.6 GB of output

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = pope_step(x, ei, n, anchors, world)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pope_step(x, ei, n, anchors, world)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0          # this rank's own time, BEFORE the closing barrier: a straggler shows in per_rank
    barrier()
    elapsed = time.perf_counter() - t0
    collective_ranks = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, device=dev, dtype=torch.int32)          # how many ranks the collective backend really joined
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        collective_ranks = int(ones.item())

    per_rank = None
    if world > 1:
        # every rank's own achieved rate and what it put on the wire, gathered so that a SCALE line checks itself: the
        # speculative path all-gathers planes[0:5] (reachability + 4 hop-bit planes) of its K_PER_GPU anchors
        wpr = _lib.load().pope_words(k_rank)
        sent = 5 * n * wpr * 8
        mine = torch.tensor([elapsed_local, float(sent)], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        src_rank = k_rank * (4.0 * e + 4.0 * (n + 1) + 4.0 * n)
        per_rank = [{"rank": i, "ms_per_step": float(t[0]) / args.steps * 1e3, "achieved_gbs_per_source_model": src_rank / (float(t[0]) / args.steps) / 1e9,
                     "all_gather_bytes_sent_per_step": int(t[1]), "all_gather_bytes_received_per_step": int(t[1]) * (world - 1)} for i, t in enumerate(allr)]
    # the same step when the library's depth hint misses (what a process's FIRST call on a graph pays: 12 speculative level
    # launches instead of the 10 this graph needs): a call with another anchor count in between invalidates the hint
    hint_miss_ms = None
    if world == 1 and cfg == 1:
        tm = []
        for _ in range(5):
            engine.geodesic_run(x, ei, n, anchors[:192], reuse_workspace=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pope_step(x, ei, n, anchors, world)
            torch.cuda.synchronize()
            tm.append(time.perf_counter() - t1)
        hint_miss_ms = float(np.median(tm)) * 1e3
    result = None
    ms = elapsed / args.steps * 1e3
    if rank == 0:
        result = {
            "metric": "POPE embeddings/sec (nodes x anchors) + SAGE nodes/sec, Flickr 256 anchors",
            "value": n * k_total / (elapsed / args.steps), "unit": "embeddings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak" if cfg == 1 else "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": {1: "configs[1]: flickr-shaped geodesic-stochastic, N=%d E=%d F=500, 256 anchors PER GPU (np.random seed 42); "
                                       "inputs resident in HBM (edge_index, x) -> [N, F+K] f32 in HBM on every rank" % (n, e),
                                    3: "configs[3]: flickr-shaped geodesic, N=%d E=%d F=500, 1024 anchors IN ALL (np.random seed 42, 1020 distinct) "
                                       "sharded %d per rank; inputs resident in HBM -> [N, 1524] f32 in HBM on every rank" % (n, e, k_rank),
                                    4: "configs[4]: R-MAT scale 22 (a,b,c,d)=(.57,.19,.19,.05), N=%d E=%d CSR slots, 512 anchors IN ALL (np.random "
                                       "seed 42) sharded %d per rank, no features; edge_index resident in HBM -> [N, 512] f32 in HBM on every rank" % (n, e, k_rank)}[cfg],
                       "baseline_config": cfg, "anchors_total": k_total, "anchors_per_rank": k_rank,
                       "parallelism": f"anchor-shard x{world} + RCCL all-gather of hop planes" if world > 1 else "single GPU"},
        }
        result["value_note"] = ("steady state of repeated identical calls, inputs and output resident in HBM: the call sizes its run of "
                                "level launches from the depth the previous call found (a process's FIRST call enqueues 12 levels "
                                "instead of 10, about +9 us).  The reference's own case -- host tensors in, host tensor out, once per "
                                "process -- is the top-level key host_to_host (first_call_ms is that single call)")
        result["ms_per_step_depth_hint_miss"] = hint_miss_ms
        if world > 1:
            result["backend"] = backend
            result["collective_ranks"] = collective_ranks
            result["per_rank"] = per_rank
    if rank == 0:
        # dominant kernel by total time: k_bfs_level (one launch per level), timed on this rank's own anchor shard with HIP
        # events on the launch stream.  Algorithmic bytes of ONE launch: per CSR slot erow + col (8 B) + the
        # neighbour's frontier words (8W B); per node seen (read) + frontier (write) (16W B).  W = 4 words for 256 anchors.
        exp_ms, launches, active_levels, hp = level_kernel_times(ei, n, anchors[:k_rank], reps=10 if cfg != 4 else 3)
        wp = hp.planes.shape[2]
        dense_bytes = e * (8.0 + 8.0 * wp) + n * 16.0 * wp           # one level that does work
        exp_bytes = dense_bytes * active_levels / launches            # per launch, the early-exit launches included
        exp_gbs = exp_bytes / (exp_ms * 1e-3) / 1e9
        pmc, pmc_src = pmc_traffic()
        pmc = pmc or {}
        result["roofline"] = {
            "kernel": level_kernel_name(n, k_rank), "bound": "hbm", "achieved": exp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": exp_gbs / HBM_PEAK_GBS, "traffic": pmc.get("k_bfs_level_hbm_bytes_per_launch") if cfg == 1 else None,
            "traffic_source": (pmc_src + " (rocprofv3 --pmc passes of this command, collected in a separate run: not measured live)") if pmc_src else None,
            "algorithmic_bytes_per_launch": exp_bytes, "avg_launch_ms": exp_ms, "launches_per_step": launches,
            "active_levels": active_levels, "algorithmic_bytes_per_active_level": dense_bytes,
            "note": "per GPU; average over EVERY launch of the kernel in a step (what rocprofv3 --stats averages): "
                    f"{active_levels} levels do work, {launches - active_levels} exit at once (the run is sized by the depth of the previous call).  The working set (CSR 7.2 MB + planes of "
                    "2.9 MB) is L2 / Infinity-Cache resident: the dense levels run at the L2 line-fill rate of 32-byte gathers, not at the "
                    "HBM rate (DESIGN.md §5); the live-bit table skips quiet neighbours, so sparse levels move fewer bytes than the model"}
        src_bytes = k_total * (4.0 * e + 4.0 * (n + 1) + 4.0 * n)           # SURVEY.md §8d per-source model, all GPUs
        geo_gbs = src_bytes / (ms * 1e-3) / 1e9
        result["roofline_per_source_model"] = {
            "scope": "whole step, all GPUs (CSR + BFS + exchange + finalise)", "bound": "hbm", "achieved": geo_gbs,
            "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": geo_gbs / (HBM_PEAK_GBS * world), "algorithmic_bytes": src_bytes,
            "note": "SURVEY 8d per-source byte model (every anchor reads the CSR once and writes one f32 column); the "
                    "bit-parallel BFS shares each CSR pass between 64 anchors per word, so this is a model, not traffic"}
        floor_bytes = 16.0 * e + 4.0 * n * feat + 4.0 * n * (feat + k_total)    # edge list in, x in, [N, F+K] out: each once (per rank)
        result["compulsory_floor"] = {"bytes": floor_bytes, "ms_at_6.29TBs": floor_bytes / 6.29e12 * 1e3,
                                      "step_over_floor": ms / (floor_bytes / 6.29e12 * 1e3) if world == 1 else None}
        result["level_kernel_ms"] = {"avg": exp_ms, "active_levels": active_levels}
    if world == 1 and cfg == 1:
        # per-phase device time with HIP events on the launch stream, separate from the wall-clock loop above
        timers = {"csr": [], "bfs": [], "finalize": []}
        for _ in range(max(10, min(args.steps, 50))):
            pope_phases(x, ei, n, anchors, timers)
        med = {p: float(np.median(timers[p])) for p in ("csr", "bfs", "finalize")}
        result["phases_ms"] = med
        result["max_hop"] = timers["max_hop"]
        fin_bytes = 4.0 * n * F + 4.0 * n * (F + K_PER_GPU) + 8.0 * n * wp * (1 + timers["n_hop_bits"])
        # the kernel's own duration: launches queued back to back between two events (the phase figure above starts at an
        # idle device behind the BFS verdict's host synchronisation, so it carries a launch latency the hot path does not pay)
        fin_ms = finalize_kernel_ms(x, ei, n, anchors)
        fin_gbs = fin_bytes / (fin_ms * 1e-3) / 1e9
        result["roofline_finalize"] = {
            "kernel": finalize_kernel_name(n, K_PER_GPU, F, True, 1), "bound": "hbm", "achieved": fin_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": fin_gbs / HBM_PEAK_GBS, "traffic": pmc.get("k_finalize_hbm_bytes_per_launch"),
            "algorithmic_bytes_per_launch": fin_bytes, "avg_launch_ms": fin_ms, "phase_ms_from_idle_device": med["finalize"],
            "achieved_from_idle_device": fin_bytes / (med["finalize"] * 1e-3) / 1e9,
            "method": "achieved / frac use avg_launch_ms = 20 launches queued back to back between two HIP events (since round 4; the kernel's own "
                      "duration, what rocprofv3 --stats averages); phase_ms_from_idle_device = the one launch of the phase loop, which starts at an "
                      "idle device behind a host synchronisation (rounds 1-3 quoted that figure)"}
        boundary = None
        if not args.no_extra:
            # the host -> host call right behind the timed steps.  It is host-side work and the box's host side is noisy: the same
            # code in the same box gave 4.9 / 1.35 ms (first / repeated call) in one run of this file and 7.9 / 2.9 ms in the next
            # (profiles/r04_bench_order.txt); its position among the legs is not what moves it.
            x_cpu = x.cpu()
            boundary = boundary_leg(ei_np, n, x_cpu, None)
        if not args.no_sage:
            result["sage"] = sage_leg(out, ei_np, n, dev, steps=max(10, min(args.steps, 50)), warmup=3)
        if not args.no_cpu_baseline:
            result.update(base)
            got = engine.hop_matrix(engine.geodesic_run(None, ei, n, anchors, want_out=False)[1]).cpu().numpy()
            result["hops_bit_exact_vs_cpu"] = bool(np.array_equal(got, want_hops))
        del out
        if not args.no_extra:
            steps = max(5, min(args.steps, 20))
            if want_hops is None:
                from oracle import oracle
                want_hops = oracle.geodesic_hops(ei_np, n, anchors)
            boundary["bit_exact_vs_cpu"] = boundary_check(boundary.pop("_out"), x_cpu, want_hops)
            result["boundary_host_to_host"] = boundary
            b = result["boundary_host_to_host"]
            result["host_to_host"] = {"what": "SURVEY 8(d) primary metric: utils.Graphpope() from CPU tensors to the returned (pageable) CPU tensor",
                                      "first_call_ms": b["first_call_ms"], "ms": b["ms"], "embeddings_per_s": b["embeddings_per_s"],
                                      "embeddings_per_s_first_call": n * K_PER_GPU / (b["first_call_ms"] * 1e-3),
                                      "pcie_floor_ms": b["pcie"]["floor_ms"], "bit_exact_vs_cpu": b["bit_exact_vs_cpu"]}
            if "cpu_baseline" in result and "value" in result["cpu_baseline"]:
                # like for like: the host -> host call against the reference's host -> host CPU path (Pool(6))
                result["speedup_vs_cpu_baseline"] = {
                    "host_to_host_vs_pool6": result["boundary_host_to_host"]["embeddings_per_s"] / result["cpu_baseline"]["value"],
                    "note": "boundary_host_to_host.embeddings_per_s / cpu_baseline.value: both from host tensors to a host tensor; "
                            "the HBM-resident `value` is not compared with a CPU figure"}
            result["pagerank"] = pagerank_leg(ei_np, n, ei)
            result["pairwise"] = pairwise_leg(n, anchors, x, dev, steps)
            result["kmeans"] = kmeans_leg(n, dev)
            result["config3"] = config3_leg(x, ei, n, steps)
            del x, ei
            engine._WORKSPACE.clear()
            torch.cuda.empty_cache()
            result["config4"] = config4_leg(dev)
    if rank == 0:
        # the numbers that matter as top-level scalars, so that a record which only keeps scalars still carries them
        sage = result.get("sage") or {}
        if "ms_per_step" in sage:
            result["sage_ms_per_step"] = sage["ms_per_step"]
            result["sage_nodes_per_s"] = sage["nodes_per_s"]
            result["sage_graph_replay_ms_per_step"] = sage.get("graph_replay", {}).get("ms_per_step")
            result["sage_sampled_ms_per_step"] = sage.get("with_gpu_sampling", {}).get("ms_per_step")
            result["sage_torch_gpu_baseline_ms_per_step"] = sage.get("torch_gpu_baseline", {}).get("ms_per_step")
            result["sage_layer0_projection_ms"] = sage.get("layer0_projection", {}).get("ms")
            result["sage_layer0_projection_vendor_ms"] = sage.get("layer0_projection", {}).get("vendor_library_ms")
        if "host_to_host" in result:
            result["host_to_host_first_call_ms"] = result["host_to_host"]["first_call_ms"]
            result["host_to_host_repeat_call_ms"] = result["host_to_host"]["ms"]
        if "pairwise" in result:
            result["pairwise_ms_per_call"] = result["pairwise"].get("ms_per_call")
            result["pairwise_embedding_only_ms"] = result["pairwise"].get("embedding_only_ms")
        for key in ("config3", "config4"):
            if key in result:
                result[key + "_one_gpu_ms_per_step"] = result[key].get("ms_per_step")
        line = json.dumps(result if args.verbose else compact(result))
        print(line)
    if world > 1:
        barrier()                              # rank 0 measured its level kernel alone meanwhile: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
