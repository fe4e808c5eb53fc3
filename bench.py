#!/usr/bin/env python3
"""bench.py -- walk-generation throughput of the MI355X engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config headline|C2|C3|C4|C5] [--scale S]

One "step" = one full pass of the hot path (MT19937 stream expansion + all walk kernels + for N > 1 the one
gather of the walk shards on rank 0) over the whole job array of the workload.  Default workload = the
configuration the BASELINE metric is quoted on: RMAT-22, SparseOTF p = 0.5 q = 2, 10 walks x 80 steps per
vertex, seed 0.  --config selects the other BASELINE.json configurations:
    C2  RMAT-18 SparseOTF p=0.5 q=2          C3  RMAT-22 SparseOTF p=0.25 q=4
    C4  ER N=100k density 0.25 DenseOTF p=0.5 q=2 (generated on the device as packed adjacency bits)
    C5  weighted RMAT-20, node2vec+ (--extend, gamma 0), SparseOTF p=0.5 q=2
Graph, shuffled start array and output buffers are resident in HBM before the timed region.  N > 1: one process
per GPU; when WORLD_SIZE is not set the script launches the N ranks itself (torch.distributed.run, 127.0.0.1).
Graph replicated, job array sharded, strong scaling (total work fixed); the walk shards are gathered once on rank 0
over RCCL/xGMI inside the timed region (--no-gather leaves every shard on the GPU that produced it).  Rank 0 prints
ONE JSON line.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
RANDOM_SECTOR_GBS = 3100.0     # measured: scattered 64-byte sectors, profiles/r02_fetch_calibration.txt

CONFIGS = {
    "headline": dict(graph="rmat", scale=22, p=0.5, q=2.0, mode="SparseOTF", weighted=False, extend=False),
    "C2": dict(graph="rmat", scale=18, p=0.5, q=2.0, mode="SparseOTF", weighted=False, extend=False),
    "C3": dict(graph="rmat", scale=22, p=0.25, q=4.0, mode="SparseOTF", weighted=False, extend=False),
    "C4": dict(graph="er", n=100000, density=0.25, p=0.5, q=2.0, mode="DenseOTF", weighted=False, extend=False),
    "C5": dict(graph="rmat", scale=20, p=0.5, q=2.0, mode="SparseOTF", weighted=True, extend=True),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="headline")
    ap.add_argument("--scale", type=int, default=None, help="RMAT scale (overrides the config's)")
    ap.add_argument("--er-nodes", type=int, default=None, help="ER vertex count (overrides C4's 100000)")
    ap.add_argument("--p", type=float, default=None)
    ap.add_argument("--q", type=float, default=None)
    ap.add_argument("--num-walks", type=int, default=10)
    ap.add_argument("--walk-length", type=int, default=80)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weighted", action="store_true", help="hashed U(0,1] edge weights (RMAT configs)")
    ap.add_argument("--extend", action="store_true", help="node2vec+ (weighted graphs)")
    ap.add_argument("--self-loops", type=int, default=0, help="add this many random self loops to the RMAT graph (the reference "
                    "accepts them, graph.py:238-268; round 6: they keep the lane kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-call", action="store_true", help="skip the host-pointer call (config.host_call)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--gather-chunks", default="auto",
                    help="N > 1: chunks per shard whose gathers overlap the next chunk's walk kernel; 'auto' = 2 / 3 / 4 for "
                         "shards below 8 M / below 16 M / from 16 M jobs (a call costs ~10 ms + 3 ms per million jobs: "
                         "DESIGN.md section 6, profiles/r04_shard_chunks_two_engines.txt)")
    ap.add_argument("--rank0-share", default="auto",
                    help="rank 0's shard as a fraction of a uniform one (it also assembles the matrix); 'auto' = the model of "
                         "DESIGN.md section 6 (0.5 at 8 GPUs), 1 = uniform")
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: leave the walk shards on their GPUs (skip the gather on rank 0)")
    return ap.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def reference_format_bytes(walks, deg, L, common=None):
    """bytes the REFERENCE's data layout moves for these walks -- per sampled step
    8*d_cur + 4*d_prev + 28 (first step of a walk: 8*d_cur + 20).  Reported for comparison only: the lane kernel
    does not stream rows, so this figure is not what its roofline is computed from.
    node2vec+ (``common`` = (sorted edge keys row * n + col, per-entry common-neighbour counts, n)): every step with a prev
    adds 4*d_prev (prev's weights) + 4*|N(cur) & N(prev)| (threshold gathers) + 4, section 8(d)'s n2v+ formula."""
    import torch

    total = 0
    n = walks.shape[0]
    chunk = 1 << 20
    for lo in range(0, n, chunk):
        w = walks[lo:lo + chunk].long() & 0xFFFFFFFF
        ln = w[:, L + 1]
        steps = (ln - 1).clamp(min=0)
        idx = torch.arange(L, device=w.device).unsqueeze(0)
        valid = idx < steps.unsqueeze(1)
        d_cur = deg[w[:, :L]] * valid
        total += int((8 * d_cur).sum().item())
        total += int((28 * valid).sum().item())
        total -= int(8 * valid[:, 0].sum().item())
        d_prev = deg[w[:, : L - 1]] * valid[:, 1:]
        total += int((4 * d_prev).sum().item())
        if common is not None:
            keys, n_in, n_nodes = common
            total += int((4 * d_prev).sum().item()) + int((4 * valid[:, 1:]).sum().item())
            # the entry (prev -> cur) of every step with a prev: its common-neighbour count (0 for the rare non-edge arrival)
            qk = (w[:, : L - 1] * n_nodes + w[:, 1:L])[valid[:, 1:]]
            pos = torch.searchsorted(keys, qk).clamp(max=keys.numel() - 1)
            hit = keys[pos] == qk
            total += int((4 * n_in[pos] * hit).sum().item())
    return total


def er_bits_gpu(n, density, dev, seed=1):
    """Packed adjacency (int64 words, bit x of row u <=> edge u-x) of an undirected ER graph, built on the GPU."""
    import torch

    gen = torch.Generator(device=dev).manual_seed(seed)
    wpr = (n + 63) // 64
    adj = torch.zeros((n, wpr * 64), dtype=torch.bool, device=dev)
    rows_per = max(1, (1 << 28) // n)
    cols = torch.arange(n, device=dev)
    for lo in range(0, n, rows_per):
        hi = min(n, lo + rows_per)
        u = torch.rand((hi - lo, n), generator=gen, device=dev) < density
        u &= cols.unsqueeze(0) > torch.arange(lo, hi, device=dev).unsqueeze(1)   # strict upper triangle
        adj[lo:hi, :n] = u
    adj[:, :n] |= adj[:, :n].t().clone()
    w32 = (adj.view(n, wpr * 2, 32).to(torch.int64) * (1 << torch.arange(32, device=dev, dtype=torch.int64))).sum(-1)
    bits = w32[:, 0::2] | (w32[:, 1::2] << 32)
    deg = adj[:, :n].sum(1)
    return bits.contiguous(), deg


def load_pmc(key):
    """HBM-side traffic and issue counters of this exact workload from the committed rocprofv3 --pmc passes
    (profiles/r06_traffic.json, written by tools/pmc_run.sh + tools/pmc_to_json.py); PMC cannot be collected
    inside a timed run."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json"):   # (a workload not re-profiled this round keeps its last entry)
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                w = json.load(f)["workloads"].get(key)
            if w:
                w = dict(w)
                w["source"] = "profiles/" + name
                return w
        except (OSError, KeyError, ValueError):
            pass
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)   # does not return
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PECANPY_BENCH_BACKEND=gloo + PECANPY_BENCH_ONE_GPU=1: dry-run of the multi-rank path on a box
    # with a single GPU (RCCL refuses two ranks on one device); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("PECANPY_BENCH_BACKEND", "nccl")
    if os.environ.get("PECANPY_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # the library's one-time start-up (pw_warmup: ~140 ms for the first stream the library creates in a process) on a helper
    # thread, beside the graph generation below -- what Base.__init__ does in front of read_edg (pecanpy_amd/pecanpy.py);
    # reported as config.library_warmup_ms and charged by value_first_call_incl_warmup
    if not os.environ.get("PECANPY_AMD_NO_WARMUP"):
        from pecanpy_amd import _lib as _pw_lib

        _pw_lib.warmup_async(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")   # where collectives exchange tensors
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from pecanpy_amd.engine import WalkEngine, auto_rank0_share, shard_bounds, tapered_bounds
    from pecanpy_amd.synth import rmat_csr

    cfg = dict(CONFIGS[args.config])
    if args.scale is not None:
        cfg["scale"] = args.scale
    if args.er_nodes is not None:
        cfg["n"] = args.er_nodes
    if args.p is not None:
        cfg["p"] = args.p
    if args.q is not None:
        cfg["q"] = args.q
    if args.weighted:
        cfg["weighted"] = True
    if args.extend:
        cfg["extend"] = cfg["weighted"] = True
    p, q, mode, extend = cfg["p"], cfg["q"], cfg["mode"], cfg["extend"]
    L, W = args.walk_length, args.num_walks

    t0 = time.time()
    indptr = indices = data = None
    if cfg["graph"] == "rmat":
        indptr, indices, data = rmat_csr(cfg["scale"], seed=1, weighted=cfg["weighted"])
        n_nodes = indptr.size - 1
        if args.self_loops and not cfg["weighted"]:
            from pecanpy_amd.synth import csr_from_edges

            lp = np.random.default_rng(7).choice(n_nodes, args.self_loops, replace=False).astype(np.int64)
            rows = np.repeat(np.arange(n_nodes, dtype=np.int64), np.diff(indptr.astype(np.int64)))
            indptr, indices, data = csr_from_edges(np.concatenate([rows, lp]), np.concatenate([indices.astype(np.int64), lp]), n_nodes)
            del rows
        t_graph = time.time() - t0
        # what is left of the HIP runtime's start-up at this point is timed APART from the handle creation (torch.cuda.set_device
        # above has usually started the runtime: ~0; the ~150 ms "runtime / streams / events" stage INSIDE pw_csr_create is the first
        # use of this library's code object and stays in graph_create_wall_ms); value_first_call charges both
        t_rt = time.perf_counter()
        torch.cuda.init()
        torch.empty(1, device=dev)
        torch.cuda.synchronize()
        hip_startup_ms = (time.perf_counter() - t_rt) * 1e3
        t_create = time.perf_counter()
        eng = WalkEngine.from_csr(indptr, indices, data, device=local_rank)
        create_wall_ms = (time.perf_counter() - t_create) * 1e3   # the whole of pw_csr_create as the caller sees it
        if extend:
            from pecanpy_amd import pecanpy as node2vec

            g = node2vec.SparseOTF.from_csr(indptr, indices, data, extend=True, gamma=0)
            with np.errstate(all="ignore"):
                eng.set_thresholds(np.nan_to_num(g.get_noise_thresholds(), nan=0.0))
        has_nbr = indptr[1:] != indptr[:-1]
        nnz = int(indices.size)
        gdesc = (f"RMAT-{cfg['scale']} (Graph500 a,b,c=.57,.19,.19, edge factor 8, symmetrised"
                 f"{', hashed U(0,1] weights' if cfg['weighted'] else ', unweighted'}"
                 f"{', + %d random self loops' % args.self_loops if args.self_loops else ''})")
        key_graph = f"rmat{cfg['scale']}{'w' if cfg['weighted'] else ''}{'_loops%d' % args.self_loops if args.self_loops else ''}"
    else:
        n_nodes = cfg["n"]
        bits, deg_t = er_bits_gpu(n_nodes, cfg["density"], dev)
        torch.cuda.synchronize()
        t_graph = time.time() - t0
        hip_startup_ms = 0.0            # (the graph was generated on the device: the runtime is up)
        t_create = time.perf_counter()
        eng = WalkEngine.from_dense_bits(bits, n_nodes, device=local_rank)
        create_wall_ms = (time.perf_counter() - t_create) * 1e3
        del bits
        has_nbr = (deg_t > 0).cpu().numpy()
        nnz = int(deg_t.sum().item())
        gdesc = f"Erdos-Renyi N={n_nodes} density {cfg['density']} (undirected, unweighted, packed adjacency bits)"
        key_graph = f"er{n_nodes}"
    info = eng.index_info()
    nodes = np.arange(n_nodes, dtype=np.uint32)
    starts = np.concatenate([nodes] * W)
    np.random.RandomState(args.seed).shuffle(starts)   # legacy seeded shuffle, as the reference
    n_jobs = starts.size
    t_prep = time.time() - t0

    do_gather = world > 1 and not args.no_gather
    rank0_share = auto_rank0_share(world, do_gather) if args.rank0_share == "auto" else float(args.rank0_share)
    all_bounds = shard_bounds(n_jobs, world, rank0_share)
    lo, hi = all_bounds[rank]
    d_starts = torch.from_numpy(starts[lo:hi].view(np.int32)).to(dev)
    # stream address of this shard (undirected graph: nominal counts are exact)
    skip = int(has_nbr[starts[:lo]].sum()) * L
    # N > 1 with gather: the shard is walked in a few chunks and the gather of chunk c (async, on RCCL's
    # stream) overlaps the walk kernel of chunk c + 1; --gather-chunks 1 = one blocking gather at the end
    if args.gather_chunks == "auto":   # (the same on every rank: from the LARGEST shard)
        largest = max(b - a for a, b in all_bounds)
        # (round 5: a chunk of a shard costs ~2.9 ms + 3.55 ms per million jobs in the CHAINS form -- three chunks even for small shards)
        auto_chunks = 3 if largest < 16_000_000 else 4
    n_chunks = (auto_chunks if args.gather_chunks == "auto" else max(1, int(args.gather_chunks))) if do_gather else 1
    chunk_bounds = tapered_bounds(hi - lo, n_chunks)   # decreasing sizes: the exposed tail is the smallest chunk's transfer
    csum = np.concatenate([[0], np.cumsum(has_nbr[starts[lo:hi]], dtype=np.int64)])
    chunk_skip = [skip + int(csum[a]) * L for a, _ in chunk_bounds]
    # The gather (N > 1): ONE preallocated [n_jobs, L + 2] matrix on rank 0; every chunk of every other rank's shard is
    # sent point to point (RCCL send/recv) straight into its row slice -- no padded staging buffers, no concatenation --
    # rank 0 walks its own shard in place, and the rows of starts without neighbours ([start, 0, ..., 0, 1]: 52 % of an
    # R-MAT job array) are not sent at all: rank 0 writes them itself.
    gather = None
    job_has_nbr = has_nbr[starts]
    if do_gather:
        from pecanpy_amd.sharding import RowGather, isolated_row_filler

        fill = isolated_row_filler(starts, L, cdev)
        gather_mode = "rccl"    # (round 6: the peer-write assembly lives below the Python layer now -- pw_simulate_multi, one process)
        gather = RowGather(n_jobs, L + 2, all_bounds, torch.int32, cdev, dst=0, known=~job_has_nbr, fill_known=fill)
        # (the matrix is allocated on rank 0 here: outside the timed region, like d_out)
    if do_gather and rank == 0 and cdev == dev:
        d_out = gather.own_rows()
    else:
        d_out = torch.empty((hi - lo, L + 2), dtype=torch.int32, device=dev)
    chunk_of = [tapered_bounds(b[1] - b[0], n_chunks) for b in all_bounds]

    acc = {k: [] for k in ("walk_kernel_ms", "lane_kernel_ms", "rng_kernel_ms", "total_steps", "list_entries_read",
                           "ambiguous_steps", "wave_chain_steps", "redo_walks", "overflow_reads", "verify_checked", "verify_mismatch")}
    param_index_ms = [0.0]   # (p, q)-dependent index built by the first call (normaliser table of weighted graphs)

    pass_no = [0]

    def one_pass():
        # every pass walks with a NEW seed (args.seed + pass number): a repeated seed would let the engine reuse the
        # MT19937 generator states of the previous pass (its cache is keyed by seed) and skip the jump-ahead launches
        # -- work a user's call with a fresh seed has to do.  The walks of different seeds sample the same number of
        # transitions on an undirected graph (every non-isolated start runs L steps).
        seed = args.seed + pass_no[0]
        pass_no[0] += 1
        tot = {k: 0 for k in acc}
        if do_gather and rank == 0:
            gather.prefill()   # the rows rank 0 writes itself (isolated starts of the other shards): part of every pass
        if len(chunk_bounds) > 1 and mode in ("SparseOTF", "DenseOTF"):
            # round 6: the stream of the rank's WHOLE shard is expanded once (one MT19937 jump-ahead tree, ~3 ms of sequential
            # launches plus one per set bit of the shard's first block) and the chunks find their draws in place
            eng.stream_hold(seed, skip, int(csum[-1]) * L)
        for c, (a, b) in enumerate(chunk_bounds):
            eng.simulate_device(mode, p, q, extend, d_starts[a:b], L, seed=seed,
                                stream_skip=chunk_skip[c], out=d_out[a:b])
            for k in tot:
                tot[k] += eng.last_stats[k]
            param_index_ms[0] += eng.last_stats["param_index_ms"]
            if do_gather:  # this chunk travels over RCCL/xGMI while the next chunk is walked
                if rank == 0:
                    if cdev != dev:
                        gather.full[lo + a: lo + b] = d_out[a:b].to(cdev)
                    gather.expect([(all_bounds[r][0] + chunk_of[r][c][0], all_bounds[r][0] + chunk_of[r][c][1], r)
                                   for r in range(1, world)])
                else:
                    gather.post(lo + a, lo + b, d_out[a:b])
        if len(chunk_bounds) > 1 and mode in ("SparseOTF", "DenseOTF"):
            eng.stream_release()
        if do_gather:
            gather.finish()                     # rank 0: the contiguous [n_jobs, L + 2] matrix is complete
        for k in tot:
            acc[k].append(tot[k])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    for k in acc:
        acc[k].clear()
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    fence()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = eng.last_stats
    shard_steps = torch.tensor([acc["total_steps"][-1]], dtype=torch.int64, device=cdev)
    per_rank_ms = torch.tensor([float(np.mean(acc["walk_kernel_ms"]))], dtype=torch.float64, device=cdev)
    rank_ms = [float(per_rank_ms.item())]
    if world > 1:
        dist.all_reduce(shard_steps)
        outs = [torch.zeros_like(per_rank_ms) for _ in range(world)]
        dist.all_gather(outs, per_rank_ms)
        rank_ms = [float(o.item()) for o in outs]
    total_steps = int(shard_steps.item())           # sampled transitions of the whole job array
    sec_per_step = elapsed / max(args.steps, 1)
    value = total_steps / sec_per_step / 1e6

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    if os.environ.get("PECANPY_BENCH_VERIFY") and do_gather:
        # dry-run check of the sharded + chunked addressing: the assembled matrix equals one whole-array run
        gathered = gather.full.to(dev)
        whole = eng.simulate_device(mode, p, q, extend, torch.from_numpy(starts.view(np.int32)).to(dev), L,
                                    seed=args.seed + pass_no[0] - 1)
        assert torch.equal(gathered, whole), "gathered shards differ from the single-stream matrix"
        print("bench.py: gathered shards verified against a whole-array run", file=sys.stderr)

    # ---- roofline of the dominant kernel (rank 0's shard) ------------------------------------------------------
    lane = int(st["lane_kernel"]) in (1, 2)          # (3 = the weighted lane form: its own accounting below)
    wlane = int(st["lane_kernel"]) == 3
    k_ms = float(np.mean(acc["lane_kernel_ms"] if (lane or wlane) else acc["walk_kernel_ms"]))
    steps0 = int(acc["total_steps"][-1])
    walks0 = int(has_nbr[starts[lo:hi]].sum())
    ref_bytes = None
    if cfg["graph"] == "rmat":
        deg_t = torch.from_numpy(np.diff(indptr.astype(np.int64))).to(dev)
        common = None
        if extend:   # section 8(d)'s node2vec+ formula needs |N(cur) & N(prev)| of every step: the lane index's per-entry counts
            try:
                cnt, _, _, _ = eng.lane_index()
                rows_np = np.repeat(np.arange(n_nodes, dtype=np.int64), np.diff(indptr.astype(np.int64)))
                common = (torch.from_numpy(rows_np * n_nodes + indices.astype(np.int64)).to(dev),
                          torch.from_numpy(cnt.astype(np.int64)).to(dev), n_nodes)
            except Exception:   # noqa: BLE001 (no lane index: the counts are left out, the figure says so)
                common = None
        ref_bytes = reference_format_bytes(d_out, deg_t, L, common)
    if lane:
        # declared format of the lane kernel: per sampled step one 64-byte edge line (the record
        # of the edge the walk arrives by and -- for lists of up to 20 entries -- the list itself), one 8-byte draw, one
        # 4-byte output cell, 2 bytes per common-neighbour list entry read (counted in the kernel; uint16 positions;
        # entries read inside the edge line are counted twice, rows beyond 65536 entries use 4 bytes); per walk: start
        # 4 + stream offset 8 + vertex record 16 + header/length cells 8
        entries = int(acc["list_entries_read"][-1])
        # ... and 128 bytes per step that needs the float32 chain (the walk's queue record, written and read back)
        chain_steps = int(acc["wave_chain_steps"][-1])
        floats_form = int(st["lane_kernel"]) == 2
        declared = (steps0 * (64 + 8 + 4 + (4 if floats_form else 0)) + entries * 2 + (0 if floats_form else chain_steps * 128) +
                    (hi - lo) * (4 + 8) + walks0 * (16 + 8) + (hi - lo - walks0) * 8)
        kernel = ("walk_lanes_kernel (QUAD form: every round of a pass) + lanes_chain_kernel" if int(st["lane_kernel"]) == 1 else
                  "walk_lanes_kernel<FLOATS> (1/p or 1/q not a power of two: float64-bounded decision from per-line row totals, "
                  "float32 chains for the steps it leaves open, per lane)")
        fmt = ("per step ONE 64 B edge line (record + inline list or pivots; fetched whole by a quad of lanes into LDS) + 8 B draw "
               "+ 4 B output; 64 B per overflow-list sector a search visits (in-kernel counter, 32 two-byte entries' worth each) "
               "and 2 B per list entry the float chains probe; 128 B per parked step (queue record out and back); 36 B per walk"
               if int(st["lane_kernel"]) == 1 else
               "64 B edge line + 8 B draw + 4 B output + 4 B row total per step, 2 B per common-neighbour list entry probed "
               "(in-kernel counter), 36 B per walk")
    elif cfg["graph"] == "er":
        wpr = (n_nodes + 63) // 64
        if wpr <= 2048:
            # rows of up to 131 072 columns: the row of cur stays in registers and serves as prev's row one step later
            declared = steps0 * (wpr * 8 + 12)
            fmt = "packed adjacency: ONE row per step (the row of cur is kept in registers for the next step) + draw + output"
        else:
            declared = steps0 * (3 * wpr * 8 + 12)
            fmt = "packed adjacency: rows of cur and prev (count pass) + cur's row again (search segment) + draw + output"
        # which kernel ran (launch_dense_bits' own conditions): the register-only kernel needs dyadic 1/p, 1/q, rows of at most
        # 131 072 columns and no PECANPY_AMD_DENSE_NO_FAST
        def _pow2(x):
            m, _ = np.frexp(np.float32(1.0 / x))
            return float(m) == 0.5
        fast = wpr <= 2048 and _pow2(p) and _pow2(q) and not os.environ.get("PECANPY_AMD_DENSE_NO_FAST")
        kernel = "walk_dense_fast_kernel" if fast else "walk_dense_bits_kernel"
        if not fast:
            declared = steps0 * (3 * wpr * 8 + 12)
            fmt = "packed adjacency: rows of cur and prev (count pass) + cur's row again (search segment) + draw + output"
    else:
        # the wave-per-walk kernel streams rows (keys of the shorter row, weights of cur's row): SURVEY 8(d)'s
        # figure in the reference's element sizes is its declared format
        declared = ref_bytes
        kernel = "walk_kernel<float,false,%s,%s>" % ("true" if not cfg["weighted"] else "false", "true" if extend else "false")
        if wlane:
            kernel = ("walk_lanes_kernel<WEIGHTED> (float64-bounded decision per lane) + lanes_eager_weighted_kernel (wave scan of the "
                      "steps the bound leaves open, from recorded chain values), every round of a pass")
            # declared format of the weighted lane form: per step the 64-byte edge line, the draw and the
            # output cell; per search probe a list entry and two float64 table values (18 B: an upper bound, probes inside a run
            # read one value); per parked step its 64-byte queue record out and back and the scan window of the wave scan
            # (at most 2 x 1024 elements from the last recorded chain value: 4-byte weights + 4-byte neighbour ids); 36 B per walk
            probes = int(acc["list_entries_read"][-1])
            eager = int(st["eager_steps"])
            declared = (steps0 * (64 + 8 + 4) + probes * 18 + eager * (128 + 2048 * 8) + (hi - lo) * (4 + 8) + walks0 * (16 + 8) +
                        (hi - lo - walks0) * 8)
            fmt = ("64 B edge line + 8 B draw + 4 B output per step, 18 B per search probe (list entry + two float64 prefix values), "
                   "per parked step 128 B of queue record + a scan window of at most 2048 elements x 8 B; 36 B per walk")
        fmt_ref = ("SURVEY 8(d): 8*d_cur + 4*d_prev + 28 per step" +
               (" + 4*d_prev + 4*|N(cur)&N(prev)| + 4 (node2vec+)" if extend else "") +
               " -- the REFERENCE's row traffic, which this kernel's rows mostly take from L2 / Infinity Cache: `frac` here is an "
               "algorithmic-byte rate, NOT an HBM utilisation (that is traffic_frac_of_peak, from the PMC passes); the kernel is "
               "instruction bound (issue_bound)")
        if not wlane:
            fmt = fmt_ref
    achieved = declared / (k_ms * 1e-3) / 1e9
    key = f"{key_graph}_{mode}_p{p:g}_q{q:g}{'_ext' if extend else ''}_w{W}_l{L}_seed{args.seed}"
    pmc = load_pmc(key) if world == 1 else None
    traffic = int(pmc["fetch_bytes"] + pmc["write_bytes"]) if pmc else None
    wide_note = ""
    if pmc and mode == "DenseOTF" and cfg["graph"] == "er":
        # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming
        # read (128-byte requests tallied at 64 B) -- the packed rows are read that way (8 B per lane, 512 B per wavefront load)
        traffic = int(2 * pmc["fetch_bytes"] + pmc["write_bytes"])
        wide_note = (" FETCH_SIZE doubled: wide coalesced row reads are tallied at half their bytes on gfx950 (the build image's "
                     "/opt/skills/guides/MI355X_MICROARCH.md, HBM section); raw counter value in traffic_raw_counter_bytes.")
    roofline = {
        "bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "declared_bytes_per_launch": int(declared), "declared_format": fmt,
        "avg_launch_ms": round(k_ms, 3), "rng_jump_and_expand_ms": round(float(np.mean(acc["rng_kernel_ms"])), 3),
        "traffic_note": ((pmc["note"] + wide_note) if pmc else "no PMC pass committed for this workload (profiles/r06_traffic.json)"),
        "random_sector_peak_GBps": RANDOM_SECTOR_GBS,
        "reference_format_bytes": ref_bytes,
    }
    if pmc and wide_note:
        roofline["traffic_raw_counter_bytes"] = int(pmc["fetch_bytes"] + pmc["write_bytes"])
    if traffic:
        roofline["traffic_GBps"] = round(traffic / (k_ms * 1e-3) / 1e9, 1)
        roofline["traffic_frac_of_peak"] = round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roofline["traffic_frac_of_random_sector_peak"] = round(traffic / (k_ms * 1e-3) / 1e9 / RANDOM_SECTOR_GBS, 4)
    if pmc and "issue" in pmc:
        roofline["issue_bound"] = pmc["issue"]
    if wlane:
        roofline["eager_step_frac"] = round(st["eager_steps"] / max(steps0, 1), 5)
        roofline["lane_rounds"] = int(st["lane_rounds"])
    if lane:
        roofline["ambiguous_step_frac"] = round(acc["ambiguous_steps"][-1] / max(steps0, 1), 5)
        roofline["float_chain_step_frac"] = round(acc["wave_chain_steps"][-1] / max(steps0, 1), 5)
        roofline["list_entries_per_step"] = round(acc["list_entries_read"][-1] / max(steps0, 1), 2)
        roofline["redo_walks"] = int(acc["redo_walks"][-1])
        roofline["lane_rounds"] = int(st["lane_rounds"])
        # the safety net under the argued bounds of the interval decision: 1/1024 of the steps it settles are
        # re-decided by the float chain inside every call -- summed over the timed passes
        roofline["verify_sampled_decisions"] = int(sum(acc["verify_checked"][-args.steps:]))
        roofline["verify_mismatches"] = int(sum(acc["verify_mismatch"][-args.steps:]))
        roofline["launch_note"] = ("one pass = lane_rounds launches of walk_lanes_kernel (walks whose step needs the float32 chain "
                                   "are parked and resumed by the next round) + one lanes_chain_kernel launch per queue; "
                                   "avg_launch_ms and declared_bytes_per_launch are per PASS (sum over these launches)")

    cpu = None
    if not args.no_cpu_baseline and world == 1 and cfg["graph"] == "rmat" and not extend:
        from oracle import pyoracle as orc

        # thread count: the box may expose more logical CPUs than the container can run on (affinity mask,
        # cgroup quota), and the allocation-heavy faithful port stops scaling well before 256 threads --
        # probe a few counts on the same 20000-job prefix and keep the fastest (the strongest baseline)
        logical = os.cpu_count() or 1
        try:
            usable = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            usable = logical
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q_us, period = f.read().split()
                if q_us != "max":
                    quota = max(1, int(int(q_us) / int(period)))
        except (OSError, ValueError):
            pass
        cand = sorted({c for c in (logical, usable, quota or usable, 128, 64, 32, 16, 8) if 1 <= c <= logical},
                      reverse=True)
        probe = starts[: 20000]
        probed = {}
        for c in cand:
            t = time.perf_counter()
            s, _ = orc.cpu_baseline_walks(indptr, indices, data, p, q, probe, L, args.seed,
                                          n_threads=c, faithful=True)
            probed[c] = max(s, 1) / (time.perf_counter() - t)
        cores = max(probed, key=probed.get)
        rate = probed[cores]
        n_sample = int(min(n_jobs, max(20000, rate * args.cpu_seconds / max(s / probe.size, 1e-9))))
        sample = starts[:n_sample]
        t = time.perf_counter()
        s_f, _ = orc.cpu_baseline_walks(indptr, indices, data, p, q, sample, L, args.seed,
                                        n_threads=cores, faithful=True)
        dt_f = time.perf_counter() - t
        t = time.perf_counter()
        s_t, _ = orc.cpu_baseline_walks(indptr, indices, data, p, q, sample, L, args.seed,
                                        n_threads=cores, faithful=False)
        dt_t = time.perf_counter() - t
        cpu_model = ""
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        cpu = {
            "value": round(s_f / dt_f / 1e6, 4), "unit": "million walk-steps/s", "cores": cores,
            "kind": "port",
            "sample": f"first {n_sample} of {n_jobs} shuffled jobs ({s_f} steps), faithful "
                      f"(per-step heap temporaries like Numba) OpenMP port, {dt_f:.1f}s",
            "tuned_value": round(s_t / dt_t / 1e6, 4), "cpu_model": cpu_model,
            "logical_cpus": logical, "usable_cpus": usable, "cgroup_cpu_quota": quota,
            "probe_msteps_per_s_by_threads": {str(c): round(v / 1e6, 3) for c, v in probed.items()},
        }

    # ---- the call Base.simulate_walks actually makes: host pointers in, the walk matrix out into pageable NumPy memory ----
    # (pw_simulate: parts walked while the finished ones leave over PCIe through a ring of pinned buffers; never `value`)
    host_call = None
    if world == 1 and not args.no_host_call and cfg["graph"] == "rmat":
        try:
            eng.simulate(mode, p, q, extend, starts[: max(1, n_jobs // 64)], L, seed=args.seed + 1000)   # (staging buffers exist)
            t = time.perf_counter()
            mat = eng.simulate(mode, p, q, extend, starts, L, seed=args.seed + 1001)
            dt = time.perf_counter() - t
            hs = int(eng.last_stats["total_steps"])
            out_bytes = int(mat.nbytes)
            del mat
            host_call = {"value_host_call": round(hs / dt / 1e6, 3), "host_call_ms": round(dt * 1e3, 2), "matrix_bytes": out_bytes,
                         "pcie_floor_ms_at_55GBps": round(out_bytes / 55e9 * 1e3, 2),
                         "note": "wall clock of ONE pw_simulate call on host pointers (H2D of the starts, walks, D2H of the matrix into "
                                 "fresh pageable memory), new seed; the device-resident `value` excludes the two copies"}
        except MemoryError:
            host_call = {"note": "not measured: the host could not allocate the walk matrix"}
        except Exception as exc:   # noqa: BLE001 (the headline line must be printed whatever this extra call does)
            host_call = {"note": f"not measured: {exc!r}"}

    build_s = info["build_ms"] * 1e-3
    library_warmup_ms = None
    try:
        from pecanpy_amd import _lib as _pw_lib2

        wm = _pw_lib2.warmup_ms()
        library_warmup_ms = round(wm, 1) if wm is not None else None
    except Exception:   # noqa: BLE001
        pass
    result = {
        "metric": f"million walk-steps/sec on {gdesc.split(' (')[0]} {mode} p={p:g} q={q:g}"
                  f"{' node2vec+' if extend else ''}",
        "value": round(value, 3),
        "unit": "million walk-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(sec_per_step * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64" if mode == "DenseOTF" else "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{gdesc} {mode}{' node2vec+ (extend, gamma 0)' if extend else ''} "
                        f"p={p:g} q={q:g}, {W} walks x {L} steps per vertex, random_state={args.seed}+pass "
                        f"(a new seed every pass: the MT19937 jump-ahead is paid inside the timed region)",
            "baseline_config": args.config,
            "n_nodes": int(n_nodes), "nnz": nnz, "n_jobs": int(n_jobs),
            "effective_steps_per_pass": total_steps, "nominal_steps_per_pass": int(n_jobs) * L,
            "nominal_value": round(int(n_jobs) * L / sec_per_step / 1e6, 3),
            "parallelism": f"jobs sharded over {world} GPU(s), graph replicated",
            "gather_on_rank0": bool(do_gather), "gather_chunks": n_chunks if do_gather else 0,
            "gather_mode": (gather_mode if do_gather else None), "rank0_share": round(rank0_share, 3),
            "shard_jobs": [b[1] - b[0] for b in all_bounds],
            "per_rank_walk_kernel_ms": [round(x, 3) for x in rank_ms],
            "overflow_reads": st["overflow_reads"], "host_prep_s": round(t_prep, 1),
            "graph_gen_s": round(t_graph, 1),
            # per-graph index built once by pw_csr_create (membership filters, adjacency index, per-edge
            # common-neighbour lists and records); NOT in the timed region -- the second figure charges it to
            # ONE pass of 10 x 80 walks (every rank builds its own replica)
            # graph_index_build_ms = device time of the index KERNELS (event pairs around them); graph_create_wall_ms =
            # wall clock of the whole handle creation as the caller sees it -- HIP runtime start-up in a fresh process,
            # host passes over the CSR, its H2D copy, device allocations (a fresh box's first multi-GB hipMalloc can take
            # > 100 ms) and the kernels; value_first_call charges THAT and one pass to one 10 x 80 run
            "graph_index_build_ms": round(info["build_ms"], 1),
            "graph_create_wall_ms": round(create_wall_ms, 1),
            "hip_runtime_startup_ms": round(hip_startup_ms, 1),
            "graph_create_note": "graph_create_wall_ms = wall clock of pw_csr_create; the ~140 ms the library's first stream costs in a process "
                                 "(rounds 4-5: inside this figure) run on a helper thread beside the graph generation since round 6 "
                                 "(library_warmup_ms; value_first_call_incl_warmup charges them in full); hip_runtime_startup_ms = what was "
                                 "left of the HIP runtime's own start-up just before pw_csr_create (torch has usually paid it)",
            "value_first_call": round(total_steps / (sec_per_step + (create_wall_ms + hip_startup_ms) * 1e-3 + param_index_ms[0] * 1e-3) / 1e6, 3),
            # round 6: the library's start-up ran on a helper thread beside the graph generation (pw_warmup); what it took, and the
            # first-call figure with it charged in full, as if nothing had overlapped it (rounds 4-5: it sat inside graph_create_wall_ms)
            "library_warmup_ms": library_warmup_ms,
            "value_first_call_incl_warmup": round(total_steps / (sec_per_step + (create_wall_ms + hip_startup_ms + (library_warmup_ms or 0.0)) * 1e-3
                                                                 + param_index_ms[0] * 1e-3) / 1e6, 3),
            # index that depends on (p, q, extend), built inside the first (warm-up) call and cached in the handle:
            # per-edge normalisers of weighted graphs
            "param_index_build_ms": round(param_index_ms[0], 1),
            "graph_index_bytes": info["index_bytes"],
            "lane_list_entries": info["lane_list_entries"],
            "value_incl_index_build": round(total_steps / (sec_per_step + build_s + param_index_ms[0] * 1e-3) / 1e6, 3),
            "host_call": host_call,
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if cpu:
        result["vs_cpu_baseline"] = round(value / cpu["value"], 2) if cpu["value"] else None
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
