#!/usr/bin/env python3
"""A/B of library builds on one workload (GPU box): the graph is generated once, every build runs in its own process
(PECANPY_AMD_LIB) on the same starts and seeds, and prints pass times, kernel times, counters and a checksum of the walk
matrix of every pass -- builds that differ in their walks show up at once.
usage: python tools/ab_bench.py [--scale 22] [--p 0.5 --q 2] [--passes 3] [--pmc-friendly] lib_a.so lib_b.so ..."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(args):
    import torch

    from pecanpy_amd.engine import WalkEngine

    z = np.load(args.graph)
    indptr, indices = z["indptr"], z["indices"]
    data = z["data"] if "data" in z else None
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * args.num_walks)
    np.random.RandomState(0).shuffle(starts)
    skip = 0
    if args.shard:                       # "i/N": the i-th of N contiguous shards, addressed into the stream like a rank of a multi-GPU run
        i, nsh = (int(t) for t in args.shard.split("/"))
        lo, hi = i * starts.size // nsh, (i + 1) * starts.size // nsh
        has = (indptr[1:] != indptr[:-1])
        skip = int(has[starts[:lo]].sum()) * args.walk_length
        starts = starts[lo:hi]
        has_sh = has[starts]
    elif args.jobs:
        starts = starts[: args.jobs]
    torch.cuda.init()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng = WalkEngine.from_csr(indptr, indices, data)
    create_wall = (time.perf_counter() - t0) * 1e3
    info = eng.index_info()
    if args.extend:
        from pecanpy_amd import pecanpy as node2vec

        g = node2vec.SparseOTF.from_csr(indptr, indices, data, extend=True, gamma=0)
        with np.errstate(all="ignore"):
            eng.set_thresholds(np.nan_to_num(g.get_noise_thresholds(), nan=0.0))
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    out = torch.empty((starts.size, args.walk_length + 2), dtype=torch.int32, device="cuda")
    res = {"lib": os.path.basename(os.environ.get("PECANPY_AMD_LIB", "libpecanpy_amd.so")), "create_wall_ms": round(create_wall, 1),
           "index_build_ms": round(info["build_ms"], 1), "index_GB": round(info["index_bytes"] / 1e9, 2), "passes": []}
    if args.chunks > 1:
        from pecanpy_amd.engine import tapered_bounds

        csum = np.concatenate([[0], np.cumsum(has_sh if args.shard else (indptr[1:] != indptr[:-1])[starts], dtype=np.int64)])
    for k in range(args.passes + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        if args.chunks > 1:              # tapered chunks, each its own call (what a rank does while chunk c travels): sum of the calls
            if args.hold:
                eng.stream_hold(k, skip, int(csum[-1]) * args.walk_length)
            agg = None
            for a, b in tapered_bounds(starts.size, args.chunks):
                eng.simulate_device("SparseOTF", args.p, args.q, args.extend, d_starts[a:b], args.walk_length, seed=k, out=out[a:b],
                                    stream_skip=skip + int(csum[a]) * args.walk_length)
                cs = dict(eng.last_stats)
                if agg is None:
                    agg = cs
                else:
                    for key in ("walk_kernel_ms", "lane_kernel_ms", "rng_kernel_ms", "lane_rounds", "total_steps", "ambiguous_steps",
                                "wave_chain_steps", "list_entries_read", "redo_walks"):
                        agg[key] += cs[key]
            if args.hold:
                eng.stream_release()
            eng.last_stats = agg
        else:
            eng.simulate_device("SparseOTF", args.p, args.q, args.extend, d_starts, args.walk_length, seed=k, out=out, stream_skip=skip)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3
        st = eng.last_stats
        ck = int(out.long().sum().item()) ^ int((out[:, 1:-1].long() * torch.arange(1, args.walk_length + 1, device="cuda")).sum().item())
        res["passes"].append({"seed": k, "ms": round(ms, 2), "walk_ms": round(st["walk_kernel_ms"], 2), "lane_ms": round(st["lane_kernel_ms"], 2),
                              "rng_ms": round(st["rng_kernel_ms"], 2), "rounds": st["lane_rounds"], "steps": st["total_steps"],
                              "Msteps_s": round(st["total_steps"] / ms / 1e3, 1), "amb": st["ambiguous_steps"], "chain": st["wave_chain_steps"],
                              "probes": st["list_entries_read"], "redo": st["redo_walks"], "param_ms": round(st["param_index_ms"], 1),
                              "lane_kernel": st["lane_kernel"], "checksum": ck})
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--p", type=float, default=0.5)
    ap.add_argument("--q", type=float, default=2.0)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--shard", default="", help="i/N: walk the i-th of N contiguous shards of the job array at its stream offset")
    ap.add_argument("--chunks", type=int, default=1, help="walk the job array in this many tapered chunks (one call each)")
    ap.add_argument("--hold", action="store_true", help="with --chunks: expand the stream of the whole array once (pw_stream_hold)")
    ap.add_argument("--num-walks", type=int, default=10)
    ap.add_argument("--walk-length", type=int, default=80)
    ap.add_argument("--weighted", action="store_true")
    ap.add_argument("--extend", action="store_true")
    ap.add_argument("--graph", default=None)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    from pecanpy_amd.synth import rmat_csr

    path = f"/tmp/ab_rmat{args.scale}{'w' if args.weighted or args.extend else ''}.npz"
    if not os.path.exists(path):
        t = time.time()
        indptr, indices, data = rmat_csr(args.scale, seed=1, weighted=args.weighted or args.extend)
        if args.weighted or args.extend:
            np.savez(path, indptr=indptr, indices=indices, data=data)
        else:
            np.savez(path, indptr=indptr, indices=indices)
        print(f"# graph RMAT-{args.scale} generated in {time.time() - t:.1f}s", flush=True)
    for lib in args.libs or ["libpecanpy_amd.so"]:
        env = dict(os.environ, PECANPY_AMD_LIB=os.path.join(REPO, "pecanpy_amd", lib))
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--graph", path, "--scale", str(args.scale), "--p", str(args.p),
               "--q", str(args.q), "--passes", str(args.passes), "--jobs", str(args.jobs), "--num-walks", str(args.num_walks),
               "--walk-length", str(args.walk_length), "--chunks", str(args.chunks)] + (["--extend"] if args.extend else []) + \
              (["--weighted"] if args.weighted else []) + (["--shard", args.shard] if args.shard else []) + (["--hold"] if args.hold else [])
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            print(line if line.startswith("{") else f"# {lib}: rc={r.returncode} {r.stderr[-600:]}", flush=True)
            for ln in r.stderr.splitlines():
                if ln.startswith("[lanes]") or ln.startswith("[lane_prof]") or ln.startswith("[watchdog]"):
                    print("#   " + ln, flush=True)
        except subprocess.TimeoutExpired:
            print(f"# {lib}: TIMEOUT", flush=True)


if __name__ == "__main__":
    main()
