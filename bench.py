#!/usr/bin/env python3
"""bench.py -- cells/s of the Snapshot+Log chunk build (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one pass of the hot path (dcdf_encoder_run: the fused HIP encoder over every chunk of the batch,
heuristic + DAC/bitmap packing + serialization into HBM) over one batch of synthetic chunks that is already
resident in HBM.

Workloads
  config2 (default, the configuration BASELINE.json's metric and target are quoted on: configs[2]/[3]):
      ONE 4096x4096x365 raster (seed 0xDCDF0003) tiled into 16x16 tiles of 256^2 and 12 time segments of <= 32
      instants = 3072 chunks.  With N ranks the SAME raster is sharded by chunk (dcdf_amd/shard.py, balanced by
      cell count) -> "scaling": "strong".  No data-path collective: torch.distributed (RCCL) is used only for the
      barrier and the max-over-ranks of the time; after the timed region every rank's encoded buffers and
      per-instant (min,max) pairs are gathered on the host side (reported under "gather", never part of `value`).
  config1 (--workload config1, BASELINE configs[1]): 1024 independent [32,256,256] chunks per GPU, seed
      0xDCDF0002 + c; every rank owns its own 1024 chunks -> "scaling": "weak".

Launch: `python bench.py --gpus N`.  When N > 1 and no launcher set WORLD_SIZE, this process starts N ranks itself
(one child process per GPU, before anything here touches the GPU) and relays rank 0's line.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are the launcher's.

Prints ONE JSON line on rank 0 (see the driver contract), with extra objects:
  roofline     : algorithmic bytes (input cells * sizeof + encoded bytes) / HIP-event time of the encode kernel
                 (events recorded on the stream the kernel is launched on, inside dcdf_encoder_run)
  cpu_baseline : the CPU oracle (C++ restatement of the reference, 1 thread like the reference) on a bounded
                 sample of the same chunks, timed on this host (N = 1 only)
  gather       : the final host-side gather of {offsets, bytes, minmax} and the SHA-256 of the concatenation of all
                 chunks in chunk order (equal for every N)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["config1", "config2"], default="config2")
    ap.add_argument("--chunks", type=int, default=1024, help="config1: chunks per GPU")
    ap.add_argument("--instants", type=int, default=32, help="config1: instants per chunk")
    ap.add_argument("--side", type=int, default=256, help="config1: tile side")
    ap.add_argument("--days", type=int, default=365, help="config2: instants of the raster (365)")
    ap.add_argument("--extent", type=int, default=4096, help="config2: rows = cols of the raster (4096)")
    ap.add_argument("--dtype", choices=["i32", "i64", "f32", "f64"], default="i32",
                    help="i32 (default): stored fixed-point integers; f32/f64: floats converted on the fly (to_fixed, --fbits)")
    ap.add_argument("--fbits", type=int, default=3, help="fractional bits of the float workloads")
    ap.add_argument("--cpu-sample", type=int, default=24, help="chunks timed on the CPU oracle (0 = skip)")
    ap.add_argument("--verify", type=int, default=4, help="chunks compared byte-for-byte with the oracle (untimed)")
    ap.add_argument("--no-gather", action="store_true", help="skip the host-side gather + SHA-256 (profiling runs)")
    ap.add_argument("--pad-elems", type=int, default=0, help="config1: extra elements between consecutive chunks in HBM")
    ap.add_argument("--dataset", choices=["model", "noise", "wide"], default="model",
                    help="model: the SURVEY 8(d) value model (default, the headline workload).  noise: iid U[0, 2^30) cells -- the "
                         "adversarial set of SURVEY 8(d): every instant's Snapshot wins chunk.rs:62, i.e. the general (re-reading) "
                         "emission path and one-instant blocks; a throughput figure for that path, never the headline.  wide: the "
                         "model raster times 300 -- log differences need three bytes and the snapshots' ranges exceed 16 bits (no compact "
                         "copy), logs still win chunk.rs:62: the general path's LOG side")
    ap.add_argument("--host-sample", type=int, default=1024, help="chunks pushed through the host-buffer entry point for the "
                    "PCIe-inclusive end-to-end figure (0 = skip); 1024 chunks = 8.6 GB of int32 input: long enough for the pinned "
                    "double buffers of the upload and of the download to reach their steady state")
    ap.add_argument("--decode-queries", type=int, default=200000,
                    help="after the timed region (full configs[2] int32 raster, N = 1 only): this many configs[4] queries against the "
                         "chunks the session left on the device -> the line's \"decode\" object (0 = skip)")
    ap.add_argument("--also", default="i64,f32,f64", help="after the timed region (full configs[2] int32 raster, N = 1 only): short runs "
                    "of the same raster in these element types -> the line's \"also\" object ('' = skip)")
    return ap.parse_args(argv)


def spawn_ranks(n):
    """--gpus N without a launcher: start N fresh rank processes (this process has not touched the GPU and never will)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Poll ALL children: if any rank dies, the others would sit in a collective until it times out -- end them at once
    # (this parent never touches the GPU, and the children are fresh processes of its own).
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        time.sleep(0.05)
    if failed is not None:
        print("bench.py: rank %d exited with code %s; stopping the other ranks" % (failed, procs[failed].returncode), file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        return 1
    reader.join(timeout=10)
    sys.stdout.write((chunks[0] if chunks else b"").decode())
    sys.stdout.flush()
    return max(abs(p.returncode) for p in procs)


def source_sha():
    """Identifies the kernel sources a committed PMC profile belongs to (roofline.traffic is only quoted for them)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dcdf_amd", "csrc")
    # the sources the fused encoder kernel is built from (the query / superchunk / universal kernels do not enter this bench)
    for f in ("Makefile", "k2r_common.h", "k2r_exec.h", "k2r_encode.h", "k2r_encode_inst.hip", "k2r_kernels.hip", "k2r_launch.h",
              "k2r_runtime.h", "k2r_capi_encode.hip", "k2r_synth.hip"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries ONE line, the JSON result: whatever libraries write to file descriptor 1 meanwhile (gloo announces its
    # connections there) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch

    # one rank per GPU; DCDF_BENCH_BACKEND=gloo lets several ranks share a card (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("DCDF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank)
    dist = None
    host_group = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        host_group = dist.group.WORLD if backend == "gloo" else dist.new_group(backend="gloo")  # the host-side gather

    from dcdf_amd import _lib as L
    from dcdf_amd.encoder import Encoder, synth_fill
    from dcdf_amd.shard import my_chunks, partition

    name = L.lib().dcdf_device_name()
    if not name:
        raise RuntimeError("libdcdf_k2r.so sees no GPU; the MI355X path has no CPU fallback")

    tdt = {"i32": torch.int32, "i64": torch.int64, "f32": torch.int32, "f64": torch.int32}[args.dtype]
    code = {"i32": L.DCDF_I32, "i64": L.DCDF_I64, "f32": L.DCDF_I32, "f64": L.DCDF_I32}[args.dtype]
    esz = 8 if args.dtype in ("i64", "f64") else 4
    fb = 0
    if args.workload == "config1":
        n, T, S = args.chunks, args.instants, args.side
        per = T * S * S + args.pad_elems
        flat = torch.empty((n * per,), dtype=tdt, device="cuda")
        data = [flat[c * per:c * per + T * S * S].view(T, S, S) for c in range(n)]
        base_seed = 0xDCDF0002 + rank * n
        for c in range(n):
            synth_fill(data[c].data_ptr(), code, base_seed + c, 0, T, 0, S, 0, S)
        ids = list(range(rank * n, rank * n + n))  # global chunk ids (for the gather order)
        n_global = n * world
        scaling = "weak"
        workload = "configs[1]: %d independent [%d,%d,%d] %s chunks per GPU, seed 0xDCDF0002+c" % (n, T, S, S, args.dtype)
    else:
        # BASELINE configs[2]/[3]: chunk (seg, i, j) = instants [32 seg, min(days, 32 seg + 32)) of rows [256 i, 256 i + 256),
        # cols [256 j, 256 j + 256), as Variable::append (dataset.rs:838) x Superchunk::build (superchunk.rs:127-181) cut it;
        # dealt to the ranks balanced by cell count
        S, T = 256, 32
        nseg, nt = (args.days + T - 1) // T, args.extent // S
        grid = [(seg, i, j) for seg in range(nseg) for i in range(nt) for j in range(nt)]
        seglen = lambda seg: min(args.days, T * seg + T) - T * seg
        owner = partition([seglen(seg) * S * S for seg, _, _ in grid], world)
        ids = my_chunks(owner, rank)
        mine = [grid[g] for g in ids]
        n = len(mine)
        n_global = len(grid)
        sizes = [seglen(seg) * S * S for seg, _, _ in mine]
        offs = [0]
        for z in sizes:
            offs.append(offs[-1] + z)
        flat = torch.empty((offs[-1],), dtype=tdt, device="cuda")
        data = []
        for (seg, i, j), o, z in zip(mine, offs, sizes):
            t0, t1 = T * seg, T * seg + seglen(seg)
            v = flat[o:o + z].view(t1 - t0, S, S)
            synth_fill(v.data_ptr(), code, 0xDCDF0003, t0, t1, S * i, S * i + S, S * j, S * j + S)
            data.append(v)
        scaling = "strong"
        workload = ("configs[2]: %dx%dx%d %s raster, seed 0xDCDF0003, tiled into %d [<=32,256,256] chunks; %d of them on "
                    "this GPU" % (args.extent, args.extent, args.days, args.dtype, n_global, n))
    if args.dataset == "noise":
        g = torch.Generator(device="cuda")
        g.manual_seed(0xDCDF0006 + rank)
        flat.copy_(torch.randint(0, 1 << 30, (flat.numel(),), generator=g, device="cuda", dtype=torch.int64).to(tdt))
        workload += "; cells replaced by iid U[0, 2^30) noise (--dataset noise)"
    if args.dataset == "wide":
        flat.mul_(300)
        workload += "; cells multiplied by 300 (--dataset wide: three-byte log values, no compact snapshot copy)"
    torch.cuda.synchronize()
    if args.dtype in ("f32", "f64"):  # the same integers as exact multiples of 2^-fbits in floating point (|v| < 2^24)
        fb = args.fbits
        assert int(flat.abs().max().item()) < (1 << 24)
        flat2 = flat.to(torch.float32 if args.dtype == "f32" else torch.float64) / float(1 << fb)
        data = [flat2[d.storage_offset():d.storage_offset() + d.numel()].view(d.shape) for d in data]
        flat = flat2
        code = L.DCDF_F32 if args.dtype == "f32" else L.DCDF_F64

    # diagnostics (config1 only): DCDF_BENCH_SAME=copy -> every chunk holds chunk 0's cells (equal work per chunk, still streamed
    # from HBM); =alias -> every descriptor points AT chunk 0 (the same work served by the caches): their difference is what the
    # memory system costs the kernel
    same = os.environ.get("DCDF_BENCH_SAME")
    if same and args.workload == "config1":
        if same == "copy":
            for d in data[1:]:
                d.copy_(data[0])
        else:
            data = [data[0]] * len(data)
        torch.cuda.synchronize()
        workload += "; DCDF_BENCH_SAME=" + same
    full_config2 = (args.workload == "config2" and args.dtype == "i32" and args.dataset == "model" and args.days == 365 and
                    args.extent == 4096 and not same)
    descs = [(d.data_ptr(), code, (S * S, S, 1), tuple(d.shape), fb, 0) for d in data]
    enc = Encoder(descs, k=2)
    cells_local = sum(d.numel() for d in data)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        enc.run()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kernel_ms.append(enc.run())  # launches on the session stream and waits for it
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    cells_total = cells_local
    if dist is not None:
        dev = "cpu" if backend == "gloo" else "cuda"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cc = torch.tensor([cells_local], dtype=torch.int64, device=dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        cells_total = int(cc.item())

    bad = sum(1 for i in range(n) if enc.result(i)[0] != 0)
    out_bytes = enc.total_bytes()
    snaps = sum(enc.result(i)[2] for i in range(n))

    # ---- the final host-side gather of {offsets, bytes, minmax} (untimed w.r.t. `value`; reported on its own) --------
    gather = None
    if not args.no_gather:
        g0 = time.perf_counter()
        buf, goffs, glens, mm = enc.gather()  # device pack + one D2H copy per rank
        g1 = time.perf_counter()
        if dist is not None:
            meta = [None] * world if rank == 0 else None
            dist.gather_object((ids, glens.tolist(), int(mm.shape[0])), meta, dst=0, group=host_group)
            if rank == 0:
                parts = {0: (buf, goffs, glens)}
                total_b, total_mm = int(glens.sum()), int(mm.shape[0])
                for r in range(1, world):
                    rl = np.array(meta[r][1], dtype=np.uint64)
                    ro = np.zeros(len(rl), dtype=np.uint64)
                    if len(rl) > 1:
                        ro[1:] = np.cumsum((rl[:-1] + np.uint64(15)) & ~np.uint64(15))
                    nb = int(((rl + np.uint64(15)) & ~np.uint64(15)).sum())
                    rb = torch.empty(max(1, nb), dtype=torch.uint8)
                    dist.recv(rb, src=r, group=host_group)
                    rm = torch.empty((max(1, meta[r][2]), 2), dtype=torch.int64)
                    dist.recv(rm, src=r, group=host_group)
                    parts[r] = (rb.numpy(), ro, rl)
                    total_b += int(rl.sum())
                    total_mm += meta[r][2]
                g2 = time.perf_counter()
                where = {}
                for r in range(world):
                    for q, cid in enumerate(meta[r][0]):
                        where[cid] = (r, q)
                h = hashlib.sha256()
                for cid in range(n_global):
                    r, q = where[cid]
                    b, o, l = parts[r]
                    h.update(memoryview(b[int(o[q]):int(o[q]) + int(l[q])]))
                sha, nchunks = h.hexdigest(), len(where)
            else:
                dist.send(torch.from_numpy(buf), dst=0, group=host_group)
                dist.send(torch.from_numpy(mm if mm.size else np.zeros((1, 2), dtype=np.int64)), dst=0, group=host_group)
        else:
            g2 = g1
            h = hashlib.sha256()
            for q in range(n):
                h.update(memoryview(buf[int(goffs[q]):int(goffs[q]) + int(glens[q])]))
            sha, nchunks, total_b, total_mm = h.hexdigest(), n, int(glens.sum()), int(mm.shape[0])
        if rank == 0:
            gather = {"device_pack_and_d2h_s": g1 - g0, "rank_exchange_s": g2 - g1, "chunks": nchunks, "bytes": total_b,
                      "minmax_pairs": total_mm, "sha256_of_concatenation_in_chunk_order": sha,
                      "note": "host-side, after the timed region; equal sha256 for every --gpus N of one workload"}
            # the golden of tests/test_gpu_configs.py (every one of the 3072 chunks compared with the oracle there)
            if full_config2:
                try:
                    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config2_sha256.json")))
                    gather["sha_matches_golden"] = sha == gold["sha256_of_concatenation_in_chunk_order"]
                except Exception:
                    gather["sha_matches_golden"] = None

    # ---- untimed parity spot check against the oracle (rank 0) -------------------------------------
    verified = 0
    cpu = None
    if rank == 0:
        import oracle_lib as O
        nv = min(args.verify, n)
        for c in sorted(set(int(round(i * (n - 1) / max(1, nv - 1))) for i in range(nv))) if nv else []:  # spread over all time segments
            host = data[c].cpu().numpy()
            assert enc.fetch(c) == O.chunk_build(host, fractional_bits=fb), "chunk %d: encoded bytes differ from the oracle" % c
            verified += 1
        if world == 1 and args.cpu_sample > 0:
            full = [d for d in data if d.shape[0] == T]
            m = min(args.cpu_sample, len(full))
            sample = torch.stack(full[:m]).cpu().numpy()
            native = O.native_lib() is not None  # rebuilt -O3 -march=native on THIS host; else the portable -O3 build
            sec, tb, _ = O.bench_build(sample, native=native)
            cpu = {"value": sample.size / sec, "unit": "cells/s", "cores": 1, "kind": "port",
                   "sample": "first %d of the %d [%d,%d,%d] %s chunks of this workload (%.1f s); C++ restatement of the Rust "
                             "reference (no Rust toolchain), built -O3 %s, serial like "
                             "superchunk.rs:166-188" % (m, n, T, S, S, args.dtype, sec,
                                                         "-march=native on this host" if native else "without -march=native")}
            # the same port with one chunk per task on every host core (the reference itself is serial);
            # rounds of 2 tasks per thread until at least 2.5 s have been measured
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, min(16, len(os.sched_getaffinity(0))))  # a one-GPU box's CPU share is 16 cores
            reps = [sample[i % m:i % m + 1] for i in range(2 * nthr)]
            done_cells, rounds = 0, 0
            w0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                while time.perf_counter() - w0 < 2.5:
                    list(ex.map(lambda r: O.bench_build(r, native=native), reps))  # ctypes releases the GIL inside the call
                    done_cells += sum(r.size for r in reps)
                    rounds += 1
            wall = time.perf_counter() - w0
            cpu["all_cores"] = {"value": done_cells / wall, "unit": "cells/s", "cores": nthr,
                                "sample": "%d chunk builds on %d threads (%.1f s)" % (rounds * len(reps), nthr, wall)}

    # ---- end to end through the host-buffer entry point ("pinned H2D + D2H gather"), rank 0, N = 1 --------
    # dcdf_chunk_build_batch on tiles in HOST memory: staging, H2D, the same kernel, D2H of the bytes -- PCIe inclusive.  A
    # bounded sample of the same chunks; reported under its own key, never `value`.
    end_to_end = None
    if rank == 0 and world == 1 and args.host_sample > 0:
        import ctypes as C
        from dcdf_amd.chunk import _desc
        hs = [data[c].cpu().numpy() for c in range(min(args.host_sample, n))]
        hd = (L.TileDesc * len(hs))()
        for i, a in enumerate(hs):
            hd[i] = _desc(a, fb, False)
        best, out_b = None, 0
        for _ in range(3):
            res = C.POINTER(L.Encoded)()
            h0 = time.perf_counter()
            L.check(L.lib().dcdf_chunk_build_batch(hd, C.c_size_t(len(hs)), 2, L.MEM_HOST, C.byref(res)), "chunk_build_batch")
            dt = time.perf_counter() - h0
            out_b = sum(int(res[i].len) for i in range(len(hs)))
            L.lib().dcdf_free_encoded(res, C.c_size_t(len(hs)))
            best = dt if best is None else min(best, dt)
        hcells = sum(a.size for a in hs)
        end_to_end = {"entry": "dcdf_chunk_build_batch, tiles and results in host memory (staging + H2D + kernel + D2H)",
                      "value": hcells / best, "unit": "cells/s", "input_GB_per_s": hcells * esz / best / 1e9,
                      "output_GB_per_s": out_b / best / 1e9, "input_bytes": hcells * esz, "output_bytes": out_b,
                      "sample": "first %d chunks of this workload, pageable numpy arrays in, malloc'ed buffers out, best of 3 (%.3f s)" % (len(hs), best)}
        del hs, hd

    # ---- "also report against a measured device-to-device copy on the same GPU" ------------------------------
    copy_gbs = None
    if rank == 0 and world == 1:
        nb = 1 << 31
        a = torch.empty(nb, dtype=torch.uint8, device="cuda")
        b = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        a.copy_(b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            a.copy_(b)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * nb / (e0.elapsed_time(e1) * 1e-3) / 1e9  # bytes read + bytes written
        del a, b

    # ---- BASELINE configs[4] (row d2): a bounded sample of the decode path against the chunks this session left on the device ----
    decode = None
    if rank == 0 and world == 1 and full_config2 and args.decode_queries > 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_query
        d0 = time.perf_counter()
        q = bench_query.run(queries=args.decode_queries, batch=min(args.decode_queries, 100000), check=4, cpu_sample=1500,
                            host_results=False, session=(enc, args.days))
        rl, cb = q["roofline"], q.get("cpu_baseline") or {}
        decode = {"workload": q["config"]["workload"], "queries": args.decode_queries,
                  "queries_per_s": q["queries_per_s"], "cells_per_s": q["value"],
                  "kernel_ms": q["fill_window"]["kernel_ms"] + q["search_window"]["kernel_ms"],
                  "fill_window": {k: q["fill_window"][k] for k in ("queries", "chunk_level_subqueries", "kernel_ms", "cells", "cells_per_s_kernel", "queries_per_s_kernel")},
                  "search_window": {k: q["search_window"][k] for k in ("queries", "chunk_level_subqueries", "kernel_ms", "hits", "queries_per_s_kernel")},
                  "answers_checked_vs_model": q["config"]["answers_checked_vs_model"],
                  "open_chunks_s": q["open"]["seconds"],
                  "end_to_end_raster_entry_points": q["end_to_end"]["raster_level"],
                  "roofline": {"bound": "hbm", "achieved": rl["achieved"], "peak": rl["peak"], "unit": "GB/s", "frac": rl["frac"],
                               "traffic": rl.get("traffic"), "traffic_detail": rl.get("traffic_detail"),
                               "algorithmic_bytes": rl["fill_window"]["algorithmic_bytes"] + rl["search_window"]["algorithmic_bytes"],
                               "algorithmic_bytes_upper_bound_whole_structures": rl["fill_window"]["encoded_bytes_upper_bound"] + rl["fill_window"]["output_bytes"] +
                               rl["search_window"]["encoded_bytes_upper_bound"] + rl["search_window"]["output_bytes"],
                               "kernel": "k2r::k_window_wave2 (+ k_search_count / k_search_emit)", "note": rl["note"]},
                  "cpu_baseline": {"kind": cb.get("kind"), "unit": cb.get("unit"), "cores": 1, "sample": cb.get("sample"),
                                   "fill_window": cb.get("fill_window"), "search_window": cb.get("search_window"),
                                   "gpu_subqueries_per_s_kernel": cb.get("gpu_subqueries_per_s_kernel")},
                  "wall_s": time.perf_counter() - d0}

    # ---- the same raster in the other element types: short runs in child processes (this one keeps its memory) ----------------
    also = None
    if rank == 0 and world == 1 and full_config2 and args.also:
        also = {}
        for dt in [x for x in args.also.split(",") if x and x != "i32"]:
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", dt, "--steps", "3", "--warmup", "1", "--no-gather", "--cpu-sample", "0",
                   "--host-sample", "0", "--verify", "2", "--decode-queries", "0", "--also", ""]
            try:
                out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True).stdout.decode()
                cl = json.loads(out.strip().splitlines()[-1])
                also[cl["dtype"]] = {"value": cl["value"], "unit": cl["unit"], "kernel_ms": cl["roofline"]["kernel_ms"],
                                     "frac": cl["roofline"]["frac"], "kernel": cl["roofline"]["kernel"],
                                     "algorithmic_bytes": cl["roofline"]["algorithmic_bytes"], "steps": cl["steps"],
                                     "failed_tiles": cl["config"]["failed_tiles_rank0"],
                                     "bytes_verified_vs_oracle": cl["config"]["bytes_verified_vs_oracle"]}
            except Exception as e:  # (a side figure must never take the headline line down)
                also[dt] = {"error": repr(e)[:200]}

    if rank == 0:
        k_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = cells_local * esz + out_bytes  # every input cell read once, every output byte written once
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch come from separate rocprofv3 PMC passes (tools/collect_profiles.sh); they are quoted only
        # when that profile was taken on this workload with these kernel sources
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        key = "%s/%s/n%d/w%d%s" % (args.workload, args.dtype, n, world, "" if args.dataset == "model" else "/" + args.dataset)
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if rec.get("workload_key") == key and rec.get("source_sha") == source_sha():
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_src = rec.get("tag")
            except Exception:
                traffic = None
        line = {
            "metric": "raster cells/s encoded (Snapshot+Log build)",
            "value": cells_total * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": {"i32": "int32", "i64": "int64", "f32": "f32->fixed(int32 arithmetic)", "f64": "f64->fixed(int32 arithmetic)"}[args.dtype],
            "data": "synthetic",
            "config": {"workload": workload, "workload_key": key,
                       "chunks_total": n_global, "chunks_on_rank0": n, "cells_total": cells_total, "k": 2,
                       "device": name.decode(), "failed_tiles_rank0": bad,
                       "encoded_bytes_rank0": out_bytes, "snapshots_rank0": snaps, "bytes_verified_vs_oracle": verified,
                       "parallelism": "chunks sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_from_profile": traffic_src,
                         "kernel": "k2r::k_encode<%d,false,%d>" % (S.bit_length() - 1, {"i32": 1, "f32": 2, "i64": 3, "f64": 4}[args.dtype]),
                         "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes, "scope": "rank 0's GPU, HIP events around the encode kernel",
                         "source_sha": source_sha(),
                         "measured_copy": None if copy_gbs is None else
                         {"GB/s": copy_gbs, "frac_of_copy": achieved / copy_gbs,
                          "what": "device-to-device copy of 2 GiB on this GPU, bytes read + written per second"}},
            "cpu_baseline": cpu,
            "end_to_end_host_buffers": end_to_end,
            "gather": gather,
            "decode": decode,
            "also": also,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
