#!/usr/bin/env python3
"""bench.py -- cells/s of the Snapshot+Log chunk build (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one pass of the hot path (dcdf_encoder_run: the fused HIP encoder over every chunk of the batch,
heuristic + DAC/bitmap packing + serialization into HBM) over one batch of synthetic chunks that is already
resident in HBM.  Workload at N=1 = BASELINE configs[1]: 1024 independent [32,256,256] chunks (seed
0xDCDF0002 + c).  With N ranks every rank owns its own 1024 chunks (independent units, no data-path
collective; torch.distributed/RCCL is used only for the barrier and the max-over-ranks of the time).

Prints ONE JSON line on rank 0 (see the driver contract), with two extra objects:
  roofline     : algorithmic bytes (input cells * sizeof + encoded bytes) / HIP-event time of the encode kernel
  cpu_baseline : the CPU oracle (C++ restatement of the reference, 1 thread like the reference) on a bounded
                 sample of the same chunks, timed on this host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=1024, help="chunks per GPU (configs[1]: 1024)")
    ap.add_argument("--instants", type=int, default=32)
    ap.add_argument("--side", type=int, default=256)
    ap.add_argument("--dtype", choices=["i32", "i64", "f32", "f64"], default="i32",
                    help="i32 (default): stored fixed-point integers; f32: floats converted on the fly (to_fixed, --fbits)")
    ap.add_argument("--fbits", type=int, default=3, help="fractional bits of the f32 workload")
    ap.add_argument("--cpu-sample", type=int, default=24, help="chunks timed on the CPU oracle (0 = skip)")
    ap.add_argument("--verify", type=int, default=4, help="chunks compared byte-for-byte with the oracle (untimed)")
    ap.add_argument("--pad-elems", type=int, default=0, help="extra elements between consecutive chunks in HBM")
    ap.add_argument("--workload", choices=["config1", "config2"], default="config1",
                    help="config1 (default, the configuration the metric is quoted on): --chunks independent chunks per GPU; "
                         "config2: the 4096x4096x365 raster (seed 0xDCDF0003) tiled into 16x16x12 chunks of <= 32 instants, "
                         "sharded over the ranks by chunk")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
    # one rank per GPU; DCDF_BENCH_BACKEND=gloo lets several ranks share a card (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("DCDF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from dcdf_amd import _lib as L
    from dcdf_amd.encoder import Encoder, synth_fill

    name = L.lib().dcdf_device_name()
    if not name:
        raise RuntimeError("libdcdf_k2r.so sees no GPU; the MI355X path has no CPU fallback")

    n, T, S = args.chunks, args.instants, args.side
    tdt = {"i32": torch.int32, "i64": torch.int64, "f32": torch.int32, "f64": torch.int32}[args.dtype]
    code = {"i32": L.DCDF_I32, "i64": L.DCDF_I64, "f32": L.DCDF_I32, "f64": L.DCDF_I32}[args.dtype]
    esz = 8 if args.dtype in ("i64", "f64") else 4
    fb = 0
    if args.workload == "config1":
        per = T * S * S + args.pad_elems
        flat = torch.empty((n * per,), dtype=tdt, device="cuda")
        data = [flat[c * per:c * per + T * S * S].view(T, S, S) for c in range(n)]
        base_seed = 0xDCDF0002 + rank * n
        for c in range(n):
            synth_fill(data[c].data_ptr(), code, base_seed + c, 0, T, 0, S, 0, S)
        workload = "configs[1]: %d independent [%d,%d,%d] %s chunks per GPU, seed 0xDCDF0002+c" % (n, T, S, S, args.dtype)
    else:
        # BASELINE configs[2]/[3]: one 4096x4096x365 raster; chunk (seg, i, j) = instants [32 seg, min(365, 32 seg + 32)) of
        # rows [256 i, 256 i + 256), cols [256 j, 256 j + 256); the chunks are dealt to the ranks balanced by cell count (dcdf_amd/shard.py)
        from dcdf_amd.shard import my_chunks, partition
        S, T = 256, 32
        grid = [(seg, i, j) for seg in range(12) for i in range(16) for j in range(16)]
        owner = partition([(min(365, 32 * seg + 32) - 32 * seg) * S * S for seg, _, _ in grid], world)
        mine = [grid[g] for g in my_chunks(owner, rank)]
        n = len(mine)
        sizes = [(min(365, 32 * seg + 32) - 32 * seg) * S * S for seg, _, _ in mine]
        offs = [0]
        for z in sizes:
            offs.append(offs[-1] + z)
        flat = torch.empty((offs[-1],), dtype=tdt, device="cuda")
        data = []
        for (seg, i, j), o, z in zip(mine, offs, sizes):
            t0, t1 = 32 * seg, min(365, 32 * seg + 32)
            v = flat[o:o + z].view(t1 - t0, S, S)
            synth_fill(v.data_ptr(), code, 0xDCDF0003, t0, t1, S * i, S * i + S, S * j, S * j + S)
            data.append(v)
        workload = "configs[2]: 4096x4096x365 %s raster, seed 0xDCDF0003, %d of 3072 [<=32,256,256] chunks on this GPU" % (args.dtype, n)
    torch.cuda.synchronize()
    if args.dtype in ("f32", "f64"):  # the same integers as exact multiples of 2^-fbits in floating point (|v| < 2^24)
        fb = args.fbits
        assert int(flat.abs().max().item()) < (1 << 24)
        flat2 = flat.to(torch.float32 if args.dtype == "f32" else torch.float64) / float(1 << fb)
        data = [flat2[d.storage_offset():d.storage_offset() + d.numel()].view(d.shape) for d in data]
        flat = flat2
        code = L.DCDF_F32 if args.dtype == "f32" else L.DCDF_F64

    descs = [(d.data_ptr(), code, (S * S, S, 1), tuple(d.shape), fb, 0) for d in data]
    enc = Encoder(descs, k=2)
    cells_per_step = sum(d.numel() for d in data)

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
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    bad = sum(1 for i in range(n) if enc.result(i)[0] != 0)
    out_bytes = enc.total_bytes()
    snaps = sum(enc.result(i)[2] for i in range(n))

    # ---- untimed parity spot check against the oracle (rank 0) -------------------------------------
    verified = 0
    cpu = None
    if rank == 0:
        import oracle_lib as O
        for c in range(min(args.verify, n)):
            host = data[c].cpu().numpy()
            assert enc.fetch(c) == O.chunk_build(host, fractional_bits=fb), "chunk %d: encoded bytes differ from the oracle" % c
            verified += 1
        if world == 1 and args.cpu_sample > 0:
            m = min(args.cpu_sample, n)
            m = min(m, sum(1 for d in data if d.shape == data[0].shape))
            sample = torch.stack([d for d in data if d.shape == data[0].shape][:m]).cpu().numpy()
            sec, tb, _ = O.bench_build(sample)
            cpu = {"value": sample.size / sec, "unit": "cells/s", "cores": 1, "kind": "port",
                   "sample": "first %d of the %d [%d,%d,%d] %s chunks (%.1f s); C++ restatement of the Rust "
                             "reference, serial like superchunk.rs:166-188" % (m, n, T, S, S, args.dtype, sec)}
            # the same port with one chunk per task on every host core (the reference itself is serial)
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, min(16, len(os.sched_getaffinity(0))))  # a one-GPU box's CPU share is 16 cores
            reps = [sample[i % m:i % m + 1] for i in range(2 * nthr)]
            w0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(O.bench_build, reps))  # ctypes releases the GIL inside the call
            wall = time.perf_counter() - w0
            cpu["all_cores"] = {"value": sum(r.size for r in reps) / wall, "unit": "cells/s", "cores": nthr,
                                "sample": "%d chunk builds on %d threads (%.1f s)" % (len(reps), nthr, wall)}

    if rank == 0:
        k_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = cells_per_step * esz + out_bytes  # every input cell read once, every output byte written once
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the committed PMC passes -- only valid for the workload they were taken on
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        default_workload = (args.workload, n, T, S, args.dtype, args.pad_elems) == ("config1", 1024, 32, 256, "i32", 0)
        if default_workload and os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "raster cells/s encoded (Snapshot+Log build)",
            "value": cells_per_step * world * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"i32": "int32", "i64": "int64", "f32": "f32->fixed(int32 arithmetic)", "f64": "f64->fixed(int32 arithmetic)"}[args.dtype],
            "data": "synthetic",
            "config": {"workload": workload,
                       "chunks_per_gpu": n, "k": 2, "device": name.decode(), "failed_tiles": bad,
                       "encoded_bytes_per_gpu": out_bytes, "snapshots": snaps, "bytes_verified_vs_oracle": verified},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k2r::k_encode<%d,false,%d>" % (S.bit_length() - 1, {"i32": 1, "f32": 2, "i64": 3, "f64": 4}[args.dtype]),
                         "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
