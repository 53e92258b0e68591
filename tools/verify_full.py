#!/usr/bin/env python3
"""Whole-workload parity evidence: every chunk of configs[1] -- 1024 independent [32,256,256] chunks,
seed 0xDCDF0002 + c -- encoded on the GPU and compared byte for byte with the CPU oracle (run on all host cores, one
chunk per task), plus the SHA-256 of the concatenation of all encoded chunks from both sides.  Untimed; the oracle
is only the checker here."""
import argparse
import hashlib
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=1024)
    ap.add_argument("--dtype", choices=["i32", "i64"], default="i32")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    from dcdf_amd import _lib as L
    from dcdf_amd.encoder import Encoder, synth_fill

    n, T, S = args.chunks, 32, 256
    tdt, code = (torch.int32, L.DCDF_I32) if args.dtype == "i32" else (torch.int64, L.DCDF_I64)
    flat = torch.empty((n * T * S * S,), dtype=tdt, device="cuda")
    data = [flat[c * T * S * S:(c + 1) * T * S * S].view(T, S, S) for c in range(n)]
    for c in range(n):
        synth_fill(data[c].data_ptr(), code, 0xDCDF0002 + c, 0, T, 0, S, 0, S)
    torch.cuda.synchronize()
    enc = Encoder([(d.data_ptr(), code, (S * S, S, 1), (T, S, S)) for d in data], k=2)
    enc.run()
    gpu = [enc.fetch(c) for c in range(n)]
    # content addressing on the device: SHA-256 of (object header + chunk bytes) for every chunk, vs hashlib
    dig, sha_ms = enc.object_sha256()
    hdr = bytes([0xDC, 0xE0, 0, 0, 0, 1, 2, 4])
    sha_bad = sum(1 for c in range(n) if hashlib.sha256(hdr + gpu[c]).digest() != dig[c].tobytes())

    def ref(c):
        return O.chunk_build(data[c].cpu().numpy())

    with ThreadPoolExecutor(args.threads) as ex:
        cpu = list(ex.map(ref, range(n)))
    bad = [c for c in range(n) if gpu[c] != cpu[c]]
    hg, hc = hashlib.sha256(), hashlib.sha256()
    for c in range(n):
        hg.update(gpu[c])
        hc.update(cpu[c])
    print(json.dumps({"workload": "configs[1], %d x [32,256,256] %s" % (n, args.dtype), "chunks_compared": n,
                      "chunks_differing": len(bad), "encoded_bytes": sum(map(len, gpu)),
                      "sha256_gpu": hg.hexdigest(), "sha256_oracle": hc.hexdigest(),
                      "object_sha256_on_device": {"digests_differing_from_hashlib": sha_bad, "kernel_ms": sha_ms,
                                                  "GB_per_s": sum(map(len, gpu)) / (sha_ms * 1e-3) / 1e9}}))
    sys.exit(1 if bad or sha_bad or hg.hexdigest() != hc.hexdigest() else 0)


if __name__ == "__main__":
    main()
