#!/usr/bin/env python3
"""Audit of the built encoder objects (dcdf_amd/csrc/_build/enc_L*.o): extracts the gfx950 code object of each and fails
if a kernel keeps more than a few spill slots in scratch memory or reaches LDS through FLAT instructions.

Why: when a per-block loop of the encoder is not unrolled, the execution context (with its pointer to the LDS state) stays
in scratch, every LDS access becomes a FLAT instruction, and a 12/16-byte FLAT access landing in LDS at 4-byte alignment
raises HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  Run by `make check` and by the CPU test-suite.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAX_SCRATCH = 256  # bytes per lane (register spills only)


def audit(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}", "--unbundle"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
        dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)
    scratch = max(int(x) for x in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", notes))
    lds = max(int(x) for x in re.findall(r"\.group_segment_fixed_size:\s*(\d+)", notes))
    n_ds = len(re.findall(r"\sds_", dis))
    wide_flat = len(re.findall(r"\sflat_(?:load|store)_dwordx[34]\s", dis))
    return scratch, lds, n_ds, wide_flat


def audit_kernels(obj):
    """Per kernel of one object: (name, LDS bytes, scratch bytes/lane, ds_ count, flat_ count, 12/16-byte flat count)."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}", "--unbundle"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
        dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)
    meta = {m.group(2): (int(m.group(1)), int(m.group(3))) for m in re.finditer(
        r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+)", notes, re.S)}
    cnt, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            cnt.setdefault(cur, [0, 0, 0])
        elif cur is not None:
            cnt[cur][0] += bool(re.search(r"\sds_", line))
            cnt[cur][1] += bool(re.search(r"\sflat_(load|store|atomic)", line))
            cnt[cur][2] += bool(re.search(r"\sflat_(?:load|store)_dwordx[34]\s", line))
    return [(k, lds, scr) + tuple(cnt.get(k, [0, 0, 0])) for k, (lds, scr) in sorted(meta.items())]


# the other kernels of the library (query, universal encoder, superchunk assembly, hashing, float ingestion): the same
# failure mode -- LDS state behind a pointer the compiler lost track of, reached through FLAT instructions, a 12/16-byte
# one of which faults when it lands in LDS at 4-byte alignment -- is excluded per kernel: a kernel with LDS state must reach
# it with ds_ instructions, and must not keep more than MAX_SCRATCH_OTHER bytes of scratch per lane (the universal encoder's
# per-thread level cursors are the largest today, 560 B).
OTHER_OBJECTS = ["k2r_query.o", "k2r_generic.o", "k2r_superchunk.o", "k2r_cid.o", "k2r_suggest.o", "k2r_synth.o"]
MAX_SCRATCH_OTHER = 1024


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dcdf_amd", "csrc", "_build")
    objs = sorted(glob.glob(os.path.join(root, "enc_L*.o")))
    if not objs:
        print("no encoder objects under", root)
        return 1
    bad = 0
    for o in objs:
        scratch, lds, n_ds, wide_flat = audit(o)
        # LDS state of tens of KB and (almost) no ds_ instruction = LDS reached through FLAT
        ok = scratch <= MAX_SCRATCH and n_ds >= 100
        print(f"{os.path.basename(o):22s} scratch {scratch:5d} B/lane  LDS {lds:6d} B  ds_ {n_ds:5d}  flat x3/x4 {wide_flat:4d}  {'ok' if ok else 'BAD'}")
        bad += 0 if ok else 1
    for name in OTHER_OBJECTS:
        o = os.path.join(root, name)
        if not os.path.exists(o):
            print("missing", o)
            bad += 1
            continue
        for k, lds, scr, n_ds, n_flat, wide in audit_kernels(o):
            lds_via_flat = lds >= 256 and n_ds == 0 and n_flat > 0
            ok = scr <= MAX_SCRATCH_OTHER and not lds_via_flat and not (lds > 0 and n_ds == 0 and wide > 0)
            print(f"{name:18s} {k[:48]:48s} scratch {scr:5d}  LDS {lds:6d}  ds_ {n_ds:4d}  flat {n_flat:4d}  flat x3/x4 {wide:3d}  {'ok' if ok else 'BAD'}")
            bad += 0 if ok else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
