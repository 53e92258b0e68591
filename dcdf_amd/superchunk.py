"""Host-side mirror of `Superchunk::build` (dcdf/src/superchunk.rs:88-270) over the MI355X library: the caller of the chunk
path.  Tiling, elision and the reference table are control flow; the per-tile (min, max), the fractional bits, every
Chunk::build, the min / max Dacs and the CIDs are computed on the GPU (dcdf_superchunk_build)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .chunk import _desc


class SuperchunkBuild:  # MMStruct3Build (mmstruct.rs:24-34) of a Superchunk, plus what resolver.save would have stored
    def __init__(self, objects, root_cid, size, elided, local, external, snapshots, logs):
        self.objects = objects      # {cid bytes: stored object bytes}, sub-objects and the Superchunk node itself
        self.cid = root_cid         # CID of the Superchunk node
        self.data = objects[root_cid]
        self.size, self.elided, self.local, self.external = size, elided, local, external
        self.snapshots, self.logs = snapshots, logs


class Superchunk:
    @staticmethod
    def build(buffer, levels, k=2, fractional_bits=0, round=False):
        """superchunk.rs:88.  `buffer`: ndarray[instants, rows, cols]; `levels`: the k2_levels of the variable."""
        a = np.asarray(buffer)
        d = _desc(a, fractional_bits, round)
        lv = (C.c_uint32 * len(levels))(*[int(x) for x in levels])
        out = C.POINTER(L.SuperchunkBuild)()
        L.check(L.lib().dcdf_superchunk_build(C.byref(d), lv, C.c_size_t(len(levels)), int(k), L.MEM_HOST, C.byref(out)),
                "Superchunk::build")
        owner = _Owner(out)  # frees the C result when the last view of it is gone
        s = out.contents
        objects = {}
        cid = None
        for i in range(s.n_objects):
            o = s.objects[i]
            cid = bytes(o.cid)
            if o.len < _VIEW_FROM:
                objects[cid] = C.string_at(o.bytes, o.len)
            else:
                # the sub-chunk objects are the bulk (hundreds of MB per slice): read-only views of the library's buffers, no copy
                arr = (C.c_uint8 * o.len).from_address(C.addressof(o.bytes.contents))
                arr._owner = owner
                objects[cid] = memoryview(arr).cast("B").toreadonly()
        return SuperchunkBuild(objects, cid, s.size, s.elided, s.local, s.external, s.snapshots, s.logs)


_VIEW_FROM = 1 << 16


class _Owner:
    def __init__(self, out):
        self._out = out

    def __del__(self):
        try:
            L.lib().dcdf_free_superchunk(self._out)
        except Exception:
            pass
