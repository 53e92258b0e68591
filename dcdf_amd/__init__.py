"""dcdf_amd -- MI355X-native Heuristic K^2-Raster chunk engine behind the reference's Chunk interface.

Only what the hot path needs: `chunk` (host mirror of dcdf::Chunk over the C ABI), `encoder` (device-resident
batch sessions), `synth` (deterministic synthetic rasters).  The compute lives in csrc/ (HIP, gfx950)."""
from .chunk import (Chunk, Cube, Rect, MMStruct3Build, build_batch, build_chunk, window, search, suggest_fraction,  # noqa: F401
                    get_batch, fill_cell_batch, fill_window_batch)
from .superchunk import Superchunk, SuperchunkBuild  # noqa: F401
from ._lib import DcdfError, LIB_PATH  # noqa: F401
