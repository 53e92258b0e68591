"""Host logic: routing of dataset-level cubes to chunk-level sub-queries (dcdf_amd/raster.py) against a direct
transcription of the loops of superchunk.rs:589-633 (`subchunks_for`) and span.rs:190-216 (time segments)."""
import numpy as np

from dcdf_amd.raster import EncodedRaster


def loops(shape, tile, cs, cube):
    t0, t1, r0, r1, c0, c1 = cube
    nti, ntj = -(-shape[1] // tile), -(-shape[2] // tile)
    out = []
    if t1 <= t0 or r1 <= r0 or c1 <= c0:
        return out
    span, instant = t0 // cs, t0 % cs  # span.rs:272-274 find_span
    start, instants = 0, t1 - t0
    while start < instants:
        span_len = min(cs - instant, instants - start)  # span.rs:193
        for row in range(r0 // tile, (r1 - 1) // tile + 1):  # superchunk.rs:592-599
            top = max(row * tile, r0) - row * tile
            bottom = min(row * tile + tile, r1) - row * tile
            for col in range(c0 // tile, (c1 - 1) // tile + 1):
                left = max(col * tile, c0) - col * tile
                right = min(col * tile + tile, c1) - col * tile
                out.append(((span * nti + row) * ntj + col, instant, instant + span_len, top, bottom, left, right))
        instant = 0
        span += 1
        start += span_len
    return out


def test_split_matches_reference_loops():
    rng = np.random.default_rng(1)
    for shape, tile, cs in [((70, 768, 768), 256, 32), ((365, 4096, 4096), 256, 32), ((10, 17, 17), 4, 3), ((5, 9, 30), 16, 8)]:
        n_chunks = -(-shape[0] // cs) * -(-shape[1] // tile) * -(-shape[2] // tile)
        er = EncodedRaster(shape, [None] * n_chunks, tile, cs)
        n = 400
        t0 = rng.integers(0, shape[0], n); t1 = np.minimum(shape[0], t0 + rng.integers(0, 3 * cs, n))
        r0 = rng.integers(0, shape[1], n); r1 = np.minimum(shape[1], r0 + rng.integers(0, 3 * tile, n))
        c0 = rng.integers(0, shape[2], n); c1 = np.minimum(shape[2], c0 + rng.integers(0, 3 * tile, n))
        q = np.stack([t0, t1, r0, r1, c0, c1], axis=1)
        q[0] = (0, shape[0], 0, shape[1], 0, shape[2])
        sub = er.split(q)
        want = []
        for i, c in enumerate(q):
            want += [(i,) + s for s in loops(shape, tile, cs, tuple(int(x) for x in c))]
        assert [tuple(int(x) for x in row) for row in sub] == want
        grid = EncodedRaster.chunk_grid(shape, tile, cs)
        assert len(grid) == n_chunks
        for row in sub[:200]:
            g = grid[int(row[1])]
            assert er._origin(int(row[1])) == (g[0], g[2], g[4])
            assert row[3] <= g[1] - g[0] and row[5] <= g[3] - g[2] and row[7] <= g[5] - g[4]
