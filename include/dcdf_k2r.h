/* dcdf_k2r.h -- C ABI of the MI355X-native Heuristic K^2-Raster chunk engine.
 *
 * Drop-in boundary for ONE path of Arbol-Project/dcdf v0.2.0: `Chunk::build` and the
 * chunk-level queries.  Every entry point names the reference interface it replaces
 * (file:line relative to /root/reference/dcdf/src/).  A Rust `dcdf` fork binds these
 * with a plain `extern "C"` block; nothing here mentions torch.
 *
 * All encoded byte strings are bit-identical to the reference's `Chunk::write_to`
 * (chunk.rs:235-243).  Panics of the reference are negative return codes here.
 * The library needs a gfx950 GPU: every compute entry point returns
 * DCDF_ERR_NO_DEVICE when none is present -- there is no CPU fallback.
 */
#ifndef DCDF_K2R_H
#define DCDF_K2R_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MMEncoding byte codes, mmstruct.rs:36-43 */
enum { DCDF_I32 = 4, DCDF_I64 = 8, DCDF_F32 = 32, DCDF_F64 = 64 };

/* memory space of caller-owned buffers */
enum { DCDF_MEM_HOST = 0, DCDF_MEM_DEVICE = 1 };

/* return / status codes (0 = ok) */
enum {
    DCDF_OK = 0,
    DCDF_ERR_BAD_ARG = -1,       /* instants == 0, bad dtype, null pointer ...                     */
    DCDF_ERR_NONFINITE = -2,     /* to_fixed on +-inf                       fixed.rs:39-41         */
    DCDF_ERR_PRECISION = -3,     /* to_fixed precision loss, round=false    fixed.rs:47-59         */
    DCDF_ERR_OVERFLOW = -4,      /* fixed-point value does not fit i64      fixed.rs:65-70         */
    DCDF_ERR_BOUNDS = -5,        /* query outside the chunk                 mmarray.rs:218-229     */
    DCDF_ERR_TOO_MANY_LOGS = -6, /* unreachable through build (254 cap)     block.rs:27-32         */
    DCDF_ERR_FORMAT = -7,        /* malformed encoded chunk on open         chunk.rs:247-266       */
    DCDF_ERR_UNSUPPORTED = -8,   /* sidelen > 1024 (fused kernel: k = 2, sidelen 16..256) */
    DCDF_ERR_NO_DEVICE = -9,     /* no gfx950 device / HIP runtime failure                         */
    DCDF_ERR_NOMEM = -10,
    DCDF_ERR_CAPACITY = -11,     /* result buffer too small (search): *n holds the needed count    */
    DCDF_ERR_INTERNAL = -12,     /* an internal consistency guard of the kernels tripped (a bug)     */
    DCDF_ERR_HIP = -13,          /* a HIP runtime call failed (launch failure, fault ...): dcdf_last_hip_error() */
    DCDF_ERR_HIP_INVALID = -14   /* HIP rejected an argument (bad device pointer / value)                        */
};

/* One `Chunk::build` input: a borrowed strided 3-D view [instants, rows, cols]
 * (mmbuffer.rs:255-260; tile slices are non-contiguous, mmbuffer.rs:517-522).
 * Not mutated, no pointer retained after the call returns. */
typedef struct dcdf_tile_desc {
    const void* base;                      /* element [0,0,0]                                   */
    int32_t dtype;                         /* DCDF_I32 / I64 / F32 / F64                         */
    int32_t _pad0;
    int64_t stride_t, stride_r, stride_c;  /* in ELEMENTS                                        */
    uint32_t instants, rows, cols;
    uint8_t fractional_bits;               /* floats only (mmbuffer.rs:554-571)                  */
    uint8_t round;                         /* floats only: lossy rounding allowed (fixed.rs:46)  */
    uint8_t _pad1[2];
} dcdf_tile_desc;

/* One `MMStruct3Build` (mmstruct.rs:24-34) for a Subchunk: serialized bytes + counters. */
typedef struct dcdf_encoded {
    uint8_t* bytes;       /* == Chunk::write_to image (chunk.rs:235-243); library-owned host memory   */
    size_t len;           /* == Chunk::size()  (chunk.rs:272-277)                                      */
    uint32_t snapshots;   /* chunk.rs:93-94                                                            */
    uint32_t logs;
    int32_t status;       /* per-tile DCDF_* code; bytes == NULL when != 0                             */
    int32_t _pad;
    int64_t* minmax;      /* [instants][2] stored-value (min,max) per instant = root of each k2 tree;  */
                          /* what Superchunk::build recomputes at superchunk.rs:144 (mmbuffer.rs:366)  */
} dcdf_encoded;

/* ---- encode -------------------------------------------------------------------------------- */

/* Replaces `Chunk::build(buffer, shape, k)` (chunk.rs:42-96) called per tile from
 * superchunk.rs:169, for n independent tiles at once.  Returns 0 if the call itself ran
 * (inspect out[i].status per tile) or a negative code for a whole-call failure.
 * `mem` says where tiles[i].base lives.  *out is an array of n records; free with
 * dcdf_free_encoded(*out, n). */
int dcdf_chunk_build_batch(const dcdf_tile_desc* tiles, size_t n, int k, int mem, dcdf_encoded** out);

/* Single tile == batch of 1 (chunk.rs:42). */
int dcdf_chunk_build(const dcdf_tile_desc* tile, int k, int mem, dcdf_encoded** out);

void dcdf_free_encoded(dcdf_encoded* out, size_t n);

/* Device-resident encode session (what bench.py times: inputs already in HBM, outputs stay in HBM).
 * tiles[i].base must be device pointers.  `out_cap_per_tile` bytes of device output are reserved
 * per tile (0 = library default = raw tile bytes + 4 KiB). */
typedef struct dcdf_encoder dcdf_encoder;
int dcdf_encoder_create(const dcdf_tile_desc* tiles, size_t n, int k, size_t out_cap_per_tile, dcdf_encoder** enc);
/* Launches the encode kernels on the session's stream and waits.  kernel_ms (may be NULL) receives
 * the HIP-event time of the encode kernel alone. */
int dcdf_encoder_run(dcdf_encoder* enc, float* kernel_ms);
/* Per-tile results of the last run (host copies of the small tables; bytes stay on the device). */
int dcdf_encoder_result(dcdf_encoder* enc, size_t i, int32_t* status, uint64_t* len, uint32_t* snapshots,
                        uint32_t* logs, const uint8_t** device_bytes);
/* Copies tile i's encoded bytes to host memory `dst` (cap >= len). */
int dcdf_encoder_fetch(dcdf_encoder* enc, size_t i, uint8_t* dst, size_t cap);
/* Sum of len over tiles with status 0 (for the algorithmic-bytes figure). */
uint64_t dcdf_encoder_total_bytes(dcdf_encoder* enc);
/* The final host-side gather of one GPU's encoded buffers (superchunk.rs:183-235 consumes them in tile order): tile i's
 * bytes land at dst + offsets[i] (16-byte aligned starts, lens[i] bytes; 0 for failed tiles), packed on the device and
 * copied back in one transfer; minmax (may be NULL) receives the per-(tile, instant) stored (min,max) pairs in tile
 * order, 2 * instants words per tile (what Superchunk::build recomputes at superchunk.rs:144).
 * dcdf_encoder_gather_size tells how large dst (bytes) and minmax (int64 words) must be. */
int dcdf_encoder_gather_size(dcdf_encoder* enc, uint64_t* packed_bytes, uint64_t* minmax_words);
int dcdf_encoder_gather(dcdf_encoder* enc, uint8_t* dst, size_t cap, uint64_t* offsets, uint64_t* lens, int64_t* minmax);
/* Content addressing of the stored chunk objects, computed where the encoded bytes lie (HBM): digests[32*i ..] =
 * SHA-256 of the object the reference would store for tile i -- u16 0xDCE0, u32 1, NODE_MMSTRUCT3 (2),
 * NODE_SUBCHUNK (4), big-endian, then the Chunk::write_to bytes (resolver.rs:17-18,126-138; mmstruct.rs:215-218) --
 * i.e. the digest inside its CID: CIDv1, codec 0x12, multihash sha2-256 (testing.rs:172-183).  Zeros for tiles that
 * failed.  `digests` is host memory; call after dcdf_encoder_run. */
int dcdf_encoder_object_sha256(dcdf_encoder* enc, uint8_t* digests, float* kernel_ms);
void dcdf_encoder_destroy(dcdf_encoder* enc);

/* ---- superchunk assembly (the caller of Chunk::build) ---------------------------------------- */

/* One object the reference would hand to `resolver.save` (resolver.rs:126-138): the 7-byte header (u16 0xDCE0, u32 1,
 * u8 node type) + the node, and its CID as MemoryMapper names it (testing.rs:172-183: CIDv1, codec 0x12, sha2-256). */
typedef struct dcdf_stored_object {
    uint8_t cid[36];
    uint8_t* bytes;
    size_t len;
} dcdf_stored_object;

/* `MMStruct3Build` of a Superchunk (mmstruct.rs:24-34) plus the objects to store: framed sub-chunks (mmstruct.rs:215-218),
 * nested Superchunk nodes, `Links` nodes (links.rs:65-76), de-duplicated by CID like `external_references`
 * (superchunk.rs:199-235), in save order; the LAST object is the Superchunk node itself (superchunk.rs:672-706), which
 * `Superchunk::build` leaves to its caller to save. */
typedef struct dcdf_superchunk {
    dcdf_stored_object* objects;
    size_t n_objects;
    uint64_t size;
    uint32_t elided, local, external, snapshots, logs;
} dcdf_superchunk;

/* Replaces `Superchunk::build(resolver, buffer, shape, levels, k)` (superchunk.rs:88-270): uniform-tile elision from
 * per-tile per-instant (min,max) computed on the device (mmbuffer.rs:366-499), per-tile fractional bits
 * (superchunk.rs:167), one batched Chunk::build launch per level, nested superchunks, the reference table, the
 * instant-major min / max Dacs, Links, object framing and CIDs.  `buffer` = the whole [instants, rows, cols] view
 * (fractional_bits / round as the caller's MMBuffer3 carries them); sum(levels) must equal ceil(log_k(max(rows, cols))). */
int dcdf_superchunk_build(const dcdf_tile_desc* buffer, const uint32_t* levels, size_t n_levels, int k, int mem,
                          dcdf_superchunk** out);
void dcdf_free_superchunk(dcdf_superchunk* s);

/* ---- query --------------------------------------------------------------------------------- */

typedef struct dcdf_chunk dcdf_chunk; /* an opened, device-resident encoded chunk */

/* geom::Cube (geom.rs:71-103): half-open bounds; reversed bounds are swapped as in the reference. */
typedef struct dcdf_cube {
    uint32_t start, end, top, bottom, left, right;
} dcdf_cube;

/* Replaces `Chunk::read_from` (chunk.rs:247-266) + upload.  `bytes` is host memory.  The stream is validated structurally
 * (DCDF_ERR_FORMAT otherwise) and copied to the device; for k = 2 chunks of sidelen 32..256 a 24-byte entry per 16 x 16
 * square and instant is built next to it on the device (where the query walks of that square start, and the square's value
 * range, which search prunes with). */
int dcdf_chunk_open(const uint8_t* bytes, size_t len, dcdf_chunk** h);
void dcdf_chunk_close(dcdf_chunk* h);
/* Chunk::shape (chunk.rs:119-123), encoding / fractional_bits (chunk.rs:33-38), block count. */
int dcdf_chunk_info(const dcdf_chunk* h, uint32_t shape[3], int32_t* encoding, uint32_t* fractional_bits,
                    uint32_t* n_blocks);

/* Byte offsets of the instants' Snapshots / Logs inside the chunk (off[instants + 1], the last = the chunk's length) and,
 * per instant, the instant of its block's snapshot (may be NULL): the encoded bytes a decode of instant i can touch are
 * [off[i], off[i + 1]) and the snapshot's range -- the decode path's algorithmic bytes. */
int dcdf_chunk_instant_layout(const dcdf_chunk* h, uint64_t* off, uint32_t* snapshot_of);

/* Replaces `Chunk::get` (chunk.rs:127-131): stored i64 value (fixed-point for float chunks). */
int dcdf_chunk_get(const dcdf_chunk* h, uint32_t instant, uint32_t row, uint32_t col, int64_t* out);
/* Replaces `Chunk::fill_cell` (chunk.rs:135-148): out[end-start] stored i64 values (host). */
int dcdf_chunk_fill_cell(const dcdf_chunk* h, uint32_t start, uint32_t end, uint32_t row, uint32_t col, int64_t* out);
/* Replaces `Chunk::fill_window` (chunk.rs:152-158) with MMBuffer3::set conversion (mmbuffer.rs:292-299,
 * 505,525,560,622): writes [end-start][bottom-top][right-left] into the caller's typed strided HOST array. */
int dcdf_chunk_fill_window(const dcdf_chunk* h, const dcdf_cube* cube, void* out, int32_t out_dtype, int64_t stride_t,
                           int64_t stride_r, int64_t stride_c);
/* Replaces `Chunk::iter_search` (chunk.rs:213-229, 336-383): (instant,row,col) triples, sorted.
 * cap = capacity of `out` in triples; *n = number found (DCDF_ERR_CAPACITY if > cap). */
int dcdf_chunk_search(const dcdf_chunk* h, const dcdf_cube* cube, int64_t lower, int64_t upper, uint32_t* out,
                      size_t cap, size_t* n);

/* Batched queries against many opened chunks (BASELINE config 5).  All arrays are host memory.
 * fill_window: query q writes its window, dense row-major i64 stored values, at out + out_offset[q]. */
int dcdf_query_fill_window_batch(dcdf_chunk* const* chunks, const dcdf_cube* cubes, size_t nq, int64_t* out,
                                 const uint64_t* out_offset, float* kernel_ms);
/* search: per-query counts in counts[q]; triples of query q at out + 3*offsets[q] (offsets = exclusive
 * prefix of counts, written by the call); cap in triples. */
int dcdf_query_search_batch(dcdf_chunk* const* chunks, const dcdf_cube* cubes, const int64_t* lower,
                            const int64_t* upper, size_t nq, uint32_t* out, size_t cap, uint64_t* counts,
                            uint64_t* offsets, float* kernel_ms);

/* The same with a typed result and a result that may stay on the device: out_dtype = DCDF_I32 / I64 / F32 / F64 is the
 * element type written (MMBuffer3::set, mmbuffer.rs:292-299: integers as stored, floats through from_fixed, fixed.rs:81-86);
 * out_mem = DCDF_MEM_DEVICE: `out` is device memory, the decode kernel writes window q at out + out_offset[q] (ELEMENTS of
 * out_dtype) itself and nothing crosses PCIe.  An int32 result of int32 chunks moves half the bytes of the i64 form. */
int dcdf_query_fill_window_batch_typed(dcdf_chunk* const* chunks, const dcdf_cube* cubes, size_t nq, void* out,
                                       int32_t out_dtype, int out_mem, const uint64_t* out_offset, float* kernel_ms);
/* dcdf_query_search_batch whose triples may stay on the device (out_mem = DCDF_MEM_DEVICE: `out` is device memory of `cap`
 * triples); counts / offsets are always host arrays. */
int dcdf_query_search_batch_mem(dcdf_chunk* const* chunks, const dcdf_cube* cubes, const int64_t* lower, const int64_t* upper,
                                size_t nq, uint32_t* out, size_t cap, int out_mem, uint64_t* counts, uint64_t* offsets,
                                float* kernel_ms);
/* Many point / cell-series queries in ONE launch -- what Superchunk::get / fill_cell (superchunk.rs:313-400) and the
 * Span / Dataset layers above them route to their chunks.  points = n x {instant, row, col}; out[i] = the stored i64 value
 * of point i of chunks[i] (Chunk::get, chunk.rs:127-131).  cells = n x {start, end, row, col}; series i goes to
 * out + out_offset[i] (Chunk::fill_cell, chunk.rs:135-148).  out_mem as above. */
int dcdf_query_get_batch(dcdf_chunk* const* chunks, const uint32_t* points, size_t n, int64_t* out, int out_mem,
                         float* kernel_ms);
int dcdf_query_fill_cell_batch(dcdf_chunk* const* chunks, const uint32_t* cells, size_t n, int64_t* out,
                               const uint64_t* out_offset, int out_mem, float* kernel_ms);
/* Opens n chunks at once.  mem = DCDF_MEM_HOST: as n calls of dcdf_chunk_open.  mem = DCDF_MEM_DEVICE: bytes[i] are DEVICE
 * pointers, e.g. dcdf_encoder_result's `device_bytes` -- the streams are packed into one slab, parsed on the device (one
 * thread per chunk, bounds-checked) and the side-16 tables of all their instants are built by one launch; nothing is copied
 * to the host but the per-chunk metadata.  Device input gets dcdf_chunk_open's structural validation as well, on the device (a wave per
 * instant: popcounts against the Dac and eqB lengths, rank indexes).  status (may be NULL): one code per chunk;
 * out[i] = NULL where it is not 0.  Every handle is closed with dcdf_chunk_close; the slab goes with the last one. */
int dcdf_chunk_open_batch(const uint8_t* const* bytes, const uint64_t* lens, size_t n, int mem, dcdf_chunk** out,
                          int32_t* status);

/* ---- a whole raster of chunks: the routing of the layers above ------------------------------------------------------- */
/* A variable as dcdf lays it out: time segments of chunk_size instants (Variable::append, dataset.rs:838), each cut into
 * tile x tile sub-arrays (Superchunk::build, superchunk.rs:127-181); chunks[(segment * ntiles_r + ti) * ntiles_c + tj], every chunk
 * with the shape its place gives it.  The handle keeps the chunk table on the device.  Lifetime: the raster refers to the chunks'
 * device buffers; it shares the ownership of the slab of chunks opened by dcdf_chunk_open_batch (closing such a chunk first is
 * fine), chunks opened one by one (dcdf_chunk_open) must stay open until dcdf_raster_destroy. */
typedef struct dcdf_raster dcdf_raster;
int dcdf_raster_create(dcdf_chunk* const* chunks, size_t n_chunks, const uint32_t shape[3], uint32_t tile, uint32_t chunk_size,
                       dcdf_raster** out);
void dcdf_raster_destroy(dcdf_raster* r);
/* fill_window of nq dataset-level cubes (raster coordinates): each is split at segment / tile boundaries where Span::fill_window
 * (span.rs:190-216) and Superchunk::subchunks_for (superchunk.rs:589-633) split it, every piece is decoded by the same launch
 * straight into its place in the window; window q is dense [instants][rows][cols] of out_dtype at out + out_offset[q] (elements),
 * host or device memory.  k * k > 64 chunks: DCDF_ERR_UNSUPPORTED (use the per-chunk entry points). */
int dcdf_raster_fill_window_batch(const dcdf_raster* r, const dcdf_cube* cubes, size_t nq, void* out, int32_t out_dtype,
                                  int out_mem, const uint64_t* out_offset, float* kernel_ms);
/* search of nq dataset-level cubes: (instant, row, col) triples in RASTER coordinates (span.rs:231-270 adds the segment offset,
 * superchunk.rs:516-585 the tile origin); query q's triples are out[3 * offsets[q] .. + 3 * counts[q]), ordered by piece
 * (segment, tile row, tile col), sorted inside a piece. */
int dcdf_raster_search_batch(const dcdf_raster* r, const dcdf_cube* cubes, const int64_t* lower, const int64_t* upper, size_t nq,
                             uint32_t* out, size_t cap, int out_mem, uint64_t* counts, uint64_t* offsets, float* kernel_ms);

/* ---- misc ---------------------------------------------------------------------------------- */
/* fixed.rs:96-159 + mmbuffer.rs:596-613: per-tile suggest_fraction on the device; out_round = 1 for
 * Fraction::Round.  Host or device data per `mem`. */
int dcdf_suggest_fraction(const dcdf_tile_desc* tile, int mem, int32_t* out_round, int32_t* out_bits);

/* Bench/test utility: fills dst[(t1-t0)][(r1-r0)][(c1-c0)] (DEVICE memory, dense) with the deterministic
 * synthetic raster of SURVEY.md section 8(d) (same integers as dcdf_amd/synth.py; costab_host = its 1024-entry
 * cosine table).  DCDF_I32 -> v, DCDF_I64 -> 2*v+1. */
int dcdf_synth_fill(void* dst_device, int32_t dtype, uint64_t seed, int64_t t0, int64_t t1, int64_t r0, int64_t r1,
                    int64_t c0, int64_t c1, const int32_t* costab_host);

/* Profiling utility: reads n_tiles dense [instants,256,256] int32 tiles (DEVICE memory) once with the encoder's
 * load pattern, so rocprofv3's FETCH_SIZE can be calibrated against a known byte count (MI355X_MICROARCH.md). */
int dcdf_calib_read(const void* tiles_device, uint32_t n_tiles, uint32_t instants);

/* Device memory for callers that have no HIP binding of their own (the ctypes mirror, tests, tools): hipMalloc /
 * hipFree / hipMemcpy (to_device != 0: host -> device, else device -> host). */
int dcdf_device_alloc(size_t bytes, void** out);
int dcdf_device_free(void* p);
int dcdf_device_copy(void* dst, const void* src, size_t bytes, int to_device);

/* The library keeps a few large device blocks between calls (an appending caller asks for the same sizes slice after
 * slice; hipMalloc / hipFree of gigabytes cost tens of milliseconds).  It gives them back by itself when one of its own
 * allocations fails; a caller that shares the device with another allocator can return them at any time: the bytes freed
 * are stored to *freed_bytes (may be NULL).  Environment K2R_POOL=0 disables the pool. */
int dcdf_device_pool_trim(uint64_t* freed_bytes);

const char* dcdf_strerror(int code);
/* "gfx950 <device name>, <CUs> CUs" or NULL when no device. */
const char* dcdf_device_name(void);
int dcdf_abi_version(void);
/* The raw hipError_t of the last HIP failure seen by this thread (0 = none), for DCDF_ERR_HIP / _HIP_INVALID /
 * _NO_DEVICE / _NOMEM returns. */
int dcdf_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DCDF_K2R_H */
