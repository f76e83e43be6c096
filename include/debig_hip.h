/*
 * debig_hip.h -- C-ABI of libdebigulator_hip.so: the MI355X (gfx950) batched
 * DEFLATE inflate / PNG de-filter path behind debigulator's header API.
 *
 * Everything here is plain C: pointers, sizes, POD structs.  No HIP or torch
 * types appear in a signature; `hip_stream` is an opaque hipStream_t passed as
 * void* (NULL = the default stream).
 *
 * What each entry point replaces in the reference (ArtOfBBQ/debigulator):
 *   debig_hip_inflate_batch      N x inflate()           src/inflate.h:51-60, src/inflate.c:786-1965
 *   debig_hip_inflate_batch_ex   the same, with the number of wavefronts per stream chosen
 *                                by the caller (few large streams vs thousands of small ones)
 *   debig_hip_png_defilter_batch the de-filter + palette loops of decode_png()
 *                                                         src/decode_png.c:1381-1564
 *   debig_hip_png_decode_fused_batch  inflate + de-filter of decode_png() in one kernel
 *                                                         src/decode_png.c:800-820 -> :1381-1564
 *   debig_hip_checksum_batch     update_crc() over PNG chunks, src/decode_png.c:313-333 (and the
 *                                gzip CRC-32 / zlib Adler-32 trailers the reference never checks)
 *   debig_hip_gather             the IDAT concatenation decode_png does in the caller's buffer,
 *                                src/decode_png.c:1285-1291, as a device-to-device copy list
 * The single-call drop-in API (inflate / decode_png / decode_gz with the
 * reference's own prototypes) is in inflate.h, decode_png.h, decode_gz.h next to
 * this file and is implemented on top of these batch calls; decode_gz.h also has
 * debig_gunzip_batch, an RFC 1952-complete gunzip that goes beyond the reference.
 */
#ifndef DEBIG_HIP_H
#define DEBIG_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One raw DEFLATE stream (what one reference inflate() call receives).  Offsets
 * are relative to the input / output arenas given to the batch call. */
typedef struct debig_stream {
    uint64_t in_off;  /* first byte of the compressed stream (any alignment)          */
    uint64_t in_len;  /* compressed_input_size (reference inflate.h:57)               */
    uint64_t out_off; /* start of the recipient (any alignment)                       */
    uint64_t out_cap; /* recipient_size (reference inflate.h:52); never written past  */
    /* decode_png() buffer-aliasing replay (reference quirk):
     * stream byte index aliased by the first scratch-table byte, and the
     * recipient size `est`.  p2_on = 0 for plain inflate()/decode_gz().          */
    int64_t p2_s0;
    uint64_t p2_est;
    uint32_t p2_on;
    uint32_t flags;   /* DEBIG_STREAM_* (0 = exactly the reference's inflate())         */
} debig_stream;

/* debig_stream.flags.  NO_REF_GATES: skip the reference's argument gates
 * (recipient_size < compressed_input_size, compressed_input_size < 5; quirk Q1) -- for
 * callers outside the reference's API whose input span is not "one stream", e.g. a gzip
 * member followed by further members (debig_gunzip_batch). */
#define DEBIG_STREAM_NO_REF_GATES 1u
/* A HINT for the host layers' choice of path, ignored by the kernels: the stream holds filtered image rows (a PNG IDAT
 * payload).  Such data is short matches a few bytes back, chained a dozen deep: a wavefront resolves it at the latency
 * of its LDS, and what helps is more wavefronts per stream -- chunk tasks (DEBIG_WAVES_CHUNKED) -- already from
 * 256 KiB of input per stream (one sample file of the reference, 1 MB of IDAT: 28.7 ms on a workgroup of eight
 * wavefronts, 4.9 ms in chunk tasks; text of the same size is faster on the workgroup).  decode_png sets it. */
#define DEBIG_STREAM_IMAGE_ROWS 2u
/* Test hook (fault injection, never set by the host layer): the multi-wavefront match resolve
 * gives up at its first idle poll instead of after DEBIG_RESOLVE_IDLE_BOUND of them, so the
 * DEBIG_E_INTERNAL path can be exercised on ordinary data. */
#define DEBIG_STREAM_FAULT_INJECT_IDLE 0x80000000u

/* status codes in debig_result.status (0 = success) */
enum {
    DEBIG_OK = 0,
    DEBIG_E_GATE_RECIPIENT_SMALL = 1, /* recipient_size < compressed_input_size (inflate.c:826) */
    DEBIG_E_GATE_INPUT_SHORT = 2,     /* compressed_input_size < 5 (inflate.c:836)              */
    DEBIG_E_STORED_NLEN = 3,          /* LEN != ~NLEN (inflate.c:949)                           */
    DEBIG_E_BAD_CODE_LENGTHS = 4,     /* code length >= table size (inflate.c:599)             */
    DEBIG_E_NO_CODE = 5,              /* bit pattern matches no code (inflate.c:465-473)        */
    DEBIG_E_DIST_SYMBOL = 6,          /* distance symbol > 29 (inflate.c:1809)                  */
    DEBIG_E_DIST_TOO_FAR = 7,         /* distance beyond start of output (inflate.c:1843)       */
    DEBIG_E_OUTPUT_FULL = 8,          /* output would exceed recipient_size (ref: overflow/assert) */
    DEBIG_E_LITLEN_286_287 = 9,       /* symbols 286/287 (ref: reads past its table)            */
    DEBIG_E_INTERNAL = 10             /* a kernel-internal guard tripped (bounded wait exhausted); never
                                         expected -- the stream is reported failed, not silently wrong */
};

typedef struct debig_result {
    uint64_t final_size; /* *final_recipient_size                                     */
    uint32_t good;       /* *out_good                                                 */
    uint32_t status;     /* DEBIG_E_* reason when good == 0                           */
    uint32_t final_set;  /* 0 when the reference leaves *final_recipient_size untouched */
    uint32_t n_blocks;
    uint32_t n_windows;  /* decode windows processed (perf counters, not API)          */
    uint32_t n_rounds;   /* speculative rounds summed over windows                     */
    /* shader-clock cycles per phase, only filled by -DDEBIG_PROFILE builds (else 0):
     * 0 stage input, 1 pass-1 scan rounds, 2 pass-2 decode, 3 LZ77 resolve, 4 flush,
     * 5 headers + tables, 6 whole stream, 7 far-match copy */
    uint32_t prof[8];
    /* bit position (relative to the stream's first byte) where decoding stopped: just past the
     * end-of-block code of the final block on success.  ceil(in_end_bits / 8) is where a
     * container's trailer starts (gzip CRC-32/ISIZE, zlib Adler-32). */
    uint64_t in_end_bits;
} debig_result;

/* Inflate n independent raw DEFLATE streams.  All pointers are DEVICE pointers
 * (streams/results included); the call is asynchronous on hip_stream.
 * Returns 0 or a hipError_t value. */
int debig_hip_inflate_batch(const void *d_in, void *d_out, const debig_stream *d_streams,
                            debig_result *d_results, uint32_t n, void *hip_stream);

/* Same, choosing how many 64-lane wavefronts cooperate on ONE stream:
 *   1          one wavefront per stream (best for many thousands of streams)
 *   2, 4, 8    one stream per workgroup of that many wavefronts (best for a few large
 *              streams, e.g. big PNG images; 8 uses half-size input segments)
 *   DEBIG_WAVES_LARGE4_SMALL1 / _SMALL2
 *              by stream: large ones (>= 256 KiB of input or >= 1 MiB of recipient) 4-wide,
 *              the others 1- or 2-wide, as two launches that run side by side (an internal
 *              HIP stream; hip_stream continues only after both)
 *   0          the library picks from n: n <= 256: 8; n <= 512: 4; n <= 768: 2; n <= 2048: DEBIG_WAVES_STRAND_PIPE;
 *              n <= 3072: DEBIG_WAVES_STRAND; else DEBIG_WAVES_SPLIT (never a mixed mode: stream sizes are in
 *              device memory; the host batch calls, which see the sizes, also take the pipeline for 257..768
 *              streams of 128 KiB of input or more on average).
 *              debig_hip_inflate_batch does this.
 *              The environment variable DEBIG_WAVES_PER_STREAM (1, 2, 4, 0x41, 0x42)
 *              replaces this choice, for measurements.
 * Results are identical for every choice.  Any other value: hipErrorInvalidValue. */
#define DEBIG_LARGE_IN_BYTES (256u << 10)  /* a stream is "large" from this much input ...   */
#define DEBIG_LARGE_OUT_BYTES (1u << 20)   /* ... or this much recipient (out_cap)            */
#define DEBIG_WAVES_AUTO 0u
#define DEBIG_WAVES_LARGE4_SMALL1 0x41u
#define DEBIG_WAVES_LARGE4_SMALL2 0x42u
/*   DEBIG_WAVES_SPLIT
 *              the throughput path for thousands of streams (what 0 picks for n > 1024): two
 *              kernels, one wavefront per stream each -- a scan kernel (block headers, tables,
 *              speculative Huffman scan; its decoded symbols go to a token workspace in HBM) and
 *              an LZ77 kernel (replays the tokens, resolves matches, writes the output) -- both
 *              at 3 wavefronts per SIMD instead of 2.  Needs device workspace
 *              (debig_hip_inflate_batch_ws; debig_hip_inflate_batch / _ex use a cached internal
 *              one of DEBIG_WORKSPACE_MB MiB, default 1024); a stream that does not fit its share
 *              is decoded by the one-kernel path in the same call.
 *              ORDER: one workgroup per stream is dealt to the shader engines by index, whatever it costs,
 *              so since round 4 the plan step also fixes the DISPATCH order: streams that look expensive
 *              (recipient larger than input + 64 bytes) first, stored / incompressible ones behind them, each
 *              class in the caller's order.  Stored and Huffman streams alternating in the descriptors
 *              (1.59 ms before, 1.8 x the sorted batch) now take what the sorted batch takes (0.91 vs 0.89 ms:
 *              bench.py reports both, roofline and roofline_interleaved).  What is left to the caller: batches
 *              whose expensive streams differ a lot among themselves (thumbnails beside full images): group
 *              them by size, or use DEBIG_WAVES_SPLIT_QUEUED. */
#define DEBIG_WAVES_SPLIT 0x10u
/*   DEBIG_WAVES_SPLIT_QUEUED
 *              DEBIG_WAVES_SPLIT for a batch whose ORDER mixes cheap and expensive streams (stored and
 *              Huffman streams alternating, thumbnails beside full images): workgroups that stay on
 *              the device and take streams from a queue instead of one workgroup per stream.  The
 *              hardware deals workgroups to its shader engines by index, whatever they cost, so with
 *              DEBIG_WAVES_SPLIT an alternating order can leave half the chip idle (4096 Huffman
 *              streams alternating with 4096 tiny ones: 1.67 ms, queued 0.98 ms, sorted by kind 0.89 ms).
 *              Never picked by 0: a batch sorted or grouped by kind / size is 2-4 % faster with
 *              DEBIG_WAVES_SPLIT.  DEBIG_SPLIT_WORKGROUPS overrides the number of resident workgroups. */
#define DEBIG_WAVES_SPLIT_QUEUED 0x11u
/*   DEBIG_WAVES_STRAND
 *              DEBIG_WAVES_SPLIT with the long-segment scan (csrc/inflate_strand_kernel.inc): a lane decodes a
 *              contiguous STRAND of a Huffman block as long as the block allows (a 64 KiB fixed-Huffman
 *              stream: one window of 64 strands) from a per-lane input ring in LDS, keeps its tokens in one
 *              pass, and only the lanes whose guessed start was wrong are decoded again, up to the point
 *              where they rejoin their first decode.  Same workspace, same results. */
#define DEBIG_WAVES_STRAND 0x12u
/*   DEBIG_WAVES_STRAND_PIPE
 *              DEBIG_WAVES_STRAND as a pipeline inside a workgroup of TWO wavefronts: one scans the stream, the other
 *              replays what the first has finished, record by record (a window, a stored block).  For batches that
 *              leave most SIMDs one or two wavefronts (a few hundred to about two thousand streams): a stream then
 *              takes max(scan, LZ77) instead of their sum.  Same workspace, same results. */
#define DEBIG_WAVES_STRAND_PIPE 0x13u
#define DEBIG_STRAND_MIN_STREAMS 768u  /* what 0 picks: up to here 2 wavefronts per stream ...            */
#define DEBIG_STRAND_MAX_STREAMS 3072u /* ... DEBIG_WAVES_STRAND up to here, DEBIG_WAVES_SPLIT beyond      */
#define DEBIG_STRAND_PIPE_MAX_STREAMS 2048u /* ... and up to here as a two-wavefront pipeline (DEBIG_WAVES_STRAND_PIPE) */
#define DEBIG_STRAND_PIPE_MEAN_IN_BYTES (128u << 10) /* host batch calls: 257..768 streams this long on average also take it */
/*   DEBIG_WAVES_CHUNKED
 *              a FEW LARGE streams (hundreds of big PNG images): every stream is cut at DEFLATE
 *              block boundaries into chunk tasks of 32..256 KiB of input, found by looking for
 *              dynamic block headers, and the tasks go through the scan / LZ77 kernels side by
 *              side; a task does not know the 32 KiB of output in front of it, so its matches
 *              are replayed against two synthetic histories and translated once the true window
 *              is known (csrc/inflate_chunk_kernel.inc).  Needs workspace:
 *              debig_hip_inflate_chunked_workspace_bytes(); callers with more data than
 *              workspace pass the batch in groups.  Streams the path cannot take (no dynamic
 *              blocks, a block longer than the workspace share, a failing stream, any doubt)
 *              are decoded by the one-kernel path in the same call.  Never picked by 0. */
#define DEBIG_WAVES_CHUNKED 0x20u
int debig_hip_inflate_batch_ex(const void *d_in, void *d_out, const debig_stream *d_streams,
                               debig_result *d_results, uint32_t n, uint32_t waves_per_stream,
                               void *hip_stream);

/* One-time set-up of the stream's device: the fixed-Huffman table images every inflate call reads
 * (three small allocations, three tiny kernels, one synchronisation of `hip_stream`).  Calls do it
 * lazily on first use; call it yourself BEFORE capturing a stream into a hipGraph -- allocations and
 * synchronisation are illegal during capture.  Thread safe, idempotent.  0 or a hipError_t. */
int debig_hip_init(void *hip_stream);

/* Concurrency: calls on different streams or from different host threads are safe.  Callers that
 * bring no workspace share ONE cached buffer per device: their groups of launches are serialised
 * on it (each group waits for the event of the group before it), so concurrent callers that want
 * overlap on the device should bring their own workspace.
 *
 * Same with caller-owned workspace for DEBIG_WAVES_SPLIT (no allocation inside the call once
 * debig_hip_init() has run on the device: safe to capture into a hipGraph).
 * debig_hip_inflate_workspace_bytes() is the size that lets ordinary
 * data through the scan/LZ77 pair (about 12 x the compressed bytes of the largest group of 16384
 * streams + 24 KiB per stream); less is legal and only sends more streams down the one-kernel
 * path.  d_workspace = NULL: the internal cached workspace.  The workspace holds no state between
 * calls. */
uint64_t debig_hip_inflate_workspace_bytes(uint64_t total_in_bytes, uint32_t n);
/* The same for a caller that also knows the recipients: total_out_cap = the sum of out_cap.  Highly compressible
 * streams (flat image areas: 30 KB for 4 MB, codes of one or two bits) write many more token units per compressed byte
 * than 12 x allows for; the plan step gives every stream a share by in_len + min(out_cap / 64, 4 in_len) + 2 KiB, and
 * this size adds the second term (at most 3/16 of total_out_cap), so that such streams are not handed back. */
uint64_t debig_hip_inflate_workspace_bytes_io(uint64_t total_in_bytes, uint64_t total_out_cap, uint32_t n);
/* workspace that lets DEBIG_WAVES_CHUNKED take every stream of a batch: total_out_bytes = the sum
 * of the recipients (out_cap), which should be close to the decoded sizes */
uint64_t debig_hip_inflate_chunked_workspace_bytes(uint64_t total_in_bytes, uint64_t total_out_bytes, uint32_t n);
int debig_hip_inflate_batch_ws(const void *d_in, void *d_out, const debig_stream *d_streams,
                               debig_result *d_results, uint32_t n, uint32_t waves_per_stream,
                               void *d_workspace, uint64_t workspace_bytes, void *hip_stream);

/* DEBIG_WAVES_SPLIT in two steps, for callers that inflate batches with the SAME descriptors again
 * and again (a decode loop over equal-sized buffers, a captured graph): the share of the workspace
 * every stream gets depends on the descriptors only, so it can be carved once.
 *   debig_hip_inflate_plan_ws     carves `d_workspace` for these n <= 16384 descriptors (one small
 *                                 kernel: what debig_hip_inflate_batch_ws does first on every call);
 *   debig_hip_inflate_planned_ws  scan + LZ77 (+ the one-kernel path for streams handed back) over a
 *                                 workspace that plan_ws carved for exactly these descriptors, this n
 *                                 and this workspace size and that nothing else has written since
 *                                 (the kernels leave the plan intact: call it any number of times).
 * Results are those of debig_hip_inflate_batch_ws.  n > 16384 or no workspace: hipErrorInvalidValue
 * (such batches go through the workspace group by group: use debig_hip_inflate_batch_ws). */
int debig_hip_inflate_plan_ws(const debig_stream *d_streams, uint32_t n, void *d_workspace, uint64_t workspace_bytes,
                              void *hip_stream);
int debig_hip_inflate_planned_ws(const void *d_in, void *d_out, const debig_stream *d_streams, debig_result *d_results,
                                 uint32_t n, void *d_workspace, uint64_t workspace_bytes, void *hip_stream);
/* the same with the dispatch named: DEBIG_WAVES_SPLIT (what debig_hip_inflate_planned_ws runs) or
 * DEBIG_WAVES_SPLIT_QUEUED (persistent workgroups over a work queue: batches whose order mixes cheap and
 * expensive streams); anything else: hipErrorInvalidValue */
int debig_hip_inflate_planned_ws_ex(const void *d_in, void *d_out, const debig_stream *d_streams,
                                    debig_result *d_results, uint32_t n, uint32_t waves_per_stream,
                                    void *d_workspace, uint64_t workspace_bytes, void *hip_stream);

/* One image for the de-filter kernel: the inflated scanline stream (filter byte
 * + w*bpp bytes per row) -> 4-channel RGBA. */
typedef struct debig_png_image {
    uint64_t stream_off; /* inflated stream, relative to d_streams_arena               */
    uint64_t rgba_off;   /* output, relative to d_rgba_arena; 4*w*h bytes              */
    uint64_t pal_off;    /* colour type 3: 768 bytes R[256] G[256] B[256], rel. to d_streams_arena */
    uint32_t width, height;
    uint32_t color_type; /* 6 (RGBA), 3 (palette), 2 (RGB)                                */
    uint32_t asserts_off;/* 0: a filter byte > 4 fails the image (reference default build) */
    /* colour type 2 only: replay_p3 = 1 reproduces the reference's output for RGB images bit
     * for bit (its RGB->RGBA expansion runs inside the row loop): rgba_off
     * must then hold the caller's PRIOR buffer contents and tmp_off a second 4*w*h byte
     * buffer (relative to d_rgba_arena).  replay_p3 = 0: spec-conforming RGB -> RGBA. */
    /* colour type 3 only: rows wider than 16384 pixels need width + 16 bytes of scratch at tmp_off
     * (relative to d_rgba_arena, 4-byte aligned, not 0): the index row handed from one band of 64
     * rows to the next; narrower palette images keep that row in LDS and ignore tmp_off. */
    uint64_t tmp_off;
    uint32_t replay_p3;
    uint32_t reserved;
} debig_png_image;

typedef struct debig_png_result {
    uint32_t good;
    uint32_t bad_row; /* first row whose filter byte was > 4 (when good == 0) */
} debig_png_result;

/* De-filter n inflated scanline streams into RGBA (device pointers, asynchronous on
 * hip_stream).  d_streams_arena must stay readable for 16 bytes past the end of every
 * stream (h * (w * bpp + 1) bytes): rows are fetched as aligned 16-byte pieces.
 * Up to 128 images: an image is spread over 16 / 8 / 4 / 2 workgroups (n <= 16 / 32 / 64 / 128; fewer when the device
 * does not hold them all at once), the bands of 64 rows handed from wavefront to wavefront through memory; images whose
 * workgroups turn out not to be resident together are decoded again by one workgroup inside the same call (never
 * failed).  That mode uses a per-device scratch of progress words shared by the callers of the device (calls are
 * ordered on it by events: such a call cannot be captured into a graph). */
int debig_hip_png_defilter_batch(const void *d_streams_arena, void *d_rgba_arena,
                                 const debig_png_image *d_images, debig_png_result *d_results,
                                 uint32_t n, void *hip_stream);

/* SURVEY.md 8(f) row 1 -- inflate AND de-filter n PNG images in ONE kernel launch (debig_png_fused_kernel):
 * replaces the pair src/decode_png.c:800-820 (inflate of the IDAT payload) -> src/decode_png.c:1381-1564 (the row
 * loops over the buffer it filled).  Stream i (d_streams[i]: compressed bytes in d_in, recipient inside
 * d_streams_arena) and image i (d_images[i]: stream_off == d_streams[i].out_off) belong together.  A workgroup owns an
 * image: one wavefront scans the DEFLATE stream, one replays it into the scanline stream, two de-filter bands of 64
 * rows as soon as their bytes are final -- read from L2, where the same CU has just put them; the scanline stream is
 * never read back from HBM, and the de-filter of an image no longer waits for the slowest inflate of the batch.
 * Results are those of debig_hip_inflate_batch_ws (d_results) followed by debig_hip_png_defilter_batch
 * (d_png_results), bit for bit: streams the scan hands back are decoded by debig_inflate_kernel and their images
 * de-filtered by the one-workgroup kernel inside the same call; colour type 2 images with replay_p3 go to the P3
 * kernel as always.  d_workspace / workspace_bytes: as debig_hip_inflate_batch_ws (NULL: the library's own).
 * Meant for hundreds to a few thousand images (a workgroup of four wavefronts and 50 KB of LDS per image). */
int debig_hip_png_decode_fused_batch(const void *d_in, void *d_streams_arena, const debig_stream *d_streams,
                                     debig_result *d_results, void *d_rgba_arena, const debig_png_image *d_images,
                                     debig_png_result *d_png_results, uint32_t n, void *d_workspace,
                                     uint64_t workspace_bytes, void *hip_stream);

/* A byte span of a device arena. */
typedef struct debig_span {
    uint64_t off;
    uint64_t len;
} debig_span;

/* CRC-32 (kind 0; PNG chunk / gzip convention) or Adler-32 (kind 1; zlib) of n spans, one
 * result word per span.  Replaces the per-byte update_crc loop the reference runs over every
 * PNG chunk (src/decode_png.c:313-333, :862-874) and provides what it never verifies (gzip
 * CRC32 trailer, zlib Adler-32: src/decode_gz.c:281-297, src/decode_png.c:393-395).
 * Device pointers, asynchronous on hip_stream. */
int debig_hip_checksum_batch(const void *d_arena, const debig_span *d_spans, uint32_t *d_out,
                             uint32_t n, uint32_t kind, void *hip_stream);

/* Copy n byte ranges between device arenas (any alignment): the IDAT concatenation of
 * decode_png (src/decode_png.c:1285-1291) done in HBM. */
typedef struct debig_copy {
    uint64_t src_off;
    uint64_t dst_off;
    uint64_t len;
} debig_copy;
int debig_hip_gather(const void *d_src_arena, void *d_dst_arena, const debig_copy *d_copies,
                     uint32_t n, void *hip_stream);

/* plain device-to-device helpers used by the host layer (no torch needed) */
int debig_hip_device_count(void);
int debig_hip_set_device(int dev);
int debig_hip_get_device(void); /* the calling thread's current device, -1 on error */
uint64_t debig_hip_mem_free(void); /* free device memory of the current device in bytes, 0 on error */
void *debig_hip_malloc(uint64_t bytes);
void debig_hip_free(void *p);
int debig_hip_memcpy_h2d(void *d, const void *h, uint64_t bytes, void *hip_stream);
int debig_hip_memcpy_d2h(void *h, const void *d, uint64_t bytes, void *hip_stream);
int debig_hip_memset(void *d, int v, uint64_t bytes, void *hip_stream);
int debig_hip_stream_sync(void *hip_stream);
/* page-locked host memory (staging arenas of the host-buffer batch calls) */
void *debig_hip_host_alloc(uint64_t bytes);
void debig_hip_host_free(void *p);
const char *debig_hip_error_string(int err);
/* kernel timing on the stream the kernels run on (hipEvent based) */
void *debig_hip_event_create(void);
int debig_hip_event_record(void *ev, void *hip_stream);
float debig_hip_event_elapsed_ms(void *start, void *stop); /* synchronises on stop */
int debig_hip_event_sync(void *ev);
void debig_hip_event_destroy(void *ev);

#ifdef __cplusplus
}
#endif
#endif
