/*
 * awpu_hip.h -- C ABI of libawpu_hip.so, the MI355X (gfx950) delay-and-sum heatmap engine
 * that replaces the body of the reference's MIMO sweep behind its aw_processing_unit API.
 *
 * The reference (acoustic-warfare/beamforming-lk @ 2024_08_07) has no FFI or plugin
 * registry: the seam is the C++ class boundary AWProcessingUnit -> Worker (MIMOWorker).
 * Each entry point below names the reference interface it replaces (file:line relative
 * to the reference tree).  The reference-side binding (a MIMOWorker whose update() calls
 * awpu_hip_process) is shown in INTEGRATION.md and implemented in
 * beamforming-lk_amd/host/mimo_worker_hip.{h,cpp}.
 *
 * Conventions
 *   - plain C: opaque handle, plain pointers and sizes, no C++/torch types.
 *   - every function returns an awpu_status (0 = OK, negative = error); nothing throws
 *     across the boundary.  The reference's own convention is int 0 / -1
 *     (src/fpga/pipeline.h:57-77) and bool for start/stop.
 *   - the caller owns every host pointer; the library copies on call and owns all device
 *     memory.  `power` buffers are caller-allocated.
 *   - one handle per worker; a handle is used by one thread at a time; handles are
 *     independent (one HIP stream each), like one MIMOWorker thread per AWPU
 *     (src/dsp/mimo.cpp:12).
 *   - there is NO CPU fallback: without a usable HIP device awpu_hip_create fails with
 *     AWPU_ERR_NO_DEVICE.
 */
#ifndef AWPU_HIP_H
#define AWPU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWPU_HIP_ABI_VERSION 4

/* compile-time constants of the reference */
#define AWPU_N_SAMPLES 256 /* src/fpga/streams.hpp:28  N_SAMPLES */
#define AWPU_HIST 1024     /* src/fpga/streams.hpp:32  N_ITEMS_BUFFER = PAGE_SIZE/4 */
#define AWPU_ELEMENTS 64   /* src/geometry/antenna.h:20 ELEMENTS */
#define AWPU_MAX_DEVICES 8 /* GPUs one handle can spread its pixels over (one node) */

typedef enum {
    AWPU_OK = 0,
    AWPU_ERR_INVALID = -1,   /* bad argument / bad configuration */
    AWPU_ERR_NO_DEVICE = -2, /* no HIP device, or not gfx950 */
    AWPU_ERR_HIP = -3,       /* a HIP runtime call failed (see awpu_hip_last_error) */
    AWPU_ERR_STATE = -4,     /* call order: table or mic list not set */
    AWPU_ERR_RANGE = -5,     /* a table entry would read outside the frame history */
    AWPU_ERR_NOMEM = -6
} awpu_status;

/* interpolation of delay(): src/dsp/delay.cpp */
typedef enum {
    AWPU_INTERP_LERP = 0, /* :16-26, the shipped (-mavx2) variant */
    AWPU_INTERP_FIR8 = 1  /* :31-40, 8-tap table variant (needs awpu_hip_set_fir_table) */
} awpu_interp;

/* arithmetic of the sweep */
typedef enum {
    /* THE DEFAULT.  fp32, the reference's operation order per sample (sub, fma, add) and mic order
     * s = 0..usable-1: the pre-epilogue sums are bit-identical to delay.cpp:19-25, the powers within 1e-5 of
     * the reference on ANY input (DC-biased included) */
    AWPU_MATH_F32_EXACT = 0,
    /* opt-in.  fp32, re-ordered (stencil first, two FMAs per sample, shared integer-delay sums): ~1.35 x the rate of
     * EXACT on batches; within 1e-5 of the reference on zero-mean input only (on DC-biased input it is closer to the exact
     * sums than the reference is, and for that reason up to 1e-4 .. 1e-3 from the REFERENCE) */
    AWPU_MATH_F32_FAST = 1,
    /* EXACT's structure with the running sum of every sample KEPT in bf16 (rounded to nearest even after
     * every mic; the interpolation term and the epilogue stay fp32).  Not in the reference and not for
     * production: it exists so that BASELINE configs[4] ("bf16 vs fp32 accumulator") is measured on the
     * device -- per-pixel power moves by about 1e-2 relative, and gfx950 has no packed bf16 add, so it is
     * also slower than the fp32 accumulator */
    AWPU_MATH_BF16_ACC = 2
} awpu_math;

/* which sweep kernel a launch ran (awpu_hip_stats.kernel_variant): the dispatcher picks by math mode, batch, grid and
 * table statistics */
typedef enum {
    AWPU_KERNEL_NONE = 0,
    AWPU_KERNEL_QUAD = 1,             /* das_quad_kernel: batches, four vertically adjacent pixels share their integer-delay sum */
    AWPU_KERNEL_PAIR = 2,             /* das_pair_kernel: batches, pixel pairs share sample reads */
    AWPU_KERNEL_PAIR_STATIONARY = 3,  /* das_pair_stationary_kernel: batches of small arrays, every mic's window resident */
    AWPU_KERNEL_QUADH = 4,            /* das_quadh_kernel behind pack_halves_kernel: single frames */
    AWPU_KERNEL_QUADH_STATIONARY = 5, /* das_quadh_stationary_kernel: single frames of small arrays (the reference's own shape) */
    AWPU_KERNEL_SINGLE_DB = 6,        /* das_fast_db_kernel: single frames without a row length */
    AWPU_KERNEL_SINGLE_SMALL = 7,     /* das_fast_kernel: small grids */
    AWPU_KERNEL_FIR8_PLANES = 8,      /* das_fir8_plane_kernel */
    AWPU_KERNEL_FIR8 = 9,             /* das_fir8_kernel: FIR8 on small launches, or in the reference's tap order (exact math) */
    AWPU_KERNEL_EXACT_PAIR = 10,      /* das_exact_pair_kernel: the reference's operation order on the frame-pair layout */
    AWPU_KERNEL_EXACT_VERIFY = 11,    /* das_exact_kernel: round-1 verification structure (bf16 accumulator mode, fallback) */
    AWPU_KERNEL_TUNING = 12,          /* a shape only -DAWPU_TUNING_BUILD builds dispatch to */
    AWPU_KERNEL_EXACT_QUAD = 13,      /* das_exact_quad_kernel: the reference's order, four vertically adjacent pixels per wave (round 4) */
    AWPU_KERNEL_EXACT_ND = 14,        /* das_exact_nd_kernel: the reference's order on the {next, d} layout (cur - next formed once per sample) */
    AWPU_KERNEL_EXACT_NDH = 15,       /* das_exact_ndh_kernel: single frames in the reference's order (the two halves of the block in the packed lanes) */
    AWPU_KERNEL_EXACT_NDH_STATIONARY = 16, /* ... with every active mic resident and staged by the workgroups themselves (the reference's own shape) */
    AWPU_KERNEL_EXACT_NDP = 17        /* das_exact_ndp_kernel: ... one pixel per wave (grids of at most 32 pixels per CU: one AWPU's 4 arrays on a 64 x 64 grid; FAST takes it up to 16) */
} awpu_kernel_id;

typedef struct awpu_hip awpu_hip_t;

typedef struct {
    int32_t struct_size; /* sizeof(awpu_hip_cfg), for ABI growth */
    int32_t device;      /* HIP device ordinal */
    int32_t n_streams;   /* mic streams per frame snapshot (Pipeline::get_n_sensors,
                            src/fpga/pipeline.h:107) */
    int32_t hist;        /* floats per stream in a snapshot; AWPU_HIST (streams.hpp:113-116) */
    int32_t n_pixels;    /* P = rows*columns of the steering grid (mimo.cpp:8 maxIndex) */
    int32_t lut_stride;  /* row length of the delay tables = number of physical mic ids
                            (ELEMENTS in mimo.cpp:26-27) */
    int32_t interp;      /* awpu_interp */
    int32_t math;        /* awpu_math */
    int32_t max_batch;   /* frames per awpu_hip_process call, >= 1 */
    /* pixel shard owned by this handle (multi-GPU: one handle per rank); the tables passed
     * to awpu_hip_set_delay_table and the power rows written cover exactly these pixels */
    int32_t pixel_begin;
    int32_t pixel_count; /* 0 = all n_pixels */
    /* optional hint: pixels per row of the steering grid (MIMOWorker's `columns`, mimo.cpp:8), 0 = unknown.
     * The sweeps use it to give a wave vertically adjacent pixels, whose integer delays coincide for most mics (shared
     * sums and sample reads: the quad shapes, the FIR8 shared-sample block).  Results with and without it agree within the
     * fast sweep's fp32 rounding (another order of the same sums; FIR8 and the exact mode: bit for bit).  Ignored unless
     * pixel_begin and pixel_count are whole rows. */
    int32_t grid_columns;
    /* Device group: with n_devices > 1 the handle spreads its pixels over devices[0 .. n_devices-1]
     * (HIP ordinals; `device` is then ignored) -- row groups of four dealt round-robin when grid_columns is set and every
     * device gets at least two of them (edge rows cost more than centre rows), contiguous slabs otherwise --
     * one internal engine, stream and table slab per device; all in this one process, no collective library.
     * Entry points keep their meaning: host frames are uploaded by every device over its own PCIe link; device
     * frames (awpu_hip_process_device, pointers on devices[0]) fan out by direct peer copies over xGMI, one per
     * destination on that destination's stream -- as PACKED frame pairs (devices[0] runs the sweep's pack pass once, the
     * others sweep the pairs as they arrive) wherever the batch is swept by a frame-pair shape, as raw windows otherwise;
     * the slabs are swept concurrently; power tiles come back to the caller's buffer.  awpu_hip_ingest_block feeds every device's ring.  0 or 1 = one device, as before. */
    int32_t n_devices;
    int32_t devices[AWPU_MAX_DEVICES];
    /* Optional: history samples [window_begin, window_end) that the handle stages per stream even where its own table
     * touches fewer (0, 0 = exactly what the table touches: [min off, max off + 257), FIR8 + 263).  Handles that own
     * different slabs of one grid and exchange PACKED frames (awpu_hip_pack_frames / awpu_hip_process_packed) must
     * stage the same window, so every rank passes the union over all slabs here.  Results do not depend on it. */
    int32_t window_begin;
    int32_t window_end;
    int32_t reserved[1];
} awpu_hip_cfg;

typedef struct {
    uint64_t frames;          /* frames processed since creation */
    uint64_t launches;        /* sweep kernel launches */
    double last_kernel_ms;    /* device time of the last sweep launch that was timed (hipEvent): every launch, except that the synchronous
                                 one-frame host call (awpu_hip_process, batch 1) times its first call and every 32nd after it -- the two
                                 event records cost 3 us of a 45 us call */
    double total_kernel_ms;   /* sum over launches */
    uint64_t alg_bytes_frame; /* 4*U*W + 8*P*U + 4*P, W = 256 + tau_max + 1 */
    uint64_t alg_flops_frame; /* 4*P*U*256 + 6*P*254 */
    int32_t tau_max;          /* largest integer delay in the table = 256 - min(off) */
    int32_t window;           /* W */
    int32_t usable;           /* U */
    int32_t kernel_variant;   /* awpu_kernel_id of the sweep kernel the LAST launch ran (AWPU_KERNEL_NONE before the first) */
    int32_t group_exchange;   /* device groups: what the last awpu_hip_process_device sent to the other devices (AWPU_EXCHANGE_*) */
    int32_t group_ranges;     /* device groups: pixel ranges per device (1 = contiguous slabs, more = row groups of four dealt round-robin) */
} awpu_hip_stats;
#define AWPU_EXCHANGE_NONE 0
#define AWPU_EXCHANGE_WINDOWS 1      /* the touched window of every stream, one 2-D copy per device; every device runs its whole sweep */
#define AWPU_EXCHANGE_PACKED_PAIRS 2 /* devices[0] packs once, one linear copy per device, every device sweeps the packed pairs */

/* fills cfg with the reference defaults (64 streams, hist 1024, LERP, batch 1) and the reference's arithmetic (AWPU_MATH_F32_EXACT:
 * results within 1e-5 of the reference on any input; AWPU_MATH_F32_FAST is an explicit opt-in, see INTEGRATION.md "Which math mode") */
void awpu_hip_default_cfg(awpu_hip_cfg *cfg);

/* replaces: MIMOWorker::MIMOWorker allocation of its state, src/dsp/mimo.cpp:7-13 /
 * src/dsp/mimo.h:74-91 */
int awpu_hip_create(awpu_hip_t **h, const awpu_hip_cfg *cfg);

/* replaces: MIMOWorker::~MIMOWorker / Worker::~Worker, src/dsp/worker.h:104-107 */
int awpu_hip_destroy(awpu_hip_t *h);

/* uploads the tables MIMOWorker::computeDelayLUT produced (src/dsp/mimo.cpp:20-59):
 * off[pixel_count][lut_stride] (offsetDelays, = 256 - floor(tau)), frac[..][..]
 * (fractionalDelays in [0,1)), pixel-major like mimo.h:86-88.  Entries of mic ids that are
 * not in the active list are ignored. */
int awpu_hip_set_delay_table(awpu_hip_t *h, const int32_t *off, const float *frac);

/* sets antenna.usable / antenna.index as filled by AWProcessingUnit::calibrate
 * (src/aw_processing_unit/aw_processing_unit.cpp:157-200, src/geometry/antenna.h:89-90):
 * index[s] is both the stream read for signals[s] (mimo.cpp:100-103) and the table column
 * (mimo.cpp:125-127).  index == NULL means identity 0..usable-1. */
int awpu_hip_set_active_mics(awpu_hip_t *h, const int32_t *index, int32_t usable);

/* Optional per-mic gain: the reference computes antenna.power_correction_mask
 * (aw_processing_unit.cpp:190-200) and never applies it; with gains set, the sweep behaves as if every
 * stream s of the snapshot had been multiplied by gains[s] first.  gains [n_streams], indexed by
 * stream id like the table columns; NULL switches it off (the default, = the reference). */
int awpu_hip_set_mic_gains(awpu_hip_t *h, const float *gains);

/* replaces: AWProcessingUnit::calibrate for one array, src/aw_processing_unit/aw_processing_unit.cpp:102-212
 * (SURVEY 8f N4), with the per-mic mean squares computed on the device: mean square of every mic of
 * array `array` (streams array*64 .. +63) over the snapshot, the median (elements 32 and 33 of the
 * sorted list, as the reference takes it), then the usable mics: those within 1e-4 of the median and
 * not below median*1e-3.  Outputs (host): index[64] (mic ids inside the array, ascending),
 * correction[64] (reference_power_level / power), *median, *usable (entries filled).
 * _device: d_frame [n_streams][hist] in device memory, on `stream` (NULL = the handle's); returns
 * after the result has reached the host.  _ring: on the current snapshot of the ingest ring. */
int awpu_hip_calibrate_device(awpu_hip_t *h, const float *d_frame, int32_t array, float reference_power_level,
                              int32_t *index, float *correction, float *median, int32_t *usable, void *stream);
int awpu_hip_calibrate_ring(awpu_hip_t *h, int32_t array, float reference_power_level, int32_t *index,
                            float *correction, float *median, int32_t *usable);
/* _host: frame [n_streams][hist] in host memory, the snapshot AWProcessingUnit::calibrate assembles with
 * Streams::read_stream (aw_processing_unit.cpp:116-122); the array's 64 streams are uploaded, the rest is as
 * above.  What the C++ mirror's AWProcessingUnit::calibrate calls when its pipeline keeps host rings. */
int awpu_hip_calibrate_host(awpu_hip_t *h, const float *frame, int32_t array, float reference_power_level,
                            int32_t *index, float *correction, float *median, int32_t *usable);

/* ---- few-beam delay-and-sum for the trackers --------------------------------- */

/* replaces: Particle::steer for a batch of directions, src/dsp/particle.cpp:37-49 (the same modf split as
 * the MIMO table): off/frac [n_dir][n] from element positions xyz[3][n] and angles theta/phi [n_dir]. */
int awpu_hip_steer_table(const float *xyz, int32_t n, const double *theta, const double *phi, int32_t n_dir,
                         int32_t *off, float *frac);

/* replaces: Particle::beam (src/dsp/particle.cpp:51-82) and Particle::das (:88-103) for n_dir steered
 * directions in one launch -- what GradientParticle::step (gradient_ascend.cpp:30-81) asks four times
 * per particle and MISOWorker::update (miso.cpp:40-46) once per block.  off/frac: host tables
 * [n_dir][lut_stride] as from awpu_hip_steer_table, read at the handle's active mics.  d_frame: one
 * snapshot [n_streams][hist] in device memory, or NULL = the current snapshot of the ingest ring.
 * power [n_dir] (host, may be NULL): sum MA^2 / 256 -- NOT divided by the mic count, as in the reference.
 * beams [n_dir][256] (host, may be NULL): the delayed-and-summed signal, bit-identical to the reference's.
 * Needs awpu_hip_set_active_mics; does not touch the heatmap table.  Synchronous. */
int awpu_hip_beams(awpu_hip_t *h, const float *d_frame, const int32_t *off, const float *frac, int32_t n_dir,
                   float *power, float *beams);

/* FIR8 coefficient table, the caller's copy of filter_coeffs[101][8] (src/dsp/filter.h:10-112) */
int awpu_hip_set_fir_table(awpu_hip_t *h, const float *coeffs);

/* replaces: the body of MIMOWorker::update, src/dsp/mimo.cpp:97-151.
 *   frames [batch][n_streams][hist] host floats; each frame is the snapshot update() takes
 *          with Streams::read_stream (mimo.cpp:100-103; oldest..newest, streams.hpp:113-116)
 *   power  [batch][pixel_count] host floats = powerdB (mimo.cpp:150)
 * Synchronous: returns when power is written. */
int awpu_hip_process(awpu_hip_t *h, const float *frames, int32_t batch, float *power);

/* The same call split in two, so that the caller's thread can do something else while the frames travel and the
 * sweep runs: _async enqueues the upload, the sweep and the read-back on the handle's stream and
 * returns; awpu_hip_wait blocks until `power` is written (and updates the kernel time in the stats).  `frames`
 * and `power` must stay valid and untouched until awpu_hip_wait returns; one call in flight per handle (a second
 * _async before the wait is AWPU_ERR_STATE).  Page-locked host buffers make the copies truly asynchronous;
 * pageable ones work, the runtime then stages them. */
int awpu_hip_process_async(awpu_hip_t *h, const float *frames, int32_t batch, float *power);
int awpu_hip_wait(awpu_hip_t *h);

/* same sweep on buffers already resident in device memory, enqueued on `stream`
 * (a hipStream_t, NULL = the handle's own stream); asynchronous.  d_frames
 * [batch][n_streams][hist], d_power [batch][pixel_count]. */
int awpu_hip_process_device(awpu_hip_t *h, const float *d_frames, int32_t batch, float *d_power,
                            void *stream);

/* Verification export (no reference counterpart; the reference's `float out[N_SAMPLES]` of src/dsp/mimo.cpp:122 is a local):
 * the same sweep with every pixel's out[0..255] AFTER the last mic's delay() and BEFORE the moving average of
 * mimo.cpp:131-137 also written to d_sums [batch][pixel_count][256].  AWPU_MATH_F32_EXACT + AWPU_INTERP_LERP on a
 * single-device handle only (AWPU_ERR_STATE otherwise): those sums are bit-identical to what the reference's delay()
 * leaves in out[] (tests/test_gpu_parity.py checks them against the goldens its compiled object code produced). */
int awpu_hip_process_device_sums(awpu_hip_t *h, const float *d_frames, int32_t batch, float *d_power, float *d_sums,
                                 void *stream);

/* ---- the sweep split at its pack pass (multi-GPU: the exchange format between ranks) --------------------------
 * The batched sweep first interleaves the touched window of two consecutive frames sample by sample ("packed frame
 * pairs": [ceil(batch/2)][usable][wp][2] floats, active-mic order) and then sweeps that buffer.  A
 * multi-GPU job whose ranks own slabs of one grid needs the pack only ONCE: the ingest rank packs, the
 * packed buffer is what travels (the same bytes as the raw window), and every rank sweeps it as it arrives --
 * no window cut on the root, no pack pass on the others.  No reference counterpart (one thread, one array:
 * src/dsp/mimo.cpp:12, :100-103 is the snapshot this replaces).
 * Available with AWPU_INTERP_LERP, usable % 4 == 0, no mic gains, single-device handles, in both fp32 modes; every rank
 * must have been created with the same math mode, n_streams, hist, active mics and cfg.window_begin/window_end.
 *   AWPU_MATH_F32_EXACT (the default): the packed rows are the {next, d} elements das_exact_nd_kernel sweeps --
 *     [ceil(batch/2)][usable][window - 1] x { X_a[t+1], X_b[t+1], X_a[t] - X_a[t+1], X_b[t] - X_b[t+1] } (16 bytes: delay.cpp:21's
 *     `cur - next` is formed once, by the pack pass) -- twice the bytes of the raw window; needs cfg.grid_columns and batch >= 2;
 *     awpu_hip_process_packed(pack_frames(x)) gives the bits of awpu_hip_process_device(x) for every such batch.
 *   AWPU_MATH_F32_FAST: pre-filtered sample pairs, the same bytes as the raw window; the same bits as
 *     awpu_hip_process_device(x) wherever that call sweeps frame pairs itself (batches that fill the chip: >= 256 workgroups);
 *     for smaller batches, where process_device prefers a single-frame shape, the two agree to fp32 rounding (a few 1e-6).
 *   _packed_bytes: *bytes = size of the packed buffer for `batch` frames, AWPU_ERR_STATE when the handle's sweep does
 *                  not take packed frames (the caller then exchanges raw windows and calls awpu_hip_process_device)
 *   _pack_frames:  d_frames [batch][n_streams][hist] -> d_packed, enqueued on `stream` (NULL = the handle's)
 *   _process_packed: d_packed -> d_power [batch][pixel_count], enqueued on `stream`; asynchronous */
int awpu_hip_packed_bytes(awpu_hip_t *h, int32_t batch, uint64_t *bytes);
int awpu_hip_pack_frames(awpu_hip_t *h, const float *d_frames, int32_t batch, float *d_packed, void *stream);
int awpu_hip_process_packed(awpu_hip_t *h, const float *d_packed, int32_t batch, float *d_power, void *stream);

/* blocks until everything enqueued through this handle has finished */
int awpu_hip_synchronize(awpu_hip_t *h);

/* replaces: MIMOWorker::populateHeatmap (USE_DB 0), src/dsp/mimo.cpp:61-95, with the
 * cv::Mat replaced by a plain rows*columns uint8 image: pix = clip(power/max*255).
 * power/pix are host buffers of `n` elements (n = whole grid).  An all-zero frame (max 0, so 0/0: the
 * reference casts a NaN to uchar there, which is undefined) gives an all-zero image, here and on the device. */
int awpu_hip_heatmap_u8(const float *power, int32_t n, uint8_t *pix);

/* the same display step on buffers resident in device memory: d_power [batch][n]
 * floats -> d_pix [batch][n] bytes, enqueued on `stream` (NULL = the handle's stream).  d_peak [batch]
 * floats receives each frame's maximum; with peak_given != 0 it is an INPUT instead (multi-GPU: the
 * all-reduced maximum over the ranks' tiles, so that every tile is scaled alike). */
int awpu_hip_heatmap_u8_device(awpu_hip_t *h, const float *d_power, int32_t n, int32_t batch, float *d_peak,
                               int32_t peak_given, uint8_t *d_pix, void *stream);

/* replaces: cv::resize(*compact, *normal, normal->size(), 0, 0, cv::INTER_LINEAR) in AWProcessingUnit::draw,
 * src/aw_processing_unit/aw_processing_unit.cpp:252 (and, with a colour table, cv::applyColorMap in the
 * GUI loop, src/aw_processing_unit/main.cpp:345), on images resident in device memory: d_pix
 * [batch][rows][cols] bytes -> d_out [batch][out_rows][out_cols] bytes, or [batch][out_rows][out_cols][3]
 * through d_colormap[256][3] (device memory; entry v = the three output bytes for level v) when
 * d_colormap is not NULL.  Same 8-bit fixed-point arithmetic as OpenCV's generic path.  Upscaling only
 * (out >= in): AWPU_ERR_INVALID otherwise.  Enqueued on `stream` (NULL = the handle's stream). */
int awpu_hip_upscale_u8_device(awpu_hip_t *h, const uint8_t *d_pix, int32_t rows, int32_t cols, int32_t batch,
                               const uint8_t *d_colormap, uint8_t *d_out, int32_t out_rows, int32_t out_cols,
                               void *stream);

/* the same resize on host buffers (one image), for callers that already hold the compact image on the host */
int awpu_hip_resize_linear_u8(const uint8_t *pix, int32_t rows, int32_t cols, uint8_t *out, int32_t out_rows,
                              int32_t out_cols);

/* ---- wire-format ingest on the device -------------------------------------- */

#define AWPU_DATAGRAM_BYTES 1032 /* sizeof(message), src/fpga/receiver.h:24-30: 8-byte header + 256 x i32 */

/* replaces: Pipeline::receive_exposure + Streams::write_stream/forward for one block,
 * src/fpga/pipeline.cpp:260-297, src/fpga/streams.hpp:103-105,136-139.  `datagrams` = 256 consecutive
 * wire datagrams (host memory, `stride_bytes` apart, normally AWPU_DATAGRAM_BYTES): sample i of
 * sensor s is stream_i[flip(s)] / 2^23.  The block is appended to a per-handle history ring in device
 * memory (cfg.hist must be AWPU_HIST, cfg.n_streams <= 256).  The ring starts zeroed. */
int awpu_hip_ingest_block(awpu_hip_t *h, const void *datagrams, int32_t stride_bytes);

/* One display step of the live path in one call and one wait -- what the reference spreads over
 * Pipeline::receive_exposure (src/fpga/pipeline.cpp:260-297), MIMOWorker::update (src/dsp/mimo.cpp:97-151),
 * populateHeatmap (mimo.cpp:61-95) and cv::resize in AWProcessingUnit::draw (aw_processing_unit.cpp:252):
 * ingest the block, sweep the ring's new snapshot, scale to 8 bits, upscale, and bring back what the
 * caller asks for.  Every output is a host buffer and may be NULL: power [n_pixels], image [rows*cols],
 * big_image [out_rows*out_cols] (x3 through d_colormap[256][3], device memory, when that is not NULL).
 * Needs the whole grid on this handle (pixel_count == n_pixels) and rows*cols == n_pixels.
 * A display loop that passes the same buffers block after block (a receive buffer refilled in place, its two images)
 * gets the call replayed as one captured HIP graph per ring position from the third call on: the copies read and
 * write the buffers' current contents; a new delay table, mic list or set of gains retires the graphs.
 * AWPU_LIVE_GRAPH=0 in the environment keeps the step-by-step path. */
int awpu_hip_live_block(awpu_hip_t *h, const void *datagrams, int32_t stride_bytes, float *power, int32_t rows,
                        int32_t cols, uint8_t *image, int32_t out_rows, int32_t out_cols, const uint8_t *d_colormap,
                        uint8_t *big_image);

/* the body of MIMOWorker::update (src/dsp/mimo.cpp:97-151) on the snapshot the ring currently holds
 * (oldest..newest, what Streams::read_stream would return for every stream): power [pixel_count]. */
int awpu_hip_process_ring(awpu_hip_t *h, float *power);

/* copies the current 1024-sample snapshot of every stream to host memory [n_streams][1024] (debug /
 * calibration: AWProcessingUnit::calibrate reads exactly this, aw_processing_unit.cpp:116-122) */
int awpu_hip_ring_snapshot(awpu_hip_t *h, float *frames);

/* ---- host-side geometry, one-off (not on the per-frame path) ------------------------- */

/* create_antenna, src/geometry/antenna.cpp:60-87: xyz[3][rows*columns] */
int awpu_hip_create_antenna(int32_t columns, int32_t rows, float distance, float *xyz);

/* tiled multi-array geometry (build-defined): arrays_x*arrays_y arrays of 8x8,
 * stream id = a*64 + r*8 + c (aw_processing_unit.cpp:120), xyz[3][64*arrays_x*arrays_y] */
int awpu_hip_create_tiled_antenna(int32_t arrays_x, int32_t arrays_y, float distance, float *xyz);

/* steering_vector_spherical, src/geometry/antenna.cpp:126-129: tau[n] in samples, min 0 */
int awpu_hip_steering_delays(const float *xyz, int32_t n, double theta, double phi, float *tau);

/* MIMOWorker::computeDelayLUT, src/dsp/mimo.cpp:20-59, for grid rows
 * [row_begin, row_begin+row_count) of a rows x columns grid:
 * off/frac [row_count*columns][n] */
int awpu_hip_build_delay_table(const float *xyz, int32_t n, int32_t rows, int32_t columns,
                               float fov_deg, int32_t row_begin, int32_t row_count, int32_t *off,
                               float *frac);

/* The same table generated on the device ("optionally ... so the LUT can be generated on device"): the
 * per-pixel angles and rotation entries (double, the host's libm: rows * columns values) stay on the host, the
 * rows * columns * n part -- steer() / compute_delays() of src/geometry/antenna.cpp:89-107 and the split of
 * src/dsp/mimo.cpp:46-54 -- runs on HIP device `device` with the host builder's operations in the host builder's order.
 * off / frac are host arrays as above and receive the SAME BITS awpu_hip_build_delay_table writes (c4, 65 536 pixels x
 * 512 mics: 0.12 s on one core of the GPU box's host, 0.022 s here, most of it the copy back).
 * AWPU_ERR_NO_DEVICE without a gfx950 device: the host builder needs none. */
int awpu_hip_build_delay_table_device(int32_t device, const float *xyz, int32_t n, int32_t rows, int32_t columns,
                                      float fov_deg, int32_t row_begin, int32_t row_count, int32_t *off, float *frac);

/* ---- introspection -------------------------------------------------------------------- */

int awpu_hip_get_stats(awpu_hip_t *h, awpu_hip_stats *stats);

/* How each device of a device group (cfg.n_devices > 1) exchanges frames and power tiles with devices[0]; the
 * reference has no counterpart (one thread, one array: src/dsp/mimo.cpp:12).  awpu_hip_create asks for peer access
 * both ways (hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess) and checks the answers: a device without it takes
 * the explicit path through pinned host memory (correct, PCIe-bound), and awpu_hip_last_error_of(h) says which
 * pair and why right after creation.  status[k] for k < n_devices; returns the number of entries written
 * (1 and AWPU_PEER_SAME_DEVICE for a single-device handle) or a negative awpu_status. */
#define AWPU_PEER_SAME_DEVICE 0 /* devices[k] == devices[0]: the caller's buffers are swept in place */
#define AWPU_PEER_DIRECT 1      /* peer copies over xGMI */
#define AWPU_PEER_HOST_STAGED 2 /* no peer access: staged through pinned host memory */
int awpu_hip_group_peer_status(awpu_hip_t *h, int32_t *status, int32_t n);
const char *awpu_hip_strerror(int status);
/* text of the last error seen by this thread ("" if none) */
const char *awpu_hip_last_error(void);
/* text of the last error any thread ran into while working on this handle ("" if none): a worker thread
 * owns the handle, the thread that asks (the reference's GUI thread) usually is another one.  The
 * pointer stays valid until the handle's next failing call. */
const char *awpu_hip_last_error_of(awpu_hip_t *h);
int awpu_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AWPU_HIP_H */
