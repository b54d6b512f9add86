// das_kernels.h -- launch interface between the C-ABI layer (awpu_hip.cpp) and the gfx950
// sweep kernels (das_kernels.hip).  Internal to libawpu_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace awpu {

constexpr int kSamples = 256;  // N_SAMPLES, src/fpga/streams.hpp:28

// AWPU_FAST_DEBUG bits.  The switches that give WRONG results (timing experiments: what does the kernel cost
// without its refill / sweep / tail pass / barrier) exist only in builds with -DAWPU_TIMING_BUILD
// (AWPU_EXTRA_HIPCC_FLAGS); in the shipping library the tests below are the constant 0 and the environment
// variable cannot reach them (awpu_hip.cpp masks it to kDebugSafeBits).
constexpr int kDebugNoRefill = 1, kDebugNoSweep = 2, kDebugNoTail = 4, kDebugNoBarrier = 8, kDebugRegStaging = 64;
constexpr int kDebugSafeBits = 16 | 256 | 512 | 4096;  // stamps, dispatch order, refill by rank, unshared block: same results
#if defined(AWPU_TIMING_BUILD) && !defined(AWPU_TUNING_BUILD)
#define AWPU_TUNING_BUILD 1  // the timing experiments need the tuning knobs and the stamped kernels
#endif
#ifdef AWPU_TIMING_BUILD
#define AWPU_DBG(a, bit) ((a).debug & (bit))
#else
#define AWPU_DBG(a, bit) 0
#endif

// One table entry per (pixel, active mic s): the integer window start relative to the
// staged window (off - wstart >= 0) and the fraction.  8 bytes, the size the reference
// keeps per (pixel, mic) (int offsetDelays + float fractionalDelays, src/dsp/mimo.h:86-88).
struct LutEntry {
    int32_t off_rel;
    float frac;
};

struct SweepArgs {
    const float *frames;   // [batch][n_streams][hist]
    const LutEntry *lut;   // [pixel_count][usable], compact active-mic order
    const int32_t *index;  // [usable] stream id per active mic
    float *power;          // [batch][pixel_count]
    const float *gain;     // [usable] per-mic gain applied while staging, or nullptr (the reference has none)
    int32_t n_streams;
    int32_t hist;
    int32_t usable;
    int32_t pixel_count;
    int32_t wstart;  // first history sample any entry touches (= min off)
    int32_t window;  // W = 256 + tau_max + 1 samples staged per mic
    int32_t batch;
};

// ---- fast kernel (das_fast.hip) ---------------------------------------------------------
constexpr int kFastLdsBytes = 78 * 1024; // one staged image; a CU holds two (+ a 4 KiB side table)
constexpr int kFastSideBytes = 4 * 1024;
constexpr int kFastLdsBytesSmall = 38 * 1024;  // image of the two-workgroups-per-CU double-buffered shape

// One 16-byte entry per (pixel, active mic): what one item needs, laid out so that f and g
// start even SGPRs after an s_load_dwordx16 (packed-FMA scalar operands are aligned pairs).
struct FastEntry {
    float f;        // weight of X[off+i]   (the reference's `fraction`)
    uint32_t addr;  // LDS byte offset of X[off] in the copy of parity (off & 1), frame 0, lane 0
    float g;        // 1 - f, weight of X[off+i+1]
    uint32_t pad;
};

// What lies behind the two pointers of a sweep launch.  The kernels read PAST the data they use, by design -- the table is
// prefetched one group of mics ahead of the sweep (a load of entries the last trip never uses), slots outside the grid sweep
// a clamped or a null table row, the samples arrive in whole 16-byte pieces -- so every launcher below states its kernel's
// REACH (the furthest entry / float any wave can load, for the last frame pair, the last chunk and the last pixel) and
// refuses, with hipErrorInvalidValue, a launch whose reach exceeds what the caller says it allocated.
struct Extents {
    size_t lut_entries;    // entries (of the table's element type) behind the table pointer
    size_t sample_floats;  // floats behind `packed` (0: the samples are the caller's frames, bounded by hist / row_limit)
};
constexpr int kPairTablePrefetch = 4;        // FastEntry: one group of four mics past the end of a pixel's row
constexpr int kQuadTablePrefetch = 16;       // QuadEntry: one group (4 pixels x 4 mics) past the end of a quad's groups
constexpr int kFir8PlaneTablePrefetch = 68;  // dwords: entries four items ahead + 64 past a chunk (block_fir8)

struct FastPlan {
    int fpi;         // frames per item (1 or 2)
    int wr;          // floats per staged row (even)
    int chunk;       // mics staged per pass (multiple of 4, <= 64)
    int usable_pad;  // table row length, usable rounded up to 4 (null entries at the end)
    int row_bytes;
    int image_bytes;  // LDS bytes of one staged image
};

struct FastArgs {
    const float *frames;   // [batch][n_streams][hist]
    const FastEntry *lut;  // [pixel_count][usable_pad] (+ one spare group)
    const int32_t *index;  // [usable]
    const int32_t *row_off;  // [2*usable_pad + spare] float offset of staged row 2*s+q in a frame
    float *power;          // [batch][pixel_count]
    int32_t n_streams, hist, usable, usable_pad, pixel_count;
    int32_t wstart, wr, chunk, batch;
    int32_t frames_per_wg;  // double-buffered shape: consecutive frames one workgroup sweeps
    unsigned long long *debug_out;  // diagnostics (debug bit 16): 4 words per wave
    int32_t debug;  // timing experiments only (AWPU_FAST_DEBUG): 1 = stage first chunk only, 2 = skip the sweep
};

// ---- frame-pair shape: frames packed two by two, sample-interleaved
struct PairArgs {
    const float *packed;   // [pairs][usable][wp][2]: (frame 2k, frame 2k+1) samples of the window, active-mic order
    const FastEntry *lut;  // [P_pad][usable_pad]; addr = slot * wp * 8 + (off - wstart) * 8
    float *power;          // [batch][pixel_count]
    int32_t usable, usable_pad, pixel_count, wp, chunk, batch;
    int32_t cols;  // > 0: the grid's row length; a wave then sweeps vertical pixel pairs (pixel_count % cols == 0)
    int32_t tiles, n_pairs, pair_group;  // das_pair_kernel: 64-pixel tiles; frame pairs, and how many an XCD works on at a time
    unsigned long long *debug_out;
    int32_t debug;
    // das_pair_stationary_kernel only: frames != null -> the workgroup stages its pair ITSELF from the caller's frames
    // [batch][n_streams][hist] (pack_pairs_kernel<true>'s expression, straight into the LDS image: the same bits, no pack
    // pre-pass, `packed` unused); index [usable] = the active streams, wstart = first history sample of the window
    const float *frames;
    const int32_t *index;
    int32_t n_streams, hist, wstart;
};
// 64-pixel tiles of the frame-pair sweep: consecutive pixels, or 2 rows x 32 columns when cols > 0
inline int pair_tiles(int pixel_count, int cols) {
    return cols > 0 ? ((pixel_count / cols + 1) / 2) * ((cols + 31) / 32) : (pixel_count + 63) / 64;
}
bool pair_plan(int window, int usable, FastPlan *plan);
// rows_out >= usable rows per pair are written (the extra ones zero); d_gain [usable] (or null) scales row s;
// filter: the rows carry the moving-average stencil of the samples, Y[t] = X[t]/2 - (X[t+1] + X[t-1])/4 -- what
// das_pair_kernel, das_pair_stationary_kernel and das_quad_kernel sweep (their epilogue then has no stencil and
// there is no 257th sample); the FIR8 pair kernel takes raw samples
hipError_t launch_pack_pairs(const float *d_frames, int n_streams, int hist, int wstart, const int32_t *d_index,
                             int usable, int rows_out, const float *d_gain, int wp, int batch, float *d_packed,
                             bool filter, hipStream_t stream);
hipError_t launch_das_pairs(const PairArgs &a, const Extents &have, hipStream_t stream);
// ---- reference-order sweep on the frame-pair layout (das_exact_pair_kernel, AWPU_MATH_F32_EXACT): raw samples
// (launch_pack_pairs with filter = false and rows_out = usable_pad: gains on the samples, padding rows zero), the pair
// shape's table with the UNSCALED fraction in .f and padding entries that point at their own (zero) row
struct ExactPairArgs {
    const float *packed;   // [pairs][usable_pad][wp][2]
    const FastEntry *lut;  // [P_pad][usable_pad]; f = the reference's fraction, addr = slot * wp * 8 + (off - wstart) * 8
    float *power;          // [batch][pixel_count]
    float *sums;           // optional [batch][pixel_count][256]: out[] of every pixel before the epilogue (tests), or null
    int32_t usable, usable_pad, pixel_count, wp, chunk, batch;
    int32_t cols;          // > 0: the grid's row length; a wave then sweeps vertical pixel pairs (pixel_count % cols == 0)
    int32_t tiles, n_pairs, pair_group;  // 64-pixel tiles (pair_tiles); frame pairs, and how many an XCD works on at a time
};
hipError_t launch_das_exact_pairs(const ExactPairArgs &a, const Extents &have, hipStream_t stream);
// the same with four vertically adjacent pixels per wave (das_exact_quad_kernel): the quad-major table of the quad shapes
// ([quad][group of 4 mics][pixel][mic] x QuadEntry) with the RAW fraction in .f; needs the grid's row length
struct ExactQuadArgs {
    const float *packed;   // [pairs][usable_pad][wp][2], padding rows zero
    const struct QuadEntry *lut;
    float *power;          // [batch][pixel_count]
    float *sums;           // optional [batch][pixel_count][256], or null
    int32_t usable, usable_pad, pixel_count, wp, chunk, batch;
    int32_t cols, rows;
    int32_t tiles, n_pairs, pair_group;  // workgroup tiles of 4 rows x 16 columns; frame pairs, and how many an XCD works on at a time
};
hipError_t launch_das_exact_quads(const ExactQuadArgs &a, const Extents &have, hipStream_t stream);
// ---- the reference's order on the {next, d} layout (das_exact_nd_kernel, round 5; the default of AWPU_MATH_F32_EXACT where the
// row length is known): the pack pass forms delay.cpp:21's `cur - next` ONCE per sample and frame -- element t of a mic's row =
// { X_a[t+1], X_b[t+1], X_a[t] - X_a[t+1], X_b[t] - X_b[t+1] }, 16 bytes, t counted from wstart -- and the sweep is left with
// fma(frac, d, next) and the add, in the reference's order (the same fp32 subtraction of the same operands: the same bits).
struct ExactNdArgs {
    const float *packed;   // [pairs][usable_pad][wq][4], padding rows zero
    const struct QuadEntry *lut;  // quad-major, raw fractions, addr = slot * wq * 16 + (off - wstart) * 16; quad rows padded to a multiple of nq
    float *power;          // [batch][pixel_count]
    float *sums;           // optional [batch][pixel_count][256]: out[] of every pixel before the epilogue (tests), or null
    int32_t usable, usable_pad, pixel_count, wq, chunk, batch;
    int32_t cols, rows;
    int32_t nq;            // quads per wave: a workgroup tile is 4 nq rows x 16 columns
    int32_t tiles, n_pairs, pair_group;  // nd_tiles(rows, cols, nq); frame pairs, and how many an XCD works on at a time
    unsigned *queue;                     // [9] item counters (one per XCD + the common tail): device memory of the handle, zeroed by the launcher on the stream
    int32_t wgs;                         // persistent workgroups to launch: the device's CUs (one fits a CU)
    int32_t tail;                        // items at the end of every XCD's run that go to the common queue (>= the run: one queue for the chip)
    const int2 *items;                   // [n_pairs * tiles] (frame pair, first table quad of the tile), item order; device memory of the handle
    int32_t build_items;                 // the launcher runs nd_items_kernel first (the list is stale: another batch size, nq or pair group)
    unsigned long long *debug_out;       // tuning builds (AWPU_FAST_DEBUG=16): 8 words per workgroup -- where and when it ran -- or null
};
inline int nd_tiles(int rows, int cols, int nq) { return (((rows + 3) / 4 + nq - 1) / nq) * ((cols + 15) / 16); }
inline int nd_quad_count(int rows, int cols, int nq) { return (((rows + 3) / 4 + nq - 1) / nq) * nq * ((cols + 15) / 16) * 16; }  // table quads incl. padding
bool exact_nd_plan(int window, int usable, FastPlan *plan);  // plan->wr = elements per row (window - 1), row_bytes = 16 wr
hipError_t launch_pack_nd(const float *d_frames, int n_streams, int hist, int wstart, const int32_t *d_index, int usable, int rows_out,
                          const float *d_gain, int wq, int batch, float *d_packed, hipStream_t stream);
hipError_t launch_das_exact_nd(const ExactNdArgs &a, const Extents &have, hipStream_t stream);
// ---- single frames in the reference's order: the HALVES form of the {next, d} layout (das_exact_ndh_kernel, round 5).  Element t of
// a mic's row = { X[t+1], X[t+129], X[t] - X[t+1], X[t+128] - X[t+129] } (t from wstart; wh = window - 129 elements): the two packed
// lanes are the two halves of the 256-sample block, as in das_quadh_kernel.  Chunked (rows packed by launch_pack_ndh, the item block
// refilling the other image) or STATIONARY (every active mic resident, the workgroup forming the elements itself from the caller's
// frame: no pre-pass, no chunks -- one array at the reference's default resolution).
// Completion flag of the synchronous one-frame host call (the resident single-frame kernels, one quad per wave; flag == null: none): a
// workgroup's 64 powers leave as four 64-byte system-scope stores, the workgroup counts itself on *counter -- which only ever grows --
// when they are acknowledged, and the one that brings it to `target` stores `seq` to *flag (pinned host memory), where the host spins.
struct DoneFlag {
    unsigned long long *counter;
    unsigned *flag;
    unsigned long long target;
    uint32_t seq;
};
struct ExactNdhArgs {
    const float *packed;   // chunked: [batch][usable_pad][wh][4], padding rows zero; stationary: unused
    const float *frames;   // stationary: [batch][n_streams][pitch]
    const struct QuadEntry *lut;  // quad-major, raw fractions, addr = slot * wh * 16 + (off - wstart) * 16; columns padded to 16 nq
    const int32_t *index;  // stationary: [usable] stream of active mic s
    const float *gain;     // stationary: [usable] or null
    float *power;          // [batch][pixel_count]
    float *sums;           // optional [batch][pixel_count][256], or null
    int32_t n_streams, pitch, wstart;  // stationary
    int32_t usable, usable_pad, pixel_count, wh, chunk, batch;
    int32_t cols, rows;
    int32_t nq;            // quads per wave: a workgroup tile is 4 rows x 16 nq columns
    int32_t tiles;         // ndh_tiles(rows, cols, nq)
    int32_t lut_cols;      // columns of the table (the grid's, padded to whole tiles of 32)
    int32_t identity;      // stationary: the active-mic list is 0 .. usable-1 (rows need no look-up)
    DoneFlag done;         // the resident kernel with one quad per wave only
};
inline int ndh_tiles(int rows, int cols, int nq) { return ((rows + 3) / 4) * ((cols + 16 * nq - 1) / (16 * nq)); }
bool exact_ndh_plan(int window, int usable, bool stationary, FastPlan *plan);  // plan->wr = wh, row_bytes = 16 wh
hipError_t launch_pack_ndh(const float *d_frames, int n_streams, int pitch, int wstart, const int32_t *d_index, int usable, int rows_out,
                           const float *d_gain, int wh, int batch, float *d_packed, hipStream_t stream);
hipError_t launch_das_exact_ndh(const ExactNdhArgs &a, bool stationary, const Extents &have, hipStream_t stream);
// ... one PIXEL per wave (das_exact_ndp_kernel: chunked, 16-wave workgroups of 4 rows x 4 columns; nq unused): grids with too few quads
// to give every SIMD more than one wave
inline int ndp_tiles(int rows, int cols) { return ((rows + 3) / 4) * ((cols + 3) / 4); }
hipError_t launch_das_exact_ndp(const ExactNdhArgs &a, const Extents &have, hipStream_t stream);
// FIR8 on the four-plane frame-pair layout (a lane owns four consecutive outputs: 11 LDS reads per 32 FMAs).  Rows packed by
// launch_pack_planes, `wr` a multiple of 4 (fir8_plane_plan); d_entries [pixel_count][usable_pad] + 4 spare dwords,
// one per (pixel, mic): fir8_plane_word(LDS byte offset of X[off] in its chunk's image, its plane, coefficient row);
// the entries that pad a row to usable_pad carry coefficient row kFir8ZeroRow and row 0's address.  d_coeffs: the
// [kFir8CoeffRows][8] coefficient table on the device, the caller's 101 rows followed by zeros.
constexpr int kFir8ZeroRow = 101, kFir8CoeffRows = 128;
inline uint32_t fir8_plane_word(uint32_t addr, uint32_t plane, uint32_t k) {
    return (addr & 0x3ffffu) | ((plane & 3u) << 18) | ((k & 0x7fu) << 20);
}
bool fir8_plane_plan(int window, int usable, FastPlan *plan);
hipError_t launch_pack_planes(const float *d_frames, int n_streams, int hist, int wstart, const int32_t *d_index, int usable,
                              const float *d_gain, int wp, int batch, float *d_packed, hipStream_t stream);
constexpr uint32_t kFirStaticPlaneBytesHost = 768;  // = kFirStaticPlaneBytes of das_fast_trip.inc (static_assert in das_fast.hip)
hipError_t launch_das_fir8_planes(const PairArgs &a, const void *d_entries, const float *d_coeffs, int variant, const Extents &have,
                                  hipStream_t stream);
// stationary shape: every active mic's window of a frame pair in LDS at once (plan->chunk = usable_pad); a
// workgroup stages the pair once and sweeps tiles_per_wg tiles from it
bool pair_plan_stationary(int window, int usable, FastPlan *plan);
hipError_t launch_das_pairs_stationary(const PairArgs &a, int tiles_per_wg, const Extents &have, hipStream_t stream);

// ---- quad shape (das_quad_kernel): the frame-pair layout swept four vertically adjacent pixels at a time with
// a shared integer-delay sum (das_fast.hip).  Needs the grid's row length.
struct QuadEntry {  // 8 bytes per (pixel, active mic)
    float f;        // the reference's `fraction` (weight of X[off+i]) MINUS 1/2; 1 - f is never needed in the sweep
    uint32_t addr;  // LDS byte offset of the element X[off] in the chunk's image
};
struct QuadArgs {
    const float *packed;   // [pairs][usable_pad][wp][2]: rows of padding mics are zero
    const QuadEntry *lut;  // [quads][usable_pad / 4][4 pixels][4 mics], quads = row quads x padded columns
    float *power;          // [batch][pixel_count]
    int32_t usable, usable_pad, pixel_count, wp, chunk, batch;
    int32_t cols, rows;    // the handle's slab of the grid: rows x cols = pixel_count
    int32_t tiles;         // workgroup tiles of 4 rows x 16 columns
    int32_t n_pairs, pair_group;  // frame pairs, and how many of them one XCD works on at a time
    int32_t variant;              // tuning builds only (AWPU_QUAD_VARIANT)
    int32_t wgs;                  // persistent workgroups to launch (0 = one workgroup per item; AWPU_FAST_WGS)
    unsigned *queue;              // [9] item counters (NdQueues: one per XCD + the common tail), zeroed by the launcher; null: the static shares
    int32_t tail;                 // with `queue`: items at the end of every XCD's run that go to the common queue
    unsigned long long *debug_out;
    int32_t debug;
};
inline int quad_tiles(int rows, int cols) { return ((rows + 3) / 4) * ((cols + 15) / 16); }
inline int quad_count(int rows, int cols) { return ((rows + 3) / 4) * ((cols + 15) / 16) * 16; }  // table quads incl. padding columns
hipError_t launch_das_quads(const QuadArgs &a, const Extents &have, hipStream_t stream);

// tiles of the single-frame quad kernel (das_quadh_kernel): 4 rows x 16 qpw columns
inline int quad1_tiles(int rows, int cols, int qpw) { return ((rows + 3) / 4) * ((cols + 16 * qpw - 1) / (16 * qpw)); }

// ---- single frames on the halves layout (das_quadh_kernel): the two halves of the 256-sample block in the two packed
// lanes, pre-filtered; rows packed by launch_pack_halves, table = the quad-major table with this layout's addresses
// (pair_plan on a window of window - 128 samples)
struct QuadhArgs {
    const float *packed;      // [batch][usable_pad][wp][2]: element t = (Y[wstart + t], Y[wstart + t + 128])
    const QuadEntry *lut;
    float *power;             // [batch][pixel_count]
    int32_t usable, usable_pad, pixel_count, wp, chunk, batch;
    int32_t cols, rows;
    unsigned long long *debug_out;
    int32_t debug;
};
hipError_t launch_das_quadh(const QuadhArgs &a, int qpw, const Extents &have, hipStream_t stream);
// the same for arrays small enough that every active mic's halves row fits the LDS at once (das_quadh_stationary_kernel): the
// workgroup stages (and filters) the window itself from the caller's frame -- no pack pre-pass, no chunks
struct QuadhStationaryArgs {
    DoneFlag done;            // one quad per wave only
    const float *frames;      // [batch][n_streams][pitch] (pitch = hist, or 2048 for a frame read in place from the ingest ring)
    const QuadEntry *lut;     // quad-major table, slot = mic: address = s * wp * 8 + (off - wstart) * 8
    const int32_t *index;     // [usable]
    const float *gain;        // [usable] or null
    float *power;             // [batch][pixel_count]
    int32_t n_streams, pitch, hist, wstart;
    int32_t usable, usable_pad, pixel_count, wp, batch;
    int32_t cols, rows;
    int32_t raw_begin, raw_wr;  // the raw rows staged by LDS-DMA: history samples [raw_begin, raw_begin + raw_wr), whole 16-byte pieces
    int32_t row_limit;          // floats of a stream's row that may be read (never past the frame's allocation)
    int32_t image_offset;       // floats from the start of the LDS to the halves image (after the raw rows)
    int32_t waves;              // waves per workgroup = columns (x qpw) per tile: 16
    int32_t identity;           // the active-mic list is 0 .. usable-1 (awpu_hip_set_active_mics(NULL)): rows need no look-up
};
bool quadh_stationary_plan(int window, int usable, FastPlan *plan);
bool quadh_stationary_raw(const FastPlan &plan, int usable, int wstart, int row_limit, int *raw_begin, int *raw_wr, int *image_offset);
hipError_t launch_das_quadh_stationary(const QuadhStationaryArgs &a, int qpw, const Extents &have, hipStream_t stream);
// `pitch` = floats between two streams of a frame (hist, or 2048 in the ingest ring), `hist` = samples of a stream's
// history (neighbours of the filter outside it count as 0), wstart = first history sample of the window
hipError_t launch_pack_halves(const float *d_frames, int n_streams, int pitch, int hist, int wstart, const int32_t *d_index, int usable,
                              int rows_out, const float *d_gain, int wp, int batch, float *d_packed, hipStream_t stream);

// LDS image geometry for a window of `window` samples; false if it cannot fit.
bool fast_plan(int window, int usable, int fpi, int image_bytes, FastPlan *plan);
int fast_image_bytes(int nw);
// fpi in {1,2} frames per item; ppw in {2,4,8} pixels per wave (8 only with fpi 1)
// nw: 8 = 8-wave workgroups (two per CU, single image); 32 = double-buffered 16-wave workgroup, one
// per CU; 24 = double-buffered 12-wave workgroups, two per CU (fpi 1 only for 24 and 32)
hipError_t launch_das_fast(const FastArgs &a, int fpi, int ppw, int nw, const Extents &have, hipStream_t stream);
bool fast_db_fits(const FastPlan &plan);

// exact-order kernel (AWPU_MATH_F32_EXACT): sub, fma, add per sample, mics in order; with
// bf16_accumulator (AWPU_MATH_BF16_ACC) the running sums are rounded to bf16 after every mic.
hipError_t launch_das_exact(const SweepArgs &a, bool bf16_accumulator, hipStream_t stream);

// FIR8 variant: LutEntry.frac holds the coefficient-row index k as an int bit pattern; window
// must cover off + 263.  d_coeffs = [101][8] floats on the device.
hipError_t launch_das_fir8(const SweepArgs &a, const float *d_coeffs, hipStream_t stream);

// populateHeatmap on the device: d_peak[batch] receives (or, if peak_given, supplies) the per-frame
// maximum; d_pix[batch][n] the 8-bit image.
hipError_t launch_heatmap(const float *d_power, int n, int batch, float *d_peak, bool peak_given, uint8_t *d_pix,
                          hipStream_t stream);

// Few-beam delay-and-sum (Particle::beam / Particle::das): entries [n_dir][usable] with off_rel = float
// offset of X[off] from `frame` (row start + off); d_power [n_dir], d_beams [n_dir][256] or null.
hipError_t launch_das_beams(const float *d_frame, const LutEntry *d_entries, int usable, int n_dir, float *d_power,
                            float *d_beams, hipStream_t stream);

// geometry_host.cpp: the per-pixel half of computeDelayLUT for the device builder (rot [row_count * columns][12])
void pixel_rotations(int rows, int columns, float fov_deg, int row_begin, int row_count, float *rot);
float samples_per_metre();  // (float) (48828 / 340), antenna.cpp:90

// computeDelayLUT's P x n part on the device (das_kernels.hip, delay_table_kernel): d_rot [n_pixels][12] = Rz(phi)
// row-major + row z of Ry(-theta), d_xyz [3][n]; d_off / d_frac [n_pixels][n]
hipError_t launch_delay_table(const float *d_xyz, int n, const float *d_rot, int n_pixels, float scale, int32_t *d_off, float *d_frac,
                              hipStream_t stream);

// mean square of `hist` samples of each of `n` rows (pitch floats apart), summed in sample order
// (calibration, aw_processing_unit.cpp:133-143)
hipError_t launch_stream_power(const float *d_rows, int pitch, int hist, int n, float *d_out, hipStream_t stream);

// One axis of the 8-bit bilinear upscale: for output coordinate d, the first source coordinate
// (may be -1 / size-1: the kernel clamps) and the two 11-bit weights (w0 + w1 = 2048).
struct ResizeTap {
    int32_t src;
    int16_t w0, w1;
};
// fills taps[dsize]; `zero_frac_at_border`: columns zero the fraction at the borders, rows only clamp
void resize_taps(int ssize, int dsize, bool zero_frac_at_border, ResizeTap *taps);
// one output pixel from the two horizontally interpolated sums
inline __host__ __device__ uint8_t resize_combine(int S0, int S1, int b0, int b1) {
    return (uint8_t) ((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}

// d_src [batch][srows][scols] u8 -> d_dst [batch][drows][dcols] u8, or [batch][drows][dcols][3] through
// d_colormap[256][3] when it is not null.  d_taps = dcols column taps followed by drows row taps.
hipError_t launch_upscale(const uint8_t *d_src, int srows, int scols, int batch, const ResizeTap *d_taps,
                          const uint8_t *d_colormap, uint8_t *d_dst, int drows, int dcols, hipStream_t stream);

// one block of 256 wire datagrams -> floats in the device ring [n_sensors][2048] at column pos (and pos+1024)
hipError_t launch_unpack_block(const void *d_datagrams, int stride_bytes, int n_sensors, float *d_ring, int pos,
                               hipStream_t stream);

// n floats (a multiple of 4, both pointers 16-byte aligned) from PINNED HOST memory to device memory by a kernel: the one-frame host
// call's upload (75 KB at the reference's shape) -- a DMA engine copy of that size is 8-9 us of start-up, a kernel reading over PCIe 3
hipError_t launch_upload_floats(const float *h_pinned, float *d_dst, size_t n, hipStream_t stream);

// LDS bytes the exact kernel asks for with the given window; 0 if the window cannot fit.
size_t das_exact_lds_bytes(int window, int usable, int *chunk_out);

}  // namespace awpu
