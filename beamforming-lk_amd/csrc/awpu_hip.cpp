// awpu_hip.cpp -- the C ABI of libawpu_hip.so (include/awpu_hip.h): handle lifetime, table
// packing, frame upload, kernel dispatch.  No CPU fallback: every compute entry point ends
// in a gfx950 kernel launch or an error status.
#include "awpu_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <array>
#include <map>
#include <vector>

#include "das_kernels.h"

namespace {

thread_local std::string g_last_error;
// the handle the calling thread is working on: errors are also kept in the handle, so that a thread
// other than the one that ran into the error (a GUI thread asking about a worker's engine) can read them
thread_local awpu_hip *g_ctx = nullptr;
void note_error(const std::string &text);

struct CtxScope {
    awpu_hip *saved;
    explicit CtxScope(awpu_hip *h) : saved(g_ctx) { g_ctx = h; }
    ~CtxScope() { g_ctx = saved; }
};
#define AWPU_CTX(h) CtxScope ctx_scope_(h)

int hip_fail(hipError_t e, const char *what) {
    note_error(std::string(what) + ": " + hipGetErrorString(e));
    return AWPU_ERR_HIP;
}

#define AWPU_HIP_TRY(call)                               \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

int invalid(const char *why) {
    note_error(why);
    return AWPU_ERR_INVALID;
}

int fail(int status, const char *why) {
    note_error(why);
    return status;
}

template <class T>
void dev_free(T *&p) {  // hipFree + forget
    if (p) (void) hipFree(p);
    p = nullptr;
}

}  // namespace

struct awpu_hip {
    awpu_hip_cfg cfg{};
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool timing = true;

    // host copies of what the reference keeps in MIMOWorker / Antenna
    std::vector<int32_t> off;   // [pixel_count][lut_stride]  offsetDelays
    std::vector<float> frac;    // [pixel_count][lut_stride]  fractionalDelays
    std::vector<int32_t> index; // [usable]                   antenna.index
    std::vector<float> gain;    // [n_streams] optional per-mic gain (awpu_hip_set_mic_gains); empty = none
    bool have_table = false, have_mics = false, prepared = false;

    // device state
    awpu::LutEntry *d_lut = nullptr;
    struct FastLut {
        awpu::FastPlan plan;
        awpu::FastEntry *d = nullptr;
        size_t entries = 0;  // allocated (the launchers check their kernel's reach against it: das_kernels.h, Extents)
    };
    std::vector<FastLut> fast_luts;  // one per (frames per item, LDS image size) in use
    awpu::FastEntry *d_exact_pair_lut = nullptr;  // reference-order sweep on the frame-pair layout (das_exact_pair_kernel)
    size_t exact_pair_lut_entries = 0, fir_plane_lut_entries = 0;  // allocated entries of the tables below and above ...
    size_t quad_lut_entries[8] = {0, 0, 0, 0, 0, 0, 0, 0};                  // ... and of the quad-major tables, by QuadLayout
    awpu::QuadEntry *d_exact_quad_lut = nullptr;  // ... four vertically adjacent pixels per wave (das_exact_quad_kernel): quad-major, raw fractions
    awpu::FastPlan exact_plan{};
    unsigned *d_nd_queue = nullptr;               // das_exact_nd_kernel's eight item counters (one per XCD)
    int2 *d_nd_items = nullptr;                   // ... and its item list (nd_items_kernel), valid for nd_items_key
    size_t nd_items_cap = 0;
    long long nd_items_key = -1;                  // (n_pairs, pair group, quads per wave) the list was built for; -1: none
    int n_cus = 0;                                // compute units of the handle's device (persistent workgroups: one per CU)
    awpu::QuadEntry *d_exact_nd_lut = nullptr;    // ... on the {next, d} layout (das_exact_nd_kernel): 16-byte elements, quad rows padded to an even count
    awpu::FastPlan exact_nd_plan{};
    bool exact_nd_ok = false;     // ... and the window fits the {next, d} image
    awpu::QuadEntry *d_exact_ndh_lut = nullptr, *d_exact_ndhs_lut = nullptr;  // single frames: the halves form of that layout, chunked / every mic resident
    awpu::FastPlan exact_ndh_plan{}, exact_ndhs_plan{};
    bool exact_ndh_ok = false, exact_ndhs_ok = false, fast_ndp_ok = false;
    bool exact_pairs_ok = false;  // AWPU_MATH_F32_EXACT + LERP and the window fits the pair image
    float *sums_out = nullptr;    // awpu_hip_process_device_sums: where the launch in progress exports out[] (else null)
    awpu::QuadEntry *d_quad_lut = nullptr;  // quad-major table of the quad shape (das_quad_kernel)
    awpu::QuadEntry *d_quadh_lut = nullptr; // the same with the halves layout's LDS addresses (das_quadh_kernel)
    awpu::QuadEntry *d_quadhs_lut = nullptr; // the same with slot = mic (das_quadh_stationary_kernel: every mic's row resident)
    void *d_fir_plane_lut = nullptr;           // FIR8 on the four-plane layout: one dword per (pixel, mic): address, plane, coefficient row
    awpu::FastPlan fir_plane_plan{};
    std::vector<float> fir;                    // host copy of the [101][8] coefficient table (baked into the plane entries)
    awpu::FastPlan quad_plan{}, quadh_plan{}, quadhs_plan{};
    bool quadh_fits = false;      // single frames on the halves layout (das_quadh_kernel)
    bool quadhs_fits = false;     // ... with every active mic's row in LDS at once (das_quadh_stationary_kernel: one 8x8 array does)
    bool quad_ok = false;         // the table's statistics favour the quad shape (decided in prepare)
    double quad_cost = 0.0;       // its expected packed VALU instructions per quad and mic (32 = no sharing at all)
    double quad_differ = 3.0;     // pixels of a vertical quad (of three) whose integer delay differs from the second pixel's, per mic (table sample)
    int32_t *d_index = nullptr;
    float *d_gain = nullptr;  // [usable] gains in active-mic order, or null
    float *d_calib = nullptr; // [64] per-mic mean squares (calibration)
    awpu::LutEntry *d_beam_lut = nullptr;  // [beam_cap][usable] entries of awpu_hip_beams
    float *d_beam_out = nullptr;           // [beam_cap] powers then [beam_cap][256] beams
    size_t beam_cap = 0, beam_lut_cap = 0;
    float *d_fir = nullptr;  // [101][8] coefficient table (AWPU_INTERP_FIR8)
    float *d_ring = nullptr;            // [n_streams][2048] history ring (awpu_hip_ingest_block)
    uint8_t *d_display = nullptr;       // awpu_hip_live_block: peak (one float), compact image, upscaled image
    size_t display_cap = 0;             // bytes
    awpu::ResizeTap *d_taps = nullptr;  // column + row taps of the display upscale, for taps_key
    int taps_key[4] = {0, 0, 0, 0};     // {srows, scols, drows, dcols}
    float *d_pack = nullptr;            // [pairs][usable][wp][2] sample-interleaved frame pairs
    size_t pack_cap = 0;                // floats
    unsigned char *d_datagrams = nullptr;  // staging for one block of wire datagrams
    int32_t *d_row_off_ring = nullptr;  // row offsets for frames read out of the ring (pitch 2048)
    int ring_pos = 0;                   // where the next block goes = start of the snapshot
    // awpu_hip_live_block as a HIP graph: the call's copies and launches captured once per (ring position, caller
    // buffers, table generation) and replayed with one hipGraphLaunch
    struct LiveGraph {
        int ring_pos, stride, rows, cols, out_rows, out_cols;
        const void *datagrams, *power, *image, *colormap, *big_image;
        unsigned long long gen;
        hipGraphExec_t exec;
        unsigned long long last_use;  // live_clock at the last replay: the least recently used graph is evicted
    };
    std::vector<LiveGraph> live_graphs;
    unsigned long long table_gen = 0;   // bumped whenever prepare() rebuilds the device tables
    int live_warm = 0;                  // plain live calls made with the current tables AND this call shape (lazy allocations done after one)
    unsigned long long live_shape = 0;  // the shape those calls had: image sizes, which outputs, colour table or not
    const void *live_bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // ... and the caller's buffers (a capture bakes them in)
    unsigned long long live_clock = 0;
    bool live_graph_broken = false;     // a capture failed on this runtime: never try again
    bool have_fir = false;
    int32_t *d_row_off = nullptr;
    int32_t *d_row_off_compact = nullptr;  // the same for frames uploaded as [streams][compact_hist] windows
    int compact_hist = 0;                  // 0 = the window cannot be cut out (it touches the newest sample)
    float *d_frames = nullptr;
    float *h_live_in = nullptr, *h_live_out = nullptr;  // pinned staging of awpu_hip_process's one-frame calls (the live path): window in, powers out
    size_t live_in_cap = 0, live_out_cap = 0;           // in floats
    unsigned live_calls = 0;                            // ... how many of them this handle has served (every 32nd is timed by events)
    // ... their completion flag (the resident single-frame kernels with one quad per wave): the device counter the workgroups count themselves on, what it will read
    // when every launch armed so far is over, the pinned flag and the sequence number of the last armed launch
    unsigned long long *d_done_counter = nullptr, done_total = 0;
    unsigned *h_done_flag = nullptr, done_seq = 0;
    bool done_arm = false, done_used = false;           // arm: the next single-frame sweep is to raise the flag; used: it will
    float *d_power = nullptr;
    size_t frames_cap = 0, power_cap = 0;  // in floats
    int wstart = 0, window = 0, tau_max = 0;
    int pair_cols = 0;  // frame-pair sweep: > 0 = waves take vertically adjacent pixels (grid row length), 0 = consecutive

    // device group (cfg.n_devices > 1): this handle owns no sweep state of its own, only one part per device
    std::vector<awpu_hip *> parts;
    // a part's pixels inside the group's range: (first pixel relative to the group's pixel_begin, count), ascending; the part's
    // own table and power rows hold them back to back.  One range = a contiguous slab; several = row groups of four dealt
    // round-robin over the devices (edge rows of the sine-space grid cost the quad shapes more than centre rows: DESIGN.md 6)
    std::vector<std::pair<int, int>> ranges;
    bool union_window_done = false;  // group: every part stages the union of the parts' windows (packed frames need one layout)
    hipEvent_t ev_fan = nullptr;            // group: recorded on the caller's stream, awaited by every part
    // a part's share of the fan-out (awpu_hip_process_device on a group): two window buffers, so that the copy of
    // call k+1 (on copy_stream) runs beside the sweep of call k (on stream)
    hipStream_t copy_stream = nullptr;
    float *d_fan[2] = {nullptr, nullptr};
    size_t fan_cap = 0;                     // floats per buffer
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_swept[2] = {nullptr, nullptr}, ev_done = nullptr;
    unsigned fan_turn = 0;
    // how a part of a group reaches devices[0]: kPeerSame (the same GPU), kPeerDirect (peer copies over xGMI) or
    // kPeerStaged (no peer access on this node: the window and the tiles cross pinned host memory, explicitly)
    int peer = 0;
    float *h_stage[2] = {nullptr, nullptr};  // group: pinned staging of the frames' window for the staged parts
    size_t stage_cap = 0;                    // floats per buffer
    int stage_lo = 0, stage_w = 0;           // group: the window [stage_lo, stage_lo + stage_w) of every stream that is staged
    hipEvent_t ev_staged[2] = {nullptr, nullptr};   // group: window b is in h_stage[b]
    unsigned stage_turn = 0;
    float *h_tile[2] = {nullptr, nullptr};   // part (staged): pinned staging of its power tile on the way back
    size_t tile_cap = 0;
    hipEvent_t ev_tile_free[2] = {nullptr, nullptr};  // part (staged): the caller's stream has read h_tile[b]
    hipEvent_t ev_staged_read[2] = {nullptr, nullptr};  // part (staged): its upload out of the group's h_stage[b] is done
    bool stage_used[2] = {false, false};
    bool tile_used[2] = {false, false}, fan_used[2] = {false, false};
    bool in_flight = false;                 // awpu_hip_process_async without its awpu_hip_wait yet

    awpu_hip_stats stats{};
    std::string last_error;                  // awpu_hip_last_error_of
    unsigned long long *d_diag = nullptr;    // AWPU_FAST_DEBUG=16 cycle stamps of the last launch
    size_t diag_cap = 0;                     // in 64-bit words

    int usable() const { return static_cast<int>(index.size()); }
};

namespace {

void note_error(const std::string &text) {
    g_last_error = text;
    if (g_ctx) g_ctx->last_error = text;
}

// diagnostics buffer of `words` 64-bit words, grown on demand (one per handle: handles on different
// devices, or launches of different sizes, must not share it)
int ensure_diag(awpu_hip *h, size_t words) {
    if (h->diag_cap >= words) return AWPU_OK;
    dev_free(h->d_diag);
    h->diag_cap = 0;
    AWPU_HIP_TRY(hipMalloc(&h->d_diag, words * sizeof(unsigned long long)));
    h->diag_cap = words;
    return AWPU_OK;
}

// What the process environment can change.  The SHIPPING library reads three variables, none of them needed in production:
//   AWPU_SHAPE             force one of the production sweep shapes wherever it can serve the call (tests sweep every shape
//                          through the oracle this way; the default rule -- launch() below -- picks by table statistics and launch size):
//                          pair | pair_vertical | pair_horizontal | quad | noquad | stationary | quadh | quadh_chunked | single_db | single_small |
//                          fir8_planes | exact_pair | exact_quad | exact_nd1 | exact_nd2 | exact_ndp | exact_verify
//   AWPU_LIVE_GRAPH=0      awpu_hip_live_block always enqueues its steps one by one (no HIP-graph replay)
//   AWPU_GROUP_FORCE_COPY  device groups: 1 = a part on devices[0] takes the window-copy path too, 2 = through pinned host
//                          memory (how one GPU exercises the paths a part on another GPU takes)
// Everything else -- chunk geometry, XCD pair groups, persistent workgroups, priority variants, cycle stamps -- exists only in
// builds with -DAWPU_TUNING_BUILD (AWPU_EXTRA_HIPCC_FLAGS; -DAWPU_TIMING_BUILD implies it), which read the round-1..3 variables
// (AWPU_FAST_*, AWPU_FIR8_*, AWPU_QUAD_VARIANT, AWPU_EXACT_PAIRS) as before.  Read once per process.
struct EnvKnobs {
    int fpi = 0, ppw = 0, nw = 0;  // single-frame shape forced: frames per item (always 1 here), pixels per wave, 8 / 32
    int pairs = -1;                // 0 / 1: never / always a frame-pair sweep (quad, stationary or pair shape) for batches >= 2
    int debug = 0;                 // AWPU_FAST_DEBUG bits (tuning builds)
    int fpw = 0;                   // tuning: consecutive frames per workgroup of the double-buffered single-frame shape
    int quads = -1;                // 0 / 1: never / always (where the row length is known) the quad shapes
    int pair_group = 0;            // tuning: frame pairs an XCD works on at a time (quad shape)
    int quad_variant = 0;          // tuning: block variant (AWPU_QUAD_VARIANT)
    int stationary = -1;           // 0 / 1: never / always (where the window fits the LDS) the stationary pair shape
    int fir_planes = 1;            // 2: the four-plane FIR8 kernel for every batch >= 2 however small the grid
    int fir_share = 1;             // tuning: 0 = the FIR8 plane kernel sweeps four consecutive pixels even where the row length is known
    int wgs = 0;                   // tuning: persistent workgroups of the quad shape (0 = default rule, -1 = one workgroup per item)
    int live_graph = 1;            // AWPU_LIVE_GRAPH
    int halves = -1;               // 1: single frames on the halves layout for every call (where the quad table is built)
    int exact_pairs = 1;           // 0: AWPU_MATH_F32_EXACT on the round-1 verification kernel (das_exact_kernel)
    int pair_cols = -1;            // 0 / 1: the pair shape pairs consecutive / vertically adjacent pixels (default: whichever coincides more)
    int group_copy = 0;            // AWPU_GROUP_FORCE_COPY
    EnvKnobs() {
        if (const char *v = std::getenv("AWPU_GROUP_FORCE_COPY")) group_copy = std::atoi(v);
        if (const char *v = std::getenv("AWPU_LIVE_GRAPH")) live_graph = std::atoi(v);
        if (const char *v = std::getenv("AWPU_SHAPE")) {
            const std::string shape(v);
            if (shape == "pair" || shape == "pair_vertical" || shape == "pair_horizontal") {
                pairs = 1, quads = 0, stationary = 0;
                if (shape != "pair") pair_cols = shape == "pair_vertical";
            } else if (shape == "quad") quads = 1;
            else if (shape == "noquad") quads = 0;
            else if (shape == "stationary") pairs = 1, quads = 0, stationary = 1;
            else if (shape == "quadh") quads = 1, pairs = 0, halves = 1;
            else if (shape == "quadh_chunked") quads = 1, pairs = 0, halves = 1, stationary = 0;  // never the resident-window variant
            else if (shape == "single_db") pairs = 0, quads = 0, fpi = 1, ppw = 8, nw = 32;
            else if (shape == "single_small") pairs = 0, quads = 0, fpi = 1, ppw = 2, nw = 8;
            else if (shape == "fir8_planes") fir_planes = 2;
            else if (shape == "exact_verify") exact_pairs = 0;
            else if (shape == "exact_pair") exact_pairs = 2;  // the two-pixel reference-order block even where quads would run
            else if (shape == "exact_quad") exact_pairs = 3;  // round 4's quad kernel on raw sample pairs (cur - next per pixel)
            else if (shape == "exact_nd1") exact_pairs = 4;   // the {next, d} kernel with one quad per wave
            else if (shape == "exact_nd2") exact_pairs = 5;   // ... with two
            else if (shape == "exact_ndp") exact_pairs = 6;   // single frames: one pixel per wave (das_exact_ndp_kernel) wherever its rows can be chunked
            else std::fprintf(stderr, "libawpu_hip: AWPU_SHAPE=%s is not a shape of this build; ignored\n", v);
        }
#ifdef AWPU_TUNING_BUILD
        if (const char *v = std::getenv("AWPU_FAST_QUADS")) quads = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FAST_PAIRGROUP")) pair_group = std::atoi(v);
        if (const char *v = std::getenv("AWPU_QUAD_VARIANT")) quad_variant = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FAST_HALVES")) halves = std::atoi(v);
        if (const char *v = std::getenv("AWPU_EXACT_PAIRS")) exact_pairs = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FAST_WGS")) wgs = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FIR8_PLANES")) fir_planes = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FIR8_SHARE")) fir_share = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FAST_STATIONARY")) stationary = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FAST_PAIRCOLS")) pair_cols = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FAST_VARIANT"))
            if (std::sscanf(v, "%d,%d,%d", &fpi, &ppw, &nw) < 2) fpi = ppw = nw = 0;
        if (const char *v = std::getenv("AWPU_FAST_PAIRS")) pairs = std::atoi(v);
        if (const char *v = std::getenv("AWPU_FAST_DEBUG")) debug = std::atoi(v);
#ifndef AWPU_TIMING_BUILD
        debug &= awpu::kDebugSafeBits;  // the wrong-result timing switches exist only with -DAWPU_TIMING_BUILD (das_kernels.h)
#endif
        if (const char *v = std::getenv("AWPU_FAST_FPW")) fpw = std::atoi(v);
#endif
    }
};
const EnvKnobs &env() {
    static const EnvKnobs knobs;  // initialised once, thread-safe
    return knobs;
}

// The captured live-block graphs hold raw device pointers (d_power, d_display, d_taps, d_ring, tables, d_pack):
// whoever frees or reallocates one of those retires the graphs first.  The next live calls run step by step and
// capture again once the buffers have settled.
void retire_live_graphs(awpu_hip *h) {
    for (auto &g : h->live_graphs) (void) hipGraphExecDestroy(g.exec);
    h->live_graphs.clear();
    h->live_warm = 0;
}

// grow-only device buffer shared by the sweep shapes that pack frames (pairs, quads, FIR8 planes)
int ensure_pack(awpu_hip *h, size_t need) {
    if (h->pack_cap >= need) return AWPU_OK;
    retire_live_graphs(h);
    dev_free(h->d_pack);
    h->pack_cap = 0;
    AWPU_HIP_TRY(hipMalloc(&h->d_pack, need * sizeof(float)));
    h->pack_cap = need;
    return AWPU_OK;
}

void release_device(awpu_hip *h) {
    retire_live_graphs(h);
    dev_free(h->d_lut);
    for (auto &l : h->fast_luts) dev_free(l.d);
    h->fast_luts.clear();
    dev_free(h->d_exact_pair_lut);
    dev_free(h->d_exact_quad_lut);
    dev_free(h->d_nd_items);
    h->nd_items_cap = 0;
    h->nd_items_key = -1;
    dev_free(h->d_nd_queue);
    dev_free(h->d_done_counter);
    if (h->h_done_flag) (void) hipHostFree(h->h_done_flag);
    h->h_done_flag = nullptr;
    h->done_total = 0;
    dev_free(h->d_exact_nd_lut);
    dev_free(h->d_exact_ndh_lut);
    dev_free(h->d_exact_ndhs_lut);
    dev_free(h->d_quad_lut);
    dev_free(h->d_quadh_lut);
    dev_free(h->d_quadhs_lut);
    dev_free(h->d_fir_plane_lut);
    dev_free(h->d_index);
    dev_free(h->d_gain);
    dev_free(h->d_calib);
    dev_free(h->d_beam_lut);
    dev_free(h->d_beam_out);
    dev_free(h->d_fir);
    dev_free(h->d_ring);
    dev_free(h->d_pack);
    dev_free(h->d_taps);
    dev_free(h->d_display);
    h->display_cap = 0;
    dev_free(h->d_datagrams);
    dev_free(h->d_row_off_ring);
    dev_free(h->d_row_off);
    dev_free(h->d_row_off_compact);
    dev_free(h->d_frames);
    dev_free(h->d_power);
    dev_free(h->d_diag);
    h->diag_cap = 0;
    dev_free(h->d_fan[0]);
    dev_free(h->d_fan[1]);
    h->fan_cap = 0;
    for (int b = 0; b < 2; b++) {
        if (b == 0) {
            if (h->h_live_in) (void) hipHostFree(h->h_live_in);
            if (h->h_live_out) (void) hipHostFree(h->h_live_out);
            h->h_live_in = h->h_live_out = nullptr;
            h->live_in_cap = h->live_out_cap = 0;
        }
        if (h->h_stage[b]) (void) hipHostFree(h->h_stage[b]);
        if (h->h_tile[b]) (void) hipHostFree(h->h_tile[b]);
        h->h_stage[b] = h->h_tile[b] = nullptr;
    }
    h->stage_cap = h->tile_cap = 0;
    h->beam_cap = h->beam_lut_cap = h->pack_cap = h->frames_cap = h->power_cap = 0;
}

// Pack the reference-format tables into the kernels' layout once both the tables and the
// active-mic list are known.  Validates that no entry reads outside the frame history:
// delay() reads signal[0..256] from &signals[s][offset] (delay.cpp:19-22).
int prepare(awpu_hip *h) {
    const auto &c = h->cfg;
    const int U = h->usable();
    const int P = c.pixel_count;
    int lo = c.hist, hi = -1;
    for (int p = 0; p < P; p++) {
        const int32_t *row = &h->off[(size_t) p * c.lut_stride];
        for (int s = 0; s < U; s++) {
            const int o = row[h->index[s]];
            lo = std::min(lo, o);
            hi = std::max(hi, o);
        }
    }
    const int reach = c.interp == AWPU_INTERP_FIR8 ? awpu::kSamples + 6 : awpu::kSamples;  // last sample read past off
    if (lo < 0 || hi + reach > c.hist - 1) {
        return fail(AWPU_ERR_RANGE, "delay table entry reads outside the frame history");
    }
    if (c.window_end > c.window_begin) {  // a wider window asked for (ranks that exchange packed frames stage the union)
        lo = std::min(lo, c.window_begin);
        hi = std::max(hi, c.window_end - reach - 1);
    }
    h->wstart = lo;
    h->window = hi - lo + reach + 1;
    h->tau_max = awpu::kSamples - lo;

    // (hipFree waits for the device: launches still reading the old tables finish first)
    dev_free(h->d_lut);
    dev_free(h->d_index);
    for (auto &l : h->fast_luts) dev_free(l.d);
    h->fast_luts.clear();
    dev_free(h->d_exact_pair_lut);
    dev_free(h->d_exact_quad_lut);
    dev_free(h->d_exact_nd_lut);
    dev_free(h->d_exact_ndh_lut);
    dev_free(h->d_exact_ndhs_lut);
    dev_free(h->d_quad_lut);
    dev_free(h->d_quadh_lut);
    dev_free(h->d_quadhs_lut);
    dev_free(h->d_fir_plane_lut);
    AWPU_HIP_TRY(hipMalloc(&h->d_index, (size_t) U * sizeof(int32_t)));
    AWPU_HIP_TRY(hipMemcpy(h->d_index, h->index.data(), (size_t) U * sizeof(int32_t),
                           hipMemcpyHostToDevice));
    dev_free(h->d_gain);
    if (!h->gain.empty()) {
        std::vector<float> compact(U);
        for (int s = 0; s < U; s++) compact[s] = h->gain[h->index[s]];
        AWPU_HIP_TRY(hipMalloc(&h->d_gain, (size_t) U * sizeof(float)));
        AWPU_HIP_TRY(hipMemcpy(h->d_gain, compact.data(), (size_t) U * sizeof(float), hipMemcpyHostToDevice));
    }
    {   // float offset, inside one frame, of staged row 2*s+q (copy q of active mic s)
        dev_free(h->d_row_off);
        const int upad = (U + 3) & ~3;
        std::vector<int32_t> ro((size_t) 2 * upad + 8, h->index[0] * c.hist + lo);
        for (int s = 0; s < U; s++)
            for (int q = 0; q < 2; q++) ro[2 * s + q] = h->index[s] * c.hist + lo + q;
        AWPU_HIP_TRY(hipMalloc(&h->d_row_off, ro.size() * sizeof(int32_t)));
        AWPU_HIP_TRY(hipMemcpy(h->d_row_off, ro.data(), ro.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        // Host-buffer calls upload only the window [lo, lo + compact_hist) of every stream (the rest
        // of the 1024-sample snapshot is never read: SURVEY 8a A10): a third of the PCIe bytes.
        dev_free(h->d_row_off_compact);
        const int ch = ((h->window + 3) & ~3) + 4;
        h->compact_hist = lo + ch <= c.hist ? ch : 0;
        if (c.hist == AWPU_HIST) {  // frames read in place from the ingest ring: rows 2048 floats apart
            dev_free(h->d_row_off_ring);
            std::vector<int32_t> rr(ro.size(), h->index[0] * 2048 + lo);
            for (int s = 0; s < U; s++)
                for (int q = 0; q < 2; q++) rr[2 * s + q] = h->index[s] * 2048 + lo + q;
            AWPU_HIP_TRY(hipMalloc(&h->d_row_off_ring, rr.size() * sizeof(int32_t)));
            AWPU_HIP_TRY(hipMemcpy(h->d_row_off_ring, rr.data(), rr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if (h->compact_hist) {
            for (int s = 0; s < U; s++)
                for (int q = 0; q < 2; q++) ro[2 * s + q] = h->index[s] * h->compact_hist + q;
            for (size_t i = 2 * (size_t) U; i < ro.size(); i++) ro[i] = h->index[0] * h->compact_hist;
            AWPU_HIP_TRY(hipMalloc(&h->d_row_off_compact, ro.size() * sizeof(int32_t)));
            AWPU_HIP_TRY(hipMemcpy(h->d_row_off_compact, ro.data(), ro.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }

    h->exact_pairs_ok = c.math == AWPU_MATH_F32_EXACT && c.interp == AWPU_INTERP_LERP && awpu::pair_plan(h->window, U, &h->exact_plan);
    h->exact_nd_ok = h->exact_pairs_ok && awpu::exact_nd_plan(h->window, U, &h->exact_nd_plan);
    h->exact_ndh_ok = h->exact_pairs_ok && awpu::exact_ndh_plan(h->window, U, false, &h->exact_ndh_plan);
    // (AWPU_MATH_F32_FAST sweeps single frames on small grids with the reference-order pixel-per-wave kernel too -- launch() -- : the
    // same plan, the same table)
    h->fast_ndp_ok = c.math == AWPU_MATH_F32_FAST && c.interp == AWPU_INTERP_LERP && awpu::exact_ndh_plan(h->window, U, false, &h->exact_ndh_plan);
    h->exact_ndhs_ok = h->exact_pairs_ok && awpu::exact_ndh_plan(h->window, U, true, &h->exact_ndhs_plan);
    if (c.math != AWPU_MATH_F32_FAST || c.interp == AWPU_INTERP_FIR8) {
        int chunk = 0;
        if (awpu::das_exact_lds_bytes(h->window, U, &chunk) == 0)
            return invalid("delay window does not fit the LDS budget");
        std::vector<awpu::LutEntry> packed((size_t) P * U);
        for (int p = 0; p < P; p++) {
            const int32_t *orow = &h->off[(size_t) p * c.lut_stride];
            const float *frow = &h->frac[(size_t) p * c.lut_stride];
            awpu::LutEntry *dst = &packed[(size_t) p * U];
            for (int s = 0; s < U; s++) {
                const int id = h->index[s];
                dst[s].off_rel = orow[id] - lo;
                dst[s].frac = frow[id];
                if (c.interp == AWPU_INTERP_FIR8) {  // delay.cpp:32-33: the coefficient row
                    const float get_filter = frow[id] * 100.0f + 0.5f;
                    const int32_t k = (int32_t) get_filter;
                    std::memcpy(&dst[s].frac, &k, sizeof(k));
                }
            }
        }
        AWPU_HIP_TRY(hipMalloc(&h->d_lut, packed.size() * sizeof(awpu::LutEntry)));
        AWPU_HIP_TRY(hipMemcpy(h->d_lut, packed.data(), packed.size() * sizeof(awpu::LutEntry),
                               hipMemcpyHostToDevice));
    } else {
        awpu::FastPlan plan;
        if (!awpu::fast_plan(h->window, U, 1, awpu::kFastLdsBytes, &plan))
            return invalid("delay window does not fit the LDS budget");
    }

    // Pairing of pixels inside a wave of the frame-pair sweep: the shared-read block saves a mic's LDS reads
    // when the two pixels' integer delays coincide.  With the grid's row length known, compare consecutive
    // pixels against vertically adjacent ones on a sample of the table and take the better.
    h->pair_cols = 0;
    {
        const int cols = c.grid_columns;
        if (cols > 0 && P % cols == 0 && c.pixel_begin % cols == 0 && P / cols >= 2) {
            long same_h = 0, same_v = 0, seen = 0;
            const int step = std::max(1, (P - cols) / 4096);
            for (int p = 0; p + cols < P; p += step) {
                if ((p % cols) + 1 >= cols) continue;
                const int32_t *o0 = &h->off[(size_t) p * c.lut_stride];
                const int32_t *oh = &h->off[(size_t) (p + 1) * c.lut_stride];
                const int32_t *ov = &h->off[(size_t) (p + cols) * c.lut_stride];
                for (int s = 0; s < U; s++) {
                    const int id = h->index[s];
                    same_h += o0[id] == oh[id];
                    same_v += o0[id] == ov[id];
                    seen++;
                }
            }
            if (seen > 0 && same_v > same_h) h->pair_cols = cols;
            if (env().pair_cols >= 0) h->pair_cols = env().pair_cols ? cols : 0;  // AWPU_SHAPE=pair_vertical / pair_horizontal: tests force either
        }
    }

    // The quad shape (das_quad_kernel) shares arithmetic between four vertically adjacent pixels wherever their
    // integer delays coincide with the second pixel's: 20 packed VALU instructions per quad and mic, +8 for every
    // pixel that differs (-4 where the third and fourth differ together), against 32 without sharing.  Count it on a sample of the table; take the shape when it
    // saves VALU work at all once its own address adds are counted (measured: a count of 30.2 -- BASELINE c2 -- is
    // 4 % faster than the pair shape, 29.4 -- c3 -- 8 %, 25.3 -- the headline -- 20 %; AWPU_FAST_QUADS=0/1 forces either).
    h->quad_ok = false;
    h->quadh_fits = false;
    h->quadhs_fits = false;
    h->quad_cost = 0.0;
    {
        const int cols = c.grid_columns;
        h->quad_differ = 3.0;
        if (c.interp == AWPU_INTERP_LERP && cols > 0 && P % cols == 0 && c.pixel_begin % cols == 0) {
            const int rows = P / cols;
            long differ = 0, together = 0, seen = 0;  // `together`: pixels 2 and 3 away from the reference as one (4 less)
            const int n_quads = ((rows + 3) / 4) * cols;
            const int step = std::max(1, n_quads / 2048);
            for (int q = 0; q < n_quads; q += step) {
                const int r0 = (q / cols) * 4, col = q % cols;
                const int32_t *o[4];
                for (int k = 0; k < 4; k++) o[k] = &h->off[((size_t) std::min(r0 + k, rows - 1) * cols + col) * c.lut_stride];
                for (int s = 0; s < U; s++) {
                    const int id = h->index[s];
                    differ += (o[0][id] != o[1][id]) + (o[2][id] != o[1][id]) + (o[3][id] != o[1][id]);
                    together += o[2][id] != o[1][id] && o[2][id] == o[3][id];
                }
                seen += U;
            }
            h->quad_cost = seen ? 20.0 + (8.0 * (double) differ - 4.0 * (double) together) / (double) seen : 32.0;
            h->quad_differ = seen ? (double) differ / (double) seen : 3.0;  // pixels of a quad (of three) that leave the reference pixel's address, per mic
            const bool fast = c.math == AWPU_MATH_F32_FAST;
            h->quad_ok = fast && h->quad_cost < 31.0 && awpu::pair_plan(h->window, U, &h->quad_plan);
            if (fast && env().quads >= 0) h->quad_ok = env().quads != 0 && awpu::pair_plan(h->window, U, &h->quad_plan);
            // the halves layout: a row holds the window less 128 samples, as (sample, sample + 128) pairs (its pack pass applies the gains)
            h->quadh_fits = h->quad_ok && awpu::pair_plan(h->window - 128, U, &h->quadh_plan);
            h->quadhs_fits = h->quadh_fits && awpu::quadh_stationary_plan(h->window, U, &h->quadhs_plan);
        }
    }

    auto &st = h->stats;
    st.tau_max = h->tau_max;
    st.window = h->window;
    st.usable = U;
    st.alg_bytes_frame = 4ull * U * h->window + 8ull * P * U + 4ull * P;
    st.alg_flops_frame = 4ull * P * U * awpu::kSamples + 6ull * P * (awpu::kSamples - 2);
    st.kernel_variant = AWPU_KERNEL_NONE;
    h->prepared = true;
    h->table_gen++;  // graphs of awpu_hip_live_block captured against the old tables are stale
    retire_live_graphs(h);
    return AWPU_OK;
}

// The fast kernel's table for `fpi` frames per item: per (pixel, active mic s) the weights and
// the LDS byte address of X[off] inside the staged image (das_fast.hip), rows padded to whole
// groups of four with null entries (zero weights, address of a staged row).
int build_fast_lut(awpu_hip *h, int fpi, int image_bytes, const awpu_hip::FastLut **out) {
    for (const auto &l : h->fast_luts)
        if (l.plan.fpi == fpi && l.plan.image_bytes == image_bytes) {
            *out = &l;
            return AWPU_OK;
        }
    const auto &c = h->cfg;
    const int U = h->usable(), P = c.pixel_count;
    awpu_hip::FastLut lut;
    const bool pairs = image_bytes < 0;  // frame-pair layout: one image row per mic, 8-byte elements
    const bool planned = image_bytes == -2 ? awpu::pair_plan_stationary(h->window, U, &lut.plan)  // every mic resident
                         : pairs         ? awpu::pair_plan(h->window, U, &lut.plan)
                                         : awpu::fast_plan(h->window, U, fpi, image_bytes, &lut.plan);
    if (!planned) return invalid("delay window does not fit the LDS budget");
    const awpu::FastPlan &plan = lut.plan;
    // rows for whole pixel tiles (the kernels sweep every pixel slot of a workgroup; slots past the
    // grid get null rows) + spare groups: the kernels prefetch entries past the row they sweep
    // (with vertical pixel pairs the partner of a pixel in the last row lies one grid row past the table)
    const int P_pad = (P + (pairs ? h->pair_cols : 0) + 127) / 128 * 128;
    const size_t n = (size_t) P_pad * plan.usable_pad + 4 * awpu::kPairTablePrefetch;
    std::vector<awpu::FastEntry> packed(n, awpu::FastEntry{0.0f, 0u, 0.0f, 0u});
    for (int p = 0; p < P; p++) {
        const int32_t *orow = &h->off[(size_t) p * c.lut_stride];
        const float *frow = &h->frac[(size_t) p * c.lut_stride];
        awpu::FastEntry *dst = &packed[(size_t) p * plan.usable_pad];
        for (int s = 0; s < U; s++) {
            const int id = h->index[s];
            const int off_rel = orow[id] - h->wstart;
            const int q = off_rel & 1;
            const int j = s % plan.chunk;  // mic slot inside its chunk
            dst[s].f = frow[id];
            dst[s].g = 1.0f - frow[id];
            if (!h->gain.empty()) {  // the per-mic gain rides on the two weights
                dst[s].f *= h->gain[id];
                dst[s].g *= h->gain[id];
            }
            dst[s].addr = pairs ? (uint32_t) (j * plan.row_bytes + off_rel * 8)
                                : (uint32_t) ((j * 2 + q) * plan.row_bytes + (off_rel - q) * 4);
        }
    }
    AWPU_HIP_TRY(hipMalloc(&lut.d, n * sizeof(awpu::FastEntry)));
    AWPU_HIP_TRY(hipMemcpy(lut.d, packed.data(), n * sizeof(awpu::FastEntry), hipMemcpyHostToDevice));
    lut.entries = n;
    h->fast_luts.reserve(8);
    h->fast_luts.push_back(lut);
    *out = &h->fast_luts.back();
    return AWPU_OK;
}

// The quad shape's table (das_fast.hip, das_quad_kernel): [quad][group of 4 mics][pixel 0..3][mic 0..3] x (f, LDS
// address), quads = groups of four grid rows x columns padded to whole 16-column tiles.  Pixels past the grid
// carry weight 0 and the address of the nearest pixel inside it (they then follow the shared path and add
// nothing); padding mics (usable rounded up to 4) carry weight 0 and the address of their own, zero, row.
enum QuadLayout { kQuadPairs = 0, kQuadExactNd = 1, kQuadHalves = 2, kQuadHalvesStationary = 3, kQuadExact = 4, kQuadExactNdh = 5, kQuadExactNdhStationary = 6 };
int build_quad_lut(awpu_hip *h, int layout) {
    awpu::QuadEntry *&d_lut = layout == kQuadHalves             ? h->d_quadh_lut
                              : layout == kQuadHalvesStationary ? h->d_quadhs_lut
                              : layout == kQuadExact            ? h->d_exact_quad_lut
                              : layout == kQuadExactNd          ? h->d_exact_nd_lut
                              : layout == kQuadExactNdh         ? h->d_exact_ndh_lut
                              : layout == kQuadExactNdhStationary ? h->d_exact_ndhs_lut
                                                                : h->d_quad_lut;
    if (d_lut) return AWPU_OK;
    const auto &c = h->cfg;
    const awpu::FastPlan &plan = layout == kQuadHalves             ? h->quadh_plan
                                 : layout == kQuadHalvesStationary ? h->quadhs_plan
                                 : layout == kQuadExact            ? h->exact_plan
                                 : layout == kQuadExactNd          ? h->exact_nd_plan
                                 : layout == kQuadExactNdh         ? h->exact_ndh_plan
                                 : layout == kQuadExactNdhStationary ? h->exact_ndhs_plan
                                                                   : h->quad_plan;
    const bool halves_nd = layout == kQuadExactNdh || layout == kQuadExactNdhStationary;
    const bool raw = layout == kQuadExact || layout == kQuadExactNd || halves_nd;
    const float centre = raw ? 0.0f : 0.5f;  // the reference-order sweeps take the fraction as it is (mimo.cpp:126)
    const int elem = layout == kQuadExactNd || halves_nd ? 16 : 8;  // bytes per LDS element: {next, d} of a frame pair / of the two halves, or one sample pair
    const int U = h->usable(), cols = c.grid_columns, rows = c.pixel_count / cols;
    const int groups = plan.usable_pad / 4;
    // (the {next, d} kernel gives a wave two quads, one quad row apart: its table has an even number of quad rows, the last one clamped;
    // ... the single-frame form two quads one COLUMN apart: its table has whole tiles of 32 columns)
    const int cols_pad = halves_nd ? (cols + 31) / 32 * 32 : (cols + 15) / 16 * 16, rows4 = layout == kQuadExactNd ? ((rows + 3) / 4 + 1) / 2 * 2 : (rows + 3) / 4;
    const size_t n = (size_t) rows4 * cols_pad * groups * 16 + 2 * awpu::kQuadTablePrefetch;  // spare groups: the sweep prefetches one past the end
    std::vector<awpu::QuadEntry> packed(n, awpu::QuadEntry{0.0f, 0u});
    for (int r4 = 0; r4 < rows4; r4++)
        for (int col = 0; col < cols_pad; col++) {
            awpu::QuadEntry *dst = &packed[((size_t) r4 * cols_pad + col) * groups * 16];
            for (int q = 0; q < 4; q++) {
                const int row = 4 * r4 + q;
                const bool inside = row < rows && col < cols;
                const size_t p = (size_t) std::min(row, rows - 1) * cols + std::min(col, cols - 1);
                const int32_t *orow = &h->off[p * c.lut_stride];
                const float *frow = &h->frac[p * c.lut_stride];
                for (int s = 0; s < plan.usable_pad; s++) {
                    awpu::QuadEntry &e = dst[((s >> 2) * 4 + q) * 4 + (s & 3)];
                    const int j = s % plan.chunk;  // mic slot inside its chunk
                    if (s < U) {
                        const int id = h->index[s];
                        const int off_rel = orow[id] - h->wstart;
                        e.f = inside ? frow[id] - centre : 0.0f;  // centred weight (das_fast.hip, das_quad_kernel); exact: as it is
                        e.addr = (uint32_t) (j * plan.row_bytes + off_rel * elem);
                    } else {  // padding mic: silence (the pack passes write zero rows)
                        e.f = 0.0f;
                        e.addr = (uint32_t) (j * plan.row_bytes);
                    }
                }
            }
        }
    AWPU_HIP_TRY(hipMalloc(&d_lut, n * sizeof(awpu::QuadEntry)));
    AWPU_HIP_TRY(hipMemcpy(d_lut, packed.data(), n * sizeof(awpu::QuadEntry), hipMemcpyHostToDevice));
    h->quad_lut_entries[layout] = n;
    return AWPU_OK;
}

// Kernel shape for a call.  AWPU_FAST_VARIANT="fpi,ppw[,nw]" overrides the heuristic (tuning
// knob, read once): fpi frames per item, ppw pixels per wave, nw = 8 (8-wave workgroups, two per
// CU) or 32 (the double-buffered 16-wave shape, one per CU).
void choose_fast_variant(awpu_hip *h, int batch, int *fpi, int *ppw, int *nw) {
    const int env_fpi = env().fpi, env_ppw = env().ppw, env_nw = env().nw;
    // Prefer the double-buffered shape with the most pixels per wave that still gives every CU
    // a workgroup; small grids fall back to 8-wave workgroups with fewer pixels per wave.
    const long P = h->cfg.pixel_count;
    auto wgs = [&](long pix_per_wg) { return (P + pix_per_wg - 1) / pix_per_wg * batch; };
    *fpi = 1;
    if (wgs(128) >= 256) {
        *nw = 32;
        *ppw = 8;
    } else if (wgs(64) >= 256) {
        *nw = 32;
        *ppw = 4;
    } else {
        *nw = 8;
        *ppw = wgs(32) >= 512 ? 4 : 2;
    }
    if (env_fpi == 1 || env_fpi == 2) *fpi = env_fpi;
    if (env_ppw == 2 || env_ppw == 4 || env_ppw == 8) *ppw = env_ppw;
    if (env_nw == 8 || env_nw == 32 || env_nw == 24) *nw = env_nw;
    if (*nw == 32) {
        *fpi = 1;
        if (*ppw < 4) *ppw = 4;
    }
    if (*nw == 24) {
        *fpi = 1;
        *ppw = 4;
    }
    if (*fpi == 2 && *ppw == 8) *ppw = 4;
    if (*fpi == 2 && batch < 2) *fpi = 1;
}

// layout of d_frames: kFull [batch][n_streams][hist]; kCompact [batch][n_streams][compact_hist] with
// sample 0 = history sample wstart; kRing one frame read in place from the ingest ring (rows 2048 apart)
enum FrameLayout { kFull = 0, kCompact = 1, kRing = 2 };
enum PeerPath { kPeerSame = 0, kPeerDirect = 1, kPeerStaged = 2 };

// a launch is over: close the timing bracket and count it (also on the diagnostic paths)
int finish_launch(awpu_hip *h, int batch, hipStream_t s, int kernel_id) {
    h->stats.kernel_variant = kernel_id;
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_end, s));
    h->stats.launches += 1;
    h->stats.frames += (uint64_t) batch;
    return AWPU_OK;
}

// AWPU_FAST_DEBUG=16: per-wave cycle stamps of the launch just enqueued (12 words per wave) to stderr
int dump_diag(awpu_hip *h, size_t n_waves, int wg_waves, const char *tag, hipStream_t s) {
    AWPU_HIP_TRY(hipStreamSynchronize(s));
    std::vector<unsigned long long> hb(n_waves * 12);
    AWPU_HIP_TRY(hipMemcpy(hb.data(), h->d_diag, hb.size() * 8, hipMemcpyDeviceToHost));
    double v[10] = {0};
    std::vector<double> sw(wg_waves, 0), ba(wg_waves, 0);
    for (size_t i = 0; i < n_waves; i++) {
        for (int k = 0; k < 10; k++) v[k] += (double) hb[12 * i + k];
        sw[i % wg_waves] += (double) hb[12 * i + 5];
        ba[i % wg_waves] += (double) hb[12 * i + 8];
    }
    std::fprintf(stderr, "[awpu diag %s] waves %zu, per wave cycles: total %.0f | dma-issue %.0f sweep %.0f (in blocks %.0f = %.1f%%, "
                 "first table wait %.0f) tail %.0f dma-wait %.0f barrier %.0f | per block %.0f\n", tag, n_waves, v[2] / n_waves,
                 v[4] / n_waves, v[5] / n_waves, v[1] / n_waves, 100 * v[1] / v[2], v[0] / n_waves, v[6] / n_waves,
                 v[7] / n_waves, v[8] / n_waves, v[1] / v[3]);
    if (v[9] > 0)  // shader cycles over the 100 MHz real-time counter, both stamped by every wave (MI355X_MICROARCH.md, DVFS give-back item 6)
        std::fprintf(stderr, "[awpu diag %s] in-kernel clock %.3f GHz (cycles %.0f / real time %.2f us per wave)\n", tag, v[2] / v[9] * 0.1,
                     v[2] / n_waves, v[9] / n_waves * 0.01);
    std::fprintf(stderr, "[awpu diag %s] wave slot sweep/barrier kcycles:", tag);
    for (int k = 0; k < wg_waves; k++)
        std::fprintf(stderr, " %d:%.0f/%.0f", k, sw[k] * wg_waves / n_waves / 1e3, ba[k] * wg_waves / n_waves / 1e3);
    std::fprintf(stderr, "\n");
    return AWPU_OK;
}

// The reference-order sweep on the frame-pair layout (das_exact_pair_kernel): the table is the pair shape's with the
// UNSCALED fraction in .f (gains go on the samples, as in das_exact_kernel) and padding entries -- mics usable ..
// usable_pad-1, and whole rows past the grid -- that read a row of zeros with fraction 0 (they add +0).
int build_exact_pair_lut(awpu_hip *h) {
    if (h->d_exact_pair_lut) return AWPU_OK;
    const auto &c = h->cfg;
    const awpu::FastPlan &plan = h->exact_plan;
    const int U = h->usable(), P = c.pixel_count;
    const int P_pad = (P + h->pair_cols + 127) / 128 * 128;  // whole tiles; with vertical pairs the partner of a last-row pixel lies one grid row past the table
    const size_t n = (size_t) P_pad * plan.usable_pad + 4 * awpu::kPairTablePrefetch;  // spare groups: the block prefetches one past a row's end
    std::vector<awpu::FastEntry> packed(n);
    for (size_t i = 0; i < n; i++)  // null entry of slot s: the zero row of its own slot in the last chunk, or any row with fraction 0 ...
        packed[i] = awpu::FastEntry{0.0f, (uint32_t) ((int) (i % plan.usable_pad) % plan.chunk * plan.row_bytes), 0.0f, 0u};
    for (int p = 0; p < P; p++) {
        const int32_t *orow = &h->off[(size_t) p * c.lut_stride];
        const float *frow = &h->frac[(size_t) p * c.lut_stride];
        awpu::FastEntry *dst = &packed[(size_t) p * plan.usable_pad];
        for (int s = 0; s < U; s++) {
            const int id = h->index[s];
            dst[s].f = frow[id];  // the reference's `fraction`, mimo.cpp:126
            dst[s].addr = (uint32_t) ((s % plan.chunk) * plan.row_bytes + (orow[id] - h->wstart) * 8);
        }
    }
    AWPU_HIP_TRY(hipMalloc(&h->d_exact_pair_lut, n * sizeof(awpu::FastEntry)));
    AWPU_HIP_TRY(hipMemcpy(h->d_exact_pair_lut, packed.data(), n * sizeof(awpu::FastEntry), hipMemcpyHostToDevice));
    h->exact_pair_lut_entries = n;
    return AWPU_OK;
}

// frame pairs an XCD works on at a time: as many as keep their samples in its 4 MiB L2 beside the table stream (launch_quads)
int xcd_pair_group_bytes(size_t pair_bytes, int n_pairs) {
    int g = (int) std::max<size_t>(1, (3u << 20) / pair_bytes);
    g = g >= 8 ? 8 : g >= 4 ? 4 : g >= 2 ? 2 : 1;
    while (g > 1 && g > n_pairs) g >>= 1;
    return g;
}
int xcd_pair_group(const awpu::FastPlan &pp, int rows_per_pair, int n_pairs) {
    const size_t pair_bytes = (size_t) rows_per_pair * pp.wr * 8;
    int g = (int) std::max<size_t>(1, (3u << 20) / pair_bytes);
    g = g >= 8 ? 8 : g >= 4 ? 4 : g >= 2 ? 2 : 1;
    while (g > 1 && g > n_pairs) g >>= 1;
    return g;
}

int launch_exact_pairs(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int hist_eff, int wstart_eff) {
    int rc = build_exact_pair_lut(h);
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &pp = h->exact_plan;
    const size_t need = (size_t) ((std::max(h->cfg.max_batch, batch) + 1) / 2) * pp.usable_pad * pp.wr * 2;
    if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    awpu::ExactPairArgs a{};
    a.packed = h->d_pack;
    a.lut = h->d_exact_pair_lut;
    a.power = d_power;
    a.sums = h->sums_out;
    a.usable = h->usable();
    a.usable_pad = pp.usable_pad;
    a.pixel_count = h->cfg.pixel_count;
    a.wp = pp.wr;
    a.chunk = pp.chunk;
    a.batch = batch;
    a.cols = h->pair_cols;
    a.tiles = awpu::pair_tiles(a.pixel_count, a.cols);
    a.n_pairs = (batch + 1) / 2;
    a.pair_group = xcd_pair_group(pp, pp.usable_pad, a.n_pairs);
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    AWPU_HIP_TRY(awpu::launch_pack_pairs(d_frames, h->cfg.n_streams, hist_eff, wstart_eff, h->d_index, h->usable(), pp.usable_pad,
                                         h->d_gain, pp.wr, batch, h->d_pack, false, s));  // raw samples: no stencil in front of the reference's order
    AWPU_HIP_TRY(awpu::launch_das_exact_pairs(a, {h->exact_pair_lut_entries, h->pack_cap}, s));
    return finish_launch(h, batch, s, AWPU_KERNEL_EXACT_PAIR);
}

// ... four vertically adjacent pixels per wave where the row length is known (das_exact_quad_kernel): same bits, fewer LDS reads
int launch_exact_quads(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int hist_eff, int wstart_eff) {
    int rc = build_quad_lut(h, kQuadExact);
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &pp = h->exact_plan;
    const size_t need = (size_t) ((std::max(h->cfg.max_batch, batch) + 1) / 2) * pp.usable_pad * pp.wr * 2;
    if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    awpu::ExactQuadArgs a{};
    a.packed = h->d_pack;
    a.lut = h->d_exact_quad_lut;
    a.power = d_power;
    a.sums = h->sums_out;
    a.usable = h->usable();
    a.usable_pad = pp.usable_pad;
    a.pixel_count = h->cfg.pixel_count;
    a.wp = pp.wr;
    a.chunk = pp.chunk;
    a.batch = batch;
    a.cols = h->cfg.grid_columns;
    a.rows = h->cfg.pixel_count / a.cols;
    a.tiles = awpu::quad_tiles(a.rows, a.cols);
    a.n_pairs = (batch + 1) / 2;
    a.pair_group = xcd_pair_group(pp, pp.usable_pad, a.n_pairs);
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    AWPU_HIP_TRY(awpu::launch_pack_pairs(d_frames, h->cfg.n_streams, hist_eff, wstart_eff, h->d_index, h->usable(), pp.usable_pad,
                                         h->d_gain, pp.wr, batch, h->d_pack, false, s));
    AWPU_HIP_TRY(awpu::launch_das_exact_quads(a, {h->quad_lut_entries[kQuadExact], h->pack_cap}, s));
    return finish_launch(h, batch, s, AWPU_KERNEL_EXACT_QUAD);
}

// ... on the {next, d} layout (das_exact_nd_kernel, round 5): cur - next formed once per sample by the pack pass; nq quads per wave
int launch_exact_nd(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int hist_eff, int wstart_eff, int nq,
                    const float *prepacked = nullptr, size_t prepacked_floats = 0) {
    int rc = build_quad_lut(h, kQuadExactNd);
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &pp = h->exact_nd_plan;
    const size_t need = (size_t) ((std::max(h->cfg.max_batch, batch) + 1) / 2) * pp.usable_pad * pp.wr * 4;
    if (!prepacked)
        if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    awpu::ExactNdArgs a{};
    a.packed = prepacked ? prepacked : h->d_pack;
    a.lut = h->d_exact_nd_lut;
    a.power = d_power;
    a.sums = h->sums_out;
    a.usable = h->usable();
    a.usable_pad = pp.usable_pad;
    a.pixel_count = h->cfg.pixel_count;
    a.wq = pp.wr;
    a.chunk = pp.chunk;
    a.batch = batch;
    a.cols = h->cfg.grid_columns;
    a.rows = h->cfg.pixel_count / a.cols;
    a.nq = nq;
    a.tiles = awpu::nd_tiles(a.rows, a.cols, nq);
    a.n_pairs = (batch + 1) / 2;
    a.pair_group = xcd_pair_group_bytes((size_t) pp.usable_pad * pp.row_bytes, a.n_pairs);
    if (!h->d_nd_queue) {
        AWPU_HIP_TRY(hipMalloc(&h->d_nd_queue, 9 * sizeof(unsigned)));
        if (hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess || h->n_cus < 1) h->n_cus = 256;
    }
    {   // the item list: rebuilt (by the launcher, on the stream) when the batch, the pair group or the tile shape changed
        const size_t items = (size_t) a.n_pairs * a.tiles;
        const long long key = ((long long) a.n_pairs << 24) | ((long long) a.pair_group << 8) | nq;
        if (h->nd_items_cap < items) {
            retire_live_graphs(h);
            AWPU_HIP_TRY(hipStreamSynchronize(s));  // (a sweep in flight may still read the old list)
            dev_free(h->d_nd_items);
            h->nd_items_cap = 0;
            AWPU_HIP_TRY(hipMalloc(&h->d_nd_items, items * sizeof(int2)));
            h->nd_items_cap = items;
            h->nd_items_key = -1;
        }
        a.items = h->d_nd_items;
        a.build_items = h->nd_items_key != key;
        h->nd_items_key = key;
    }
    a.queue = h->d_nd_queue;
    a.wgs = h->n_cus;
    {   // an eighth of every XCD's run goes to the common queue (the XCDs' speeds differ by ~6 %); short runs: one queue for the chip
        const int per = (a.n_pairs * a.tiles + 7) / 8;
        a.tail = per >= 32 ? (per + 7) / 8 : per;
#ifdef AWPU_TUNING_BUILD
        if (const char *v = std::getenv("AWPU_ND_TAIL")) a.tail = std::max(1, std::atoi(v) >= 100 ? per : per * std::atoi(v) / 100);  // percent of a run
#endif
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    if (!prepacked)
        AWPU_HIP_TRY(awpu::launch_pack_nd(d_frames, h->cfg.n_streams, hist_eff, wstart_eff, h->d_index, h->usable(), pp.usable_pad, h->d_gain,
                                          pp.wr, batch, h->d_pack, s));
#ifdef AWPU_TUNING_BUILD
    const size_t n_wgs = std::min<size_t>((size_t) a.n_pairs * a.tiles, (size_t) a.wgs);
    if (env().debug & 16) {  // per-workgroup timeline (where, when, phases): printed below
        if (const int drc = ensure_diag(h, n_wgs * 8); drc != AWPU_OK) return drc;
        AWPU_HIP_TRY(hipMemsetAsync(h->d_diag, 0, n_wgs * 8 * sizeof(unsigned long long), s));
        a.debug_out = h->d_diag;
    }
#endif
    AWPU_HIP_TRY(awpu::launch_das_exact_nd(a, {h->quad_lut_entries[kQuadExactNd], prepacked ? prepacked_floats : h->pack_cap}, s));
    rc = finish_launch(h, batch, s, AWPU_KERNEL_EXACT_ND);
#ifdef AWPU_TUNING_BUILD
    if (rc == AWPU_OK && (env().debug & 16)) {
        AWPU_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> hb(n_wgs * 8);
        AWPU_HIP_TRY(hipMemcpy(hb.data(), h->d_diag, hb.size() * 8, hipMemcpyDeviceToHost));
        // group by compute unit (XCC, SE, CU of HW_ID), order by start: busy time, gaps between consecutive workgroups
        std::map<unsigned long long, std::vector<std::array<unsigned long long, 5>>> by_cu;
        unsigned long long first = ~0ull, last = 0;
        double ph[3] = {0, 0, 0};
        size_t n = 0;
        for (size_t w = 0; w < n_wgs; w++) {
            const unsigned long long *o = &hb[8 * w];
            if (!o[7]) continue;
            const unsigned hw = (unsigned) o[5], xcc = (unsigned) (o[5] >> 32) & 0xf;
            const unsigned long long cu = ((unsigned long long) xcc << 16) | ((hw >> 8) & 0xff);  // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            by_cu[cu].push_back({o[0], o[1], o[2], o[3], o[4]});
            first = std::min(first, o[0]);
            last = std::max(last, o[1]);
            for (int k = 0; k < 3; k++) ph[k] += (double) o[2 + k];
            n++;
        }
        double busy = 0, gaps = 0, head = 0, tail = 0;
        size_t n_gaps = 0;
        for (auto &kv : by_cu) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end());
            head += (double) (v.front()[0] - first);
            tail += (double) (last - v.back()[1]);
            for (size_t i = 0; i < v.size(); i++) {
                busy += (double) (v[i][1] - v[i][0]);
                if (i) gaps += (double) v[i][0] - (double) v[i - 1][1], n_gaps++;
            }
        }
        {   // per XCD: when its last workgroup ended (relative to the launch's first stamp), mean busy time of its CUs, and the
            // spread of workgroup durations by position in the item order
            std::map<unsigned, std::array<double, 4>> xs;  // last end, busy sum, CU count, workgroups
            for (auto &kv : by_cu) {
                auto &x = xs[(unsigned) (kv.first >> 16)];
                x[0] = std::max(x[0], (double) (kv.second.back()[1] - first));
                for (auto &w : kv.second) x[1] += (double) (w[1] - w[0]);
                x[2] += 1;
                x[3] += (double) kv.second.size();
            }
            std::fprintf(stderr, "[awpu diag nd] per XCD (last end us / mean busy us / CUs / workgroups):");
            for (auto &kv : xs) std::fprintf(stderr, " %u: %.0f/%.0f/%.0f/%.0f", kv.first, kv.second[0] * 0.01, kv.second[1] / kv.second[2] * 0.01, kv.second[2], kv.second[3]);
            std::vector<double> dur;
            for (size_t w = 0; w < n_wgs; w++) if (hb[8 * w + 7]) dur.push_back((double) (hb[8 * w + 1] - hb[8 * w]) * 0.01);
            std::sort(dur.begin(), dur.end());
            if (!dur.empty()) std::fprintf(stderr, "\n[awpu diag nd] workgroup duration us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f\n", dur.front(), dur[dur.size() / 10],
                                           dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur.back());
        }
        const double cus = (double) by_cu.size();
        std::fprintf(stderr, "[awpu diag nd] %zu workgroups on %zu CUs, span %.1f us | per CU: busy %.1f us, gaps %.1f us (%.2f us each), idle before first %.1f us, "
                     "after last %.1f us | per workgroup cycles: outside the block %.0f, in the sweep block %.0f, exit %.0f\n", n, by_cu.size(), (double) (last - first) * 0.01,
                     busy / cus * 0.01, gaps / cus * 0.01, n_gaps ? gaps / (double) n_gaps * 0.01 : 0.0, head / cus * 0.01, tail / cus * 0.01, ph[0] / n, ph[1] / n, ph[2] / n);
    }
#endif
    return rc;
}

// Does launch() sweep a batch of AWPU_MATH_F32_EXACT with das_exact_nd_kernel, and with how many quads per wave?  (The rule of launch()
// and of the packed-frame entry points: asked before anything is packed.)
bool takes_exact_nd(awpu_hip *h, int batch, int *nq) {
    if (!h->exact_pairs_ok || !h->exact_nd_ok || h->cfg.grid_columns < 1) return false;
    const int ex = env().exact_pairs;
    if (ex == 0 || ex == 2 || ex == 3) return false;  // AWPU_SHAPE=exact_verify / exact_pair / exact_quad
    const int cols = h->cfg.grid_columns, rows = h->cfg.pixel_count / cols;
    // (quad_differ < 1.5: on average fewer than half of a quad's pixels leave the reference pixel's address for a mic; a square
    // array's vertical and horizontal neighbours coincide equally often -- pair_cols stays 0 there -- and quads still pay)
    const bool quads_pay = h->cfg.pixel_count % cols == 0 && h->cfg.pixel_begin % cols == 0 && h->quad_differ < 1.5;
    if ((!(h->pair_cols > 0 || quads_pay) && ex != 4 && ex != 5 && ex != 6) || rows < 4) return false;  // (a forced shape runs on any table: the random tests)
    // two quads per wave where that still fills the chip (AWPU_SHAPE=exact_nd1 / exact_nd2: one / two everywhere)
    const long wgs2 = (long) awpu::nd_tiles(rows, cols, 2) * ((batch + 1) / 2);
    *nq = ex == 4 ? 1 : ex == 5 ? 2 : (rows >= 8 && wgs2 >= 512 ? 2 : 1);
    return true;
}

// The completion flag of the synchronous one-frame host call (das_kernels.h: DoneFlag; das_fast.hip: store_tile_and_signal): what the
// next armed launch of `workgroups` workgroups gets, and what the handle remembers once that launch has gone out.
int arm_done_flag(awpu_hip *h, unsigned long long workgroups, awpu::DoneFlag *out) {
    if (!h->d_done_counter) {
        AWPU_HIP_TRY(hipMalloc(&h->d_done_counter, sizeof(unsigned long long)));
        AWPU_HIP_TRY(hipMemsetAsync(h->d_done_counter, 0, sizeof(unsigned long long), h->stream));
        AWPU_HIP_TRY(hipStreamSynchronize(h->stream));  // (the handle's own stream: nothing device-wide from inside a sweep call)
        AWPU_HIP_TRY(hipHostMalloc(&h->h_done_flag, 64, hipHostMallocDefault));
        *h->h_done_flag = 0;
        h->done_total = 0;
    }
    out->counter = h->d_done_counter;
    out->flag = h->h_done_flag;
    out->target = h->done_total + workgroups;
    out->seq = h->done_seq + 1;
    return AWPU_OK;
}
void done_flag_armed(awpu_hip *h, const awpu::DoneFlag &d) {
    h->done_total = d.target;
    h->done_seq = d.seq;
    h->done_used = true;
}

// single frames in the reference's order: the halves form of the {next, d} layout (das_exact_ndh_kernel) -- every mic resident and
// staged by the workgroups themselves (one array), or chunked behind a pack pre-pass.  `pitch` = floats between two streams of a frame
int launch_exact_ndh(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int pitch, int wstart_eff, bool stationary,
                     int nq, bool pixel_per_wave = false) {
    int rc = build_quad_lut(h, stationary ? kQuadExactNdhStationary : kQuadExactNdh);
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &pp = stationary ? h->exact_ndhs_plan : h->exact_ndh_plan;
    if (!stationary) {
        const size_t need = std::max((size_t) h->cfg.max_batch, (size_t) batch) * pp.usable_pad * pp.wr * 4;
        if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    }
    awpu::ExactNdhArgs a{};
    a.packed = stationary ? nullptr : h->d_pack;
    a.frames = d_frames;
    a.lut = stationary ? h->d_exact_ndhs_lut : h->d_exact_ndh_lut;
    a.index = h->d_index;
    a.gain = h->d_gain;
    a.power = d_power;
    a.sums = h->sums_out;
    a.n_streams = h->cfg.n_streams;
    a.pitch = pitch;
    a.wstart = wstart_eff;
    a.usable = h->usable();
    a.usable_pad = pp.usable_pad;
    a.pixel_count = h->cfg.pixel_count;
    a.wh = pp.wr;
    a.chunk = pp.chunk;
    a.batch = batch;
    a.cols = h->cfg.grid_columns;
    a.rows = h->cfg.pixel_count / a.cols;
    a.nq = nq;
    a.tiles = pixel_per_wave ? awpu::ndp_tiles(a.rows, a.cols) : awpu::ndh_tiles(a.rows, a.cols, nq);
    a.lut_cols = (a.cols + 31) / 32 * 32;
    a.identity = 1;
    for (int k = 0; k < a.usable && a.identity; k++) a.identity = h->index[k] == k;
    h->done_used = false;
    if (h->done_arm && s == h->stream && stationary && nq == 1 && !pixel_per_wave) {  // (the one-frame host call: live_host_call)
        rc = arm_done_flag(h, (unsigned long long) batch * a.tiles, &a.done);
        if (rc != AWPU_OK) return rc;
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    if (!stationary)
        AWPU_HIP_TRY(awpu::launch_pack_ndh(d_frames, h->cfg.n_streams, pitch, wstart_eff, a.identity ? nullptr : h->d_index, h->usable(), pp.usable_pad, h->d_gain, pp.wr,
                                           batch, h->d_pack, s));
    if (pixel_per_wave) {
        AWPU_HIP_TRY(awpu::launch_das_exact_ndp(a, {h->quad_lut_entries[kQuadExactNdh], h->pack_cap}, s));
    } else {
        AWPU_HIP_TRY(awpu::launch_das_exact_ndh(a, stationary, {h->quad_lut_entries[stationary ? kQuadExactNdhStationary : kQuadExactNdh], stationary ? 0 : h->pack_cap}, s));
    }
    if (a.done.flag) done_flag_armed(h, a.done);  // (the launch went out: its workgroups will count themselves)
    return finish_launch(h, batch, s, pixel_per_wave ? AWPU_KERNEL_EXACT_NDP : stationary ? AWPU_KERNEL_EXACT_NDH_STATIONARY : AWPU_KERNEL_EXACT_NDH);
}

int launch_exact(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int hist_eff, int wstart_eff) {
    awpu::SweepArgs a{};
    a.frames = d_frames;
    a.lut = h->d_lut;
    a.index = h->d_index;
    a.power = d_power;
    a.gain = h->d_gain;
    a.n_streams = h->cfg.n_streams;
    a.hist = hist_eff;
    a.usable = h->usable();
    a.pixel_count = h->cfg.pixel_count;
    a.wstart = wstart_eff;
    a.window = h->window;
    a.batch = batch;
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    if (h->cfg.interp == AWPU_INTERP_FIR8) {
        AWPU_HIP_TRY(awpu::launch_das_fir8(a, h->d_fir, s));
    } else {
        AWPU_HIP_TRY(awpu::launch_das_exact(a, h->cfg.math == AWPU_MATH_BF16_ACC, s));
    }
    return finish_launch(h, batch, s, h->cfg.interp == AWPU_INTERP_FIR8 ? AWPU_KERNEL_FIR8 : AWPU_KERNEL_EXACT_VERIFY);
}

// FIR8 on the four-plane frame-pair layout (das_fir8_plane_kernel): a lane owns four consecutive outputs
int launch_fir8_planes(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int hist_eff, int wstart_eff) {
    const awpu::FastPlan &pp = h->fir_plane_plan;
    const int U = h->usable(), P = h->cfg.pixel_count;
    const int row_entries = pp.usable_pad;  // (a multiple of 4, like the chunk: the block sweeps groups of four items)
    if (!h->d_fir_plane_lut) {
        const uint32_t plane_bytes = (uint32_t) pp.row_bytes / 4;
        // one dword per (pixel, mic); four spare: the block requests entries four items ahead.  Null entries (the
        // padding of a row, the spares) read row 0 with the zero coefficient row.
        std::vector<uint32_t> packed((size_t) P * row_entries + awpu::kFir8PlaneTablePrefetch, awpu::fir8_plane_word(0, 0, awpu::kFir8ZeroRow));
        for (int p = 0; p < P; p++) {
            const int32_t *orow = &h->off[(size_t) p * h->cfg.lut_stride];
            const float *frow = &h->frac[(size_t) p * h->cfg.lut_stride];
            for (int m = 0; m < U; m++) {
                const int id = h->index[m];
                const int32_t k = (int32_t) (frow[id] * 100.0f + 0.5f);  // delay.cpp:32-33: the coefficient row
                const uint32_t first = (uint32_t) (orow[id] - h->wstart);  // row element of X[off]
                const uint32_t addr = (uint32_t) (m % pp.chunk) * pp.row_bytes + (first & 3) * plane_bytes + (first >> 2) * 8;
                packed[(size_t) p * row_entries + m] = awpu::fir8_plane_word(addr, first & 3, (uint32_t) k);
            }
        }
        AWPU_HIP_TRY(hipMalloc(&h->d_fir_plane_lut, packed.size() * sizeof(uint32_t)));
        AWPU_HIP_TRY(hipMemcpy(h->d_fir_plane_lut, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        h->fir_plane_lut_entries = packed.size();
    }
    const size_t need = (size_t) ((h->cfg.max_batch + 1) / 2) * U * pp.wr * 2;
    if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    awpu::PairArgs pa{};
    pa.packed = h->d_pack;
    pa.power = d_power;
    pa.usable = U;
    pa.usable_pad = row_entries;
    pa.pixel_count = P;
    pa.wp = pp.wr;
    pa.chunk = pp.chunk;
    pa.batch = batch;
    // vertical pixel quads (samples shared between pixels of one column with the same integer delay) where the grid's row
    // length is known and the rows are staged at the pitch that block is generated for; AWPU_FIR8_SHARE=0: consecutive pixels
    {
        const bool allow = env().fir_share != 0;
        const int cols = h->cfg.grid_columns;
        if (allow && cols > 0 && P % cols == 0 && h->cfg.pixel_begin % cols == 0 && (uint32_t) pp.wr * 2u == awpu::kFirStaticPlaneBytesHost)
            pa.cols = cols;
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    AWPU_HIP_TRY(awpu::launch_pack_planes(d_frames, h->cfg.n_streams, hist_eff, wstart_eff, h->d_index, U, h->d_gain, pp.wr,
                                          batch, h->d_pack, s));
    AWPU_HIP_TRY(awpu::launch_das_fir8_planes(pa, h->d_fir_plane_lut, h->d_fir, env().quad_variant, {h->fir_plane_lut_entries, h->pack_cap}, s));
    return finish_launch(h, batch, s, AWPU_KERNEL_FIR8_PLANES);
}

// frame-pair shape: two frames per item, for batches on grids that fill the chip
int launch_pairs(awpu_hip *h, const awpu_hip::FastLut *plut, const float *d_frames, int batch, float *d_power,
                 hipStream_t s, int hist_eff, int wstart_eff, int stationary_tiles = 0, const float *prepacked = nullptr,
                 size_t prepacked_floats = 0) {
    const awpu::FastPlan &pp = plut->plan;
    const size_t need = (size_t) ((h->cfg.max_batch + 1) / 2) * h->usable() * pp.wr * 2;
    // the stationary shape stages its pairs itself, straight from the caller's frames (no pack pre-pass, no packed buffer)
    const bool self_staged = stationary_tiles > 0 && !prepacked;
    if (!prepacked && !self_staged)
        if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    awpu::PairArgs pa{};
    pa.packed = prepacked ? prepacked : (self_staged ? nullptr : h->d_pack);
    if (self_staged) {
        pa.frames = d_frames;
        pa.index = h->d_index;
        pa.n_streams = h->cfg.n_streams;
        pa.hist = hist_eff;
        pa.wstart = wstart_eff;
    }
    pa.lut = plut->d;
    pa.power = d_power;
    pa.usable = h->usable();
    pa.usable_pad = pp.usable_pad;
    pa.pixel_count = h->cfg.pixel_count;
    pa.wp = pp.wr;
    pa.chunk = pp.chunk;
    pa.batch = batch;
    pa.cols = h->pair_cols;
    pa.tiles = awpu::pair_tiles(h->cfg.pixel_count, h->pair_cols);
    pa.n_pairs = (batch + 1) / 2;
    pa.pair_group = xcd_pair_group(pp, h->usable(), pa.n_pairs);
    pa.debug = env().debug;
    pa.debug_out = nullptr;
    size_t n_waves = 0;
    if (stationary_tiles > 0) pa.debug &= ~16;  // (no stamped build of the stationary shape)
    if (pa.debug & 16) {
        n_waves = (size_t) 16 * ((batch + 1) / 2) * awpu::pair_tiles(h->cfg.pixel_count, h->pair_cols);
        const int rc = ensure_diag(h, n_waves * 12);
        if (rc != AWPU_OK) return rc;
        pa.debug_out = h->d_diag;
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    if (!prepacked && !self_staged)
        AWPU_HIP_TRY(awpu::launch_pack_pairs(d_frames, h->cfg.n_streams, hist_eff, wstart_eff, h->d_index, h->usable(),
                                             h->usable(), nullptr, pp.wr, batch, h->d_pack, true, s));  // gains ride on the table weights here
    const awpu::Extents have{plut->entries, prepacked ? prepacked_floats : h->pack_cap};
    if (stationary_tiles > 0) {
        AWPU_HIP_TRY(awpu::launch_das_pairs_stationary(pa, stationary_tiles, have, s));
    } else {
        AWPU_HIP_TRY(awpu::launch_das_pairs(pa, have, s));
    }
    const int rc = finish_launch(h, batch, s, stationary_tiles > 0 ? AWPU_KERNEL_PAIR_STATIONARY : AWPU_KERNEL_PAIR);
    if (rc != AWPU_OK || !(pa.debug & 16)) return rc;
    return dump_diag(h, n_waves, 16, "pairs", s);
}

// quad shape: the frame-pair layout swept four vertically adjacent pixels at a time (das_quad_kernel)
int launch_quads(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int hist_eff, int wstart_eff,
                 const float *prepacked = nullptr, size_t prepacked_floats = 0) {
    int rc = build_quad_lut(h, kQuadPairs);
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &pp = h->quad_plan;
    const size_t need = (size_t) ((h->cfg.max_batch + 1) / 2) * pp.usable_pad * pp.wr * 2;
    if (!prepacked)
        if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    awpu::QuadArgs qa{};
    qa.packed = prepacked ? prepacked : h->d_pack;
    qa.lut = h->d_quad_lut;
    qa.power = d_power;
    qa.usable = h->usable();
    qa.usable_pad = pp.usable_pad;
    qa.pixel_count = h->cfg.pixel_count;
    qa.wp = pp.wr;
    qa.chunk = pp.chunk;
    qa.batch = batch;
    qa.cols = h->cfg.grid_columns;
    qa.rows = h->cfg.pixel_count / qa.cols;
    qa.tiles = awpu::quad_tiles(qa.rows, qa.cols);
    qa.n_pairs = (batch + 1) / 2;
    {   // frame pairs one XCD works on at a time: as many as keep their samples in its 4 MiB L2 beside the table stream.
        // (Eight pairs at the headline shape, 5.9 MB of samples, run 1.2 % faster than four -- fewer table passes --
        // but the samples then stream from beyond the L2: 3.2 GB of L2 misses per launch instead of 0.72 GB.  Not taken.)
        const size_t pair_bytes = (size_t) pp.usable_pad * pp.wr * 8;
        int g = env().pair_group > 0 ? env().pair_group : (int) std::max<size_t>(1, (3u << 20) / pair_bytes);
        g = g >= 8 ? 8 : g >= 4 ? 4 : g >= 2 ? 2 : 1;
        while (g > 1 && g > qa.n_pairs) g >>= 1;
        qa.pair_group = g;
    }
    qa.debug = env().debug;
    qa.variant = env().quad_variant;
    // Persistent workgroups (one per CU, each walking its share of the items with the next item's first chunk
    // prefetched) are 1.5 % faster than one workgroup per item on a chip they have to themselves, and fragile on one
    // they share: their share of the items is static.  A handle that owns the whole grid has the GPU to itself; a
    // handle that owns a slab is a rank of a multi-GPU run, beside which the broadcast of the next batch holds CUs.
    qa.wgs = env().wgs;
    if (qa.wgs == 0 && h->cfg.pixel_count == h->cfg.n_pixels) {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->cfg.device) == hipSuccess) qa.wgs = n_cu;
    }
    if (qa.wgs < 0) qa.wgs = 0;  // AWPU_FAST_WGS=-1: one workgroup per item everywhere
    // Round 5: the persistent workgroups take their items from queues (one per XCD, a run's last eighth common to the chip) instead of
    // static shares -- balanced whatever the XCDs' clocks and whoever else holds CUs, so a rank's slab takes them too
    if (env().wgs == 0) {
        if (!h->d_nd_queue) {
            AWPU_HIP_TRY(hipMalloc(&h->d_nd_queue, 9 * sizeof(unsigned)));
            if (hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess || h->n_cus < 1) h->n_cus = 256;
        }
        if (h->n_cus < 1 && (hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess || h->n_cus < 1))
            h->n_cus = 256;
        qa.queue = h->d_nd_queue;
        qa.wgs = h->n_cus;
        const int per = (qa.n_pairs * qa.tiles + 7) / 8;
        qa.tail = per >= 32 ? (per + 7) / 8 : per;
    }
    qa.debug_out = nullptr;
    size_t n_waves = 0;
    if (qa.debug & 16) {
        const long per_xcd = ((long) qa.n_pairs * qa.tiles + 7) / 8;
        n_waves = (size_t) 16 * 8 * (size_t) (qa.wgs > 0 ? std::min<long>(per_xcd, std::max(1, qa.wgs / 8)) : per_xcd);
        rc = ensure_diag(h, n_waves * 12);
        if (rc != AWPU_OK) return rc;
        AWPU_HIP_TRY(hipMemsetAsync(h->d_diag, 0, n_waves * 12 * sizeof(unsigned long long), s));
        qa.debug_out = h->d_diag;
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    if (!prepacked)
        AWPU_HIP_TRY(awpu::launch_pack_pairs(d_frames, h->cfg.n_streams, hist_eff, wstart_eff, h->d_index, h->usable(),
                                             pp.usable_pad, h->d_gain, pp.wr, batch, h->d_pack, true, s));
    AWPU_HIP_TRY(awpu::launch_das_quads(qa, {h->quad_lut_entries[kQuadPairs], prepacked ? prepacked_floats : h->pack_cap}, s));
    rc = finish_launch(h, batch, s, AWPU_KERNEL_QUAD);
    if (rc != AWPU_OK || !(qa.debug & 16)) return rc;
    return dump_diag(h, n_waves, 16, "quads", s);
}

// quad shape for single frames on the halves layout (das_quadh_kernel): a pack + filter pre-pass, then the sweep
int launch_quadsh(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int pitch, int hist_eff, int wstart_eff,
                  int qpw) {
    int rc = build_quad_lut(h, kQuadHalves);
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &pp = h->quadh_plan;
    const size_t need = std::max((size_t) h->cfg.max_batch, (size_t) batch) * pp.usable_pad * pp.wr * 2;
    if (const int prc = ensure_pack(h, need); prc != AWPU_OK) return prc;
    awpu::QuadhArgs qa{};
    qa.packed = h->d_pack;
    qa.lut = h->d_quadh_lut;
    qa.power = d_power;
    qa.usable = h->usable();
    qa.usable_pad = pp.usable_pad;
    qa.pixel_count = h->cfg.pixel_count;
    qa.wp = pp.wr;
    qa.chunk = pp.chunk;
    qa.batch = batch;
    qa.cols = h->cfg.grid_columns;
    qa.rows = h->cfg.pixel_count / qa.cols;
    qa.debug = env().debug;
    qa.debug_out = nullptr;
    size_t n_waves = 0;
    if (qa.debug & 16) {
        n_waves = (size_t) 16 * batch * awpu::quad1_tiles(qa.rows, qa.cols, qpw);
        rc = ensure_diag(h, n_waves * 12);
        if (rc != AWPU_OK) return rc;
        qa.debug_out = h->d_diag;
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    bool identity = true;  // the active-mic list is 0 .. usable-1 (awpu_hip_set_active_mics(NULL)): the pre-pass needs no look-up
    for (int k = 0; k < h->usable() && identity; k++) identity = h->index[k] == k;
    AWPU_HIP_TRY(awpu::launch_pack_halves(d_frames, h->cfg.n_streams, pitch, hist_eff, wstart_eff, identity ? nullptr : h->d_index, h->usable(),
                                          pp.usable_pad, h->d_gain, pp.wr, batch, h->d_pack, s));
    AWPU_HIP_TRY(awpu::launch_das_quadh(qa, qpw, {h->quad_lut_entries[kQuadHalves], h->pack_cap}, s));
    rc = finish_launch(h, batch, s, AWPU_KERNEL_QUADH);
    if (rc != AWPU_OK || !(qa.debug & 16)) return rc;
    return dump_diag(h, n_waves, 16, "quadsh", s);
}

// single frames of small arrays: every mic's halves row resident, the workgroup stages the window itself (das_quadh_stationary_kernel)
int launch_quadsh_stationary(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int pitch, int hist_eff,
                             int wstart_eff, int qpw) {
    int rc = build_quad_lut(h, kQuadHalvesStationary);
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &pp = h->quadhs_plan;
    awpu::QuadhStationaryArgs qa{};
    qa.frames = d_frames;
    qa.lut = h->d_quadhs_lut;
    qa.index = h->d_index;
    qa.gain = h->d_gain;
    qa.power = d_power;
    qa.n_streams = h->cfg.n_streams;
    qa.pitch = pitch;
    qa.hist = hist_eff;
    qa.wstart = wstart_eff;
    qa.usable = h->usable();
    qa.usable_pad = pp.usable_pad;
    qa.pixel_count = h->cfg.pixel_count;
    qa.wp = pp.wr;
    qa.batch = batch;
    qa.cols = h->cfg.grid_columns;
    qa.rows = h->cfg.pixel_count / qa.cols;
    qa.waves = 16;
    qa.identity = 1;
    for (int k = 0; k < qa.usable && qa.identity; k++) qa.identity = h->index[k] == k;
    qa.row_limit = pitch;  // (the ring's rows are 2048 floats of which any 1024 + window are valid: double-written)
    if (!awpu::quadh_stationary_raw(pp, qa.usable, wstart_eff, qa.row_limit, &qa.raw_begin, &qa.raw_wr, &qa.image_offset))
        return invalid("the raw window does not fit the LDS beside the halves image");  // (launch() asks before it comes here)
    h->done_used = false;
    if (h->done_arm && s == h->stream && qpw == 1) {  // (the one-frame host call: live_host_call)
        rc = arm_done_flag(h, (unsigned long long) batch * awpu::quad1_tiles(qa.rows, qa.cols, 1), &qa.done);
        if (rc != AWPU_OK) return rc;
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    AWPU_HIP_TRY(awpu::launch_das_quadh_stationary(qa, qpw, {h->quad_lut_entries[kQuadHalvesStationary], 0}, s));
    if (qa.done.flag) done_flag_armed(h, qa.done);
    return finish_launch(h, batch, s, AWPU_KERNEL_QUADH_STATIONARY);
}

int launch(awpu_hip *h, const float *d_frames, int batch, float *d_power, hipStream_t s, int layout = kFull) {
    const bool compact = layout == kCompact;
    const int hist_eff = compact ? h->compact_hist : (layout == kRing ? 2048 : h->cfg.hist);
    const int wstart_eff = compact ? 0 : h->wstart;
    // FIR8 with fast math on a launch that fills the chip: the frame-pair kernels.  A single frame (or the odd last
    // one) is swept as a pair with itself -- half the packed lanes idle, still 1.6 x the rate of das_fir8_kernel.
    const long fir_wgs = (long) ((h->cfg.pixel_count + 63) / 64) * ((batch + 1) / 2);
    if (h->cfg.interp == AWPU_INTERP_FIR8 && h->cfg.math == AWPU_MATH_F32_FAST && env().pairs != 0 &&
        (fir_wgs >= (batch >= 2 ? 256 : 192) || (env().fir_planes == 2 && batch >= 2))) {
        if (env().fir_planes && awpu::fir8_plane_plan(h->window, h->usable(), &h->fir_plane_plan))
            return launch_fir8_planes(h, d_frames, batch, d_power, s, hist_eff, wstart_eff);
    }
    if (h->exact_pairs_ok && env().exact_pairs != 0) {
        // vertical pixel quads on the {next, d} layout (round 5) where the row length is known and the table's statistics favour them
        // (takes_exact_nd; AWPU_SHAPE=exact_quad: round 4's kernel on raw sample pairs; exact_pair: the two-pixel block everywhere)
        int nq = 1;
        const bool nd = takes_exact_nd(h, batch, &nq);
        // one frame per call (MIMOWorker::update's regime): the halves form of the layout -- the two packed lanes are the two halves of
        // the block, not a frame and its copy; every mic resident where one array's rows fit the LDS (no pre-pass)
        const int ex = env().exact_pairs;
        const bool grid_known = h->cfg.grid_columns >= 1 && h->cfg.pixel_count % h->cfg.grid_columns == 0;
        if (batch == 1 && (ex == 1 || ex == 6) && grid_known && (h->exact_ndhs_ok || h->exact_ndh_ok)) {
            const int rows = h->cfg.pixel_count / h->cfg.grid_columns, cols = h->cfg.grid_columns;
            if (h->n_cus < 1 && (hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess || h->n_cus < 1))
                h->n_cus = 256;
            // grids of at most 32 pixels per CU (two rounds of 16-wave workgroups): one PIXEL per wave (das_exact_ndp_kernel) -- a quad
            // kernel leaves such a grid one or two waves per SIMD, and the frame then takes as long as one wave's instruction issue.
            // Measured, 256 mics, one frame per call, quads -> pixels: 64 x 64 69.1 -> 31.6 us; 72^2 69.3 -> 52.1; 80^2 67.2 -> 53.3;
            // 88^2 66.1 -> 56.5; 96^2 (a third round) 81.1 -> 76.4; 100^2 69.3 -> 77.9 (profiles/r05_single_frame_ablation.txt).  It shares
            // no reads between pixels, so it also serves tables whose quads do not share (where the quad kernels are not chosen at all)
            // One array (every mic resident in the quad kernel, no pre-pass) against pixels behind the pre-pass: 32^2 (its quads do not
            // share: das_exact_pair_kernel) 36.9 -> 10.4 us; 48^2 22.3 -> 10.8; 64^2 21.1 -> 11.5; 80^2 20.3 -> 18.6; 100^2 (three rounds) 21.3 -> 26.0
            bool solo = h->exact_ndh_ok && awpu::ndp_tiles(rows, cols) * (long) batch <= 2L * h->n_cus;
            if (ex == 6) solo = h->exact_ndh_ok;
#ifdef AWPU_TUNING_BUILD
            if (const char *v = std::getenv("AWPU_NDH_WAVES")) solo = std::atoi(v) == 1 && h->exact_ndh_ok;
#endif
            if (solo) return launch_exact_ndh(h, d_frames, batch, d_power, s, hist_eff, wstart_eff, false, 1, true);
            // (smaller workgroups of quads -- 8 or 4 waves, one round -- were the first answer to such grids: c2 76.6 -> 71.7 / 68.9 us;
            // one pixel per wave replaced them)
            if (nd) return launch_exact_ndh(h, d_frames, batch, d_power, s, hist_eff, wstart_eff, h->exact_ndhs_ok, (long) awpu::quad1_tiles(rows, cols, 2) >= 256 ? 2 : 1);
        }
        if (nd) return launch_exact_nd(h, d_frames, batch, d_power, s, hist_eff, wstart_eff, nq);
        if (env().exact_pairs == 3 && h->pair_cols > 0 && h->cfg.pixel_count / h->cfg.grid_columns >= 4)
            return launch_exact_quads(h, d_frames, batch, d_power, s, hist_eff, wstart_eff);
        return launch_exact_pairs(h, d_frames, batch, d_power, s, hist_eff, wstart_eff);
    }
    if (h->sums_out) return fail(AWPU_ERR_STATE, "the pre-epilogue sums are exported by the frame-pair reference-order kernel only");
    if (h->cfg.math != AWPU_MATH_F32_FAST || h->cfg.interp == AWPU_INTERP_FIR8)
        return launch_exact(h, d_frames, batch, d_power, s, hist_eff, wstart_eff);

    // ---- frame-pair shape: batches on grids that fill the chip (AWPU_FAST_PAIRS=0/1 overrides)
    const int env_pairs = env().pairs;
    const long pair_wgs = (long) awpu::pair_tiles(h->cfg.pixel_count, h->pair_cols) * ((batch + 1) / 2);
    if (layout != kRing && batch >= 2 && h->quad_ok && env_pairs != 0 &&
        ((long) awpu::quad_tiles(h->cfg.pixel_count / h->cfg.grid_columns, h->cfg.grid_columns) * ((batch + 1) / 2) >= 256 ||
         env().quads == 1))
        return launch_quads(h, d_frames, batch, d_power, s, hist_eff, wstart_eff);
    // ---- stationary shape: the whole window of every active mic of a frame pair fits the LDS (one 8x8 array does)
    if (layout != kRing && batch >= 2 && env_pairs != 0 && env().stationary != 0) {
        awpu::FastPlan sp;
        const long tiles = awpu::pair_tiles(h->cfg.pixel_count, h->pair_cols), pairs = (batch + 1) / 2;
        if (awpu::pair_plan_stationary(h->window, h->usable(), &sp) && (pairs * tiles >= 128 || env().stationary == 1)) {
            const awpu_hip::FastLut *slut = nullptr;
            const int rc = build_fast_lut(h, 2, -2, &slut);
            if (rc == AWPU_OK) {
                // tiles per workgroup: enough to amortise the staging, few enough to leave every CU a workgroup
#ifndef AWPU_STATIONARY_WGS
#define AWPU_STATIONARY_WGS 256  // (tuning builds: -DAWPU_STATIONARY_WGS=512 ... through tools/build_variant.sh)
#endif
                const int tpw = (int) std::max<long>(1, std::min<long>(tiles, pairs * tiles / AWPU_STATIONARY_WGS));
                return launch_pairs(h, slut, d_frames, batch, d_power, s, hist_eff, wstart_eff, tpw);
            }
            if (rc != AWPU_ERR_INVALID) return rc;
        }
    }
    if (layout != kRing && batch >= 2 && env_pairs != 0 && (pair_wgs >= 256 || env_pairs == 1)) {
        const awpu_hip::FastLut *plut = nullptr;
        const int rc = build_fast_lut(h, 2, -1, &plut);
        if (rc == AWPU_OK) return launch_pairs(h, plut, d_frames, batch, d_power, s, hist_eff, wstart_eff);
        // a window too wide for the pair image is served by the single-frame shapes below; anything
        // else (an allocation or copy that failed) is the caller's to know about
        if (rc != AWPU_ERR_INVALID) return rc;
    }
    // ---- single frames on grids of at most 16 pixels per CU: the reference-order kernel with one pixel per wave
    // (das_exact_ndp_kernel) is the fastest sweep this library has for them in ANY mode (c2, one frame per call: 47.6 us on this
    // mode's 8-wave shape, 29.9 us there; c1 11.7 -> 10.1) and its powers are the reference's own arithmetic -- inside this
    // mode's contract on any input.  Not when a shape of this mode is forced (AWPU_SHAPE: the tests sweep each through the oracle)
    if (h->fast_ndp_ok && h->cfg.grid_columns >= 1 && h->cfg.pixel_count % h->cfg.grid_columns == 0 && env().fpi == 0 &&
        env().halves == -1 && env().quads == -1 && env_pairs == -1 && env().stationary == -1) {
        if (h->n_cus < 1 && (hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess || h->n_cus < 1))
            h->n_cus = 256;
        if (awpu::ndp_tiles(h->cfg.pixel_count / h->cfg.grid_columns, h->cfg.grid_columns) * (long) batch <= h->n_cus)
            return launch_exact_ndh(h, d_frames, batch, d_power, s, hist_eff, wstart_eff, false, 1, true);
    }
    // ---- single frames on a grid whose table favours the quad shape (a forced single-frame shape goes past): the halves
    // layout behind a pack + filter pre-pass
    if (h->quadhs_fits && env().fpi == 0 && env().halves != 0 && env().stationary != 0) {
        // one 8x8 array (every mic's halves row fits the LDS): one launch, no pack pre-pass, no chunks.  A call this small is
        // latency, not throughput: taken from 48 workgroups on (below that the 8-wave shapes spread a tiny grid over more CUs)
        const int rows = h->cfg.pixel_count / h->cfg.grid_columns, cols = h->cfg.grid_columns;
        const int qpw = (long) awpu::quad1_tiles(rows, cols, 2) * batch >= 256 ? 2 : 1;
        int rb = 0, rw = 0, io = 0;
        if (((long) awpu::quad1_tiles(rows, cols, qpw) * batch >= 48 || env().quads == 1 || env().halves == 1 || env().stationary == 1) &&
            awpu::quadh_stationary_raw(h->quadhs_plan, h->usable(), wstart_eff, hist_eff, &rb, &rw, &io))
            return launch_quadsh_stationary(h, d_frames, batch, d_power, s, hist_eff, layout == kRing ? AWPU_HIST : hist_eff, wstart_eff, qpw);
    }
    if (h->quadh_fits && env().fpi == 0 && env().halves != 0) {
        const int rows = h->cfg.pixel_count / h->cfg.grid_columns, cols = h->cfg.grid_columns;
        const int qpw = (long) awpu::quad1_tiles(rows, cols, 2) * batch >= 256 ? 2 : 1;
        // from 96 workgroups on.  (Measured, one frame per call, 256 mics: a 100 x 100 grid = 175 workgroups 54.5 us here against
        // 98.9 us for the 8-wave shape; 64 x 64 = 64 workgroups 56.5 against 47.8 -- this kernel's time is its 256-stage
        // dependent chain whatever the grid, the 8-wave shape's grows with the pixels: they cross near 80 workgroups.  The
        // threshold of rounds 2-3 was 192 and sent the four-array, default-resolution case to the slower shape.)
        if ((long) awpu::quad1_tiles(rows, cols, qpw) * batch >= 96 || env().quads == 1 || env().halves == 1) {
            // pitch between streams: the ring's rows are 2048 floats apart; the history a stream offers the filter is hist_eff
            const int pitch = hist_eff;
            return launch_quadsh(h, d_frames, batch, d_power, s, pitch, layout == kRing ? AWPU_HIST : hist_eff, wstart_eff, qpw);
        }
    }
    int fpi = 1, ppw = 8, nw = 8;
    choose_fast_variant(h, batch, &fpi, &ppw, &nw);
    const awpu_hip::FastLut *lut = nullptr;
    int rc = build_fast_lut(h, fpi, awpu::fast_image_bytes(nw), &lut);
    if (rc != AWPU_OK && rc != AWPU_ERR_INVALID) return rc;
    // the double-buffered shapes read whole 16-byte pieces of every staged row
    if (nw != 8 && (rc != AWPU_OK || !awpu::fast_db_fits(lut->plan) || wstart_eff + 1 + lut->plan.wr > hist_eff)) {
        nw = 8;
        if (ppw > 4) ppw = 4;  // (the 8-wave shape is built for 2 and 4 pixels per wave)
        rc = build_fast_lut(h, fpi, awpu::fast_image_bytes(nw), &lut);
    }
    if (rc != AWPU_OK) return rc;
    const awpu::FastPlan &plan = lut->plan;
    awpu::FastArgs a{};
    a.frames = d_frames;
    a.lut = lut->d;
    a.index = h->d_index;
    a.row_off = compact ? h->d_row_off_compact : (layout == kRing ? h->d_row_off_ring : h->d_row_off);
    a.power = d_power;
    a.n_streams = h->cfg.n_streams;
    a.hist = hist_eff;
    a.usable = h->usable();
    a.usable_pad = plan.usable_pad;
    a.pixel_count = h->cfg.pixel_count;
    a.wstart = wstart_eff;
    a.wr = plan.wr;
    a.chunk = plan.chunk;
    a.batch = batch;
    // frames per persistent workgroup: measured, persistence over frames buys nothing
    a.frames_per_wg = std::min(env().fpw > 0 ? env().fpw : 1, batch);
    a.debug = env().debug;
    a.debug_out = nullptr;
    size_t n_waves = 0;
    const int wg_waves = nw == 24 ? 12 : 16;
    if ((a.debug & 16) && nw != 8) {  // the double-buffered shapes have a stamped build
        n_waves = (size_t) wg_waves * ((batch + a.frames_per_wg - 1) / a.frames_per_wg) *
                  ((h->cfg.pixel_count + wg_waves * ppw - 1) / (wg_waves * ppw));
        rc = ensure_diag(h, n_waves * 12);
        if (rc != AWPU_OK) return rc;
        a.debug_out = h->d_diag;
    }
    if (h->timing) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, s));
    AWPU_HIP_TRY(awpu::launch_das_fast(a, fpi, ppw, nw, {lut->entries, 0}, s));
    rc = finish_launch(h, batch, s, nw == 32 ? AWPU_KERNEL_SINGLE_DB : (nw == 8 && fpi == 1 && ppw <= 4 ? AWPU_KERNEL_SINGLE_SMALL : AWPU_KERNEL_TUNING));
    if (rc != AWPU_OK || !a.debug_out) return rc;
    return dump_diag(h, n_waves, wg_waves, "single", s);
}

int check_ready(awpu_hip *h, int batch) {
    if (!h) return invalid("null handle");
    if (batch < 1 || batch > h->cfg.max_batch) return invalid("batch outside [1, max_batch]");
    if (!h->have_table || !h->have_mics) {
        return fail(AWPU_ERR_STATE, "delay table and active mics must be set before processing");
    }
    if (h->cfg.interp == AWPU_INTERP_FIR8 && !h->have_fir) {
        return fail(AWPU_ERR_STATE, "AWPU_INTERP_FIR8 needs awpu_hip_set_fir_table");
    }
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    if (!h->prepared) {
        const int rc = prepare(h);
        if (rc != AWPU_OK) return rc;
    }
    return AWPU_OK;
}

int ensure_power(awpu_hip *h, size_t need_power) {
    if (h->power_cap < need_power) {
        retire_live_graphs(h);  // (they write through the old pointer)
        dev_free(h->d_power);
        h->power_cap = 0;
        AWPU_HIP_TRY(hipMalloc(&h->d_power, need_power * sizeof(float)));
        h->power_cap = need_power;
    }
    return AWPU_OK;
}

// ------------------------------------------------------------------------------------------------
// Device group (cfg.n_devices > 1): one handle, one part (an ordinary single-device engine) per GPU,
// each owning a contiguous slab of the handle's pixels.  Everything below runs in the caller's thread; the parts'
// streams run concurrently.  No collective library: host frames are uploaded by every device itself, device
// frames fan out from devices[0] by one peer copy per destination (a different xGMI link each).
// ------------------------------------------------------------------------------------------------
int create_group(awpu_hip_t **out, const awpu_hip_cfg &c) {
    if (c.n_devices > AWPU_MAX_DEVICES) return invalid("n_devices above AWPU_MAX_DEVICES");
    const int G = c.n_devices;
    // slabs: whole grid rows when the row length is known and the handle's range is whole rows, else pixels
    const bool by_rows = c.grid_columns > 0 && c.pixel_count % c.grid_columns == 0 && c.pixel_begin % c.grid_columns == 0;
    int unit = by_rows ? c.grid_columns : 1;
    // groups of four rows where that divides (the quad shapes sweep four rows at a time: slabs that start on a
    // multiple of four rows sweep the same quads as one device would, and give the same bits)
    if (by_rows && c.pixel_count % (4 * unit) == 0 && c.pixel_count / (4 * unit) >= G) unit *= 4;
    const int units = c.pixel_count / unit;
    if (units < G) return invalid("fewer grid rows (or pixels) than devices");
    awpu_hip *g = new (std::nothrow) awpu_hip();
    if (!g) return AWPU_ERR_NOMEM;
    g->cfg = c;
    g->cfg.device = c.devices[0];
    // Row groups of four dealt round-robin (device k owns groups k, k + G, ...) where every device gets at least two of them:
    // the sweep's cost per row grows from the centre of the sine-space grid outwards (fewer shared integer delays), and a
    // group's call takes as long as its slowest device; contiguous slabs otherwise.  Either way a quad is four adjacent grid rows.
    const bool interleave = by_rows && unit == 4 * c.grid_columns && units >= 2 * G;
    int begin = 0;
    for (int k = 0; k < G; k++) {
        awpu_hip_cfg pc = c;
        pc.n_devices = 1;
        pc.device = c.devices[k];
        const int n = units / G + (k < units % G ? 1 : 0);  // the first units % G devices take one more
        std::vector<std::pair<int, int>> ranges;
        if (interleave) {
            for (int u = k; u < units; u += G) ranges.emplace_back(u * unit, unit);
        } else {
            ranges.emplace_back(begin * unit, n * unit);
        }
        pc.pixel_begin = c.pixel_begin + (interleave ? 0 : begin * unit);  // (a part's pixels are what `ranges` says; this only has to be a row start)
        pc.pixel_count = n * unit;
        begin += n;
        awpu_hip *part = nullptr;
        int rc = awpu_hip_create(&part, &pc);
        if (rc == AWPU_OK) part->ranges = ranges;
        if (rc == AWPU_OK) {  // what the fan-out needs on top of an ordinary engine
            hipError_t e = hipStreamCreateWithFlags(&part->copy_stream, hipStreamNonBlocking);
            for (int b = 0; b < 2 && e == hipSuccess; b++) {
                e = hipEventCreateWithFlags(&part->ev_copied[b], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&part->ev_swept[b], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&part->ev_staged_read[b], hipEventDisableTiming);
            }
            if (e == hipSuccess) e = hipEventCreateWithFlags(&part->ev_done, hipEventDisableTiming);
            if (e != hipSuccess) rc = hip_fail(e, "group stream/event creation");
            g->parts.push_back(part);
        }
        if (rc != AWPU_OK) {
            const std::string why = g_last_error;
            awpu_hip_destroy(g);
            note_error(why);
            return rc;
        }
    }
    // Direct copies between devices[0] and the others need peer access both ways.  Asked for and CHECKED: a pair
    // without it (another PCIe root, IOMMU settings, a container that hides the links) takes the explicit staged path
    // through pinned host memory -- slower, correct, and said so in awpu_hip_last_error_of / awpu_hip_group_peer_status.
    std::string staged_note;
    for (int k = 0; k < G; k++) {
        awpu_hip *part = g->parts[k];
        if (c.devices[k] == c.devices[0]) {
            part->peer = env().group_copy >= 2 ? kPeerStaged : kPeerSame;
            continue;
        }
        int can_out = 0, can_in = 0;
        hipError_t e_out = hipDeviceCanAccessPeer(&can_out, c.devices[0], c.devices[k]);
        hipError_t e_in = hipDeviceCanAccessPeer(&can_in, c.devices[k], c.devices[0]);
        if (e_out == hipSuccess && e_in == hipSuccess && can_out && can_in) {
            e_out = hipSetDevice(c.devices[0]);
            if (e_out == hipSuccess) e_out = hipDeviceEnablePeerAccess(c.devices[k], 0);
            e_in = hipSetDevice(c.devices[k]);
            if (e_in == hipSuccess) e_in = hipDeviceEnablePeerAccess(c.devices[0], 0);
        }
        const auto enabled = [](hipError_t e) { return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled; };
        part->peer = can_out && can_in && enabled(e_out) && enabled(e_in) && env().group_copy < 2 ? kPeerDirect : kPeerStaged;
        if (part->peer == kPeerStaged) {
            staged_note += "device " + std::to_string(c.devices[0]) + " <-> " + std::to_string(c.devices[k]) + ": " +
                           (!(can_out && can_in) ? std::string("hipDeviceCanAccessPeer says no")
                                                 : std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(enabled(e_out) ? e_in : e_out)) + "; ";
        }
    }
    (void) hipGetLastError();
    if (!staged_note.empty())
        g->last_error = "device group without peer access (" + staged_note + "): frames and tiles are staged through pinned host memory";
    AWPU_HIP_TRY(hipSetDevice(c.devices[0]));
    hipError_t e = hipEventCreateWithFlags(&g->ev_fan, hipEventDisableTiming);
    for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipEventCreateWithFlags(&g->ev_staged[b], hipEventDisableTiming);
    // a staged part's ev_tile_free[] is recorded on the CALLER's stream (devices[0]) and only waited for on the part's own:
    // an event must be recorded on a stream of the device it was created on, so these belong to devices[0], not the part's
    for (awpu_hip *part : g->parts)
        for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipEventCreateWithFlags(&part->ev_tile_free[b], hipEventDisableTiming);
    if (e != hipSuccess) {
        awpu_hip_destroy(g);
        return hip_fail(e, "group event creation");
    }
    *out = g;
    return AWPU_OK;
}

// a part ran into an error: the group reports it as its own
int part_failed(awpu_hip *g, awpu_hip *part, int rc) {
    g->last_error = part->last_error.empty() ? g_last_error : part->last_error;
    g_last_error = g->last_error;
    return rc;
}

template <class F>
int for_each_part(awpu_hip *g, F f) {
    for (awpu_hip *part : g->parts) {
        const int rc = f(part);
        if (rc != AWPU_OK) return part_failed(g, part, rc);
    }
    return AWPU_OK;
}

// switches a handle's event bracket off for one asynchronous call and back on whichever way the call ends
struct TimingOff {
    awpu_hip *h;
    bool keep;
    explicit TimingOff(awpu_hip *h_) : h(h_), keep(h_->timing) { h->timing = false; }
    ~TimingOff() { h->timing = keep; }
};


// upload of host frames + the sweep into h->d_power, all on h->stream, nothing waited for
int enqueue_host_process(awpu_hip *h, const float *frames, int batch) {
    int rc = check_ready(h, batch);
    if (rc != AWPU_OK) return rc;
    const bool compact = h->compact_hist > 0;
    const int dev_hist = compact ? h->compact_hist : h->cfg.hist;
    const size_t need_frames = (size_t) h->cfg.n_streams * dev_hist * batch;
    if (h->frames_cap < need_frames) {
        dev_free(h->d_frames);
        h->frames_cap = 0;
        AWPU_HIP_TRY(hipMalloc(&h->d_frames, need_frames * sizeof(float)));
        h->frames_cap = need_frames;
    }
    rc = ensure_power(h, (size_t) h->cfg.pixel_count * batch);
    if (rc != AWPU_OK) return rc;
    // Large batches go up in pieces on a second stream, so that piece k+1 crosses PCIe while piece k is swept (the
    // pieces are whole frame pairs: the same arithmetic as one launch).  last_kernel_ms then spans all the sweeps.
    const int n_pieces = batch >= 128 ? 4 : (batch >= 64 ? 2 : 1);
    const int piece = ((batch + n_pieces - 1) / n_pieces + 1) & ~1;
    if (n_pieces > 1) {
        if (!h->copy_stream) AWPU_HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t *ev : {&h->ev_copied[0], &h->ev_copied[1]})
            if (!*ev) AWPU_HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    }
    const bool keep_timing = h->timing;
    int turn = 0;
    for (int b0 = 0; b0 < batch; b0 += piece, turn++) {
        const int nb = std::min(piece, batch - b0);
        hipStream_t up = n_pieces > 1 ? h->copy_stream : h->stream;
        float *dst = h->d_frames + (size_t) b0 * h->cfg.n_streams * dev_hist;
        const float *src = frames + (size_t) b0 * h->cfg.n_streams * h->cfg.hist;
        if (compact) {  // rows of compact_hist floats cut out of rows of hist floats: a third of the PCIe bytes
            AWPU_HIP_TRY(hipMemcpy2DAsync(dst, (size_t) dev_hist * sizeof(float), src + h->wstart, (size_t) h->cfg.hist * sizeof(float),
                                          (size_t) dev_hist * sizeof(float), (size_t) nb * h->cfg.n_streams, hipMemcpyHostToDevice, up));
        } else {
            AWPU_HIP_TRY(hipMemcpyAsync(dst, src, (size_t) nb * h->cfg.n_streams * dev_hist * sizeof(float), hipMemcpyHostToDevice, up));
        }
        if (n_pieces > 1) {
            AWPU_HIP_TRY(hipEventRecord(h->ev_copied[turn & 1], up));
            AWPU_HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_copied[turn & 1], 0));
            if (keep_timing && b0 == 0) AWPU_HIP_TRY(hipEventRecord(h->ev_begin, h->stream));
            h->timing = false;
        }
        rc = launch(h, dst, nb, h->d_power + (size_t) b0 * h->cfg.pixel_count, h->stream, compact ? kCompact : kFull);
        h->timing = keep_timing;
        if (rc != AWPU_OK) return rc;
    }
    if (n_pieces > 1 && keep_timing) AWPU_HIP_TRY(hipEventRecord(h->ev_end, h->stream));
    return AWPU_OK;
}

// h->d_power [batch][pixel_count] -> host image `power` [batch][pitch] (pitch = this handle's pixels, or the whole group's: a
// part's pixel ranges then land where they belong in the wider image), on h->stream
int enqueue_power_to_host(awpu_hip *h, int batch, float *power, size_t pitch) {
    const size_t row = (size_t) h->cfg.pixel_count * sizeof(float);
    if (h->ranges.empty()) {
        if (pitch == (size_t) h->cfg.pixel_count) {
            AWPU_HIP_TRY(hipMemcpyAsync(power, h->d_power, row * batch, hipMemcpyDeviceToHost, h->stream));
        } else {
            AWPU_HIP_TRY(hipMemcpy2DAsync(power, pitch * sizeof(float), h->d_power, row, row, (size_t) batch, hipMemcpyDeviceToHost, h->stream));
        }
        return AWPU_OK;
    }
    size_t done = 0;  // pixels of the part's own rows already sent
    for (const auto &r : h->ranges) {
        AWPU_HIP_TRY(hipMemcpy2DAsync(power + r.first, pitch * sizeof(float), h->d_power + done, row, (size_t) r.second * sizeof(float),
                                      (size_t) batch, hipMemcpyDeviceToHost, h->stream));
        done += (size_t) r.second;
    }
    return AWPU_OK;
}

int wait_and_time(awpu_hip *h);

// One frame in, one heatmap out, synchronously: the call MIMOWorker::update makes once per 256-sample block (mimo.cpp:100-103 is the
// snapshot it replaces).  The caller's buffers are pageable (std::vector, mimo.h:83-88): a device copy straight out of / into them goes
// through the runtime's own bounce buffers in several synchronous steps (measured at the reference's default shape: 58-64 us per call
// around a 20 us sweep).  Here the touched window of every stream is gathered into a PINNED buffer of the handle by the CPU (64 rows x
// 1.2 KB), crosses PCIe in ONE piece read by a small kernel (a DMA-engine copy of 75 KB is mostly start-up), and the sweep stores its
// powers straight into a pinned buffer.  Measured at the reference's shape (C level; Python adds ~5 us): 53 -> 43 us exact, 48 -> 38 us
// fast, and -- with the event bracket around the sweep sampled instead of recorded on every call -- 42 / 37 us: gather 1.3, the two launches
// 2.8 + ~3, then ~33 us until the stream is idle (upload 3 + sweep 21.6 / 19.5 + dispatch latencies + the end-of-kernel release and its
// signal), copy out 2.0; and in the default mode at the reference's shape -- completion by a flag the resident kernel's last workgroup
// stores behind its powers (launch_exact_ndh: done_*) instead of by the stream's signal -- 36.5 us.  Measured and not kept: spinning on
// hipStreamQuery instead of hipStreamSynchronize (equal); a stream-written flag (hipStreamWriteValue32: 18 us slower); the workgroup flag
// with the powers written through as they are stored (10 000 acknowledged four-byte PCIe writes: +60 us; gathered into 64-byte lines
// first: what ships).
int live_host_call(awpu_hip *h, const float *frames, float *power) {
    int rc = check_ready(h, 1);
    if (rc != AWPU_OK) return rc;
    const bool compact = h->compact_hist > 0;
    const int dev_hist = compact ? h->compact_hist : h->cfg.hist;
    const size_t need_frames = (size_t) h->cfg.n_streams * dev_hist;
    if (h->frames_cap < need_frames) {
        dev_free(h->d_frames);
        h->frames_cap = 0;
        AWPU_HIP_TRY(hipMalloc(&h->d_frames, need_frames * sizeof(float)));
        h->frames_cap = need_frames;
    }
    rc = ensure_power(h, (size_t) h->cfg.pixel_count);
    if (rc != AWPU_OK) return rc;
    if (h->live_in_cap < need_frames) {
        if (h->h_live_in) (void) hipHostFree(h->h_live_in);
        h->h_live_in = nullptr;
        h->live_in_cap = 0;
        AWPU_HIP_TRY(hipHostMalloc(&h->h_live_in, need_frames * sizeof(float), hipHostMallocDefault));
        h->live_in_cap = need_frames;
    }
    if (h->live_out_cap < (size_t) h->cfg.pixel_count) {
        if (h->h_live_out) (void) hipHostFree(h->h_live_out);
        h->h_live_out = nullptr;
        h->live_out_cap = 0;
        AWPU_HIP_TRY(hipHostMalloc(&h->h_live_out, (size_t) h->cfg.pixel_count * sizeof(float), hipHostMallocDefault));
        h->live_out_cap = (size_t) h->cfg.pixel_count;
    }
#ifdef AWPU_TUNING_BUILD
    static const bool live_timing = std::getenv("AWPU_LIVE_TIMING") != nullptr;
    static double t_acc[5] = {0, 0, 0, 0, 0};
    static long t_calls = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](int k, std::chrono::steady_clock::time_point &from) {
        const auto now = std::chrono::steady_clock::now();
        t_acc[k] += std::chrono::duration<double, std::micro>(now - from).count();
        from = now;
    };
    auto t = t0;
#endif
    if (compact) {  // rows of compact_hist floats cut out of rows of hist floats
        for (int s = 0; s < h->cfg.n_streams; s++)
            std::memcpy(h->h_live_in + (size_t) s * dev_hist, frames + (size_t) s * h->cfg.hist + h->wstart, (size_t) dev_hist * sizeof(float));
    } else {
        std::memcpy(h->h_live_in, frames, need_frames * sizeof(float));
    }
#ifdef AWPU_TUNING_BUILD
    if (live_timing) lap(0, t);
#endif
    // the upload: by a kernel that reads the pinned buffer over PCIe (a DMA-engine copy of 75 KB is mostly start-up: 8.6 us measured);
    // windows that are no whole number of 16-byte pieces (never, with compact rows) take the DMA copy
    if ((need_frames & 3) == 0) {
        AWPU_HIP_TRY(awpu::launch_upload_floats(h->h_live_in, h->d_frames, need_frames, h->stream));
    } else {
        AWPU_HIP_TRY(hipMemcpyAsync(h->d_frames, h->h_live_in, need_frames * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
#ifdef AWPU_TUNING_BUILD
    if (live_timing) lap(1, t);
#endif
    // (the sweep stores its powers straight into the pinned buffer -- one 4-byte store per pixel over PCIe, complete when the kernel
    // is: a device-to-host copy behind the sweep would be one more DMA start-up, ~10 us, for 40 KB)
    // the event bracket around the sweep (awpu_hip_stats.last_kernel_ms) is two more packets on the stream and two more runtime calls:
    // 3.3 us of a call of 50 (measured from Python, both modes).  This path brackets its first call and every 32nd after it; the other
    // calls leave last_kernel_ms / total_kernel_ms as they are
    {
        const bool timed = (h->live_calls++ & 31) == 0;
        const bool keep = h->timing;
        h->timing = keep && timed;
        h->done_arm = !h->timing;  // (a timed call waits for the stream: its end event)
        rc = launch(h, h->d_frames, 1, h->h_live_out, h->stream, compact ? kCompact : kFull);
        h->done_arm = false;
        if (rc == AWPU_OK) {
#ifdef AWPU_TUNING_BUILD
            if (live_timing) lap(2, t);
#endif
            bool seen = false;
            if (h->done_used) {
                // the sweep's last workgroup stores the call's number behind its powers (das_fast.hip: store_tile_and_signal): ~7 us
                // sooner than the stream's completion signal.  Should it not arrive within 20 ms (it arrives within the sweep's ~25 us),
                // the stream's own completion decides
                const auto spin_from = std::chrono::steady_clock::now();
                for (unsigned spins = 0;; spins++) {
                    if (__atomic_load_n(h->h_done_flag, __ATOMIC_ACQUIRE) == h->done_seq) {
                        seen = true;
                        break;
                    }
                    __builtin_ia32_pause();
                    if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - spin_from > std::chrono::milliseconds(20)) break;
                }
            }
            if (!seen) rc = wait_and_time(h);
        }
        h->timing = keep;
    }
    if (rc != AWPU_OK) return rc;
#ifdef AWPU_TUNING_BUILD
    if (live_timing) lap(3, t);
#endif
    std::memcpy(power, h->h_live_out, (size_t) h->cfg.pixel_count * sizeof(float));
#ifdef AWPU_TUNING_BUILD
    if (live_timing) {
        lap(4, t);
        if (++t_calls == 5) for (double &v : t_acc) v = 0;  // (the first calls build tables and raise limits)
        if (t_calls > 5 && (t_calls - 5) % 100 == 0) {
            std::fprintf(stderr, "[awpu live] per call us: gather %.1f | hipMemcpyAsync H2D %.1f | launch() %.1f | wait %.1f | copy out %.1f\n", t_acc[0] / 100,
                         t_acc[1] / 100, t_acc[2] / 100, t_acc[3] / 100, t_acc[4] / 100);
            for (double &v : t_acc) v = 0;
        }
    }
#endif
    return AWPU_OK;
}

int wait_and_time(awpu_hip *h) {
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->timing) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, h->ev_begin, h->ev_end) == hipSuccess) {
            h->stats.last_kernel_ms = ms;
            h->stats.total_kernel_ms += ms;
        }
    }
    return AWPU_OK;
}

int group_process(awpu_hip *g, const float *frames, int batch, float *power) {
    const size_t pitch = (size_t) g->cfg.pixel_count;
    int rc = for_each_part(g, [&](awpu_hip *part) {
        AWPU_CTX(part);
        const int r = enqueue_host_process(part, frames, batch);
        return r != AWPU_OK ? r : enqueue_power_to_host(part, batch, power, pitch);
    });
    if (rc != AWPU_OK) return rc;
    return for_each_part(g, [&](awpu_hip *part) { return wait_and_time(part); });
}

// Would launch() sweep this batch with one of the frame-pair shapes that read the packed layout (quad or pair)?  The rule of
// awpu_hip_process_packed, asked before anything is packed.
bool takes_packed_pairs(awpu_hip *h, int batch, awpu::FastPlan *plan) {
    if (batch < 2 || h->cfg.interp != AWPU_INTERP_LERP || h->usable() % 4 != 0 || !h->gain.empty()) return false;
    if (h->cfg.math == AWPU_MATH_F32_EXACT) {  // the reference's order: the {next, d} rows of das_exact_nd_kernel
        int nq = 1;
        if (!takes_exact_nd(h, batch, &nq)) return false;
        *plan = h->exact_nd_plan;
        return true;
    }
    if (h->cfg.math != AWPU_MATH_F32_FAST || env().pairs == 0) return false;
    if (!awpu::pair_plan(h->window, h->usable(), plan)) return false;
    const long pairs = (batch + 1) / 2;
    const bool quad_fills = h->quad_ok && h->quad_plan.wr == plan->wr && h->quad_plan.usable_pad == h->usable() &&
                            ((long) awpu::quad_tiles(h->cfg.pixel_count / h->cfg.grid_columns, h->cfg.grid_columns) * pairs >= 256 || env().quads == 1);
    const bool pair_fills = (long) awpu::pair_tiles(h->cfg.pixel_count, h->pair_cols) * pairs >= 256 || env().pairs == 1;
    return quad_fills || pair_fills;
}

// floats of `batch` frames in the packed layout of `plan` (rows of plan.row_bytes: sample pairs of a frame pair, or their {next, d}
// elements), `usable` rows per pair (the packed entry points ask for usable % 4 == 0: no padding rows)
size_t packed_floats_of(const awpu_hip *h, const awpu::FastPlan &plan, int batch) {
    return (size_t) ((batch + 1) / 2) * h->usable() * (size_t) (plan.row_bytes / 4);
}
// the sweep's pack pass into a caller's buffer: pre-filtered sample pairs (FAST) or {next, d} elements (EXACT)
int pack_for_sweep(awpu_hip *h, const awpu::FastPlan &plan, const float *d_frames, int batch, float *d_packed, hipStream_t s) {
    if (h->cfg.math == AWPU_MATH_F32_EXACT)
        AWPU_HIP_TRY(awpu::launch_pack_nd(d_frames, h->cfg.n_streams, h->cfg.hist, h->wstart, h->d_index, h->usable(), h->usable(), nullptr, plan.wr,
                                          batch, d_packed, s));
    else
        AWPU_HIP_TRY(awpu::launch_pack_pairs(d_frames, h->cfg.n_streams, h->cfg.hist, h->wstart, h->d_index, h->usable(), h->usable(), nullptr,
                                             plan.wr, batch, d_packed, true, s));
    return AWPU_OK;
}

// the sweep of packed frame pairs (what awpu_hip_process_packed does once its arguments are checked)
int sweep_packed(awpu_hip *h, const awpu::FastPlan &plan, const float *d_packed, size_t packed_floats, int batch, float *d_power,
                 hipStream_t s) {
    if (h->cfg.math == AWPU_MATH_F32_EXACT) {
        int nq = 1;
        if (!takes_exact_nd(h, batch, &nq)) return fail(AWPU_ERR_STATE, "packed frames: this batch is not swept by the {next, d} kernel");
        return launch_exact_nd(h, nullptr, batch, d_power, s, h->cfg.hist, h->wstart, nq, d_packed, packed_floats);
    }
    const bool quad_fills = h->quad_ok && ((long) awpu::quad_tiles(h->cfg.pixel_count / h->cfg.grid_columns, h->cfg.grid_columns) *
                                               ((batch + 1) / 2) >= 256 || env().quads == 1);
    if (quad_fills && env().pairs != 0 && h->quad_plan.wr == plan.wr && h->quad_plan.usable_pad == h->usable())
        return launch_quads(h, nullptr, batch, d_power, s, h->cfg.hist, h->wstart, d_packed, packed_floats);
    const awpu_hip::FastLut *plut = nullptr;
    const int rc = build_fast_lut(h, 2, -1, &plut);
    if (rc != AWPU_OK) return rc;
    return launch_pairs(h, plut, nullptr, batch, d_power, s, h->cfg.hist, h->wstart, 0, d_packed, packed_floats);
}

// Every part of a group stages the same window -- the union of what the parts' own rows touch -- so that ONE packed buffer
// serves them all (the layout's row length and first sample follow the window).  Results do not depend on the window.
int group_union_window(awpu_hip *g, int batch) {
    if (g->union_window_done) return AWPU_OK;
    int lo = g->cfg.hist, hi = 0;
    for (awpu_hip *part : g->parts) {
        lo = std::min(lo, part->wstart);
        hi = std::max(hi, part->wstart + part->window);
    }
    for (awpu_hip *part : g->parts) {
        if (part->wstart == lo && part->wstart + part->window == hi) continue;
        AWPU_CTX(part);
        part->cfg.window_begin = lo;
        part->cfg.window_end = hi;
        part->prepared = false;
        const int rc = check_ready(part, batch);
        if (rc != AWPU_OK) return part_failed(g, part, rc);
    }
    g->union_window_done = true;
    return AWPU_OK;
}

// a part's tile [batch][its pixels, back to back] -> the group's image [batch][pitch floats]: one 2-D copy per pixel range
int tile_to_image(awpu_hip *part, const float *tile, float *image, size_t pitch_floats, int batch, hipMemcpyKind kind, hipStream_t s) {
    const size_t row = (size_t) part->cfg.pixel_count * sizeof(float);
    size_t done = 0;
    for (const auto &r : part->ranges) {
        AWPU_HIP_TRY(hipMemcpy2DAsync(image + r.first, pitch_floats * sizeof(float), tile + done, row, (size_t) r.second * sizeof(float),
                                      (size_t) batch, kind, s));
        done += (size_t) r.second;
    }
    return AWPU_OK;
}

// Frames and power in the memory of devices[0], on the caller's stream there.  What travels to the other devices:
//   * batches that the parts sweep with a frame-pair shape (takes_packed_pairs): devices[0] runs the sweep's pack pass ONCE
//     (two frames interleaved, filtered: the packed frame pairs of awpu_hip_pack_frames, the exchange format of the
//     one-process-per-GPU path too) and every other device gets that buffer by ONE linear peer copy on its copy stream and
//     sweeps it as it arrives -- no window cut on devices[0], no pack pass anywhere else;
//   * everything else (single frames, FIR8, exact math, gains): the window of every stream that the tables touch, by one 2-D
//     peer copy per device, and every device runs its whole sweep.
// The parts' pixel ranges are swept concurrently; the tiles return by peer copies on the caller's stream, which thereby waits
// for all of it.  Parts without peer access to devices[0] (kPeerStaged) get the same bytes through pinned host memory: ONE
// copy down on the caller's stream for all of them, one copy up per part on its copy stream; their tiles return the same
// way.  Two buffers everywhere, so that call k+1's copies run beside call k's sweeps.
int group_process_device(awpu_hip *g, const float *d_frames, int batch, float *d_power, hipStream_t stream) {
    const int dev0 = g->cfg.devices[0];
    AWPU_HIP_TRY(hipSetDevice(dev0));
    hipStream_t s = stream ? stream : g->parts[0]->stream;
    int rc = for_each_part(g, [&](awpu_hip *part) {
        AWPU_CTX(part);
        return check_ready(part, batch);  // (tables packed: every part's window is known)
    });
    if (rc != AWPU_OK) return rc;
    rc = group_union_window(g, batch);
    if (rc != AWPU_OK) return rc;
    g->stats.group_exchange = AWPU_EXCHANGE_WINDOWS;
    auto in_place = [&](const awpu_hip *part) { return part->cfg.device == dev0 && !env().group_copy; };

    // ---- packed frame pairs, where every part sweeps them
    awpu::FastPlan pplan{};
    bool packed = true;
    for (awpu_hip *part : g->parts) {
        awpu::FastPlan one{};
        packed = packed && takes_packed_pairs(part, batch, &one);
        if (packed && pplan.wr && (one.wr != pplan.wr || one.usable_pad != pplan.usable_pad)) packed = false;
        pplan = one;
    }
    const size_t packed_floats = packed ? packed_floats_of(g->parts[0], pplan, batch) : 0;
    int pb = 0;  // which of the group's two packed buffers this call fills
    AWPU_HIP_TRY(hipSetDevice(dev0));
    if (packed) {
        g->stats.group_exchange = AWPU_EXCHANGE_PACKED_PAIRS;
        const size_t cap = packed_floats_of(g->parts[0], pplan, g->cfg.max_batch);
        if (g->fan_cap < cap) {  // (nobody may still be reading the old buffers)
            for (awpu_hip *part : g->parts) {
                AWPU_HIP_TRY(hipSetDevice(part->cfg.device));
                AWPU_HIP_TRY(hipStreamSynchronize(part->copy_stream));
                AWPU_HIP_TRY(hipStreamSynchronize(part->stream));
            }
            AWPU_HIP_TRY(hipSetDevice(dev0));
            AWPU_HIP_TRY(hipStreamSynchronize(s));
            dev_free(g->d_fan[0]);
            dev_free(g->d_fan[1]);
            g->fan_cap = 0;
            AWPU_HIP_TRY(hipMalloc(&g->d_fan[0], cap * sizeof(float)));
            AWPU_HIP_TRY(hipMalloc(&g->d_fan[1], cap * sizeof(float)));
            g->fan_cap = cap;
            g->fan_used[0] = g->fan_used[1] = false;
        }
        pb = (int) (g->fan_turn++ & 1);
        if (g->fan_used[pb])  // buffer pb was read two calls ago: by the peers' copies and by the in-place parts' sweeps
            for (awpu_hip *part : g->parts) AWPU_HIP_TRY(hipStreamWaitEvent(s, in_place(part) ? part->ev_swept[pb] : part->ev_copied[pb], 0));
        awpu_hip *p0 = g->parts[0];
        if (const int prc = pack_for_sweep(p0, pplan, d_frames, batch, g->d_fan[pb], s); prc != AWPU_OK) return prc;
        g->fan_used[pb] = true;
    }
    AWPU_HIP_TRY(hipEventRecord(g->ev_fan, s));  // the frames (or their packed pairs) are in place once the caller's stream gets here

    // ---- staged parts: what they need goes down to pinned memory once -- the packed buffer, or the union of their windows
    int gb = 0;
    bool any_staged = false;
    for (awpu_hip *part : g->parts) any_staged |= part->peer == kPeerStaged && !in_place(part);
    if (any_staged) {
        int lo = 0, w = 0;
        size_t need = 0;
        if (packed) {
            need = packed_floats_of(g->parts[0], pplan, g->cfg.max_batch);
            lo = -1;  // (marks the packed payload: a change of payload re-sizes the staging like a change of window)
            w = (int) pplan.wr;
        } else {
            lo = g->cfg.hist;
            int hi = 0;
            for (awpu_hip *part : g->parts) {
                if (part->peer != kPeerStaged) continue;
                const bool compact = part->compact_hist > 0;
                lo = std::min(lo, compact ? part->wstart : 0);
                hi = std::max(hi, compact ? part->wstart + part->compact_hist : part->cfg.hist);
            }
            w = hi - lo;
            need = (size_t) g->cfg.n_streams * w * g->cfg.max_batch;
        }
        // (round-4 advisor) Only a buffer that is too SMALL is replaced, behind a synchronize of everybody who may still read it.  A
        // change of payload -- packed pairs one call, raw windows the next: batches alternating with single frames -- keeps the
        // buffers and their turn: every reuse of h_stage[gb] already waits for the uploads that read it two calls ago
        // (ev_staged_read below), whatever they carried.
        if (g->stage_cap < need) {
            for (awpu_hip *part : g->parts) {  // nobody may still be reading the old staging buffers
                AWPU_HIP_TRY(hipSetDevice(part->cfg.device));
                AWPU_HIP_TRY(hipStreamSynchronize(part->copy_stream));
            }
            AWPU_HIP_TRY(hipSetDevice(dev0));
            AWPU_HIP_TRY(hipStreamSynchronize(s));
            for (int b = 0; b < 2; b++) {
                if (g->h_stage[b]) (void) hipHostFree(g->h_stage[b]);
                g->h_stage[b] = nullptr;
            }
            g->stage_cap = 0;
            for (int b = 0; b < 2; b++) AWPU_HIP_TRY(hipHostMalloc(&g->h_stage[b], need * sizeof(float), hipHostMallocPortable));
            g->stage_cap = need;
            g->stage_turn = 0;
            for (awpu_hip *part : g->parts) part->stage_used[0] = part->stage_used[1] = false;
        }
        g->stage_lo = lo;  // what THIS call's payload is (read by the uploads enqueued below, in this call)
        g->stage_w = w;
        gb = g->stage_turn++ & 1;
        for (awpu_hip *part : g->parts)  // h_stage[gb] was read by the staged parts' uploads two calls ago
            if (part->peer == kPeerStaged && !in_place(part) && part->stage_used[gb]) AWPU_HIP_TRY(hipStreamWaitEvent(s, part->ev_staged_read[gb], 0));
        if (packed) {
            AWPU_HIP_TRY(hipMemcpyAsync(g->h_stage[gb], g->d_fan[pb], packed_floats * sizeof(float), hipMemcpyDeviceToHost, s));
        } else {
            AWPU_HIP_TRY(hipMemcpy2DAsync(g->h_stage[gb], (size_t) w * sizeof(float), d_frames + lo, (size_t) g->cfg.hist * sizeof(float),
                                          (size_t) w * sizeof(float), (size_t) batch * g->cfg.n_streams, hipMemcpyDeviceToHost, s));
        }
        AWPU_HIP_TRY(hipEventRecord(g->ev_staged[gb], s));
    }

    rc = for_each_part(g, [&](awpu_hip *part) {
        AWPU_CTX(part);
        AWPU_HIP_TRY(hipSetDevice(part->cfg.device));
        int r = ensure_power(part, (size_t) part->cfg.pixel_count * batch);
        if (r != AWPU_OK) return r;
        TimingOff untimed(part);  // asynchronous path: the caller times its own stream
        const bool staged = part->peer == kPeerStaged && !in_place(part);
        if (in_place(part)) {  // same GPU: sweep the caller's frames (or the group's packed buffer) in place
            AWPU_HIP_TRY(hipStreamWaitEvent(part->stream, g->ev_fan, 0));
            if (packed) {
                r = sweep_packed(part, pplan, g->d_fan[pb], g->fan_cap, batch, part->d_power, part->stream);
                if (r == AWPU_OK) AWPU_HIP_TRY(hipEventRecord(part->ev_swept[pb], part->stream));
            } else {
                r = launch(part, d_frames, batch, part->d_power, part->stream, kFull);
            }
        } else {
            const bool compact = part->compact_hist > 0;
            const int dev_hist = compact ? part->compact_hist : part->cfg.hist;
            const size_t need_window = (size_t) part->cfg.n_streams * dev_hist * part->cfg.max_batch;
            const size_t need_packed = packed ? packed_floats_of(part, pplan, part->cfg.max_batch) : 0;
            const size_t need = std::max(need_window, need_packed);
            if (part->fan_cap < need) {
                AWPU_HIP_TRY(hipStreamSynchronize(part->stream));
                AWPU_HIP_TRY(hipStreamSynchronize(part->copy_stream));
                dev_free(part->d_fan[0]);
                dev_free(part->d_fan[1]);
                part->fan_cap = 0;
                AWPU_HIP_TRY(hipMalloc(&part->d_fan[0], need * sizeof(float)));
                AWPU_HIP_TRY(hipMalloc(&part->d_fan[1], need * sizeof(float)));
                part->fan_cap = need;
                part->fan_used[0] = part->fan_used[1] = false;
            }
            // the part's own receive buffer follows the buffer it reads from: the group's packed buffer (pb) or, for a staged
            // part, the staging buffer (gb); a window copy out of the caller's frames takes its own turns
            const int b = staged ? gb : (packed ? pb : (int) (part->fan_turn++ & 1));
            if (part->fan_used[b]) AWPU_HIP_TRY(hipStreamWaitEvent(part->copy_stream, part->ev_swept[b], 0));  // buffer b is free again
            const size_t row = (size_t) dev_hist * sizeof(float);
            if (staged) {
                AWPU_HIP_TRY(hipStreamWaitEvent(part->copy_stream, g->ev_staged[gb], 0));
                if (packed) {
                    AWPU_HIP_TRY(hipMemcpyAsync(part->d_fan[b], g->h_stage[gb], packed_floats * sizeof(float), hipMemcpyHostToDevice, part->copy_stream));
                } else {
                    AWPU_HIP_TRY(hipMemcpy2DAsync(part->d_fan[b], row, g->h_stage[gb] + ((compact ? part->wstart : 0) - g->stage_lo),
                                                  (size_t) g->stage_w * sizeof(float), row, (size_t) batch * part->cfg.n_streams,
                                                  hipMemcpyHostToDevice, part->copy_stream));
                }
                AWPU_HIP_TRY(hipEventRecord(part->ev_staged_read[gb], part->copy_stream));
                part->stage_used[gb] = true;
            } else {
                AWPU_HIP_TRY(hipStreamWaitEvent(part->copy_stream, g->ev_fan, 0));
                if (packed) {  // ONE linear copy: the packed pairs of the whole batch
                    AWPU_HIP_TRY(hipMemcpyAsync(part->d_fan[b], g->d_fan[pb], packed_floats * sizeof(float), hipMemcpyDeviceToDevice, part->copy_stream));
                } else {
                    AWPU_HIP_TRY(hipMemcpy2DAsync(part->d_fan[b], row, d_frames + (compact ? part->wstart : 0),
                                                  (size_t) part->cfg.hist * sizeof(float), row, (size_t) batch * part->cfg.n_streams,
                                                  hipMemcpyDeviceToDevice, part->copy_stream));
                }
            }
            part->fan_used[b] = true;
            AWPU_HIP_TRY(hipEventRecord(part->ev_copied[b], part->copy_stream));
            AWPU_HIP_TRY(hipStreamWaitEvent(part->stream, part->ev_copied[b], 0));
            r = packed ? sweep_packed(part, pplan, part->d_fan[b], part->fan_cap, batch, part->d_power, part->stream)
                       : launch(part, part->d_fan[b], batch, part->d_power, part->stream, compact ? kCompact : kFull);
            if (r == AWPU_OK) AWPU_HIP_TRY(hipEventRecord(part->ev_swept[b], part->stream));
            if (r == AWPU_OK && staged) {  // the tile's way back starts on the part's own stream: device -> pinned
                const size_t tile = (size_t) part->cfg.pixel_count * part->cfg.max_batch;
                if (part->tile_cap < tile) {
                    AWPU_HIP_TRY(hipStreamSynchronize(part->stream));
                    for (int k = 0; k < 2; k++) {
                        if (part->h_tile[k]) (void) hipHostFree(part->h_tile[k]);
                        part->h_tile[k] = nullptr;
                    }
                    part->tile_cap = 0;
                    for (int k = 0; k < 2; k++) AWPU_HIP_TRY(hipHostMalloc(&part->h_tile[k], tile * sizeof(float), hipHostMallocPortable));
                    part->tile_cap = tile;
                    part->tile_used[0] = part->tile_used[1] = false;
                }
                if (part->tile_used[gb]) AWPU_HIP_TRY(hipStreamWaitEvent(part->stream, part->ev_tile_free[gb], 0));
                AWPU_HIP_TRY(hipMemcpyAsync(part->h_tile[gb], part->d_power, (size_t) part->cfg.pixel_count * batch * sizeof(float),
                                            hipMemcpyDeviceToHost, part->stream));
                part->tile_used[gb] = true;
            }
        }
        if (r == AWPU_OK) AWPU_HIP_TRY(hipEventRecord(part->ev_done, part->stream));
        return r;
    });
    if (rc != AWPU_OK) return rc;
    AWPU_HIP_TRY(hipSetDevice(dev0));
    const size_t pitch = (size_t) g->cfg.pixel_count;
    for (awpu_hip *part : g->parts) {  // tiles back into the caller's [batch][pixel_count] image, range by range
        AWPU_HIP_TRY(hipStreamWaitEvent(s, part->ev_done, 0));
        const bool staged = part->peer == kPeerStaged && !in_place(part);
        if (staged) {
            rc = tile_to_image(part, part->h_tile[gb], d_power, pitch, batch, hipMemcpyHostToDevice, s);
            if (rc != AWPU_OK) return rc;
            AWPU_HIP_TRY(hipEventRecord(part->ev_tile_free[gb], s));
        } else {
            rc = tile_to_image(part, part->d_power, d_power, pitch, batch, hipMemcpyDeviceToDevice, s);
            if (rc != AWPU_OK) return rc;
        }
    }
    return AWPU_OK;
}

int group_stats(awpu_hip *g, awpu_hip_stats *out) {
    awpu_hip_stats st = g->parts[0]->stats;
    st.group_exchange = g->stats.group_exchange;
    st.group_ranges = (int32_t) g->parts[0]->ranges.size();
    for (size_t k = 1; k < g->parts.size(); k++) {
        const awpu_hip_stats &p = g->parts[k]->stats;
        st.launches += p.launches;
        st.last_kernel_ms = std::max(st.last_kernel_ms, p.last_kernel_ms);   // the slabs run side by side
        st.total_kernel_ms = std::max(st.total_kernel_ms, p.total_kernel_ms);
        st.alg_bytes_frame += p.alg_bytes_frame;
        st.alg_flops_frame += p.alg_flops_frame;
        st.tau_max = std::max(st.tau_max, p.tau_max);
        st.window = std::max(st.window, p.window);
    }
    *out = st;
    return AWPU_OK;
}

}  // namespace

extern "C" {

void awpu_hip_default_cfg(awpu_hip_cfg *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t) sizeof(*cfg);
    cfg->device = 0;
    cfg->n_streams = AWPU_ELEMENTS;
    cfg->hist = AWPU_HIST;
    cfg->n_pixels = 0;
    cfg->lut_stride = AWPU_ELEMENTS;
    cfg->interp = AWPU_INTERP_LERP;
    cfg->math = AWPU_MATH_F32_EXACT;  // the reference has one arithmetic (delay.cpp:19-25 inside mimo.cpp:121-137): that one; FAST is an opt-in
    cfg->max_batch = 1;
    cfg->pixel_begin = 0;
    cfg->pixel_count = 0;
}

int awpu_hip_create(awpu_hip_t **out, const awpu_hip_cfg *cfg) {
    if (!out || !cfg) return invalid("null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t) sizeof(awpu_hip_cfg)) return invalid("cfg.struct_size");
    if (cfg->n_streams < 1 || cfg->lut_stride < 1 || cfg->n_pixels < 1 || cfg->max_batch < 1)
        return invalid("n_streams, lut_stride, n_pixels and max_batch must be >= 1");
    if (cfg->hist < AWPU_N_SAMPLES + 1) return invalid("hist must hold at least 257 samples");
    if (cfg->max_batch > 65535) return invalid("max_batch above 65535");
    if (cfg->interp != AWPU_INTERP_LERP && cfg->interp != AWPU_INTERP_FIR8) return invalid("cfg.interp");
    if (cfg->math != AWPU_MATH_F32_EXACT && cfg->math != AWPU_MATH_F32_FAST && cfg->math != AWPU_MATH_BF16_ACC)
        return invalid("cfg.math");
    if (cfg->math == AWPU_MATH_BF16_ACC && cfg->interp != AWPU_INTERP_LERP)
        return invalid("the bf16 accumulator is built for the linear interpolation only");
    awpu_hip_cfg c = *cfg;
    if (c.pixel_count == 0) {
        c.pixel_begin = 0;
        c.pixel_count = c.n_pixels;
    }
    if (c.pixel_begin < 0 || c.pixel_count < 1 || c.pixel_begin + c.pixel_count > c.n_pixels)
        return invalid("pixel shard outside the grid");
    if (c.window_begin != 0 || c.window_end != 0) {
        const int reach = c.interp == AWPU_INTERP_FIR8 ? 263 : 257;
        if (c.window_begin < 0 || c.window_end > c.hist || c.window_end - c.window_begin < reach)
            return invalid("window_begin/window_end outside the history or narrower than one delay() read");
    }

    if (c.n_devices > 1) return create_group(out, c);
    c.n_devices = 1;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1 || c.device < 0 || c.device >= n_dev) {
        return fail(AWPU_ERR_NO_DEVICE, "no usable HIP device (this library has no CPU path)");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.device) != hipSuccess ||
        std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        return fail(AWPU_ERR_NO_DEVICE, "device is not gfx950 (MI355X); kernels are built for gfx950 only");
    }
    AWPU_HIP_TRY(hipSetDevice(c.device));

    awpu_hip *h = new (std::nothrow) awpu_hip();
    if (!h) return AWPU_ERR_NOMEM;
    h->cfg = c;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_end);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fan, hipEventDisableTiming);
    if (e != hipSuccess) {
        awpu_hip_destroy(h);
        return hip_fail(e, "stream/event creation");
    }
    *out = h;
    return AWPU_OK;
}

int awpu_hip_destroy(awpu_hip_t *h) {
    if (!h) return AWPU_OK;
    for (awpu_hip *part : h->parts) awpu_hip_destroy(part);
    h->parts.clear();
    (void) hipSetDevice(h->cfg.device);
    if (h->stream) (void) hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void) hipStreamSynchronize(h->copy_stream);
    release_device(h);
    for (hipEvent_t ev : {h->ev_begin, h->ev_end, h->ev_fan, h->ev_copied[0], h->ev_copied[1], h->ev_swept[0], h->ev_swept[1], h->ev_done,
                          h->ev_staged[0], h->ev_staged[1], h->ev_tile_free[0], h->ev_tile_free[1], h->ev_staged_read[0], h->ev_staged_read[1]})
        if (ev) (void) hipEventDestroy(ev);
    if (h->stream) (void) hipStreamDestroy(h->stream);
    if (h->copy_stream) (void) hipStreamDestroy(h->copy_stream);
    delete h;
    return AWPU_OK;
}

int awpu_hip_set_delay_table(awpu_hip_t *h, const int32_t *off, const float *frac) {
    AWPU_CTX(h);
    if (!h || !off || !frac) return invalid("null argument");
    if (!h->parts.empty()) {  // every device gets the rows of its pixel ranges, back to back
        h->union_window_done = false;
        return for_each_part(h, [&](awpu_hip *part) {
            part->cfg.window_begin = h->cfg.window_begin;  // (the union window of the OLD table is void: back to the caller's, if any)
            part->cfg.window_end = h->cfg.window_end;
            const size_t stride = (size_t) h->cfg.lut_stride;
            if (part->ranges.size() == 1) {
                const size_t first = (size_t) part->ranges[0].first * stride;
                return awpu_hip_set_delay_table(part, off + first, frac + first);
            }
            std::vector<int32_t> o((size_t) part->cfg.pixel_count * stride);
            std::vector<float> f(o.size());
            size_t done = 0;
            for (const auto &r : part->ranges) {
                std::memcpy(&o[done * stride], off + (size_t) r.first * stride, (size_t) r.second * stride * sizeof(int32_t));
                std::memcpy(&f[done * stride], frac + (size_t) r.first * stride, (size_t) r.second * stride * sizeof(float));
                done += (size_t) r.second;
            }
            return awpu_hip_set_delay_table(part, o.data(), f.data());
        });
    }
    const size_t n = (size_t) h->cfg.pixel_count * h->cfg.lut_stride;
    for (size_t i = 0; i < n; i++) {
        if (!(frac[i] >= 0.0f && frac[i] <= 1.0f)) return invalid("fraction outside [0, 1]");
    }
    h->off.assign(off, off + n);
    h->frac.assign(frac, frac + n);
    h->have_table = true;
    h->prepared = false;
    return AWPU_OK;
}

int awpu_hip_set_active_mics(awpu_hip_t *h, const int32_t *index, int32_t usable) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!h->parts.empty()) {
        h->union_window_done = false;
        return for_each_part(h, [&](awpu_hip *part) {
            part->cfg.window_begin = h->cfg.window_begin;  // (as in awpu_hip_set_delay_table: the union is taken anew)
            part->cfg.window_end = h->cfg.window_end;
            return awpu_hip_set_active_mics(part, index, usable);
        });
    }
    const int limit = std::min(h->cfg.n_streams, h->cfg.lut_stride);
    if (usable < 1 || usable > limit) return invalid("usable outside [1, min(n_streams, lut_stride)]");
    std::vector<int32_t> idx(usable);
    for (int s = 0; s < usable; s++) {
        idx[s] = index ? index[s] : s;
        if (idx[s] < 0 || idx[s] >= limit) return invalid("mic id outside the streams / table");
    }
    h->index.swap(idx);
    h->have_mics = true;
    h->prepared = false;
    return AWPU_OK;
}

int awpu_hip_set_mic_gains(awpu_hip_t *h, const float *gains) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!h->parts.empty()) return for_each_part(h, [&](awpu_hip *part) { return awpu_hip_set_mic_gains(part, gains); });
    if (!gains) {
        h->gain.clear();
    } else {
        for (int s = 0; s < h->cfg.n_streams; s++)
            if (!std::isfinite(gains[s])) return invalid("gain not finite");
        h->gain.assign(gains, gains + h->cfg.n_streams);
    }
    h->prepared = false;
    return AWPU_OK;
}

namespace {

// aw_processing_unit.cpp:145-200 on the 64 mean squares of one array
int usable_from_power(const float *power, float reference_power_level, int32_t *index, float *correction,
                      float *median_out) {
    float sorted[AWPU_ELEMENTS];
    std::copy(power, power + AWPU_ELEMENTS, sorted);
    std::sort(sorted, sorted + AWPU_ELEMENTS);
    // the reference averages elements 32 and 33 of the sorted list (.cpp:150), in double, then rounds
    const float median = (float) ((sorted[AWPU_ELEMENTS / 2] + sorted[AWPU_ELEMENTS / 2 + 1]) / 2.0);
    int count = 0;
    for (int s = 0; s < AWPU_ELEMENTS; s++) {
        const bool far_off = std::fabs(power[s] - median) > 1e-4;  // float promoted against a double bound
        const bool dead = power[s] < median * 1e-3;
        if (far_off || dead) continue;
        index[count] = s;
        correction[count] = reference_power_level / power[s];
        count++;
    }
    if (median_out) *median_out = median;
    return count;
}

int calibrate_rows(awpu_hip *h, const float *d_rows, int pitch, int hist, float reference_power_level, int32_t *index,
                   float *correction, float *median, int32_t *usable, hipStream_t s) {
    if (!h->d_calib) AWPU_HIP_TRY(hipMalloc(&h->d_calib, AWPU_ELEMENTS * sizeof(float)));
    AWPU_HIP_TRY(awpu::launch_stream_power(d_rows, pitch, hist, AWPU_ELEMENTS, h->d_calib, s));
    float power[AWPU_ELEMENTS];
    AWPU_HIP_TRY(hipMemcpyAsync(power, h->d_calib, sizeof(power), hipMemcpyDeviceToHost, s));
    AWPU_HIP_TRY(hipStreamSynchronize(s));
    *usable = usable_from_power(power, reference_power_level, index, correction, median);
    return AWPU_OK;
}

}  // namespace

int awpu_hip_calibrate_device(awpu_hip_t *h, const float *d_frame, int32_t array, float reference_power_level,
                              int32_t *index, float *correction, float *median, int32_t *usable, void *stream) {
    if (h && !h->parts.empty()) h = h->parts[0];  // not pixel-sharded: a device group answers with its first device
    AWPU_CTX(h);
    if (!h || !d_frame || !index || !correction || !usable) return invalid("null argument");
    if (array < 0 || (array + 1) * AWPU_ELEMENTS > h->cfg.n_streams) return invalid("array outside the streams");
    if (h->cfg.hist > 16384) return invalid("history too long for the calibration kernel");
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    return calibrate_rows(h, d_frame + (size_t) array * AWPU_ELEMENTS * h->cfg.hist, h->cfg.hist, h->cfg.hist,
                          reference_power_level, index, correction, median, usable, s);
}

int awpu_hip_calibrate_host(awpu_hip_t *h, const float *frame, int32_t array, float reference_power_level,
                             int32_t *index, float *correction, float *median, int32_t *usable) {
    const bool busy = h && h->in_flight;  // (the staging buffer below is the one an asynchronous call uploads into)
    if (h && !h->parts.empty()) h = h->parts[0];  // not pixel-sharded: a device group answers with its first device
    AWPU_CTX(h);
    if (!h || !frame || !index || !correction || !usable) return invalid("null argument");
    if (array < 0 || (array + 1) * AWPU_ELEMENTS > h->cfg.n_streams) return invalid("array outside the streams");
    if (h->cfg.hist > 16384) return invalid("history too long for the calibration kernel");
    if (busy) return fail(AWPU_ERR_STATE, "an awpu_hip_process_async call is in flight on this handle: awpu_hip_wait first");
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    // only the array's 64 streams travel; they share the frame staging buffer of awpu_hip_process
    const size_t need = (size_t) AWPU_ELEMENTS * h->cfg.hist;
    if (h->frames_cap < need) {
        dev_free(h->d_frames);
        h->frames_cap = 0;
        AWPU_HIP_TRY(hipMalloc(&h->d_frames, need * sizeof(float)));
        h->frames_cap = need;
    }
    AWPU_HIP_TRY(hipMemcpyAsync(h->d_frames, frame + (size_t) array * AWPU_ELEMENTS * h->cfg.hist, need * sizeof(float),
                                hipMemcpyHostToDevice, h->stream));
    return calibrate_rows(h, h->d_frames, h->cfg.hist, h->cfg.hist, reference_power_level, index, correction, median,
                          usable, h->stream);
}

int awpu_hip_calibrate_ring(awpu_hip_t *h, int32_t array, float reference_power_level, int32_t *index,
                            float *correction, float *median, int32_t *usable) {
    if (h && !h->parts.empty()) h = h->parts[0];  // not pixel-sharded: a device group answers with its first device
    AWPU_CTX(h);
    if (!h || !index || !correction || !usable) return invalid("null argument");
    if (array < 0 || (array + 1) * AWPU_ELEMENTS > h->cfg.n_streams) return invalid("array outside the streams");
    if (!h->d_ring) {
        return fail(AWPU_ERR_STATE, "no block ingested yet");
    }
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    return calibrate_rows(h, h->d_ring + (size_t) array * AWPU_ELEMENTS * 2048 + h->ring_pos, 2048, AWPU_HIST,
                          reference_power_level, index, correction, median, usable, h->stream);
}

int awpu_hip_beams(awpu_hip_t *h, const float *d_frame, const int32_t *off, const float *frac, int32_t n_dir,
                   float *power, float *beams) {
    if (h && !h->parts.empty()) h = h->parts[0];  // not pixel-sharded: a device group answers with its first device
    AWPU_CTX(h);
    if (!h || !off || !frac || (!power && !beams)) return invalid("null argument");
    if (n_dir < 1 || n_dir > 65535) return invalid("n_dir outside [1, 65535]");
    if (!h->have_mics) {
        return fail(AWPU_ERR_STATE, "active mics not set");
    }
    int pitch = h->cfg.hist;
    const float *frame = d_frame;
    if (!frame) {  // the current snapshot of the ingest ring
        if (!h->d_ring) {
            return fail(AWPU_ERR_STATE, "no block ingested yet");
        }
        frame = h->d_ring + h->ring_pos;
        pitch = 2048;
    }
    const int U = h->usable(), stride = h->cfg.lut_stride;
    std::vector<awpu::LutEntry> entries((size_t) n_dir * U);
    for (int d = 0; d < n_dir; d++) {
        for (int s = 0; s < U; s++) {
            const int id = h->index[s];
            const int o = off[(size_t) d * stride + id];
            const float f = frac[(size_t) d * stride + id];
            if (o < 0 || o + awpu::kSamples > h->cfg.hist - 1) {  // delay() reads X[off .. off+256]
                return fail(AWPU_ERR_RANGE, "delay table entry reads outside the frame history");
            }
            if (!(f >= 0.0f && f <= 1.0f)) return invalid("fraction outside [0, 1]");
            entries[(size_t) d * U + s] = awpu::LutEntry{id * pitch + o, f};
        }
    }
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->beam_lut_cap < entries.size()) {
        dev_free(h->d_beam_lut);
        h->beam_lut_cap = 0;
        AWPU_HIP_TRY(hipMalloc(&h->d_beam_lut, entries.size() * sizeof(awpu::LutEntry)));
        h->beam_lut_cap = entries.size();
    }
    if (h->beam_cap < (size_t) n_dir) {
        dev_free(h->d_beam_out);
        h->beam_cap = 0;
        AWPU_HIP_TRY(hipMalloc(&h->d_beam_out, (size_t) n_dir * (1 + awpu::kSamples) * sizeof(float)));
        h->beam_cap = n_dir;
    }
    float *d_power = h->d_beam_out, *d_beams = h->d_beam_out + h->beam_cap;
    AWPU_HIP_TRY(hipMemcpyAsync(h->d_beam_lut, entries.data(), entries.size() * sizeof(awpu::LutEntry),
                                hipMemcpyHostToDevice, h->stream));
    AWPU_HIP_TRY(awpu::launch_das_beams(frame, h->d_beam_lut, U, n_dir, d_power, beams ? d_beams : nullptr, h->stream));
    if (power)
        AWPU_HIP_TRY(hipMemcpyAsync(power, d_power, (size_t) n_dir * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (beams)
        AWPU_HIP_TRY(hipMemcpyAsync(beams, d_beams, (size_t) n_dir * awpu::kSamples * sizeof(float),
                                    hipMemcpyDeviceToHost, h->stream));
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));  // `entries` must outlive the upload
    return AWPU_OK;
}

int awpu_hip_set_fir_table(awpu_hip_t *h, const float *coeffs) {
    AWPU_CTX(h);
    if (!h || !coeffs) return invalid("null argument");
    if (!h->parts.empty()) return for_each_part(h, [&](awpu_hip *part) { return awpu_hip_set_fir_table(part, coeffs); });
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    // (on the device: the caller's 101 rows followed by zero rows up to kFir8CoeffRows -- the plane kernel's padding
    // entries name row 101, and its entry requests run a few items past a row's end, where any 7-bit row may stand)
    if (!h->d_fir) AWPU_HIP_TRY(hipMalloc(&h->d_fir, awpu::kFir8CoeffRows * 8 * sizeof(float)));
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));  // (a sweep still reading the old coefficients)
    AWPU_HIP_TRY(hipMemset(h->d_fir, 0, awpu::kFir8CoeffRows * 8 * sizeof(float)));
    AWPU_HIP_TRY(hipMemcpy(h->d_fir, coeffs, 101 * 8 * sizeof(float), hipMemcpyHostToDevice));
    h->fir.assign(coeffs, coeffs + 101 * 8);
    h->have_fir = true;  // (the plane table names coefficient rows, it does not carry them: nothing to rebuild)
    return AWPU_OK;
}

int awpu_hip_process(awpu_hip_t *h, const float *frames, int32_t batch, float *power) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!frames || !power) return invalid("null argument");
    if (h->in_flight) return fail(AWPU_ERR_STATE, "an awpu_hip_process_async call is in flight on this handle: awpu_hip_wait first");
    if (!h->parts.empty()) return group_process(h, frames, batch, power);
    if (batch == 1 && h->ranges.empty()) return live_host_call(h, frames, power);
    int rc = enqueue_host_process(h, frames, batch);
    if (rc != AWPU_OK) return rc;
    rc = enqueue_power_to_host(h, batch, power, (size_t) h->cfg.pixel_count);
    if (rc != AWPU_OK) return rc;
    return wait_and_time(h);
}

int awpu_hip_process_async(awpu_hip_t *h, const float *frames, int32_t batch, float *power) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!frames || !power) return invalid("null argument");
    if (h->in_flight) return fail(AWPU_ERR_STATE, "a call is in flight already: awpu_hip_wait first");
    int rc;
    if (!h->parts.empty()) {
        const size_t pitch = (size_t) h->cfg.pixel_count;
        rc = for_each_part(h, [&](awpu_hip *part) {
            AWPU_CTX(part);
            const int r = enqueue_host_process(part, frames, batch);
            return r != AWPU_OK ? r : enqueue_power_to_host(part, batch, power, pitch);
        });
        if (rc != AWPU_OK) {  // parts before the failing one hold copies from `frames` and into `power` in flight, and
            const std::string why = h->last_error;  // the caller is about to hear "failed": finish them before it does
            for (awpu_hip *part : h->parts)
                if (hipSetDevice(part->cfg.device) == hipSuccess) {
                    if (part->copy_stream) (void) hipStreamSynchronize(part->copy_stream);
                    (void) hipStreamSynchronize(part->stream);
                }
            (void) hipGetLastError();
            note_error(why);
        }
    } else {
        rc = enqueue_host_process(h, frames, batch);
        if (rc == AWPU_OK) rc = enqueue_power_to_host(h, batch, power, (size_t) h->cfg.pixel_count);
    }
    h->in_flight = rc == AWPU_OK;
    return rc;
}

int awpu_hip_wait(awpu_hip_t *h) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!h->in_flight) return AWPU_OK;  // nothing to wait for
    h->in_flight = false;
    if (!h->parts.empty()) return for_each_part(h, [&](awpu_hip *part) { return wait_and_time(part); });
    return wait_and_time(h);
}

int awpu_hip_process_device(awpu_hip_t *h, const float *d_frames, int32_t batch, float *d_power,
                            void *stream) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!d_frames || !d_power) return invalid("null argument");
    if (!h->parts.empty()) return group_process_device(h, d_frames, batch, d_power, static_cast<hipStream_t>(stream));
    const int rc = check_ready(h, batch);
    if (rc != AWPU_OK) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    TimingOff untimed(h);  // asynchronous path: the caller times its own stream
    return launch(h, d_frames, batch, d_power, s);
}

int awpu_hip_process_device_sums(awpu_hip_t *h, const float *d_frames, int32_t batch, float *d_power, float *d_sums, void *stream) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!d_frames || !d_power || !d_sums) return invalid("null argument");
    if (!h->parts.empty()) return invalid("the pre-epilogue sums are exported by single-device handles only");
    const int rc = check_ready(h, batch);
    if (rc != AWPU_OK) return rc;
    if (!h->exact_pairs_ok) return fail(AWPU_ERR_STATE, "the pre-epilogue sums need AWPU_MATH_F32_EXACT with AWPU_INTERP_LERP");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    TimingOff untimed(h);
    h->sums_out = d_sums;
    const int lrc = launch(h, d_frames, batch, d_power, s);
    h->sums_out = nullptr;
    return lrc;
}

namespace {

// H2D of one block of raw datagrams + the unpack launch, enqueued on the handle's stream (no wait)
int enqueue_ingest(awpu_hip *h, const void *datagrams, int32_t stride_bytes) {
    if (!h || !datagrams) return invalid("null argument");
    if (h->cfg.hist != AWPU_HIST || h->cfg.n_streams > 256) return invalid("ingest needs hist 1024 and <= 256 streams");
    if (stride_bytes < AWPU_DATAGRAM_BYTES) return invalid("datagram stride below 1032 bytes");
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    const size_t ring_bytes = (size_t) h->cfg.n_streams * 2048 * sizeof(float);
    if (!h->d_ring) {
        AWPU_HIP_TRY(hipMalloc(&h->d_ring, ring_bytes));
        AWPU_HIP_TRY(hipMemsetAsync(h->d_ring, 0, ring_bytes, h->stream));
        AWPU_HIP_TRY(hipMalloc(&h->d_datagrams, (size_t) awpu::kSamples * AWPU_DATAGRAM_BYTES));
        h->ring_pos = 0;
    }
    // tight copy of the 256 datagrams (the header travels too: 8 bytes each, ignored like the
    // reference ignores msg.counter, pipeline.cpp:264-267)
    AWPU_HIP_TRY(hipMemcpy2DAsync(h->d_datagrams, AWPU_DATAGRAM_BYTES, datagrams, (size_t) stride_bytes,
                                  AWPU_DATAGRAM_BYTES, awpu::kSamples, hipMemcpyHostToDevice, h->stream));
    AWPU_HIP_TRY(awpu::launch_unpack_block(h->d_datagrams, AWPU_DATAGRAM_BYTES, h->cfg.n_streams, h->d_ring,
                                           h->ring_pos, h->stream));
    h->ring_pos = (h->ring_pos + awpu::kSamples) % AWPU_HIST;  // Streams::forward
    return AWPU_OK;
}

}  // namespace

int awpu_hip_ingest_block(awpu_hip_t *h, const void *datagrams, int32_t stride_bytes) {
    AWPU_CTX(h);
    if (h && !h->parts.empty()) {  // every device keeps the whole ring (264 KB per block each, over its own PCIe link)
        int rc = for_each_part(h, [&](awpu_hip *part) {
            AWPU_CTX(part);
            return enqueue_ingest(part, datagrams, stride_bytes);
        });
        if (rc != AWPU_OK) return rc;
        return for_each_part(h, [&](awpu_hip *part) {
            AWPU_HIP_TRY(hipSetDevice(part->cfg.device));
            AWPU_HIP_TRY(hipStreamSynchronize(part->stream));
            return (int) AWPU_OK;
        });
    }
    const int rc = enqueue_ingest(h, datagrams, stride_bytes);
    if (rc != AWPU_OK) return rc;
    // the staging buffer is reused by the next call: finish the copy before returning
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));
    return AWPU_OK;
}

namespace {

// the steps of one live block on h->stream, without the final wait
int enqueue_live_block(awpu_hip *h, const void *datagrams, int32_t stride_bytes, float *power, int32_t rows, int32_t cols,
                       uint8_t *image, int32_t out_rows, int32_t out_cols, const uint8_t *d_colormap, uint8_t *big_image) {
    const int n = h->cfg.n_pixels;
    int rc = enqueue_ingest(h, datagrams, stride_bytes);
    if (rc != AWPU_OK) return rc;
    rc = ensure_power(h, (size_t) n);
    if (rc != AWPU_OK) return rc;
    rc = launch(h, h->d_ring + h->ring_pos, 1, h->d_power, h->stream, kRing);
    if (rc != AWPU_OK) return rc;
    if (power) AWPU_HIP_TRY(hipMemcpyAsync(power, h->d_power, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (image || big_image) {
        const size_t channels = d_colormap ? 3 : 1;
        const size_t need = sizeof(float) + (size_t) n + (big_image ? (size_t) out_rows * out_cols * channels : 0);
        if (h->display_cap < need) {
            retire_live_graphs(h);
            dev_free(h->d_display);
            h->display_cap = 0;
            AWPU_HIP_TRY(hipMalloc(&h->d_display, need));
            h->display_cap = need;
        }
        float *d_peak = reinterpret_cast<float *>(h->d_display);
        uint8_t *d_small = h->d_display + sizeof(float), *d_big = d_small + n;
        AWPU_HIP_TRY(awpu::launch_heatmap(h->d_power, n, 1, d_peak, false, d_small, h->stream));
        if (image) AWPU_HIP_TRY(hipMemcpyAsync(image, d_small, (size_t) n, hipMemcpyDeviceToHost, h->stream));
        if (big_image) {
            rc = awpu_hip_upscale_u8_device(h, d_small, rows, cols, 1, d_colormap, d_big, out_rows, out_cols, h->stream);
            if (rc != AWPU_OK) return rc;
            AWPU_HIP_TRY(hipMemcpyAsync(big_image, d_big, (size_t) out_rows * out_cols * channels, hipMemcpyDeviceToHost,
                                        h->stream));
        }
    }
    return AWPU_OK;
}

}  // namespace

int awpu_hip_live_block(awpu_hip_t *h, const void *datagrams, int32_t stride_bytes, float *power, int32_t rows,
                        int32_t cols, uint8_t *image, int32_t out_rows, int32_t out_cols, const uint8_t *d_colormap,
                        uint8_t *big_image) {
    AWPU_CTX(h);
    if (h && !h->parts.empty()) return invalid("the display step needs the whole grid on one device");
    if (h && h->in_flight) return fail(AWPU_ERR_STATE, "an awpu_hip_process_async call is in flight on this handle: awpu_hip_wait first");
    int rc = check_ready(h, 1);
    if (rc != AWPU_OK) return rc;
    const int n = h->cfg.n_pixels;
    if (h->cfg.pixel_count != n) return invalid("the display step needs the whole grid on this handle");
    if ((image || big_image) && (rows < 1 || cols < 1 || rows * cols != n)) return invalid("rows x cols must be the grid");
    if (big_image && (out_rows < rows || out_cols < cols || out_rows > 65535)) return invalid("upscale only: out >= in");
    if (!datagrams) return invalid("null argument");

    // A live block is eight small copies and launches: launch-latency bound.  Once every lazily allocated buffer
    // exists (after two plain calls) the sequence is captured into a HIP graph -- one per ring position and set of
    // caller buffers, a display loop reuses its own -- and replayed with a single launch.
    const bool graphs = env().live_graph != 0 && !h->live_graph_broken && h->d_ring != nullptr;
    // (a call of another shape allocates: upscale taps, a larger display buffer -- not while a capture is open)
    const unsigned long long shape = ((unsigned long long) (unsigned) rows << 48) ^ ((unsigned long long) (unsigned) cols << 36) ^
                                     ((unsigned long long) (unsigned) out_rows << 20) ^ ((unsigned long long) (unsigned) out_cols << 4) ^
                                     (power ? 1u : 0u) ^ (image ? 2u : 0u) ^ (big_image ? 4u : 0u) ^ (d_colormap ? 8u : 0u);
    // Only a caller that comes round with the SAME buffers gains from a graph: a call with other buffers than the
    // one before it (fresh arrays every block) starts the count again and is never captured.
    const void *bufs[5] = {datagrams, power, image, d_colormap, big_image};
    if (shape != h->live_shape || std::memcmp(bufs, h->live_bufs, sizeof(bufs)) != 0) {
        h->live_shape = shape;
        std::memcpy(h->live_bufs, bufs, sizeof(bufs));
        h->live_warm = 0;
    }
    if (graphs && h->live_warm >= 2) {
        awpu_hip::LiveGraph key{h->ring_pos, stride_bytes, rows, cols, out_rows, out_cols, datagrams, power, image, d_colormap,
                                big_image, h->table_gen, nullptr, 0};
        for (auto &g : h->live_graphs)
            if (g.ring_pos == key.ring_pos && g.stride == key.stride && g.rows == key.rows && g.cols == key.cols &&
                g.out_rows == key.out_rows && g.out_cols == key.out_cols && g.datagrams == key.datagrams && g.power == key.power &&
                g.image == key.image && g.colormap == key.colormap && g.big_image == key.big_image && g.gen == key.gen) {
                g.last_use = ++h->live_clock;
                AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
                AWPU_HIP_TRY(hipGraphLaunch(g.exec, h->stream));
                h->ring_pos = (h->ring_pos + awpu::kSamples) % AWPU_HIST;  // (what enqueue_ingest does on the plain path)
                h->stats.launches += 1;
                h->stats.frames += 1;
                AWPU_HIP_TRY(hipStreamSynchronize(h->stream));
                return AWPU_OK;
            }
        if (h->live_graphs.size() >= 64) {  // full: the least recently replayed graph makes room
            auto lru = std::min_element(h->live_graphs.begin(), h->live_graphs.end(),
                                        [](const awpu_hip::LiveGraph &a, const awpu_hip::LiveGraph &b) { return a.last_use < b.last_use; });
            (void) hipGraphExecDestroy(lru->exec);
            h->live_graphs.erase(lru);
        }
        {   // capture this variant (the stream is idle: every call ends with a wait)
            AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
            const bool keep_timing = h->timing;
            const int keep_pos = h->ring_pos;
            const auto keep_stats = h->stats;
            h->timing = false;  // (event records inside a graph would not bracket anything)
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                rc = enqueue_live_block(h, datagrams, stride_bytes, power, rows, cols, image, out_rows, out_cols, d_colormap, big_image);
                e = hipStreamEndCapture(h->stream, &graph);  // (also on failure: it takes the stream out of capture mode)
            } else {
                rc = AWPU_ERR_HIP;
            }
            h->timing = keep_timing;
            h->ring_pos = keep_pos;  // nothing ran yet
            h->stats = keep_stats;
            hipGraphExec_t exec = nullptr;
            if (rc == AWPU_OK && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                key.exec = exec;
                key.last_use = ++h->live_clock;
                h->live_graphs.push_back(key);
            } else {
                hipStreamCaptureStatus status = hipStreamCaptureStatusNone;  // (an invalidated capture must not outlive this call)
                if (hipStreamIsCapturing(h->stream, &status) == hipSuccess && status != hipStreamCaptureStatusNone) {
                    hipGraph_t dead = nullptr;
                    (void) hipStreamEndCapture(h->stream, &dead);
                    if (dead) (void) hipGraphDestroy(dead);
                }
                (void) hipGetLastError();
                h->live_graph_broken = true;  // this runtime does not capture the sequence: stay on the plain path
            }
            if (graph) (void) hipGraphDestroy(graph);
            if (!h->live_graph_broken) return awpu_hip_live_block(h, datagrams, stride_bytes, power, rows, cols, image, out_rows, out_cols, d_colormap, big_image);
        }
    }
    rc = enqueue_live_block(h, datagrams, stride_bytes, power, rows, cols, image, out_rows, out_cols, d_colormap, big_image);
    if (rc != AWPU_OK) return rc;
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));
    h->live_warm++;
    return AWPU_OK;
}

int awpu_hip_process_ring(awpu_hip_t *h, float *power) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!power) return invalid("null argument");
    if (!h->parts.empty()) {  // every device sweeps its slab of its own ring's snapshot
        int grc = for_each_part(h, [&](awpu_hip *part) {
            AWPU_CTX(part);
            int r = check_ready(part, 1);
            if (r != AWPU_OK) return r;
            if (!part->d_ring) return fail(AWPU_ERR_STATE, "no block ingested yet");
            r = ensure_power(part, (size_t) part->cfg.pixel_count);
            if (r == AWPU_OK) r = launch(part, part->d_ring + part->ring_pos, 1, part->d_power, part->stream, kRing);
            return r != AWPU_OK ? r : enqueue_power_to_host(part, 1, power, (size_t) h->cfg.pixel_count);
        });
        if (grc != AWPU_OK) return grc;
        return for_each_part(h, [&](awpu_hip *part) { return wait_and_time(part); });
    }
    int rc = check_ready(h, 1);
    if (rc != AWPU_OK) return rc;
    if (!h->d_ring) {
        return fail(AWPU_ERR_STATE, "no block ingested yet");
    }
    const size_t need_power = (size_t) h->cfg.pixel_count;
    rc = ensure_power(h, need_power);
    if (rc != AWPU_OK) return rc;
    rc = launch(h, h->d_ring + h->ring_pos, 1, h->d_power, h->stream, kRing);
    if (rc != AWPU_OK) return rc;
    AWPU_HIP_TRY(hipMemcpyAsync(power, h->d_power, need_power * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));
    return AWPU_OK;
}

int awpu_hip_ring_snapshot(awpu_hip_t *h, float *frames) {
    if (h && !h->parts.empty()) h = h->parts[0];  // not pixel-sharded: a device group answers with its first device
    AWPU_CTX(h);
    if (!h || !frames) return invalid("null argument");
    if (!h->d_ring) {
        return fail(AWPU_ERR_STATE, "no block ingested yet");
    }
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    AWPU_HIP_TRY(hipMemcpy2DAsync(frames, AWPU_HIST * sizeof(float), h->d_ring + h->ring_pos, 2048 * sizeof(float),
                                  AWPU_HIST * sizeof(float), h->cfg.n_streams, hipMemcpyDeviceToHost, h->stream));
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));
    return AWPU_OK;
}

int awpu_hip_heatmap_u8_device(awpu_hip_t *h, const float *d_power, int32_t n, int32_t batch, float *d_peak,
                               int32_t peak_given, uint8_t *d_pix, void *stream) {
    if (h && !h->parts.empty()) h = h->parts[0];  // not pixel-sharded: a device group answers with its first device
    AWPU_CTX(h);
    if (!h || !d_power || !d_peak || !d_pix || n < 1 || batch < 1 || batch > 65535) return invalid("bad argument");
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    AWPU_HIP_TRY(awpu::launch_heatmap(d_power, n, batch, d_peak, peak_given != 0, d_pix, s));
    return AWPU_OK;
}

int awpu_hip_upscale_u8_device(awpu_hip_t *h, const uint8_t *d_pix, int32_t rows, int32_t cols, int32_t batch,
                               const uint8_t *d_colormap, uint8_t *d_out, int32_t out_rows, int32_t out_cols,
                               void *stream) {
    if (h && !h->parts.empty()) h = h->parts[0];  // not pixel-sharded: a device group answers with its first device
    AWPU_CTX(h);
    if (!h || !d_pix || !d_out || rows < 1 || cols < 1 || batch < 1 || batch > 65535) return invalid("bad argument");
    if (out_rows < rows || out_cols < cols || out_rows > 65535) return invalid("upscale only: out >= in, out_rows <= 65535");
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    const int key[4] = {rows, cols, out_rows, out_cols};
    if (!h->d_taps || std::memcmp(key, h->taps_key, sizeof(key)) != 0) {
        std::vector<awpu::ResizeTap> taps((size_t) out_cols + out_rows);
        awpu::resize_taps(cols, out_cols, true, taps.data());
        awpu::resize_taps(rows, out_rows, false, taps.data() + out_cols);
        AWPU_HIP_TRY(hipStreamSynchronize(s));  // an earlier launch may still read the old taps
        retire_live_graphs(h);
        dev_free(h->d_taps);
        AWPU_HIP_TRY(hipMalloc(&h->d_taps, taps.size() * sizeof(awpu::ResizeTap)));
        AWPU_HIP_TRY(hipMemcpy(h->d_taps, taps.data(), taps.size() * sizeof(awpu::ResizeTap), hipMemcpyHostToDevice));
        std::memcpy(h->taps_key, key, sizeof(key));
    }
    AWPU_HIP_TRY(awpu::launch_upscale(d_pix, rows, cols, batch, h->d_taps, d_colormap, d_out, out_rows, out_cols, s));
    return AWPU_OK;
}

int awpu_hip_resize_linear_u8(const uint8_t *pix, int32_t rows, int32_t cols, uint8_t *out, int32_t out_rows,
                              int32_t out_cols) {
    if (!pix || !out || rows < 1 || cols < 1) return invalid("bad argument");
    if (out_rows < rows || out_cols < cols) return invalid("upscale only: out >= in");
    std::vector<awpu::ResizeTap> taps((size_t) out_cols + out_rows);
    awpu::resize_taps(cols, out_cols, true, taps.data());
    awpu::resize_taps(rows, out_rows, false, taps.data() + out_cols);
    std::vector<int> sums((size_t) 2 * out_cols);  // the two source rows of the current output row, widened
    int have0 = -1, have1 = -1;
    for (int dy = 0; dy < out_rows; dy++) {
        const awpu::ResizeTap ty = taps[(size_t) out_cols + dy];
        const int r0 = std::min(std::max(ty.src, 0), rows - 1), r1 = std::min(std::max(ty.src + 1, 0), rows - 1);
        const int want[2] = {r0, r1};
        int *have[2] = {&have0, &have1};
        for (int k = 0; k < 2; k++) {
            if (*have[k] == want[k]) continue;
            const uint8_t *row = pix + (size_t) want[k] * cols;
            int *line = sums.data() + (size_t) k * out_cols;
            for (int dx = 0; dx < out_cols; dx++) {
                const awpu::ResizeTap tx = taps[dx];
                line[dx] = row[tx.src] * tx.w0 + row[std::min(tx.src + 1, cols - 1)] * tx.w1;
            }
            *have[k] = want[k];
        }
        for (int dx = 0; dx < out_cols; dx++)
            out[(size_t) dy * out_cols + dx] = awpu::resize_combine(sums[dx], sums[(size_t) out_cols + dx], ty.w0, ty.w1);
    }
    return AWPU_OK;
}

namespace {

// The layout both frame-pair shapes read when usable is a multiple of four and no gains are set:
// [ceil(batch/2)][usable][wr][2] floats.  AWPU_ERR_STATE when this handle's sweep does not take packed frames.
int packed_plan(awpu_hip *h, int batch, awpu::FastPlan *plan) {
    if (!h) return invalid("null handle");
    if (!h->parts.empty()) return fail(AWPU_ERR_STATE, "packed frames: a device group exchanges its frames itself");
    const int rc = check_ready(h, batch);
    if (rc != AWPU_OK) return rc;
    if ((h->cfg.math != AWPU_MATH_F32_FAST && h->cfg.math != AWPU_MATH_F32_EXACT) || h->cfg.interp != AWPU_INTERP_LERP)
        return fail(AWPU_ERR_STATE, "packed frames need AWPU_MATH_F32_EXACT or AWPU_MATH_F32_FAST, and AWPU_INTERP_LERP");
    if (h->usable() % 4 != 0 || !h->gain.empty())
        return fail(AWPU_ERR_STATE, "packed frames need usable % 4 == 0 and no mic gains (the shapes of a mode then read one layout)");
    if (h->cfg.math == AWPU_MATH_F32_EXACT) {  // the {next, d} rows of das_exact_nd_kernel: where launch() takes that kernel for this batch
        int nq = 1;
        if (batch < 2 || !takes_exact_nd(h, batch, &nq))
            return fail(AWPU_ERR_STATE, "packed frames in the reference's order need the grid's row length (grid_columns) and a batch of two or more");
        *plan = h->exact_nd_plan;
        return AWPU_OK;
    }
    if (!awpu::pair_plan(h->window, h->usable(), plan)) return fail(AWPU_ERR_STATE, "the window does not fit the frame-pair image");
    return AWPU_OK;
}

}  // namespace

int awpu_hip_packed_bytes(awpu_hip_t *h, int32_t batch, uint64_t *bytes) {
    AWPU_CTX(h);
    if (!bytes) return invalid("null argument");
    awpu::FastPlan plan;
    const int rc = packed_plan(h, batch, &plan);
    if (rc != AWPU_OK) return rc;
    *bytes = (uint64_t) packed_floats_of(h, plan, batch) * sizeof(float);
    return AWPU_OK;
}

int awpu_hip_pack_frames(awpu_hip_t *h, const float *d_frames, int32_t batch, float *d_packed, void *stream) {
    AWPU_CTX(h);
    if (!d_frames || !d_packed) return invalid("null argument");
    awpu::FastPlan plan;
    const int rc = packed_plan(h, batch, &plan);
    if (rc != AWPU_OK) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    // (Measured: a throttled variant of this pass -- few persistent workgroups, non-temporal accesses -- meant to be gentler
    // on the sweep it runs beside, slowed that sweep MORE the longer it lasted: 256 / 512 / 1024 workgroups cost the ingest
    // rank 0.90 / 0.55 / 0.35 ms per 1024-frame step against 0.23 ms for this full-speed pass.  Short and fast wins.)
    return pack_for_sweep(h, plan, d_frames, batch, d_packed, s);
}

int awpu_hip_process_packed(awpu_hip_t *h, const float *d_packed, int32_t batch, float *d_power, void *stream) {
    AWPU_CTX(h);
    if (!d_packed || !d_power) return invalid("null argument");
    awpu::FastPlan plan;
    int rc = packed_plan(h, batch, &plan);
    if (rc != AWPU_OK) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    TimingOff untimed(h);  // asynchronous path: the caller times its own stream
    // the shape awpu_hip_process_device takes for this batch (same rule: same bits), as long as that is a frame-pair shape
    // (the buffer is the caller's: awpu_hip_packed_bytes(batch) of it are taken to be there, and the sweep reads no further)
    return sweep_packed(h, plan, d_packed, packed_floats_of(h, plan, batch), batch, d_power, s);
}

int awpu_hip_synchronize(awpu_hip_t *h) {
    AWPU_CTX(h);
    if (!h) return invalid("null handle");
    if (!h->parts.empty())
        return for_each_part(h, [&](awpu_hip *part) {
            AWPU_HIP_TRY(hipSetDevice(part->cfg.device));
            AWPU_HIP_TRY(hipStreamSynchronize(part->copy_stream));
            AWPU_HIP_TRY(hipStreamSynchronize(part->stream));
            return (int) AWPU_OK;
        });
    AWPU_HIP_TRY(hipSetDevice(h->cfg.device));
    AWPU_HIP_TRY(hipStreamSynchronize(h->stream));
    return AWPU_OK;
}

int awpu_hip_group_peer_status(awpu_hip_t *h, int32_t *status, int32_t n) {
    AWPU_CTX(h);
    if (!h || !status || n < 1) return invalid("null argument");
    const int have = h->parts.empty() ? 1 : (int) h->parts.size();
    if (n < have) return invalid("status array shorter than the device group");
    if (h->parts.empty()) {
        status[0] = AWPU_PEER_SAME_DEVICE;
    } else {
        for (int k = 0; k < have; k++)
            status[k] = h->parts[k]->peer == kPeerSame     ? AWPU_PEER_SAME_DEVICE
                        : h->parts[k]->peer == kPeerDirect ? AWPU_PEER_DIRECT
                                                           : AWPU_PEER_HOST_STAGED;
    }
    return have;
}

int awpu_hip_build_delay_table_device(int32_t device, const float *xyz, int32_t n, int32_t rows, int32_t columns, float fov_deg,
                                      int32_t row_begin, int32_t row_count, int32_t *off, float *frac) {
    if (!xyz || !off || !frac || n <= 0 || rows <= 0 || columns <= 0) return invalid("null or non-positive argument");
    if (row_begin < 0 || row_count < 0 || row_begin + row_count > rows) return invalid("rows outside the grid");
    if (row_count == 0) return AWPU_OK;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return fail(AWPU_ERR_NO_DEVICE, "no such HIP device (the host builder awpu_hip_build_delay_table needs none)");
    hipDeviceProp_t prop;
    AWPU_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(AWPU_ERR_NO_DEVICE, "device is not gfx950 (MI355X); kernels are built for gfx950 only");
    AWPU_HIP_TRY(hipSetDevice(device));
    const size_t P = (size_t) row_count * columns;
    std::vector<float> rot(P * 12);
    awpu::pixel_rotations(rows, columns, fov_deg, row_begin, row_count, rot.data());
    float *d_xyz = nullptr, *d_rot = nullptr, *d_frac = nullptr;
    int32_t *d_off = nullptr;
    auto body = [&]() -> int {
        AWPU_HIP_TRY(hipMalloc(&d_xyz, (size_t) 3 * n * sizeof(float)));
        AWPU_HIP_TRY(hipMalloc(&d_rot, rot.size() * sizeof(float)));
        AWPU_HIP_TRY(hipMalloc(&d_off, P * n * sizeof(int32_t)));
        AWPU_HIP_TRY(hipMalloc(&d_frac, P * n * sizeof(float)));
        AWPU_HIP_TRY(hipMemcpy(d_xyz, xyz, (size_t) 3 * n * sizeof(float), hipMemcpyHostToDevice));
        AWPU_HIP_TRY(hipMemcpy(d_rot, rot.data(), rot.size() * sizeof(float), hipMemcpyHostToDevice));
        // one launch per 32 768 pixels (the grid's x dimension is not the limit; this bounds a launch's run time)
        for (size_t p0 = 0; p0 < P; p0 += 32768) {
            const int np = (int) std::min<size_t>(32768, P - p0);
            AWPU_HIP_TRY(awpu::launch_delay_table(d_xyz, n, d_rot + p0 * 12, np, awpu::samples_per_metre(), d_off + p0 * n, d_frac + p0 * n,
                                                  nullptr));
        }
        AWPU_HIP_TRY(hipMemcpy(off, d_off, P * n * sizeof(int32_t), hipMemcpyDeviceToHost));
        AWPU_HIP_TRY(hipMemcpy(frac, d_frac, P * n * sizeof(float), hipMemcpyDeviceToHost));
        return AWPU_OK;
    };
    const int rc = body();
    dev_free(d_xyz);
    dev_free(d_rot);
    dev_free(d_off);
    dev_free(d_frac);
    return rc;
}

int awpu_hip_get_stats(awpu_hip_t *h, awpu_hip_stats *stats) {
    AWPU_CTX(h);
    if (!h || !stats) return invalid("null argument");
    if (!h->parts.empty()) return group_stats(h, stats);
    *stats = h->stats;
    return AWPU_OK;
}

const char *awpu_hip_strerror(int status) {
    switch (status) {
        case AWPU_OK: return "ok";
        case AWPU_ERR_INVALID: return "invalid argument or configuration";
        case AWPU_ERR_NO_DEVICE: return "no gfx950 HIP device (no CPU fallback exists)";
        case AWPU_ERR_HIP: return "HIP runtime error";
        case AWPU_ERR_STATE: return "delay table / active mics not set";
        case AWPU_ERR_RANGE: return "delay table reads outside the frame history";
        case AWPU_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char *awpu_hip_last_error(void) { return g_last_error.c_str(); }

const char *awpu_hip_last_error_of(awpu_hip_t *h) { return h ? h->last_error.c_str() : ""; }

int awpu_hip_abi_version(void) { return AWPU_HIP_ABI_VERSION; }

}  // extern "C"

#ifdef AWPU_TIMING_BUILD
// marker of a build whose AWPU_FAST_DEBUG timing switches are live (wrong results on request): never shipped,
// tests/test_abi.py::test_shipping_build_has_no_wrong_result_switches looks for it
extern "C" int awpu_hip_timing_build(void) { return 1; }
#endif
