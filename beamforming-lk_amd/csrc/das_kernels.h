// das_kernels.h -- launch interface between the C-ABI layer (awpu_hip.cpp) and the gfx950
// sweep kernels (das_kernels.hip).  Internal to libawpu_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace awpu {

constexpr int kSamples = 256;  // N_SAMPLES, src/fpga/streams.hpp:28

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
    int32_t n_streams;
    int32_t hist;
    int32_t usable;
    int32_t pixel_count;
    int32_t wstart;  // first history sample any entry touches (= min off)
    int32_t window;  // W = 256 + tau_max + 1 samples staged per mic
    int32_t batch;
};

// exact-order kernel (AWPU_MATH_F32_EXACT): sub, fma, add per sample, mics in order.
hipError_t launch_das_exact(const SweepArgs &a, hipStream_t stream);

// LDS bytes the exact kernel asks for with the given window; 0 if the window cannot fit.
size_t das_exact_lds_bytes(int window, int usable, int *chunk_out);

}  // namespace awpu
