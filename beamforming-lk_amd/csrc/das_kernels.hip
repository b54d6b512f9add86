// das_kernels.hip -- gfx950 (MI355X, CDNA4) delay-and-sum sweep kernels.
//
// The path replaced is MIMOWorker::update, src/dsp/mimo.cpp:121-151, whose inner kernel is
// delay(), src/dsp/delay.cpp:16-26 (reference tree acoustic-warfare/beamforming-lk):
//
//   for pixel p:  out[0..255] = 0
//     for active mic s:  out[i] += X_s[off+i+1] + frac * (X_s[off+i] - X_s[off+i+1])
//     power[p] = sum_{i=1..254} (0.5 out[i] - 0.25 (out[i+1] + out[i-1]))^2 / (256 * usable)
//
// Written for wave64 / 160 KiB LDS / gfx950 only.  No MFMA: this is a gather-and-reduce.
#include "das_kernels.h"

namespace awpu {

// ---------------------------------------------------------------------------------------
// Epilogue shared by the kernels whose lanes own samples {l, l+64, l+128, l+192}.
// MA[i] = 0.5 out[i] - 0.25 (out[i+1] + out[i-1]) for i in [1, 254]  (mimo.cpp:132-135);
// returns the wave-wide sum of MA^2 in every lane.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float epilogue_interleaved(const float (&o)[4], int lane) {
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float prev = __shfl_up(o[k], 1);    // lane l-1, same k
        float next = __shfl_down(o[k], 1);  // lane l+1, same k
        // sample l+64k-1 of lane 0 lives in lane 63 of register k-1; likewise the far end
        const float wrap_prev = __shfl(o[k > 0 ? k - 1 : 0], 63);
        const float wrap_next = __shfl(o[k < 3 ? k + 1 : 3], 0);
        if (lane == 0) prev = wrap_prev;
        if (lane == 63) next = wrap_next;
        const int i = lane + 64 * k;
        const float ma = o[k] * 0.5f - 0.25f * (next + prev);
        if (i >= 1 && i <= kSamples - 2) sum += ma * ma;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s);
    return sum;
}

// ---------------------------------------------------------------------------------------
// Exact-order kernel (AWPU_MATH_F32_EXACT).
//
// One workgroup = 4 waves; wave w sweeps PPW consecutive pixels.  The touched window of a
// chunk of mics ([chunk][W] fp32) is staged once in LDS; a table entry is wave-uniform, so
// (off, frac) travel in SGPRs; lane l owns samples l, l+64, l+128, l+192, so every LDS
// read of a wave is 64 consecutive dwords (conflict-free).  Per sample the operations are
// those of delay.cpp:19-25 in the same order -- d = cur - next; t = fma(frac, d, next);
// out += t -- and mics are visited in the reference's order s = 0..usable-1, so the
// pre-epilogue sums are bit-identical to the reference kernel.
// ---------------------------------------------------------------------------------------
constexpr int kExactThreads = 256;
constexpr int kExactPPW = 4;
constexpr size_t kExactLdsBudget = 64 * 1024;

// AWPU_MATH_BF16_ACC: the running sum of a sample is KEPT in bf16 -- rounded to nearest even after every
// mic's term -- while the term itself is computed in fp32 as above.  gfx950 has no packed bf16 add: a bf16
// accumulate is an fp32 add plus v_cvt_pk_bf16_f32 (and the unpack shift), i.e. strictly more VALU work than
// the fp32 accumulator it replaces.  Built so that BASELINE configs[4]'s "bf16 vs fp32 accumulator" is a
// measurement (error and rate, bench.py "bf16"), not an argument.
__device__ __forceinline__ float round_to_bf16(float x) { return (float) (__bf16) x; }

template <int PPW, bool BF16ACC>
__global__ __launch_bounds__(kExactThreads) void das_exact_kernel(SweepArgs a, int chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int W = a.window;
    const int pix0 = (blockIdx.x * (kExactThreads / 64) + wave) * PPW;
    const float *frame = a.frames + (size_t) b * a.n_streams * a.hist;

    float acc[PPW][4];
#pragma unroll
    for (int pp = 0; pp < PPW; pp++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[pp][k] = 0.0f;

    for (int m0 = 0; m0 < a.usable; m0 += chunk) {
        const int mc = min(chunk, a.usable - m0);
        __syncthreads();  // previous chunk fully consumed
        for (int m = wave; m < mc; m += kExactThreads / 64) {
            const float *src = frame + (size_t) a.index[m0 + m] * a.hist + a.wstart;
            if (a.gain) {  // optional per-mic gain (awpu_hip_set_mic_gains); absent in the reference
                const float gm = a.gain[m0 + m];
                for (int t = lane; t < W; t += 64) lds[m * W + t] = src[t] * gm;
            } else
            for (int t = lane; t < W; t += 64) lds[m * W + t] = src[t];
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < PPW; pp++) {
            const int p = pix0 + pp;
            if (p < a.pixel_count) {
                const LutEntry *row = a.lut + (size_t) p * a.usable + m0;
                for (int m = 0; m < mc; m++) {
                    const LutEntry e = row[m];  // wave-uniform address: scalar load
                    const float f = e.frac;
                    const float *x = lds + m * W + e.off_rel + lane;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float cur = x[64 * k];
                        const float nxt = x[64 * k + 1];
                        const float d = cur - nxt;
                        const float t = __builtin_fmaf(f, d, nxt);
                        acc[pp][k] = BF16ACC ? round_to_bf16(acc[pp][k] + t) : acc[pp][k] + t;
                    }
                }
            }
        }
    }

#pragma unroll
    for (int pp = 0; pp < PPW; pp++) {
        const int p = pix0 + pp;
        if (p < a.pixel_count) {
            const float sum = epilogue_interleaved(acc[pp], lane);
            if (lane == 0) {
                a.power[(size_t) b * a.pixel_count + p] = sum / (float) (kSamples * a.usable);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// FIR8 kernel (AWPU_INTERP_FIR8): the 8-tap table variant of delay(), src/dsp/delay.cpp:31-40,
//     k = (int)(frac * 100 + 0.5);  out[n] += sum_{t<8} C[k][t] * X[off + n + t]
// inside the same sweep and epilogue (mimo.cpp:121-151).  Not compiled in the reference's shipped
// configuration (-mavx2 selects the linear variant); provided for completeness, same structure as
// the exact kernel: lane l owns samples l+64k, taps accumulate in the reference's order
// t = 0..7 (one FMA per tap where the reference has a multiply and an add).  The table entry
// carries k (computed on the host with the reference's expression); the 8 coefficients of a
// (pixel, mic) are wave-uniform and arrive by one scalar load.
// ---------------------------------------------------------------------------------------
template <int PPW>
__global__ __launch_bounds__(kExactThreads) void das_fir8_kernel(SweepArgs a, const float *coeffs, int chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int W = a.window;
    const int pix0 = (blockIdx.x * (kExactThreads / 64) + wave) * PPW;
    const float *frame = a.frames + (size_t) b * a.n_streams * a.hist;

    float acc[PPW][4];
#pragma unroll
    for (int pp = 0; pp < PPW; pp++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[pp][k] = 0.0f;

    for (int m0 = 0; m0 < a.usable; m0 += chunk) {
        const int mc = min(chunk, a.usable - m0);
        __syncthreads();
        for (int m = wave; m < mc; m += kExactThreads / 64) {
            const float *src = frame + (size_t) a.index[m0 + m] * a.hist + a.wstart;
            if (a.gain) {  // optional per-mic gain (awpu_hip_set_mic_gains); absent in the reference
                const float gm = a.gain[m0 + m];
                for (int t = lane; t < W; t += 64) lds[m * W + t] = src[t] * gm;
            } else
            for (int t = lane; t < W; t += 64) lds[m * W + t] = src[t];
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < PPW; pp++) {
            const int p = pix0 + pp;
            if (p < a.pixel_count) {
                const LutEntry *row = a.lut + (size_t) p * a.usable + m0;
                for (int m = 0; m < mc; m++) {
                    const LutEntry e = row[m];
                    const float *c = coeffs + 8 * __float_as_int(e.frac);  // .frac carries the row index k
                    const float *x = lds + m * W + e.off_rel + lane;
#pragma unroll
                    for (int k = 0; k < 4; k++)
#pragma unroll
                        for (int t = 0; t < 8; t++) acc[pp][k] = __builtin_fmaf(c[t], x[64 * k + t], acc[pp][k]);
                }
            }
        }
    }
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) {
        const int p = pix0 + pp;
        if (p < a.pixel_count) {
            const float sum = epilogue_interleaved(acc[pp], lane);
            if (lane == 0) a.power[(size_t) b * a.pixel_count + p] = sum / (float) (kSamples * a.usable);
        }
    }
}

hipError_t launch_das_fir8(const SweepArgs &a, const float *d_coeffs, hipStream_t stream) {
    int chunk = 0;
    const size_t lds = das_exact_lds_bytes(a.window, a.usable, &chunk);
    if (lds == 0) return hipErrorInvalidValue;
    const int pix_per_block = (kExactThreads / 64) * kExactPPW;
    dim3 grid((a.pixel_count + pix_per_block - 1) / pix_per_block, a.batch);
    hipLaunchKernelGGL(das_fir8_kernel<kExactPPW>, grid, dim3(kExactThreads), lds, stream, a, d_coeffs, chunk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Display step on the device: MIMOWorker::populateHeatmap with USE_DB 0, src/dsp/mimo.cpp:61-95.
//   max_v = max(0, max_p power[p]);  pix[p] = (uint8) clip(power[p] / max_v * 255, 0, 255)
// Two launches per call: a grid-stride maximum into one float per frame (powers are >= 0, so
// the unsigned bit pattern orders like the value and atomicMax on it is exact), then the scaling.
// `peak` may be supplied by the caller instead (multi-GPU: the all-reduced maximum of the tiles).
// ---------------------------------------------------------------------------------------
__global__ void heatmap_max_kernel(const float *power, int n, unsigned *peak_bits) {
    const float *p = power + (size_t) blockIdx.y * n;
    float m = 0.0f;  // mimo.cpp:62: the running maximum starts at 0
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, p[i]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) atomicMax(peak_bits + blockIdx.y, __float_as_uint(m));
}

__global__ void heatmap_scale_kernel(const float *power, int n, const float *peak, uint8_t *pix) {
    const float max_v = peak[blockIdx.y];
    const float *p = power + (size_t) blockIdx.y * n;
    uint8_t *o = pix + (size_t) blockIdx.y * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        // mimo.cpp:85-91: float division, then double scaling and clipping, then the uchar cast
        double level = (double) (p[i] / max_v) * 255.0;
        // all-zero frame: 0/0 = NaN, defined as level 0 (awpu_hip_heatmap_u8 on the host does the same)
        level = !(level >= 0.0) ? 0.0 : (level > 255.0 ? 255.0 : level);
        o[i] = (uint8_t) level;
    }
}

hipError_t launch_heatmap(const float *d_power, int n, int batch, float *d_peak, bool peak_given, uint8_t *d_pix,
                          hipStream_t stream) {
    const int blocks = (n + 1023) / 1024 < 64 ? (n + 1023) / 1024 : 64;
    if (!peak_given) {
        hipError_t e = hipMemsetAsync(d_peak, 0, sizeof(float) * batch, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(heatmap_max_kernel, dim3(blocks, batch), dim3(256), 0, stream, d_power, n, (unsigned *) d_peak);
    }
    hipLaunchKernelGGL(heatmap_scale_kernel, dim3(blocks, batch), dim3(256), 0, stream, d_power, n, d_peak, d_pix);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Few-beam delay-and-sum for the trackers: Particle::beam and Particle::das,
// src/dsp/particle.cpp:51-82 and :88-103, for a batch of steered directions in one launch (the
// reference runs 4 monopulse directions x (seekers + trackers) one after the other,
// gradient_ascend.cpp:30-81).  One workgroup per direction, thread i owns output sample i; mics are
// visited in the reference's order with delay()'s operation order (d = cur - next; t = fma(frac, d,
// next); out += t), so the 256-sample beam -- the signal MISOWorker hands to the audio path,
// miso.cpp:46 -- is bit-identical to the reference's.  A direction reads 64 x 257 floats that all
// directions share, straight from L2: no LDS staging at this size.  power = sum MA^2 / N_SAMPLES
// (particle.cpp:68-77: not divided by the mic count, unlike the MIMO sweep).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSamples) void das_beam_kernel(const float *frame, const LutEntry *entries, int usable,
                                                            float *power, float *beams) {
    __shared__ float line[kSamples];
    __shared__ float partial[kSamples / 64];
    const int i = threadIdx.x;
    const LutEntry *row = entries + (size_t) blockIdx.x * usable;
    float out = 0.0f;
    for (int s = 0; s < usable; s++) {
        const LutEntry e = row[s];  // uniform: scalar load
        const float *x = frame + e.off_rel + i;
        const float cur = x[0], nxt = x[1];
        const float d = cur - nxt;
        const float t = __builtin_fmaf(e.frac, d, nxt);
        out = out + t;
    }
    if (beams) beams[(size_t) blockIdx.x * kSamples + i] = out;
    line[i] = out;
    __syncthreads();
    float sq = 0.0f;
    if (i >= 1 && i <= kSamples - 2) {
        const float ma = out * 0.5f - 0.25f * (line[i + 1] + line[i - 1]);
        sq = ma * ma;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sq += __shfl_xor(sq, s);
    if ((i & 63) == 0) partial[i >> 6] = sq;
    __syncthreads();
    if (i == 0 && power) power[blockIdx.x] = (partial[0] + partial[1] + partial[2] + partial[3]) / (float) kSamples;
}

hipError_t launch_das_beams(const float *d_frame, const LutEntry *d_entries, int usable, int n_dir, float *d_power,
                            float *d_beams, hipStream_t stream) {
    hipLaunchKernelGGL(das_beam_kernel, dim3(n_dir), dim3(kSamples), 0, stream, d_frame, d_entries, usable, d_power,
                       d_beams);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Calibration on the device: the per-mic mean square of AWProcessingUnit::calibrate,
// src/aw_processing_unit/aw_processing_unit.cpp:133-143:  power = (sum_i x_i * x_i) / hist,
// accumulated in float in sample order (multiply, then add: no FMA), so the 64 values -- and the
// median / usable-mic decisions the host takes on them -- equal the scalar loop bit for bit.  One
// workgroup per mic: the row is fetched coalesced into LDS, then one lane walks it.  A one-off
// call on 256 KiB; latency-bound by design (about 1024 dependent adds).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stream_power_kernel(const float *rows, int pitch, int hist, float *out) {
    extern __shared__ float samples[];
    const float *row = rows + (size_t) blockIdx.x * pitch;
    for (int i = threadIdx.x; i < hist; i += blockDim.x) samples[i] = row[i];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma clang fp contract(off)  // hipcc contracts a*b+c by default; the scalar loop it mirrors does not
        float sum = 0.0f;
        for (int i = 0; i < hist; i++) {
            const float square = samples[i] * samples[i];
            sum = sum + square;
        }
        out[blockIdx.x] = __fdiv_rn(sum, (float) hist);
    }
}

// ---------------------------------------------------------------------------------------
// MIMOWorker::computeDelayLUT on the device (src/dsp/mimo.cpp:20-59; SURVEY 8b, "optionally ... so the LUT can be
// generated on device").  The per-pixel part -- the sine-space grid, asin / atan2 and the rotation entries, all in
// double with the host's libm -- stays on the host (a few thousand pixels); what the device does is the P x n part:
// steer() and compute_delays() of src/geometry/antenna.cpp:89-107 for every (pixel, mic) with the host builder's
// operations in the host builder's order (contraction into FMAs switched off for this kernel),
// the minimum over the mics (exact in any order), and the split of mimo.cpp:46-54.  Bit-identical to
// awpu_hip_build_delay_table by construction; one workgroup per pixel.
// rot: [P][12] = the nine entries of Rz(phi) row-major, then row z of Ry(-theta).
// ---------------------------------------------------------------------------------------
__global__ void delay_table_kernel(const float *xyz, int n, const float *rot, float scale, int32_t *off, float *frac) {
#pragma clang fp contract(off)  // the host builder's x86 code has no FMA: a contracted a*b + c would round once instead of twice
    __shared__ float lowest_of_wave[4];
    const int p = blockIdx.x;
    const float *m = rot + (size_t) p * 12;
    auto delay = [&](int i) {  // steer() + compute_delays() before the minimum is removed, antenna.cpp:89-107
        const float p0 = xyz[i], p1 = xyz[n + i], p2 = xyz[2 * n + i];
        float t[3];
#pragma unroll
        for (int r = 0; r < 3; r++) t[r] = m[3 * r] * p0 + m[3 * r + 1] * p1 + m[3 * r + 2] * p2;
        const float z = m[9] * t[0] + m[10] * t[1] + m[11] * t[2];
        return z * scale;
    };
    float lowest = __builtin_inff();
    for (int i = threadIdx.x; i < n; i += blockDim.x) lowest = fminf(lowest, delay(i));
    for (int d = 32; d >= 1; d >>= 1) lowest = fminf(lowest, __shfl_xor(lowest, d));
    if ((threadIdx.x & 63) == 0) lowest_of_wave[threadIdx.x >> 6] = lowest;
    __syncthreads();
    lowest = fminf(fminf(lowest_of_wave[0], lowest_of_wave[1]), fminf(lowest_of_wave[2], lowest_of_wave[3]));
    for (int i = threadIdx.x; i < n; i += blockDim.x) {  // (recomputed: the same operations give the same bits)
        const float tau = delay(i) - lowest;
        const float whole = truncf(tau);  // modf((double) tau, &whole): exact in float as well
        frac[(size_t) p * n + i] = tau - whole;
        off[(size_t) p * n + i] = kSamples - (int) whole;
    }
}

hipError_t launch_delay_table(const float *d_xyz, int n, const float *d_rot, int n_pixels, float scale, int32_t *d_off, float *d_frac,
                              hipStream_t stream) {
    hipLaunchKernelGGL(delay_table_kernel, dim3(n_pixels), dim3(256), 0, stream, d_xyz, n, d_rot, scale, d_off, d_frac);
    return hipGetLastError();
}

hipError_t launch_stream_power(const float *d_rows, int pitch, int hist, int n, float *d_out, hipStream_t stream) {
    hipLaunchKernelGGL(stream_power_kernel, dim3(n), dim3(256), sizeof(float) * hist, stream, d_rows, pitch, hist, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// The display upscale, cv::resize(compact, normal, ..., INTER_LINEAR) in AWProcessingUnit::draw
// (src/aw_processing_unit/aw_processing_unit.cpp:252), with an optional colour table
// (cv::applyColorMap, src/aw_processing_unit/main.cpp:345) fused into the store.  Arithmetic is OpenCV's
// 8-bit fixed-point path (11-bit weights, horizontal sums in int, vertical (b*(S>>4))>>16) so that the
// image equals what the reference's GUI thread would show.  The compact image is a few KiB and stays
// in cache; the kernel is bound by the writes of the large image (one thread per output pixel, rows
// contiguous across lanes).
// ---------------------------------------------------------------------------------------
void resize_taps(int ssize, int dsize, bool zero_frac_at_border, ResizeTap *taps) {
    const double scale = 1.0 / ((double) dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float frac = (float) ((d + 0.5) * scale - 0.5);
        int first = (int) std::floor(frac);
        frac -= (float) first;
        if (zero_frac_at_border && first < 0) frac = 0.f, first = 0;
        if (zero_frac_at_border && first >= ssize - 1) frac = 0.f, first = ssize - 1;
        taps[d].src = first;
        taps[d].w0 = (int16_t) std::lrint((1.f - frac) * 2048.f);
        taps[d].w1 = (int16_t) std::lrint(frac * 2048.f);
    }
}

__global__ void upscale_kernel(const uint8_t *src, int srows, int scols, const ResizeTap *taps, const uint8_t *colormap,
                               uint8_t *dst, int drows, int dcols) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= dcols) return;
    const uint8_t *img = src + (size_t) blockIdx.z * srows * scols;
    const ResizeTap tx = taps[dx], ty = taps[dcols + dy];
    const int c0 = tx.src, c1 = min(tx.src + 1, scols - 1);
    const int r0 = min(max(ty.src, 0), srows - 1), r1 = min(max(ty.src + 1, 0), srows - 1);
    const int S0 = img[r0 * scols + c0] * tx.w0 + img[r0 * scols + c1] * tx.w1;
    const int S1 = img[r1 * scols + c0] * tx.w0 + img[r1 * scols + c1] * tx.w1;
    const uint8_t v = resize_combine(S0, S1, ty.w0, ty.w1);
    const size_t o = ((size_t) blockIdx.z * drows + dy) * dcols + dx;
    if (colormap) {
        dst[3 * o + 0] = colormap[3 * v + 0];
        dst[3 * o + 1] = colormap[3 * v + 1];
        dst[3 * o + 2] = colormap[3 * v + 2];
    } else {
        dst[o] = v;
    }
}

hipError_t launch_upscale(const uint8_t *d_src, int srows, int scols, int batch, const ResizeTap *d_taps,
                          const uint8_t *d_colormap, uint8_t *d_dst, int drows, int dcols, hipStream_t stream) {
    hipLaunchKernelGGL(upscale_kernel, dim3((dcols + 255) / 256, drows, batch), dim3(256), 0, stream, d_src, srows,
                       scols, d_taps, d_colormap, d_dst, drows, dcols);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Ingest on the device: Pipeline::receive_exposure, src/fpga/pipeline.cpp:260-297.
// Input: one block = 256 datagrams as they come off the wire, each
//     { u16 frequency; u8 n_arrays; u8 version; u32 counter; i32 stream[256]; }  (src/fpga/receiver.h:24-30)
// Output: for sensor s and sample i   x = (float) stream_i[flip(s)] / 8388608.0f   (pipeline.cpp:277-290,
// arrays are daisy-chained, every other group of 8 columns is mirrored) written into the per-mic
// history ring at `pos + i`.  The ring is [n_sensors][2048]: every block is written twice, 1024
// floats apart, so that any 1024-sample snapshot is contiguous -- the reference gets the same
// effect by mapping one page twice (src/fpga/streams.hpp:152-182).
// A 64 x 64 tile goes through LDS so that both the datagram reads (along s) and the ring writes
// (along i) are coalesced.
// ---------------------------------------------------------------------------------------
__global__ void unpack_block_kernel(const unsigned char *datagrams, int stride_bytes, int n_sensors, float *ring,
                                    int pos) {
    __shared__ float tile[64][65];
    const int s0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
    for (int r = ty; r < 64; r += 4) {
        const int s = s0 + tx, i = i0 + r;
        float v = 0.0f;
        if (s < n_sensors) {
            // pipeline.cpp:277-287: `inverted` toggles at every multiple of 8, starting inverted
            const bool inverted = ((s >> 3) & 1) == 0;
            const int idx = inverted ? 8 * (1 + (s >> 3)) - 1 - (s & 7) : s;
            const int32_t raw = *(const int32_t *) (datagrams + (size_t) i * stride_bytes + 8 + 4 * idx);
            v = (float) raw / 8388608.0f;  // MAX_VALUE_FLOAT, src/fpga/pipeline.h:25
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int s = s0 + r, i = i0 + tx;
        if (s < n_sensors) {
            const float v = tile[tx][r];
            float *row = ring + (size_t) s * 2048;
            row[pos + i] = v;
            row[(pos + i + 1024) & 2047] = v;
        }
    }
}

hipError_t launch_unpack_block(const void *d_datagrams, int stride_bytes, int n_sensors, float *d_ring, int pos,
                               hipStream_t stream) {
    dim3 grid((n_sensors + 63) / 64, kSamples / 64);
    hipLaunchKernelGGL(unpack_block_kernel, grid, dim3(256), 0, stream, (const unsigned char *) d_datagrams,
                       stride_bytes, n_sensors, d_ring, pos);
    return hipGetLastError();
}

size_t das_exact_lds_bytes(int window, int usable, int *chunk_out) {
    const size_t row = (size_t) window * sizeof(float);
    if (row == 0 || row > kExactLdsBudget) return 0;
    int chunk = (int) (kExactLdsBudget / row);
    if (chunk > usable) chunk = usable;
    if (chunk < 1) return 0;
    if (chunk_out) *chunk_out = chunk;
    return (size_t) chunk * row;
}

hipError_t launch_das_exact(const SweepArgs &a, bool bf16_accumulator, hipStream_t stream) {
    int chunk = 0;
    const size_t lds = das_exact_lds_bytes(a.window, a.usable, &chunk);
    if (lds == 0) return hipErrorInvalidValue;
    const int pix_per_block = (kExactThreads / 64) * kExactPPW;
    dim3 grid((a.pixel_count + pix_per_block - 1) / pix_per_block, a.batch);
    if (grid.y > 65535) return hipErrorInvalidValue;
    if (bf16_accumulator)
        hipLaunchKernelGGL((das_exact_kernel<kExactPPW, true>), grid, dim3(kExactThreads), lds, stream, a, chunk);
    else
        hipLaunchKernelGGL((das_exact_kernel<kExactPPW, false>), grid, dim3(kExactThreads), lds, stream, a, chunk);
    return hipGetLastError();
}

__global__ void upload_floats_kernel(const float4 *src, float4 *dst, size_t n4) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}

hipError_t launch_upload_floats(const float *h_pinned, float *d_dst, size_t n, hipStream_t stream) {
    if ((n & 3) || ((uintptr_t) h_pinned & 15) || ((uintptr_t) d_dst & 15)) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    if (n4 == 0) return hipSuccess;
    const unsigned blocks = (unsigned) std::min<size_t>((n4 + 255) / 256, 256);  // 16 bytes per lane: 64 KB per pass of 16 workgroups
    hipLaunchKernelGGL(upload_floats_kernel, dim3(blocks), dim3(256), 0, stream, (const float4 *) h_pinned, (float4 *) d_dst, n4);
    return hipGetLastError();
}

}  // namespace awpu
