/*
 * das_oracle.h -- CPU restatement of the reference delay-and-sum heatmap path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (beamforming-lk_amd/,
 * include/, the C-ABI library) may include, link or call this.  It is used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
 *
 * Every function names the reference file:line it restates (paths relative to
 * the reference tree, acoustic-warfare/beamforming-lk @ 2024_08_07).
 *
 * Pinning: the reference ships no golden vector or known-answer test for this
 * path (its tests/ import Cython modules that are not in the tree).  The
 * restatement is pinned instead against outputs of the reference's own
 * src/dsp/delay.cpp compiled in place (oracle/_ref, see oracle/Makefile) and
 * against fixtures generated from that build (tests/golden/).
 */
#ifndef DAS_ORACLE_H
#define DAS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* constants: src/geometry/antenna.h:16-21, src/fpga/streams.hpp:28-32 */
#define ORACLE_N_SAMPLES 256
#define ORACLE_HIST 1024
#define ORACLE_SAMPLE_RATE 48828.0
#define ORACLE_PROPAGATION_SPEED 340.0
#define ORACLE_ARRAY_COLUMNS 8
#define ORACLE_ARRAY_ROWS 8
#define ORACLE_ELEMENTS 64
#define ORACLE_DISTANCE 0.02

/* create_antenna, src/geometry/antenna.cpp:60-87.  xyz is 3 x (rows*columns),
 * row-major by coordinate (xyz[0*n+i]=x_i ...), element i = r*columns + c. */
void oracle_create_antenna(int columns, int rows, float distance, float *xyz);

/* Build-defined multi-array geometry (the reference only ever
 * beamforms antennas[0]): arrays_x * arrays_y tiles of one 8x8 array at the
 * same pitch, stream index = a*64 + r*8 + c with a = ay*arrays_x + ax
 * (src/aw_processing_unit/aw_processing_unit.cpp:120).  For 1x1 this equals
 * oracle_create_antenna(8, 8, distance). */
void oracle_create_tiled_antenna(int arrays_x, int arrays_y, float distance, float *xyz);

/* steering_vector_spherical = steer + compute_delays,
 * src/geometry/antenna.cpp:126-129, 99-107, 89-97 with rotateY/rotateZ of
 * src/geometry/geometry.cpp:219-233.  fp32 like the reference. */
void oracle_steering_delays_f32(const float *xyz, int n, double theta, double phi, float *tau);
/* same closed form in fp64 (tie-breaker). */
void oracle_steering_delays_f64(const float *xyz, int n, double theta, double phi, double *tau);

/* MIMOWorker::computeDelayLUT, src/dsp/mimo.cpp:20-59.  off/frac are
 * [rows*columns][n] pixel-major, k = r*columns + c. */
void oracle_compute_delay_lut(const float *xyz, int n, int rows, int columns, float fov_deg,
                              int32_t *off, float *frac);
/* the un-split fp64 delays for the same grid, [rows*columns][n] */
void oracle_compute_delays_f64(const float *xyz, int n, int rows, int columns, float fov_deg,
                               double *tau);

/* delay(), AVX2 variant: src/dsp/delay.cpp:16-26 (scalar form :44-48). */
void oracle_delay_lerp(float *out, const float *signal, float fraction);
/* delay(), FIR variant: src/dsp/delay.cpp:31-40; coeffs = the caller's
 * [101][8] table (the reference's is src/dsp/filter.h:10-112). */
void oracle_delay_fir8(float *out, const float *signal, float fraction, const float *coeffs);

/* MIMOWorker::update sweep + epilogue, src/dsp/mimo.cpp:121-151.
 *   X          [n_streams][hist] snapshot, oldest..newest (streams.hpp:113-116)
 *   off, frac  [P][lut_stride], indexed by physical mic id (mimo.cpp:126-127)
 *   index      [usable] physical mic ids (antenna.index); signals[s] is the
 *              snapshot of stream index[s] (mimo.cpp:100-103)
 *   power      [P]
 *   out_dbg    optional [P][256] pre-epilogue sums (NULL to skip)
 * fp32, mic order s = 0..usable-1, operation order of delay.cpp:19-25. */
void oracle_das_f32(const float *X, int hist, const int32_t *off, const float *frac, int P,
                    int lut_stride, const int32_t *index, int usable, float *power,
                    float *out_dbg);
/* the build's own bf16-accumulator mode (AWPU_MATH_BF16_ACC; not reference code): every running sum
 * rounded to bfloat16 after each mic, term and epilogue in fp32. */
void oracle_das_bf16acc(const float *X, int hist, const int32_t *off, const float *frac, int P,
                        int lut_stride, const int32_t *index, int usable, float *power);
/* same with every sum in fp64 (inputs still the fp32 tables). */
void oracle_das_f64(const float *X, int hist, const int32_t *off, const float *frac, int P,
                    int lut_stride, const int32_t *index, int usable, double *power);
/* FIR8 interpolation variant of the sweep (delay.cpp:31-40 inside mimo.cpp:121-151). */
void oracle_das_fir8_f32(const float *X, int hist, const int32_t *off, const float *frac, int P,
                         int lut_stride, const int32_t *index, int usable, const float *coeffs,
                         float *power);
/* the same with every sum in double (tie-breaker only) */
void oracle_das_fir8_f64(const float *X, int hist, const int32_t *off, const float *frac, int P,
                         int lut_stride, const int32_t *index, int usable, const float *coeffs,
                         double *power);

/* MIMOWorker::populateHeatmap with USE_DB 0, src/dsp/mimo.cpp:61-95. */
/* Particle::beam / Particle::das (src/dsp/particle.cpp:51-103) for n_dir directions; power and beams may be NULL */
void oracle_particle_beams(const float *X, int hist, const int32_t *off, const float *frac, int n_dir,
                           int lut_stride, const int32_t *index, int usable, float *power, float *beams);
void oracle_heatmap_u8(const float *power, int P, uint8_t *pix);
/* cv::resize(..., INTER_LINEAR) on an 8-bit single-channel image (aw_processing_unit.cpp:252), upscaling only */
int oracle_resize_linear_u8(const uint8_t *src, int srows, int scols, uint8_t *dst, int drows, int dcols);

/* AWProcessingUnit::calibrate per-array mic selection,
 * src/aw_processing_unit/aw_processing_unit.cpp:128-200.  X is the 64 streams
 * of one array [64][hist]; returns usable and fills index[<=64], corr[<=64]. */
int oracle_calibrate(const float *X, int hist, float reference_power_level, int32_t *index,
                     float *corr, float *median_out);

/* Pipeline::receive_exposure de-interleave, src/fpga/pipeline.cpp:260-292:
 * 256 datagrams of int32 stream[n_sensors] -> float block[n_sensors][256]. */
void oracle_unpack_exposure(const int32_t *stream, int stream_stride, int n_sensors, float *block);

#ifdef __cplusplus
}
#endif
#endif
