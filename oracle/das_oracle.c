/*
 * das_oracle.c -- CPU restatement of the reference delay-and-sum heatmap path.
 *
 * TEST INFRASTRUCTURE ONLY (see das_oracle.h).  Plain C, scalar, one thread.
 * Build with -ffp-contract=off so that the only fused multiply-adds are the
 * explicit fmaf() calls that mirror the reference's _mm256_fmadd_ps.
 *
 * Parity pinning: validated against the reference's own src/dsp/delay.cpp
 * compiled in place (oracle/_ref) by tests/test_oracle_golden.py and against
 * tests/golden/ (generated from that build by tests/golden/make_golden.py).
 */
#include "das_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ geometry */

/* src/geometry/antenna.cpp:60-87.  Note the reference centres x with `rows`
 * and y with `columns` (swapped; harmless for square arrays) -- kept. */
void oracle_create_antenna(int columns, int rows, float distance, float *xyz) {
    const int n = rows * columns;
    const float half = distance / 2;
    int i = 0;
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < columns; c++) {
            xyz[0 * n + i] = (float) c * distance - (float) rows * half + half;
            xyz[1 * n + i] = (float) r * distance - (float) columns * half + half;
            xyz[2 * n + i] = 0.f;
            i++;
        }
    }
}

/* Build-defined.  Tile (ax, ay) holds the 8x8 array whose
 * streams are a*64 .. a*64+63, a = ay*arrays_x + ax
 * (aw_processing_unit.cpp:120 stream numbering).  Positions follow the same
 * formula as create_antenna evaluated on the global (row, column) of the
 * element, so the 1x1 case is bit-identical to oracle_create_antenna(8, 8). */
void oracle_create_tiled_antenna(int arrays_x, int arrays_y, float distance, float *xyz) {
    const int columns = arrays_x * ORACLE_ARRAY_COLUMNS;
    const int rows = arrays_y * ORACLE_ARRAY_ROWS;
    const int n = rows * columns;
    const float half = distance / 2;
    for (int ay = 0; ay < arrays_y; ay++) {
        for (int ax = 0; ax < arrays_x; ax++) {
            const int a = ay * arrays_x + ax;
            for (int rr = 0; rr < ORACLE_ARRAY_ROWS; rr++) {
                for (int cc = 0; cc < ORACLE_ARRAY_COLUMNS; cc++) {
                    const int i = a * ORACLE_ELEMENTS + rr * ORACLE_ARRAY_COLUMNS + cc;
                    const int r = ay * ORACLE_ARRAY_ROWS + rr;
                    const int c = ax * ORACLE_ARRAY_COLUMNS + cc;
                    xyz[0 * n + i] = (float) c * distance - (float) rows * half + half;
                    xyz[1 * n + i] = (float) r * distance - (float) columns * half + half;
                    xyz[2 * n + i] = 0.f;
                }
            }
        }
    }
}

/* steer(): rotated = Ry(-(float)theta) * (Rz((float)phi) * points),
 * src/geometry/antenna.cpp:99-107; matrix entries are (float)cos((double)a)
 * etc., src/geometry/geometry.cpp:219-233.  compute_delays(): z row times
 * (float)(SAMPLE_RATE / PROPAGATION_SPEED), minus the minimum,
 * src/geometry/antenna.cpp:89-97. */
void oracle_steering_delays_f32(const float *xyz, int n, double theta, double phi, float *tau) {
    const float az = (float) phi;
    const float ay = -(float) theta;
    /* rotateZ(az) */
    const float rz00 = (float) cos((double) az), rz01 = -(float) sin((double) az);
    const float rz10 = (float) sin((double) az), rz11 = (float) cos((double) az);
    /* rotateY(ay), bottom row: (-sin, 0, cos) */
    const float ry20 = -(float) sin((double) ay), ry21 = 0.0f, ry22 = (float) cos((double) ay);
    const float k = (float) (ORACLE_SAMPLE_RATE / ORACLE_PROPAGATION_SPEED);
    float mn = INFINITY;
    for (int i = 0; i < n; i++) {
        const float x = xyz[0 * n + i], y = xyz[1 * n + i], z = xyz[2 * n + i];
        const float tx = rz00 * x + rz01 * y + 0.0f * z;
        const float ty = rz10 * x + rz11 * y + 0.0f * z;
        const float tz = 0.0f * x + 0.0f * y + 1.0f * z;
        const float rz = ry20 * tx + ry21 * ty + ry22 * tz;
        tau[i] = rz * k;
        if (tau[i] < mn) mn = tau[i];
    }
    for (int i = 0; i < n; i++) tau[i] -= mn;
}

void oracle_steering_delays_f64(const float *xyz, int n, double theta, double phi, double *tau) {
    const double k = ORACLE_SAMPLE_RATE / ORACLE_PROPAGATION_SPEED;
    const double st = sin(theta), cp = cos(phi), sp = sin(phi);
    double mn = INFINITY;
    for (int i = 0; i < n; i++) {
        const double x = xyz[0 * n + i], y = xyz[1 * n + i];
        tau[i] = k * st * (cp * x - sp * y);
        if (tau[i] < mn) mn = tau[i];
    }
    for (int i = 0; i < n; i++) tau[i] -= mn;
}

/* pixel (r, c) -> (theta, phi), src/dsp/mimo.cpp:21-43 (all double). */
static void pixel_direction(int r, int c, int rows, int columns, float fov_deg, double *theta,
                            double *phi) {
    const double fovRadian = (double) fov_deg * (M_PI / 180.0);
    const double sepR = sin(fovRadian / 2.0) / ((double) rows / 2.0);
    const double sepC = sin(fovRadian / 2.0) / ((double) columns / 2.0);
    double y = (double) r * sepR - (double) rows * sepR / 2.0 + sepR / 2.0;
    double x = (double) c * sepC - (double) columns * sepC / 2.0 + sepC / 2.0;
    double norm = sqrt(pow(x, 2) + pow(y, 2));
    x /= norm;
    y /= norm;
    if (norm > 1.0) norm = 1.0;
    *theta = asin(norm);
    *phi = atan2(y, x);
}

/* src/dsp/mimo.cpp:20-59 */
void oracle_compute_delay_lut(const float *xyz, int n, int rows, int columns, float fov_deg,
                              int32_t *off, float *frac) {
    float *tau = (float *) malloc(sizeof(float) * (size_t) n);
    int k = 0;
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < columns; c++) {
            double theta, phi;
            pixel_direction(r, c, rows, columns, fov_deg, &theta, &phi);
            oracle_steering_delays_f32(xyz, n, theta, phi, tau);
            for (int i = 0; i < n; i++) {
                double whole;
                const float fraction = (float) modf((double) tau[i], &whole);
                frac[(size_t) k * n + i] = fraction;
                off[(size_t) k * n + i] = ORACLE_N_SAMPLES - (int) whole;
            }
            k++;
        }
    }
    free(tau);
}

void oracle_compute_delays_f64(const float *xyz, int n, int rows, int columns, float fov_deg,
                               double *tau) {
    int k = 0;
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < columns; c++) {
            double theta, phi;
            pixel_direction(r, c, rows, columns, fov_deg, &theta, &phi);
            oracle_steering_delays_f64(xyz, n, theta, phi, tau + (size_t) k * n);
            k++;
        }
    }
}

/* --------------------------------------------------------------------- delay */

/* src/dsp/delay.cpp:16-26: out += fma(f, cur - next, next) per sample. */
void oracle_delay_lerp(float *out, const float *signal, float fraction) {
    for (int i = 0; i < ORACLE_N_SAMPLES; i++) {
        const float d = signal[i] - signal[i + 1];
        const float t = fmaf(fraction, d, signal[i + 1]);
        out[i] = out[i] + t;
    }
}

/* src/dsp/delay.cpp:31-40 */
void oracle_delay_fir8(float *out, const float *signal, float fraction, const float *coeffs) {
    const float get_filter = fraction * 100.0f + 0.5f;
    const int delay_int = (int) get_filter;
    for (int n = 0; n < ORACLE_N_SAMPLES; ++n) {
        for (int i = 0; i < 8; ++i) {
            out[n] += coeffs[delay_int * 8 + i] * signal[n + i];
        }
    }
}

/* --------------------------------------------------------------------- sweep */

/* epilogue, src/dsp/mimo.cpp:131-137 */
static float epilogue_f32(const float *out, int count) {
    float power = 0.0f;
    for (int i = 1; i < ORACLE_N_SAMPLES - 1; i++) {
        const float MA = out[i] * 0.5f - 0.25f * (out[i + 1] + out[i - 1]);
        power += MA * MA;
    }
    power /= (float) (ORACLE_N_SAMPLES * count);
    return power;
}

/* src/dsp/mimo.cpp:97-151.  signals[s] is stream index[s] (mimo.cpp:100-103);
 * the table column is the physical id i = index[s] (mimo.cpp:125-127). */
void oracle_das_f32(const float *X, int hist, const int32_t *off, const float *frac, int P,
                    int lut_stride, const int32_t *index, int usable, float *power,
                    float *out_dbg) {
    for (int m = 0; m < P; m++) {
        float out[ORACLE_N_SAMPLES] = {0.0f};
        int count = 0;
        for (int s = 0; s < usable; s++) {
            const int i = index[s];
            const float fraction = frac[(size_t) m * lut_stride + i];
            const int offset = off[(size_t) m * lut_stride + i];
            oracle_delay_lerp(out, X + (size_t) i * hist + offset, fraction);
            count++;
        }
        if (out_dbg) memcpy(out_dbg + (size_t) m * ORACLE_N_SAMPLES, out, sizeof(out));
        power[m] = epilogue_f32(out, count);
    }
}

/* Particle::beam and Particle::das, src/dsp/particle.cpp:51-82 and :88-103 (USE_BANDPASS 1,
 * particle.h:17), for n_dir steered directions with tables as Particle::steer fills them
 * (particle.cpp:37-49: same split as the MIMO table).  Differences from the MIMO sweep: the power is
 * divided by N_SAMPLES only (`norm` is computed and never used), and the 256-sample beam itself is a
 * result (MISOWorker hands it to the audio path, src/dsp/miso.cpp:46). */
void oracle_particle_beams(const float *X, int hist, const int32_t *off, const float *frac, int n_dir,
                           int lut_stride, const int32_t *index, int usable, float *power, float *beams) {
    for (int m = 0; m < n_dir; m++) {
        float out[ORACLE_N_SAMPLES] = {0.0f};
        for (int s = 0; s < usable; s++) {
            const int i = index[s];
            oracle_delay_lerp(out, X + (size_t) i * hist + off[(size_t) m * lut_stride + i],
                              frac[(size_t) m * lut_stride + i]);
        }
        float power_accumulator = 0.0f;
        for (int i = 1; i < ORACLE_N_SAMPLES - 1; i++) {
            const float MA = out[i] * 0.5f - 0.25f * (out[i + 1] + out[i - 1]);
            power_accumulator += MA * MA;
        }
        power_accumulator /= (float) ORACLE_N_SAMPLES;
        if (power) power[m] = power_accumulator;
        if (beams) memcpy(beams + (size_t) m * ORACLE_N_SAMPLES, out, sizeof(out));
    }
}

void oracle_das_fir8_f32(const float *X, int hist, const int32_t *off, const float *frac, int P,
                         int lut_stride, const int32_t *index, int usable, const float *coeffs,
                         float *power) {
    for (int m = 0; m < P; m++) {
        float out[ORACLE_N_SAMPLES] = {0.0f};
        int count = 0;
        for (int s = 0; s < usable; s++) {
            const int i = index[s];
            oracle_delay_fir8(out, X + (size_t) i * hist + off[(size_t) m * lut_stride + i],
                              frac[(size_t) m * lut_stride + i], coeffs);
            count++;
        }
        power[m] = epilogue_f32(out, count);
    }
}

/* NOT a restatement of reference code: the checker of the build's own AWPU_MATH_BF16_ACC mode (BASELINE
 * configs[4], "bf16 vs fp32 accumulator").  The sweep of oracle_das_f32 with the running sum of every sample
 * kept in bfloat16: after each mic  acc = bf16(acc + term), round to nearest even; term and epilogue in fp32. */
static float oracle_round_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, sizeof(u));
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; /* finite inputs only */
    memcpy(&x, &u, sizeof(u));
    return x;
}

void oracle_das_bf16acc(const float *X, int hist, const int32_t *off, const float *frac, int P,
                        int lut_stride, const int32_t *index, int usable, float *power) {
    for (int m = 0; m < P; m++) {
        float out[ORACLE_N_SAMPLES];
        for (int i = 0; i < ORACLE_N_SAMPLES; i++) out[i] = 0.0f;
        for (int s = 0; s < usable; s++) {
            const int id = index[s];
            const float f = frac[(size_t) m * lut_stride + id];
            const float *sig = X + (size_t) id * hist + off[(size_t) m * lut_stride + id];
            for (int i = 0; i < ORACLE_N_SAMPLES; i++) {
                const float t = fmaf(f, sig[i] - sig[i + 1], sig[i + 1]);
                out[i] = oracle_round_bf16(out[i] + t);
            }
        }
        power[m] = epilogue_f32(out, usable);
    }
}

void oracle_das_f64(const float *X, int hist, const int32_t *off, const float *frac, int P,
                    int lut_stride, const int32_t *index, int usable, double *power) {
    for (int m = 0; m < P; m++) {
        double out[ORACLE_N_SAMPLES];
        for (int i = 0; i < ORACLE_N_SAMPLES; i++) out[i] = 0.0;
        for (int s = 0; s < usable; s++) {
            const int id = index[s];
            const double f = (double) frac[(size_t) m * lut_stride + id];
            const float *sig = X + (size_t) id * hist + off[(size_t) m * lut_stride + id];
            for (int i = 0; i < ORACLE_N_SAMPLES; i++) {
                const double cur = sig[i], nxt = sig[i + 1];
                out[i] += nxt + f * (cur - nxt);
            }
        }
        double p = 0.0;
        for (int i = 1; i < ORACLE_N_SAMPLES - 1; i++) {
            const double MA = out[i] * 0.5 - 0.25 * (out[i + 1] + out[i - 1]);
            p += MA * MA;
        }
        power[m] = p / (double) (ORACLE_N_SAMPLES * usable);
    }
}

/* The FIR8 sweep (delay.cpp:31-40 inside mimo.cpp:121-151) with every sum in double: the tie-breaker that shows how
 * far the reference's own fp32 arithmetic is from exact sums (not a restatement: the reference has no fp64 path). */
void oracle_das_fir8_f64(const float *X, int hist, const int32_t *off, const float *frac, int P,
                         int lut_stride, const int32_t *index, int usable, const float *coeffs,
                         double *power) {
    for (int m = 0; m < P; m++) {
        double out[ORACLE_N_SAMPLES];
        for (int i = 0; i < ORACLE_N_SAMPLES; i++) out[i] = 0.0;
        for (int s = 0; s < usable; s++) {
            const int id = index[s];
            const float get_filter = frac[(size_t) m * lut_stride + id] * 100.0f + 0.5f; /* delay.cpp:32-33 */
            const float *c = coeffs + (size_t) ((int) get_filter) * 8;
            const float *sig = X + (size_t) id * hist + off[(size_t) m * lut_stride + id];
            for (int n = 0; n < ORACLE_N_SAMPLES; n++)
                for (int t = 0; t < 8; t++) out[n] += (double) c[t] * (double) sig[n + t];
        }
        double p = 0.0;
        for (int i = 1; i < ORACLE_N_SAMPLES - 1; i++) {
            const double MA = out[i] * 0.5 - 0.25 * (out[i + 1] + out[i - 1]);
            p += MA * MA;
        }
        power[m] = p / (double) (ORACLE_N_SAMPLES * usable);
    }
}

/* ------------------------------------------------------------------- display */

/* src/dsp/mimo.cpp:61-95 with USE_DB 0: db = pow(p/maxV, 1) * 255, clipped. */
void oracle_heatmap_u8(const float *power, int P, uint8_t *pix) {
    float maxV = 0.0f;
    for (int i = 0; i < P; i++) {
        if (power[i] > maxV) maxV = power[i];
    }
    for (int i = 0; i < P; i++) {
        double db = pow((double) (power[i] / maxV), 1);
        db *= 255.0;
        if (db < 0.0) db = 0.0;
        if (db > 255.0) db = 255.0;
        /* An all-zero frame makes db NaN (0/0); the reference then casts NaN to uchar, which C and C++
         * leave undefined (0 on x86-64).  The product defines that case as level 0; so does the checker. */
        if (db != db) db = 0.0;
        pix[i] = (uint8_t) db;
    }
}

/* cv::resize(compact, normal, size, 0, 0, cv::INTER_LINEAR) on CV_8UC1, the step after populateHeatmap in
 * AWProcessingUnit::draw (src/aw_processing_unit/aw_processing_unit.cpp:252).  OpenCV itself is a
 * third-party dependency that is absent from the reference tree and from this image (the reference's
 * CMakeLists finds the system OpenCV 4), so this restates OpenCV 4.x's published generic 8-bit path
 * (modules/imgproc/src/resize.cpp: resize -> ResizeFunc with HResizeLinear<uchar,int,short,2048> and
 * VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>): 11-bit fixed-point coefficients,
 * horizontal pass into ints, vertical pass ((b*(S>>4))>>16 summed, +2, >>2).  PARITY UNPINNED for this
 * one function: there is no OpenCV here to check it against; the tests pin hand-computed vectors only.
 * Upscaling only (dst >= src in both directions; the decimation special cases are not restated). */
static void resize_coeffs(int ssize, int dsize, int clamp_frac, int *ofs, short *coef) {
    const double inv_scale = (double) dsize / ssize;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float) ((d + 0.5) * scale - 0.5);
        int s0 = (int) floorf(f);
        f -= s0;
        if (clamp_frac) { /* columns: the fraction is zeroed at the borders; rows are clamped instead */
            if (s0 < 0) { f = 0.f; s0 = 0; }
            if (s0 >= ssize - 1) { f = 0.f; s0 = ssize - 1; }
        }
        ofs[d] = s0;
        coef[2 * d] = (short) lrintf((1.f - f) * 2048.f);
        coef[2 * d + 1] = (short) lrintf(f * 2048.f);
    }
}

int oracle_resize_linear_u8(const uint8_t *src, int srows, int scols, uint8_t *dst, int drows, int dcols) {
    if (drows < srows || dcols < scols || srows < 1 || scols < 1) return -1;
    int *xofs = (int *) malloc(sizeof(int) * (size_t) (dcols + drows));
    short *alpha = (short *) malloc(sizeof(short) * 2 * (size_t) (dcols + drows));
    if (!xofs || !alpha) { free(xofs); free(alpha); return -1; }
    int *yofs = xofs + dcols;
    short *beta = alpha + 2 * dcols;
    resize_coeffs(scols, dcols, 1, xofs, alpha);
    resize_coeffs(srows, drows, 0, yofs, beta);
    for (int dy = 0; dy < drows; dy++) {
        int r0 = yofs[dy], r1 = yofs[dy] + 1;
        r0 = r0 < 0 ? 0 : (r0 < srows ? r0 : srows - 1);
        r1 = r1 < 0 ? 0 : (r1 < srows ? r1 : srows - 1);
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        for (int dx = 0; dx < dcols; dx++) {
            const int sx = xofs[dx], sx1 = sx + 1 < scols ? sx + 1 : sx;
            const int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            const int S0 = src[r0 * scols + sx] * a0 + src[r0 * scols + sx1] * a1;
            const int S1 = src[r1 * scols + sx] * a0 + src[r1 * scols + sx1] * a1;
            dst[(size_t) dy * dcols + dx] = (uint8_t) ((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
        }
    }
    free(xofs);
    free(alpha);
    return 0;
}

/* --------------------------------------------------------------- calibration */

static int cmp_float(const void *a, const void *b) {
    const float x = *(const float *) a, y = *(const float *) b;
    return (x > y) - (x < y);
}

/* src/aw_processing_unit/aw_processing_unit.cpp:128-200 (one array). */
int oracle_calibrate(const float *X, int hist, float reference_power_level, int32_t *index,
                     float *corr, float *median_out) {
    float power[ORACLE_ELEMENTS];
    for (int s = 0; s < ORACLE_ELEMENTS; s++) {
        float pv = 0.0f;
        for (int i = 0; i < hist; i++) pv += X[(size_t) s * hist + i] * X[(size_t) s * hist + i];
        pv /= (float) hist;
        power[s] = pv;
    }
    float med[ORACLE_ELEMENTS];
    memcpy(med, power, sizeof(med));
    qsort(med, ORACLE_ELEMENTS, sizeof(float), cmp_float);
    /* the reference averages elements 32 and 33 (not 31 and 32) */
    const float median = (float) ((med[ORACLE_ELEMENTS / 2] + med[ORACLE_ELEMENTS / 2 + 1]) / 2.0);
    int count = 0;
    for (int s = 0; s < ORACLE_ELEMENTS; s++) {
        const float diff = fabsf(power[s] - median);
        if (diff > 1e-4) {
        } else if (power[s] < median * 1e-3) {
        } else {
            index[count] = s;
            corr[count] = reference_power_level / power[s];
            count++;
        }
    }
    if (median_out) *median_out = median;
    return count;
}

/* -------------------------------------------------------------------- ingest */

/* src/fpga/pipeline.cpp:260-292; MAX_VALUE_FLOAT src/fpga/pipeline.h:25 */
void oracle_unpack_exposure(const int32_t *stream, int stream_stride, int n_sensors, float *block) {
    for (int i = 0; i < ORACLE_N_SAMPLES; i++) {
        const int32_t *msg = stream + (size_t) i * stream_stride;
        int inverted = 0;
        for (int s = 0; s < n_sensors; s++) {
            unsigned idx;
            if (s % ORACLE_ARRAY_COLUMNS == 0) inverted = !inverted;
            if (inverted) {
                idx = ORACLE_ARRAY_COLUMNS * (1 + s / ORACLE_ARRAY_COLUMNS) - 1 -
                      s % ORACLE_ARRAY_COLUMNS;
            } else {
                idx = (unsigned) s;
            }
            block[(size_t) s * ORACLE_N_SAMPLES + i] = (float) msg[idx] / (float) 8388608.0;
        }
    }
}
