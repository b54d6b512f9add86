/* live_call_rate.c -- what ONE synchronous awpu_hip_process call on a pageable host frame costs at the C level at the shape the
 * reference ships (one 8x8 array, --mimo-res 100: src/main.cpp:38-41): the call MIMOWorker::update makes once per 256-sample block
 * (src/dsp/mimo.cpp:97-151; 5.24 ms apart in the live system, back to back here).  Both fp32 modes, 2000 calls each, median and mean.
 *
 *   gcc -O2 -Iinclude examples/live_call_rate.c -Lbeamforming-lk_amd -lawpu_hip -lm \
 *       -Wl,-rpath,$PWD/beamforming-lk_amd -Wl,-rpath,/opt/rocm/lib -o examples/live_call_rate
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "awpu_hip.h"

#define ROWS 100
#define COLS 100
#define CALLS 2000

static int cmp(const void *a, const void *b) { return (*(const double *) a > *(const double *) b) - (*(const double *) a < *(const double *) b); }
static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

int main(void) {
    static float xyz[3 * AWPU_ELEMENTS], tau[AWPU_ELEMENTS];
    int32_t *off = malloc(sizeof(int32_t) * ROWS * COLS * AWPU_ELEMENTS);
    float *frac = malloc(sizeof(float) * ROWS * COLS * AWPU_ELEMENTS);
    float *frames = malloc(sizeof(float) * 4 * AWPU_ELEMENTS * AWPU_HIST), *power = malloc(sizeof(float) * ROWS * COLS);
    static double us[CALLS];
    awpu_hip_create_antenna(8, 8, 0.02f, xyz);
    awpu_hip_build_delay_table(xyz, AWPU_ELEMENTS, ROWS, COLS, 180.0f, 0, ROWS, off, frac);
    awpu_hip_steering_delays(xyz, AWPU_ELEMENTS, 20.0 * M_PI / 180.0, 35.0 * M_PI / 180.0, tau);
    for (int f = 0; f < 4; f++)
        for (int s = 0; s < AWPU_ELEMENTS; s++)
            for (int i = 0; i < AWPU_HIST; i++)
                frames[(f * AWPU_ELEMENTS + s) * AWPU_HIST + i] = (float) ((1 + f) * 1e-2 * sin(2.0 * M_PI * 9e3 * (i + tau[s]) / 48828.0));
    for (int mode = 0; mode < 2; mode++) {
        awpu_hip_cfg cfg;
        awpu_hip_default_cfg(&cfg);
        cfg.n_streams = AWPU_ELEMENTS;
        cfg.n_pixels = ROWS * COLS;
        cfg.lut_stride = AWPU_ELEMENTS;
        cfg.grid_columns = COLS;
        cfg.max_batch = 1;
        if (mode == 1) cfg.math = AWPU_MATH_F32_FAST;
        awpu_hip_t *engine = NULL;
        int rc = awpu_hip_create(&engine, &cfg);
        if (rc != AWPU_OK) {
            fprintf(stderr, "awpu_hip_create: %s (%s)\n", awpu_hip_strerror(rc), awpu_hip_last_error());
            return rc == AWPU_ERR_NO_DEVICE ? 2 : 1;
        }
        rc = awpu_hip_set_delay_table(engine, off, frac);
        if (rc == AWPU_OK) rc = awpu_hip_set_active_mics(engine, NULL, AWPU_ELEMENTS);
        for (int k = 0; k < 40 && rc == AWPU_OK; k++) rc = awpu_hip_process(engine, frames + (size_t) (k & 3) * AWPU_ELEMENTS * AWPU_HIST, 1, power);
        const double t_begin = now_us();
        for (int k = 0; k < CALLS && rc == AWPU_OK; k++) {
            const double t0 = now_us();
            rc = awpu_hip_process(engine, frames + (size_t) (k & 3) * AWPU_ELEMENTS * AWPU_HIST, 1, power);
            us[k] = now_us() - t0;
        }
        const double mean = (now_us() - t_begin) / CALLS;
        if (rc != AWPU_OK) {
            fprintf(stderr, "sweep failed: %s (%s)\n", awpu_hip_strerror(rc), awpu_hip_last_error());
            return 1;
        }
        awpu_hip_stats st;
        awpu_hip_get_stats(engine, &st);
        qsort(us, CALLS, sizeof(double), cmp);
        printf("%s: awpu_hip_process, one frame per call, %d calls: median %.2f us, mean %.2f us, min %.2f us (sweep kernel %d, last timed launch %.1f us)\n",
               mode ? "AWPU_MATH_F32_FAST" : "AWPU_MATH_F32_EXACT (default)", CALLS, us[CALLS / 2], mean, us[0], st.kernel_variant, st.last_kernel_ms * 1e3);
        awpu_hip_destroy(engine);
    }
    return 0;
}
