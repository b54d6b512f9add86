/* heatmap_min.c -- the smallest complete use of the C ABI (include/awpu_hip.h) from plain C:
 * one 8x8 array, a 32x32 steering grid (BASELINE config 1), one synthetic snapshot -> power per pixel and
 * the 8-bit heatmap.  Mirrors what MIMOWorker does in the reference (src/dsp/mimo.cpp): computeDelayLUT once,
 * then update() + populateHeatmap() per block.
 *
 *   gcc -O2 -Iinclude examples/heatmap_min.c -Lbeamforming-lk_amd -lawpu_hip -lm \
 *       -Wl,-rpath,$PWD/beamforming-lk_amd -Wl,-rpath,/opt/rocm/lib -o examples/heatmap_min
 *
 * Needs an MI355X (gfx950): without one awpu_hip_create reports AWPU_ERR_NO_DEVICE and the program says so. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "awpu_hip.h"

#define ROWS 32
#define COLS 32

int main(void) {
    static float xyz[3 * AWPU_ELEMENTS];
    static int32_t off[ROWS * COLS * AWPU_ELEMENTS];
    static float frac[ROWS * COLS * AWPU_ELEMENTS];
    static float frame[AWPU_ELEMENTS * AWPU_HIST], power[ROWS * COLS], tau[AWPU_ELEMENTS];
    static uint8_t image[ROWS * COLS];

    /* geometry and tables: create_antenna (antenna.cpp:60-87), computeDelayLUT (mimo.cpp:20-59) */
    awpu_hip_create_antenna(8, 8, 0.02f, xyz);
    awpu_hip_build_delay_table(xyz, AWPU_ELEMENTS, ROWS, COLS, 180.0f, 0, ROWS, off, frac);

    /* a 9 kHz plane wave from theta = 20 deg, phi = 35 deg, as the reference's synthetic producer makes it */
    const double theta = 20.0 * M_PI / 180.0, phi = 35.0 * M_PI / 180.0;
    awpu_hip_steering_delays(xyz, AWPU_ELEMENTS, theta, phi, tau);
    for (int s = 0; s < AWPU_ELEMENTS; s++)
        for (int i = 0; i < AWPU_HIST; i++)
            frame[s * AWPU_HIST + i] = (float) (1e-2 * sin(2.0 * M_PI * 9e3 * (i + tau[s]) / 48828.0));

    awpu_hip_cfg cfg;
    awpu_hip_default_cfg(&cfg);
    cfg.n_streams = AWPU_ELEMENTS;
    cfg.n_pixels = ROWS * COLS;
    cfg.lut_stride = AWPU_ELEMENTS;
    cfg.grid_columns = COLS;
    awpu_hip_t *engine = NULL;
    int rc = awpu_hip_create(&engine, &cfg);
    if (rc != AWPU_OK) {
        fprintf(stderr, "awpu_hip_create: %s (%s)\n", awpu_hip_strerror(rc), awpu_hip_last_error());
        return rc == AWPU_ERR_NO_DEVICE ? 2 : 1;
    }
    rc = awpu_hip_set_delay_table(engine, off, frac);
    if (rc == AWPU_OK) rc = awpu_hip_set_active_mics(engine, NULL, AWPU_ELEMENTS); /* all 64 mics usable */
    if (rc == AWPU_OK) rc = awpu_hip_process(engine, frame, 1, power);               /* MIMOWorker::update */
    if (rc == AWPU_OK) rc = awpu_hip_heatmap_u8(power, ROWS * COLS, image);          /* populateHeatmap */
    if (rc != AWPU_OK) {
        fprintf(stderr, "sweep failed: %s (%s)\n", awpu_hip_strerror(rc), awpu_hip_last_error());
        awpu_hip_destroy(engine);
        return 1;
    }
    int peak = 0;
    for (int p = 1; p < ROWS * COLS; p++)
        if (power[p] > power[peak]) peak = p;
    printf("peak pixel (%d,%d) power %.3e, image value there %d\n", peak / COLS, peak % COLS, power[peak], image[peak]);
    awpu_hip_destroy(engine);
    return 0;
}
