// geometry_host.cpp -- one-off host geometry behind the C ABI: array element positions,
// plane-wave steering delays and the (offset, fraction) tables of the MIMO sweep.
//
// Mirrors (file:line in the reference tree):
//   create_antenna               src/geometry/antenna.cpp:60-87
//   steer / rotateY / rotateZ    src/geometry/antenna.cpp:99-107, src/geometry/geometry.cpp:219-233
//   compute_delays               src/geometry/antenna.cpp:89-97
//   steering_vector_spherical    src/geometry/antenna.cpp:126-129
//   MIMOWorker::computeDelayLUT  src/dsp/mimo.cpp:20-59
// The reference does this arithmetic with Eigen (3x3 * 3xN fp32 products); here the same
// rotations are written out, keeping its precisions: the pixel grid and the angles in
// double, the rotation entries rounded to float, the delays in float.
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "awpu_hip.h"

namespace {

constexpr double kSampleRate = 48828.0;       // src/geometry/antenna.h:17
constexpr double kPropagationSpeed = 340.0;   // src/geometry/antenna.h:16
constexpr int kArrayColumns = 8, kArrayRows = 8;  // src/geometry/antenna.h:18-19

struct Rot {
    float m[3][3];
};

Rot rotate_z(float angle) {  // geometry.cpp:219-225
    const float c = static_cast<float>(std::cos(static_cast<double>(angle)));
    const float s = static_cast<float>(std::sin(static_cast<double>(angle)));
    return Rot{{{c, -s, 0.0f}, {s, c, 0.0f}, {0.0f, 0.0f, 1.0f}}};
}

Rot rotate_y(float angle) {  // geometry.cpp:227-233
    const float c = static_cast<float>(std::cos(static_cast<double>(angle)));
    const float s = static_cast<float>(std::sin(static_cast<double>(angle)));
    return Rot{{{c, 0.0f, s}, {0.0f, 1.0f, 0.0f}, {-s, 0.0f, c}}};
}

// z row of Ry(-theta) * (Rz(phi) * points), scaled to samples, minimum removed.
void steering_delays(const float *xyz, int n, double theta, double phi, float *tau) {
    const Rot rz = rotate_z(static_cast<float>(phi));
    const Rot ry = rotate_y(-static_cast<float>(theta));
    const float scale = static_cast<float>(kSampleRate / kPropagationSpeed);
    float lowest = std::numeric_limits<float>::infinity();
    for (int i = 0; i < n; i++) {
        const float p[3] = {xyz[i], xyz[n + i], xyz[2 * n + i]};
        float t[3];
        for (int r = 0; r < 3; r++) t[r] = rz.m[r][0] * p[0] + rz.m[r][1] * p[1] + rz.m[r][2] * p[2];
        const float z = ry.m[2][0] * t[0] + ry.m[2][1] * t[1] + ry.m[2][2] * t[2];
        tau[i] = z * scale;
        lowest = std::fmin(lowest, tau[i]);
    }
    for (int i = 0; i < n; i++) tau[i] -= lowest;
}

void element_position(int r, int c, int rows, int columns, float distance, float *x, float *y) {
    // antenna.cpp:66-73 -- x is centred with `rows`, y with `columns`, as in the reference
    const float half = distance / 2;
    *x = static_cast<float>(c) * distance - static_cast<float>(rows) * half + half;
    *y = static_cast<float>(r) * distance - static_cast<float>(columns) * half + half;
}

}  // namespace

namespace awpu {

// The per-pixel half of computeDelayLUT (mimo.cpp:21-43) for the device builder: for every pixel of grid rows
// [row_begin, row_begin + row_count) the twelve floats steering_delays() would build from (theta, phi) -- Rz((float) phi)
// row-major, then row z of Ry(-(float) theta).  Same expressions, same libm as awpu_hip_build_delay_table.
void pixel_rotations(int rows, int columns, float fov_deg, int row_begin, int row_count, float *rot) {
    const double fov = static_cast<double>(fov_deg) * (M_PI / 180.0);
    const double sep_rows = std::sin(fov / 2.0) / (static_cast<double>(rows) / 2.0);
    const double sep_cols = std::sin(fov / 2.0) / (static_cast<double>(columns) / 2.0);
    size_t k = 0;
    for (int r = row_begin; r < row_begin + row_count; r++) {
        for (int c = 0; c < columns; c++, k++) {
            double y = r * sep_rows - rows * sep_rows / 2.0 + sep_rows / 2.0;
            double x = c * sep_cols - columns * sep_cols / 2.0 + sep_cols / 2.0;
            double norm = std::sqrt(x * x + y * y);
            x /= norm;
            y /= norm;
            if (norm > 1.0) norm = 1.0;
            const double theta = std::asin(norm);
            const double phi = std::atan2(y, x);
            const Rot rz = rotate_z(static_cast<float>(phi));
            const Rot ry = rotate_y(-static_cast<float>(theta));
            float *m = rot + k * 12;
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) m[3 * a + b] = rz.m[a][b];
            for (int b = 0; b < 3; b++) m[9 + b] = ry.m[2][b];
        }
    }
}

float samples_per_metre() { return static_cast<float>(kSampleRate / kPropagationSpeed); }

}  // namespace awpu

extern "C" {

int awpu_hip_create_antenna(int32_t columns, int32_t rows, float distance, float *xyz) {
    if (columns <= 0 || rows <= 0 || !xyz) return AWPU_ERR_INVALID;
    const int n = rows * columns;
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < columns; c++) {
            const int i = r * columns + c;
            element_position(r, c, rows, columns, distance, &xyz[i], &xyz[n + i]);
            xyz[2 * n + i] = 0.0f;
        }
    }
    return AWPU_OK;
}

int awpu_hip_create_tiled_antenna(int32_t arrays_x, int32_t arrays_y, float distance, float *xyz) {
    if (arrays_x <= 0 || arrays_y <= 0 || !xyz) return AWPU_ERR_INVALID;
    const int columns = arrays_x * kArrayColumns, rows = arrays_y * kArrayRows;
    const int n = rows * columns;
    for (int a = 0; a < arrays_x * arrays_y; a++) {
        const int ax = a % arrays_x, ay = a / arrays_x;
        for (int s = 0; s < AWPU_ELEMENTS; s++) {
            const int r = ay * kArrayRows + s / kArrayColumns;
            const int c = ax * kArrayColumns + s % kArrayColumns;
            const int i = a * AWPU_ELEMENTS + s;  // stream id, aw_processing_unit.cpp:120
            element_position(r, c, rows, columns, distance, &xyz[i], &xyz[n + i]);
            xyz[2 * n + i] = 0.0f;
        }
    }
    return AWPU_OK;
}

int awpu_hip_steering_delays(const float *xyz, int32_t n, double theta, double phi, float *tau) {
    if (!xyz || !tau || n <= 0) return AWPU_ERR_INVALID;
    steering_delays(xyz, n, theta, phi, tau);
    return AWPU_OK;
}

int awpu_hip_steer_table(const float *xyz, int32_t n, const double *theta, const double *phi, int32_t n_dir,
                         int32_t *off, float *frac) {
    if (!xyz || !theta || !phi || !off || !frac || n <= 0 || n_dir <= 0) return AWPU_ERR_INVALID;
    std::vector<float> tau(n);
    for (int d = 0; d < n_dir; d++) {
        steering_delays(xyz, n, theta[d], phi[d], tau.data());
        for (int i = 0; i < n; i++) {  // particle.cpp:39-47, the same split as mimo.cpp:46-54
            double whole;
            frac[(size_t) d * n + i] = static_cast<float>(std::modf(static_cast<double>(tau[i]), &whole));
            off[(size_t) d * n + i] = AWPU_N_SAMPLES - static_cast<int>(whole);
        }
    }
    return AWPU_OK;
}

int awpu_hip_build_delay_table(const float *xyz, int32_t n, int32_t rows, int32_t columns,
                               float fov_deg, int32_t row_begin, int32_t row_count, int32_t *off,
                               float *frac) {
    if (!xyz || !off || !frac || n <= 0 || rows <= 0 || columns <= 0) return AWPU_ERR_INVALID;
    if (row_begin < 0 || row_count < 0 || row_begin + row_count > rows) return AWPU_ERR_INVALID;
    // mimo.cpp:21-24: sine-space grid, everything in double
    const double fov = static_cast<double>(fov_deg) * (M_PI / 180.0);
    const double sep_rows = std::sin(fov / 2.0) / (static_cast<double>(rows) / 2.0);
    const double sep_cols = std::sin(fov / 2.0) / (static_cast<double>(columns) / 2.0);
    std::vector<float> tau(n);
    size_t k = 0;
    for (int r = row_begin; r < row_begin + row_count; r++) {
        for (int c = 0; c < columns; c++, k++) {
            double y = r * sep_rows - rows * sep_rows / 2.0 + sep_rows / 2.0;      // mimo.cpp:34
            double x = c * sep_cols - columns * sep_cols / 2.0 + sep_cols / 2.0;  // mimo.cpp:35
            double norm = std::sqrt(x * x + y * y);
            x /= norm;
            y /= norm;
            if (norm > 1.0) norm = 1.0;
            const double theta = std::asin(norm);  // mimo.cpp:41
            const double phi = std::atan2(y, x);   // mimo.cpp:43
            steering_delays(xyz, n, theta, phi, tau.data());
            for (int i = 0; i < n; i++) {  // mimo.cpp:46-54
                double whole;
                frac[k * n + i] = static_cast<float>(std::modf(static_cast<double>(tau[i]), &whole));
                off[k * n + i] = AWPU_N_SAMPLES - static_cast<int>(whole);
            }
        }
    }
    return AWPU_OK;
}

int awpu_hip_heatmap_u8(const float *power, int32_t n, uint8_t *pix) {
    if (!power || !pix || n <= 0) return AWPU_ERR_INVALID;
    // mimo.cpp:62-73: running maximum starting at 0
    float max_v = 0.0f;
    for (int i = 0; i < n; i++) max_v = power[i] > max_v ? power[i] : max_v;
    for (int i = 0; i < n; i++) {  // mimo.cpp:85-91 with USE_DB 0
        double level = static_cast<double>(power[i] / max_v) * 255.0;
        // An all-zero frame gives 0/0 = NaN here; the reference then casts NaN to uchar (undefined, 0 on
        // x86).  Defined here: a level that is not a number is level 0.
        level = !(level >= 0.0) ? 0.0 : (level > 255.0 ? 255.0 : level);
        pix[i] = static_cast<uint8_t>(level);
    }
    return AWPU_OK;
}

}  // extern "C"
