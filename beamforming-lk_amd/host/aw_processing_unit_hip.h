// aw_processing_unit_hip.h -- C++ mirror of the reference's AWProcessingUnit
// (src/aw_processing_unit/aw_processing_unit.h:26-151, .cpp) for the heatmap path: the class its
// callers hold (AWControlUnit::Start, aw_control_unit.cpp:206-213,300,436-438; awpu_test main).
// Same methods, same order of operations (connect -> setupAntennas -> calibrate -> start(MIMO)), with
// the MIMO worker running on the GPU through the C ABI.
//
// Narrowed types (none of Eigen / OpenCV / PortAudio is available to this build):
//   Pipeline*            -> awpu_host::FrameSource*      (the caller keeps ownership, unlike the
//                                                          reference, which deletes its pipeline)
//   cv::Mat* heatmaps    -> uint8_t* images (rows*cols, CV_8UC1 layout)
//   Spherical, Target    -> plain structs below
// Not on this path and therefore inert here: audio (play_audio/stop_audio), steer() (MIMO ignores
// it, worker.h:157), trackers (start(GRADIENT/MISO/PSO) returns false), targets() (MIMO has none).
#pragma once

#include <memory>
#include <vector>

#include "mimo_worker_hip.h"

namespace awpu_host {

struct Spherical {  // src/geometry/geometry.h
    double theta = 0.0, phi = 0.0, radius = 1.0;
};

struct Target {  // src/dsp/worker.h:36-61
    Spherical direction;
    float power = 0.f, probability = 0.f;
};

class AWProcessingUnitHip {
public:
    // aw_processing_unit.h:45: AWProcessingUnit(Pipeline *pipeline, int verbose = 1, bool use_audio = false)
    // (+ the grid parameters of the address/port constructor, :37: fov = FOV, small_res = MIMO_SIZE)
    // `devices` (optional, not in the reference): the GPUs the MIMO worker spreads its grid over
    AWProcessingUnitHip(FrameSource *pipeline, float fov = 180.0f, int small_res = 256, int verbose = 1,
                        bool use_audio = false, int device = 0, std::vector<int> devices = {});
    ~AWProcessingUnitHip();

    void setupAntennas();                 // .cpp:58-65
    bool start(const worker_t worker);    // .cpp:67-95
    void steer(Spherical direction);      // forwarded to workers; MIMO ignores it
    bool stop(const worker_t worker);     // .cpp:214-232 (without the reference's erase-then-delete bug)
    void pause();                         // .cpp:234-236
    void resume();                        // .cpp:238-240
    void draw_heatmap(uint8_t *heatmap) const;  // .cpp:242-244, image small_res x small_res
    void play_audio() {}                  // audio is out of scope
    void stop_audio() {}
    void calibrate(const float reference_power_level = 1e-5);  // .cpp:102-212
    std::vector<Target> targets() { return {}; }               // MIMO tracks nothing
    // .cpp:245-259: compact = the MIMO heatmap; normal = compact resized (cv::resize INTER_LINEAR
    // arithmetic) to normal_res^2, normal_res >= small_res
    void draw(uint8_t *compact, uint8_t *normal, int normal_res) const;
    void draw(uint8_t *compact, uint8_t *normal, int normal_rows, int normal_cols) const;  // a non-square big image
    void set_devices(std::vector<int> list) { devices = std::move(list); }  // GPUs of the next start(MIMO)

    int n_antennas() const { return (int) antennas.size(); }
    int usable(int a = 0) const { return antennas[a].usable; }
    const std::vector<int> &index(int a = 0) const { return antennas[a].index; }
    int status() const;
    int calibration_status() const { return calibrate_status; }

protected:
    struct AntennaState {  // src/geometry/antenna.h:80-103
        std::vector<float> points;  // xyz[3][64]
        int usable = 0;
        std::vector<int> index;
        std::vector<float> power_correction_mask;
        float mean = 0.f, median = 0.f;
    };
    float fov;
    int small_res;
    int verbose;
    int device;
    std::vector<int> devices;
    std::vector<std::unique_ptr<MIMOWorkerHip>> workers;
    FrameSource *pipeline;
    bool running = false;
    int calibrate_status = AWPU_OK;  // last C-ABI status of calibrate()
    std::vector<AntennaState> antennas;
};

}  // namespace awpu_host
