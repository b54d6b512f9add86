// mimo_worker_hip.h -- C++ host mirror of the reference's MIMO worker and processing unit for the
// heatmap path, on top of the C ABI (include/awpu_hip.h).  This is the code a maintainer of
// acoustic-warfare/beamforming-lk would drop in place of src/dsp/mimo.{h,cpp}: same constructor
// shape, same update()/draw() roles, same state (offsetDelays, fractionalDelays, powerdB), with the
// sweep of MIMOWorker::update (src/dsp/mimo.cpp:97-151) replaced by one awpu_hip_process call.
//
// The reference classes depend on Eigen (Antenna::points), OpenCV (cv::Mat) and its Pipeline; none
// of those is available to this build, so the three touch points are narrowed to plain interfaces:
//   FrameSource  what MIMOWorker uses of Pipeline/Streams   (src/fpga/pipeline.h:40-107,
//                                                            src/fpga/streams.hpp:113-116)
//   AntennaView  what it uses of Antenna                    (src/geometry/antenna.h:80-103)
//   uint8_t*     the rows x columns CV_8UC1 heatmap         (src/dsp/mimo.cpp:61-95)
#pragma once

#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

#include "awpu_hip.h"

namespace awpu_host {

// src/dsp/worker.h:66-73
enum worker_t { GENERIC, PSO, MIMO, MISO, SOUND, GRADIENT };

// The slice of Pipeline + Streams the MIMO worker touches.
class FrameSource {
public:
    virtual ~FrameSource() = default;
    virtual int get_n_sensors() = 0;                          // pipeline.h:107
    virtual int isRunning() = 0;                              // pipeline.h:71
    virtual void barrier() = 0;                               // pipeline.h:83, blocks until a new block
    virtual void read_stream(unsigned index, float *data) = 0;  // streams.hpp:113-116: 1024 floats, oldest first
    // A source that can deliver its blocks straight into an engine's device ring (PipelineHip) says so;
    // the worker then attaches its engine and sweeps with awpu_hip_process_ring, no host snapshot.
    virtual bool feeds_device_ring() { return false; }
    virtual void attach(awpu_hip_t *engine, std::mutex *guard) {
        (void) engine;
        (void) guard;
    }
    // The engine is about to be destroyed: after detach() returns the source touches neither the engine
    // nor its guard again (an ingest in flight finishes first).
    virtual void detach(awpu_hip_t *engine) { (void) engine; }
};

// src/geometry/antenna.h:80-103 without Eigen: points is xyz[3][n] row-major by coordinate.
struct AntennaView {
    const float *points = nullptr;
    int n = 0;
    int usable = 0;              // number of usable elements
    const int *index = nullptr;  // [usable] physical ids, as AWProcessingUnit::calibrate fills them
};

// Mirrors class MIMOWorker : public Worker  (src/dsp/mimo.h:25-92, src/dsp/worker.h:81-233).
class MIMOWorkerHip {
public:
    // src/dsp/mimo.h:36.  `autostart` = the reference starts its thread in the constructor
    // (mimo.cpp:12); tests pass false and call update() themselves.
    // `devices` (optional): spread the grid's rows over these GPUs of the node (one engine handle, one slab per
    // device: awpu_hip_cfg.n_devices); empty = the single `device`.
    MIMOWorkerHip(FrameSource *pipeline, const AntennaView &antenna, bool *running, int rows, int columns,
                  float fov, int device = 0, bool autostart = true, int math = AWPU_MATH_F32_EXACT,
                  const std::vector<int> &devices = {});
    ~MIMOWorkerHip();  // worker.h:104-107: looping = false; join

    worker_t get_type() { return worker_t::MIMO; }  // mimo.h:47-49

    // Worker::draw (worker.h:148-152): populateHeatmap under the worker lock.
    void draw(uint8_t *heatmap);

    // mimo.cpp:97-151.  Protected in the reference; public here so tests can step it.
    void update();

    // The tables of a caller that keeps the reference's own computeDelayLUT (mimo.cpp:20-59, Eigen arithmetic): off / frac
    // [rows * columns][antenna.n], pixel-major as mimo.h:86-88 flattens.  They replace the tables the constructor built with the
    // Eigen-free restatement (awpu_hip_build_delay_table, whose last ulp of tau is unpinned): with the reference's own bits in, the
    // heatmap is within 1e-5 of the reference's on every pixel, deep nulls included (INTEGRATION.md "Which math mode").
    // Call between blocks (it takes the worker lock).  Returns the C-ABI status.
    int setDelayLUT(const int32_t *off, const float *frac);

    int status() const { return last_status; }                 // last C-ABI status (0 = OK)
    const std::vector<float> &power() const { return powerdB; }  // mimo.h:91
    const std::vector<int32_t> &offsets() const { return offsetDelays; }
    const std::vector<float> &fractions() const { return fractionalDelays; }

private:
    void computeDelayLUT();               // mimo.cpp:20-59
    void populateHeatmap(uint8_t *heatmap);  // mimo.cpp:61-95
    void loop();                          // worker.h:212-224

    FrameSource *pipeline;
    AntennaView antenna;
    bool *running;
    bool looping = true;
    bool from_ring = false;  // the source fills the engine's device ring
    std::thread thread_loop;
    std::mutex lock;

    const int rows, columns;
    const float fov;
    int maxIndex;
    int last_status = AWPU_OK;

    std::vector<int32_t> offsetDelays;    // [maxIndex][n]  (vector<vector<int>> in mimo.h:86)
    std::vector<float> fractionalDelays;  // [maxIndex][n]  (mimo.h:88)
    std::vector<float> powerdB;           // mimo.h:91
    std::vector<float> signals;           // the snapshot update() takes: [n_sensors][1024] (mimo.cpp:100-103)
    awpu_hip_t *engine = nullptr;
};

}  // namespace awpu_host
