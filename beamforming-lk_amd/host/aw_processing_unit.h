// aw_processing_unit.h -- the reference's AWProcessingUnit, signature for signature
// (src/aw_processing_unit/aw_processing_unit.h:26-151), for callers that are to compile unchanged:
// AWControlUnit::Start news it with (ip, port, fov, small_res, verbose, use_audio), calls start(MIMO), draw(&small,
// &big), targets() and deletes it (src/aw_control_unit/aw_control_unit.cpp:206-213,300,348,436-438); awpu_test does
// the same (src/aw_processing_unit/main.cpp:247-253,337).
//
// Compiled only with AWPU_WITH_OPENCV defined: the cv::Mat-typed methods need OpenCV's core headers, which this
// repo's build hosts do not have (its own tests compile this file against a few-line stand-in for cv::Mat,
// tests/host/mock_opencv).  Without the macro use awpu_host::AWProcessingUnitHip (aw_processing_unit_hip.h), the same
// unit with uint8_t* images.  Everything here forwards to that class; no arithmetic lives in this file.
//
// The reference's own type names are mapped onto the mirrors of this directory:
//   Pipeline   -> awpu_host::PipelineHip   (pipeline_hip.h: UDP receiver + producer thread + device rings)
//   worker_t, Spherical, Target -> awpu_host::…
#pragma once
#ifdef AWPU_WITH_OPENCV

#include <opencv2/core.hpp>

#include <memory>
#include <vector>

#include "aw_processing_unit_hip.h"
#include "pipeline_hip.h"

#ifndef FOV
#define FOV 180.0        // aw_processing_unit.h:19
#endif
#ifndef MIMO_SIZE
#define MIMO_SIZE 256    // aw_processing_unit.h:20
#endif

using Pipeline = awpu_host::PipelineHip;
using awpu_host::worker_t;
using awpu_host::GENERIC;
using awpu_host::PSO;
using awpu_host::MIMO;
using awpu_host::MISO;
using awpu_host::SOUND;
using awpu_host::GRADIENT;
using awpu_host::Spherical;
using awpu_host::Target;

class AWProcessingUnit {
public:
    // aw_processing_unit.h:37
    AWProcessingUnit(const char *address, const int port, float fov = FOV, int small_res = MIMO_SIZE, int verbose = 1,
                     bool use_audio = false);
    // aw_processing_unit.h:45 (the reference leaves fov / small_res uninitialised here; this one takes the defaults)
    AWProcessingUnit(Pipeline *pipeline, int verbose = 1, bool use_audio = false);
    ~AWProcessingUnit();  // :50 -- disconnects and deletes the pipeline, like the reference (.cpp:37-55)

    void setupAntennas();                                       // :55
    bool start(const worker_t worker);                          // :62
    void steer(Spherical direction);                            // :68
    bool stop(const worker_t worker);                           // :75
    void pause();                                               // :80
    void resume();                                              // :85
    void draw_heatmap(cv::Mat *heatmap) const;                  // :91
    void play_audio();                                          // :96
    void stop_audio();                                          // :101
    void calibrate(const float reference_power_level = 1e-5);   // :107
    void synthetic_calibration();                               // :112 (declared, never defined, in the reference)
    std::vector<Target> targets();                              // :118
    void draw(cv::Mat *compact, cv::Mat *normal) const;         // :125

    // not in the reference: which GPU(s) the MIMO worker runs on (before start(); default device 0)
    void set_devices(std::vector<int> devices) { this->devices = std::move(devices); }
    int status() const { return unit ? unit->status() : AWPU_ERR_STATE; }

protected:
    float fov;
    int small_res;
    int verbose;
    Pipeline *pipeline;
    std::vector<int> devices;
    std::unique_ptr<awpu_host::AWProcessingUnitHip> unit;  // workers, antennas, running: the mirror's
};

#endif  // AWPU_WITH_OPENCV
