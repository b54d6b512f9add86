// aw_processing_unit.cpp -- see aw_processing_unit.h.  Order of operations as in the reference's
// src/aw_processing_unit/aw_processing_unit.cpp:10-55.
#ifdef AWPU_WITH_OPENCV
#include "aw_processing_unit.h"

#include <cstdio>

namespace {

// the rows x cols CV_8UC1 image the reference's workers draw into, as one contiguous byte buffer
uint8_t *bytes_of(cv::Mat *m, int rows, int cols) {
    if (m->rows != rows || m->cols != cols || m->type() != CV_8UC1 || !m->isContinuous()) m->create(rows, cols, CV_8UC1);
    return m->data;
}

}  // namespace

AWProcessingUnit::AWProcessingUnit(const char *address, const int port, float fov, int small_res, int verbose, bool use_audio)
    : fov(fov), small_res(small_res), verbose(verbose) {
    pipeline = new Pipeline(address, port);  // .cpp:13-14
    pipeline->connect();
    // setupAntennas() and calibrate() run in the mirror's constructor (.cpp:16-23)
    unit = std::make_unique<awpu_host::AWProcessingUnitHip>(pipeline, fov, small_res, verbose, use_audio, 0);
}

AWProcessingUnit::AWProcessingUnit(Pipeline *pipeline, int verbose, bool use_audio)
    : fov((float) FOV), small_res(MIMO_SIZE), verbose(verbose), pipeline(pipeline) {
    this->pipeline->connect();  // .cpp:26
    unit = std::make_unique<awpu_host::AWProcessingUnitHip>(pipeline, fov, small_res, verbose, use_audio, 0);
}

AWProcessingUnit::~AWProcessingUnit() {  // .cpp:37-55
    pause();
    if (verbose) std::printf("Destructing AWPU\n");
    unit.reset();  // the workers let go of the pipeline (detach) before it goes away
    pipeline->disconnect();
    delete pipeline;
}

void AWProcessingUnit::setupAntennas() { unit->setupAntennas(); }

bool AWProcessingUnit::start(const worker_t worker) {
    if (!devices.empty()) unit->set_devices(devices);
    return unit->start(worker);
}

void AWProcessingUnit::steer(Spherical direction) { unit->steer(direction); }
bool AWProcessingUnit::stop(const worker_t worker) { return unit->stop(worker); }
void AWProcessingUnit::pause() { unit->pause(); }
void AWProcessingUnit::resume() { unit->resume(); }

void AWProcessingUnit::draw_heatmap(cv::Mat *heatmap) const {  // .cpp:242-244
    unit->draw_heatmap(bytes_of(heatmap, small_res, small_res));
}

void AWProcessingUnit::play_audio() { unit->play_audio(); }
void AWProcessingUnit::stop_audio() { unit->stop_audio(); }
void AWProcessingUnit::calibrate(const float reference_power_level) { unit->calibrate(reference_power_level); }
void AWProcessingUnit::synthetic_calibration() {}  // declared and never defined in the reference (aw_processing_unit.h:112)
std::vector<Target> AWProcessingUnit::targets() { return unit->targets(); }

// .cpp:245-259: compact = the MIMO heatmap, normal = compact resized to normal's size (cv::INTER_LINEAR arithmetic)
void AWProcessingUnit::draw(cv::Mat *compact, cv::Mat *normal) const {
    const int out_rows = normal->rows > 0 ? normal->rows : small_res, out_cols = normal->cols > 0 ? normal->cols : small_res;
    uint8_t *small = bytes_of(compact, small_res, small_res);
    uint8_t *big = bytes_of(normal, out_rows, out_cols);
    unit->draw(small, big, out_rows, out_cols);
}

#endif  // AWPU_WITH_OPENCV
