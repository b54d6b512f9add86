// aw_processing_unit_hip.cpp -- see the header.  Order of operations follows
// src/aw_processing_unit/aw_processing_unit.cpp.
#include "aw_processing_unit_hip.h"

#include <cstdio>

namespace awpu_host {

AWProcessingUnitHip::AWProcessingUnitHip(FrameSource *pipeline, float fov, int small_res, int verbose,
                                         bool use_audio, int device)
    : fov(fov), small_res(small_res), verbose(verbose), device(device), pipeline(pipeline) {
    (void) use_audio;
    setupAntennas();  // .cpp:28
    calibrate();      // .cpp:34
}

AWProcessingUnitHip::~AWProcessingUnitHip() {
    pause();
    if (verbose) std::printf("Destructing AWPU\n");
    workers.clear();  // joins the worker threads (worker.h:104-107)
}

void AWProcessingUnitHip::setupAntennas() {  // .cpp:58-65: one identical 8x8 array per 64 sensors
    const int n_antennas = pipeline->get_n_sensors() / AWPU_ELEMENTS;
    antennas.clear();
    for (int a = 0; a < n_antennas; a++) {
        AntennaState ant;
        ant.points.resize(3 * AWPU_ELEMENTS);
        awpu_hip_create_antenna(8, 8, 0.02f, ant.points.data());
        antennas.push_back(std::move(ant));
    }
}

void AWProcessingUnitHip::calibrate(const float reference_power_level) {  // .cpp:102-212
    for (int i = 0; i < AWPU_HIST / AWPU_N_SAMPLES; i++) pipeline->barrier();  // wait for full buffers
    std::vector<float> signals((size_t) AWPU_ELEMENTS * AWPU_HIST);
    for (size_t a = 0; a < antennas.size(); a++) {
        for (int s = 0; s < AWPU_ELEMENTS; s++)
            pipeline->read_stream((unsigned) (s + a * AWPU_ELEMENTS), &signals[(size_t) s * AWPU_HIST]);
        AntennaState &ant = antennas[a];
        ant.index.assign(AWPU_ELEMENTS, 0);
        ant.power_correction_mask.assign(AWPU_ELEMENTS, 0.f);
        ant.usable = calibrate_array(signals.data(), AWPU_HIST, reference_power_level, ant.index.data(),
                                     ant.power_correction_mask.data(), &ant.median);
        ant.index.resize(ant.usable);
        ant.power_correction_mask.resize(ant.usable);
        if (verbose)
            std::printf("Calibrated antenna %zu Usable: %d Median: %g\n", a, ant.usable, (double) ant.median);
    }
}

bool AWProcessingUnitHip::start(const worker_t worker) {  // .cpp:67-95
    if (worker != MIMO || antennas.empty() || antennas[0].usable == 0) return false;
    // the reference beamforms antennas[0] only (.cpp:74)
    AntennaView view{antennas[0].points.data(), AWPU_ELEMENTS, antennas[0].usable, antennas[0].index.data()};
    auto job = std::make_unique<MIMOWorkerHip>(pipeline, view, &running, small_res, small_res, fov, device);
    if (job->status() != AWPU_OK) return false;
    workers.push_back(std::move(job));
    return true;
}

void AWProcessingUnitHip::steer(Spherical direction) { (void) direction; }

bool AWProcessingUnitHip::stop(const worker_t worker) {
    for (auto it = workers.begin(); it != workers.end(); ++it) {
        if ((*it)->get_type() == worker) {
            if (verbose) std::printf("Stopping worker from AWPU Workers\n");
            workers.erase(it);  // destroys (joins) the worker, then removes it
            return true;
        }
    }
    return false;
}

void AWProcessingUnitHip::pause() { running = false; }
void AWProcessingUnitHip::resume() { running = true; }

void AWProcessingUnitHip::draw_heatmap(uint8_t *heatmap) const {
    if (!workers.empty()) workers[0]->draw(heatmap);
}

int AWProcessingUnitHip::status() const { return workers.empty() ? AWPU_ERR_STATE : workers[0]->status(); }

void AWProcessingUnitHip::draw(uint8_t *compact, uint8_t *normal, int normal_res) const {
    for (auto &w : workers)
        if (w->get_type() == MIMO) w->draw(compact);
    // cv::resize(*compact, *normal, normal->size(), 0, 0, cv::INTER_LINEAR), .cpp:252
    if (normal && normal_res >= small_res) awpu_hip_resize_linear_u8(compact, small_res, small_res, normal, normal_res, normal_res);
}

}  // namespace awpu_host
