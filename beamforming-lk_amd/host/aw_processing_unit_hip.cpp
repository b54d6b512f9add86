// aw_processing_unit_hip.cpp -- see the header.  Order of operations follows
// src/aw_processing_unit/aw_processing_unit.cpp.
#include "aw_processing_unit_hip.h"

#include <cstdio>
#include <mutex>

namespace awpu_host {

AWProcessingUnitHip::AWProcessingUnitHip(FrameSource *pipeline, float fov, int small_res, int verbose,
                                         bool use_audio, int device, std::vector<int> devices)
    : fov(fov), small_res(small_res), verbose(verbose), device(device), devices(std::move(devices)), pipeline(pipeline) {
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
    // The per-mic mean squares are computed on the device (awpu_hip_calibrate_ring / _host) by a
    // short-lived engine of this unit's own: a pipeline that feeds device rings fills that engine's ring
    // while we wait for full buffers; any other source hands over the snapshot read_stream assembles.
    const int n_sensors = pipeline->get_n_sensors();
    awpu_hip_cfg cfg;
    awpu_hip_default_cfg(&cfg);
    cfg.device = device;
    cfg.n_streams = n_sensors;
    cfg.n_pixels = 1;  // calibration sweeps nothing
    awpu_hip_t *engine = nullptr;
    calibrate_status = awpu_hip_create(&engine, &cfg);
    if (calibrate_status != AWPU_OK) {
        std::fprintf(stderr, "AWProcessingUnitHip::calibrate: %s (%s)\n", awpu_hip_strerror(calibrate_status),
                     awpu_hip_last_error());
        return;  // no device, no calibration: every antenna stays at usable = 0 and start() refuses
    }
    std::mutex guard;
    const bool from_ring = pipeline->feeds_device_ring();
    if (from_ring) pipeline->attach(engine, &guard);
    for (int i = 0; i < AWPU_HIST / AWPU_N_SAMPLES; i++) pipeline->barrier();  // wait for full buffers (.cpp:106-109)
    std::vector<float> signals;
    if (!from_ring) {
        signals.resize((size_t) n_sensors * AWPU_HIST);
        for (int s = 0; s < n_sensors; s++) pipeline->read_stream((unsigned) s, &signals[(size_t) s * AWPU_HIST]);
    }
    for (size_t a = 0; a < antennas.size(); a++) {
        AntennaState &ant = antennas[a];
        ant.index.assign(AWPU_ELEMENTS, 0);
        ant.power_correction_mask.assign(AWPU_ELEMENTS, 0.f);
        int32_t usable = 0;
        {
            std::lock_guard<std::mutex> hold(guard);  // the producer ingests into this engine from its own thread
            calibrate_status = from_ring
                ? awpu_hip_calibrate_ring(engine, (int32_t) a, reference_power_level, ant.index.data(),
                                          ant.power_correction_mask.data(), &ant.median, &usable)
                : awpu_hip_calibrate_host(engine, signals.data(), (int32_t) a, reference_power_level, ant.index.data(),
                                          ant.power_correction_mask.data(), &ant.median, &usable);
        }
        if (calibrate_status != AWPU_OK) usable = 0;
        ant.usable = usable;
        ant.index.resize(ant.usable);
        ant.power_correction_mask.resize(ant.usable);
        if (verbose)
            std::printf("Calibrated antenna %zu Usable: %d Median: %g\n", a, ant.usable, (double) ant.median);
    }
    if (from_ring) pipeline->detach(engine);
    awpu_hip_destroy(engine);
}

bool AWProcessingUnitHip::start(const worker_t worker) {  // .cpp:67-95
    if (worker != MIMO || antennas.empty() || antennas[0].usable == 0) return false;
    // the reference beamforms antennas[0] only (.cpp:74)
    AntennaView view{antennas[0].points.data(), AWPU_ELEMENTS, antennas[0].usable, antennas[0].index.data()};
    auto job = std::make_unique<MIMOWorkerHip>(pipeline, view, &running, small_res, small_res, fov, device,
                                               /*autostart=*/true, AWPU_MATH_F32_EXACT, devices);
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
    draw(compact, normal, normal_res, normal_res);
}

void AWProcessingUnitHip::draw(uint8_t *compact, uint8_t *normal, int normal_rows, int normal_cols) const {
    for (auto &w : workers)
        if (w->get_type() == MIMO) w->draw(compact);
    // cv::resize(*compact, *normal, normal->size(), 0, 0, cv::INTER_LINEAR), .cpp:252
    if (normal && normal_rows >= small_res && normal_cols >= small_res)
        awpu_hip_resize_linear_u8(compact, small_res, small_res, normal, normal_rows, normal_cols);
}

}  // namespace awpu_host
