// mimo_worker_hip.cpp -- see mimo_worker_hip.h.  Every hot-path operation goes through the C ABI;
// there is no CPU implementation of the sweep here.
#include "mimo_worker_hip.h"

#include <cstdio>

namespace awpu_host {

MIMOWorkerHip::MIMOWorkerHip(FrameSource *pipeline, const AntennaView &antenna, bool *running, int rows,
                             int columns, float fov, int device, bool autostart, int math, const std::vector<int> &devices)
    : pipeline(pipeline), antenna(antenna), running(running), rows(rows), columns(columns), fov(fov) {
    maxIndex = rows * columns;                       // mimo.cpp:8
    powerdB = std::vector<float>(maxIndex, 0.0f);    // mimo.cpp:10
    const int n_sensors = pipeline->get_n_sensors();
    signals.resize((size_t) n_sensors * AWPU_HIST);

    awpu_hip_cfg cfg;
    awpu_hip_default_cfg(&cfg);
    cfg.device = device;
    cfg.n_streams = n_sensors;
    cfg.n_pixels = maxIndex;
    cfg.lut_stride = antenna.n;
    cfg.grid_columns = columns;  // lets the batched sweep pair vertically adjacent pixels (results unchanged)
    cfg.math = math;
    if (devices.size() > 1 && devices.size() <= AWPU_MAX_DEVICES) {  // a device group: rows of the grid per GPU
        cfg.n_devices = (int32_t) devices.size();
        for (size_t k = 0; k < devices.size(); k++) cfg.devices[k] = devices[k];
    }
    last_status = awpu_hip_create(&engine, &cfg);
    if (last_status != AWPU_OK) {
        // the reference reports on std::cerr and carries on (pipeline.cpp:33-36); same here
        std::fprintf(stderr, "MIMOWorkerHip: %s (%s)\n", awpu_hip_strerror(last_status), awpu_hip_last_error());
        return;
    }
    computeDelayLUT();  // mimo.cpp:11
    if (last_status == AWPU_OK) last_status = awpu_hip_set_active_mics(engine, antenna.index, antenna.usable);
    if (last_status == AWPU_OK && pipeline->feeds_device_ring()) {
        from_ring = true;
        pipeline->attach(engine, &lock);  // blocks arrive in the engine's ring from now on
    }
    if (autostart) thread_loop = std::thread(&MIMOWorkerHip::loop, this);  // mimo.cpp:12
}

MIMOWorkerHip::~MIMOWorkerHip() {
    // Order matters: the source's producer thread ingests into `engine` under `lock`.  Cut that link first
    // (detach waits for an ingest in flight), then stop our own loop, and only then free the engine.
    if (from_ring && engine) pipeline->detach(engine);
    looping = false;
    if (thread_loop.joinable()) thread_loop.join();
    std::lock_guard<std::mutex> guard(lock);
    awpu_hip_destroy(engine);
    engine = nullptr;
}

// mimo.cpp:20-59: the tables are built on the host exactly as the reference does (one-off), kept in
// the same members, and handed to the engine.
void MIMOWorkerHip::computeDelayLUT() {
    offsetDelays.assign((size_t) maxIndex * antenna.n, 0);
    fractionalDelays.assign((size_t) maxIndex * antenna.n, 0.f);
    last_status = awpu_hip_build_delay_table(antenna.points, antenna.n, rows, columns, fov, 0, rows,
                                             offsetDelays.data(), fractionalDelays.data());
    if (last_status == AWPU_OK)
        last_status = awpu_hip_set_delay_table(engine, offsetDelays.data(), fractionalDelays.data());
}

int MIMOWorkerHip::setDelayLUT(const int32_t *off, const float *frac) {
    if (!engine || !off || !frac) return AWPU_ERR_INVALID;
    std::lock_guard<std::mutex> guard(lock);
    offsetDelays.assign(off, off + (size_t) maxIndex * antenna.n);
    fractionalDelays.assign(frac, frac + (size_t) maxIndex * antenna.n);
    last_status = awpu_hip_set_delay_table(engine, offsetDelays.data(), fractionalDelays.data());
    return last_status;
}

// mimo.cpp:97-151.  The snapshot loop is the reference's (every stream, so that antenna.index can
// address any of them); the pixel x mic x sample sweep and the epilogue run on the GPU.
void MIMOWorkerHip::update() {
    if (!engine) return;
    if (from_ring) {  // the snapshot is already in device memory (awpu_hip_ingest_block)
        last_status = awpu_hip_process_ring(engine, powerdB.data());
        return;
    }
    const int n_sensors = pipeline->get_n_sensors();
    for (int l = 0; l < n_sensors; l++) {
        pipeline->read_stream((unsigned) l, &signals[(size_t) l * AWPU_HIST]);  // mimo.cpp:100-103
    }
    last_status = awpu_hip_process(engine, signals.data(), 1, powerdB.data());  // mimo.cpp:121-151
}

// mimo.cpp:61-95 with USE_DB 0
void MIMOWorkerHip::populateHeatmap(uint8_t *heatmap) { awpu_hip_heatmap_u8(powerdB.data(), maxIndex, heatmap); }

void MIMOWorkerHip::draw(uint8_t *heatmap) {  // worker.h:148-152
    lock.lock();
    populateHeatmap(heatmap);
    lock.unlock();
}

void MIMOWorkerHip::loop() {  // worker.h:212-224
    while (looping && pipeline->isRunning()) {
        pipeline->barrier();
        lock.lock();
        update();
        lock.unlock();
    }
}

}  // namespace awpu_host
