// test_mimo_worker.cpp -- exercises the C++ host mirror (beamforming-lk_amd/host) the way the
// reference's own process would: a producer publishes 256-sample blocks into per-mic rings, the MIMO
// worker snapshots them and updates its heatmap.  Results are checked against the CPU oracle
// (tests may link oracle/; the product never does).
//
//   test_mimo_worker          full run (needs an MI355X)
//   test_mimo_worker --nogpu  only checks that, without a device, the worker reports the failure
//                             and computes nothing (there is no CPU path)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "das_oracle.h"
#include "aw_processing_unit_hip.h"
#include "mimo_worker_hip.h"
#include "pipeline_hip.h"

#include <arpa/inet.h>
#include <sys/socket.h>
#include <unistd.h>

using namespace awpu_host;

static int failures = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                     \
            std::printf("\n");                            \
            failures++;                                   \
        }                                                 \
    } while (0)

// Per-mic rings with the reference's semantics (src/fpga/streams.hpp:103-116,136-139): 4 blocks of
// 256 floats; the producer writes at `position` then forwards it; a read starts at the new position,
// so a snapshot is oldest..newest.  Signal: the synthetic producer of src/fpga/pipeline.cpp:105-135,
// a 9 kHz plane wave of amplitude 1e-2, here from an arbitrary direction.
class SyntheticSource : public FrameSource {
public:
    SyntheticSource(const float *xyz, int n, double theta, double phi) : n(n), ring((size_t) n * 1024, 0.f), tau(n) {
        oracle_steering_delays_f32(xyz, n, theta, phi, tau.data());
    }
    int get_n_sensors() override { return n; }
    int isRunning() override { return running.load(); }
    void barrier() override {
        std::unique_lock<std::mutex> lk(m);
        const int seen = published;
        cv.wait(lk, [&] { return published != seen || !running.load(); });
    }
    void read_stream(unsigned index, float *data) override {
        std::lock_guard<std::mutex> lk(m);
        for (int i = 0; i < 1024; i++) data[i] = ring[(size_t) index * 1024 + (position + i) % 1024];
    }
    void publish_block() {  // Pipeline::synthetic_producer, pipeline.cpp:119-147
        std::lock_guard<std::mutex> lk(m);
        for (int s = 0; s < n; s++)
            for (int i = 0; i < 256; i++) {
                const double t = (double) (p + i) + tau[s];
                ring[(size_t) s * 1024 + (position + i) % 1024] = (float) (1e-2 * std::sin(2.0 * M_PI * 9e3 * t / 48828.0));
            }
        p += 256;
        position = (position + 256) % 1024;  // Streams::forward
        published++;
        cv.notify_all();
    }
    void stop() {
        running = false;
        cv.notify_all();
    }

private:
    int n;
    std::vector<float> ring;
    std::vector<float> tau;
    int position = 0, p = 0, published = 0;
    std::atomic<int> running{1};
    std::mutex m;
    std::condition_variable cv;
};

static double rel_err(const std::vector<float> &got, const std::vector<float> &want) {
    float peak = 0;
    for (float v : want) peak = std::max(peak, v);
    double worst = 0;
    for (size_t i = 0; i < got.size(); i++) {
        const double den = std::max((double) want[i], 1e-4 * peak);
        worst = std::max(worst, std::fabs((double) got[i] - want[i]) / den);
    }
    return worst;
}

int main(int argc, char **argv) {
    const bool nogpu = argc > 1 && std::strcmp(argv[1], "--nogpu") == 0;
    const int rows = 24, cols = 24;
    std::vector<float> xyz(3 * 64);
    awpu_hip_create_antenna(8, 8, 0.02f, xyz.data());
    std::vector<int> all(64);
    for (int i = 0; i < 64; i++) all[i] = i;
    const double theta = 0.35, phi = -2.0;
    bool run = true;

    if (nogpu) {
        SyntheticSource src(xyz.data(), 64, theta, phi);
        AntennaView ant{xyz.data(), 64, 64, all.data()};
        MIMOWorkerHip w(&src, ant, &run, rows, cols, 180.f, 0, /*autostart=*/false);
        CHECK(w.status() == AWPU_ERR_NO_DEVICE, "expected AWPU_ERR_NO_DEVICE, got %d", w.status());
        w.update();
        float sum = 0;
        for (float v : w.power()) sum += v;
        CHECK(sum == 0.0f, "power must stay untouched without a device");
        std::printf(failures ? "FAILED\n" : "OK nogpu\n");
        return failures ? 1 : 0;
    }

    // ---- 1. stepped worker, all 64 mics, against the oracle on the same snapshot and tables
    {
        SyntheticSource src(xyz.data(), 64, theta, phi);
        AntennaView ant{xyz.data(), 64, 64, all.data()};
        MIMOWorkerHip w(&src, ant, &run, rows, cols, 180.f, 0, false);
        CHECK(w.status() == AWPU_OK, "create: %s", awpu_hip_last_error());
        for (int b = 0; b < 5; b++) src.publish_block();
        w.update();
        CHECK(w.status() == AWPU_OK, "update: %s", awpu_hip_last_error());
        std::vector<float> snap((size_t) 64 * 1024);
        for (int s = 0; s < 64; s++) src.read_stream(s, &snap[(size_t) s * 1024]);
        std::vector<int32_t> off((size_t) rows * cols * 64);
        std::vector<float> frac(off.size()), want(rows * cols);
        oracle_compute_delay_lut(xyz.data(), 64, rows, cols, 180.f, off.data(), frac.data());
        CHECK(off == w.offsets() && frac == w.fractions(), "tables differ from computeDelayLUT restatement");
        oracle_das_f32(snap.data(), 1024, off.data(), frac.data(), rows * cols, 64, all.data(), 64, want.data(), nullptr);
        const double e = rel_err(w.power(), want);
        CHECK(e < 1e-5, "power rel err %.3e", e);
        const int k = (int) (std::max_element(w.power().begin(), w.power().end()) - w.power().begin());
        const double sep = 1.0 / (rows / 2.0);
        const int er = (int) std::lround(std::sin(theta) * std::sin(phi) / sep + rows / 2.0 - 0.5);
        const int ec = (int) std::lround(std::sin(theta) * std::cos(phi) / sep + cols / 2.0 - 0.5);
        CHECK(std::abs(k / cols - er) <= 1 && std::abs(k % cols - ec) <= 1, "peak at (%d,%d), source at (%d,%d)", k / cols, k % cols, er, ec);
        std::vector<uint8_t> img(rows * cols), img_want(rows * cols);
        w.draw(img.data());
        oracle_heatmap_u8(w.power().data(), rows * cols, img_want.data());
        CHECK(img == img_want && img[k] == 255, "heatmap differs from populateHeatmap restatement");
        std::printf("1. stepped worker: rel err %.2e, peak (%d,%d)\n", e, k / cols, k % cols);
    }

    // ---- 2. calibration drops a dead and a loud mic; the worker runs on the usable list
    {
        SyntheticSource src(xyz.data(), 64, theta, phi);
        for (int b = 0; b < 4; b++) src.publish_block();
        std::vector<float> snap((size_t) 64 * 1024);
        for (int s = 0; s < 64; s++) src.read_stream(s, &snap[(size_t) s * 1024]);
        std::fill(snap.begin() + 9 * 1024, snap.begin() + 10 * 1024, 0.f);  // dead mic 9
        for (int i = 0; i < 1024; i++) snap[(size_t) 40 * 1024 + i] *= 3.0f;  // loud mic 40
        int index[64], index_o[64];
        float corr[64], corr_o[64], med = 0, med_o = 0;
        int32_t usable = 0;
        {   // the C ABI's calibration (device mean squares) on the host snapshot, through an engine of its own
            awpu_hip_cfg cfg;
            awpu_hip_default_cfg(&cfg);
            cfg.n_pixels = 1;
            awpu_hip_t *cal = nullptr;
            CHECK(awpu_hip_create(&cal, &cfg) == AWPU_OK, "create: %s", awpu_hip_last_error());
            CHECK(awpu_hip_calibrate_host(cal, snap.data(), 0, 1e-5f, index, corr, &med, &usable) == AWPU_OK,
                  "calibrate_host: %s", awpu_hip_last_error_of(cal));
            awpu_hip_destroy(cal);
        }
        const int usable_o = oracle_calibrate(snap.data(), 1024, 1e-5f, index_o, corr_o, &med_o);
        CHECK(usable == usable_o && usable == 62 && std::equal(index, index + usable, index_o) && med == med_o,
              "calibrate: usable %d vs %d", usable, usable_o);
        AntennaView ant{xyz.data(), 64, usable, index};
        MIMOWorkerHip w(&src, ant, &run, rows, cols, 180.f, 0, false);
        w.update();
        for (int s = 0; s < 64; s++) src.read_stream(s, &snap[(size_t) s * 1024]);
        std::vector<float> want(rows * cols);
        oracle_das_f32(snap.data(), 1024, w.offsets().data(), w.fractions().data(), rows * cols, 64, index, usable, want.data(), nullptr);
        const double e = rel_err(w.power(), want);
        CHECK(e < 1e-5, "usable-subset power rel err %.3e", e);
        std::printf("2. calibrated subset (%d mics): rel err %.2e\n", usable, e);
    }

    // ---- 3. threaded like the reference: producer -> barrier -> update under the lock; draw() from here
    {
        SyntheticSource src(xyz.data(), 64, theta, phi);
        AntennaView ant{xyz.data(), 64, 64, all.data()};
        for (int b = 0; b < 4; b++) src.publish_block();
        std::vector<uint8_t> img(rows * cols, 0);
        {
            MIMOWorkerHip w(&src, ant, &run, rows, cols, 180.f, 0, /*autostart=*/true);
            for (int b = 0; b < 6; b++) {
                src.publish_block();
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
                w.draw(img.data());
            }
            src.stop();
        }
        const int k = (int) (std::max_element(img.begin(), img.end()) - img.begin());
        CHECK(img[k] == 255, "threaded worker produced no heatmap");
        std::printf("3. threaded worker: heatmap peak at (%d,%d)\n", k / cols, k % cols);
    }
    // ---- 4. the processing unit as AWControlUnit drives it: construct (calibrates), start(MIMO), draw, stop
    {
        SyntheticSource src(xyz.data(), 64, theta, phi);
        std::thread producer([&] {
            for (int b = 0; b < 40 && src.isRunning(); b++) {
                src.publish_block();
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
        });
        AWProcessingUnitHip awpu(&src, 180.f, rows, /*verbose=*/0);
        CHECK(awpu.n_antennas() == 1 && awpu.usable() == 64, "calibrate kept %d mics", awpu.usable());
        CHECK(!awpu.start(GRADIENT), "only the MIMO worker is on this path");
        CHECK(awpu.start(MIMO), "start(MIMO): %s", awpu_hip_last_error());
        awpu.resume();
        std::this_thread::sleep_for(std::chrono::milliseconds(80));
        std::vector<uint8_t> small(rows * cols, 0), big(96 * 96, 0);
        awpu.draw(small.data(), big.data(), 96);
        const int k = (int) (std::max_element(small.begin(), small.end()) - small.begin());
        const int kb = (int) (std::max_element(big.begin(), big.end()) - big.begin());
        CHECK(small[k] == 255 && awpu.status() == AWPU_OK, "no heatmap from the processing unit");
        CHECK(std::abs(kb / 96 - (k / cols) * 4) <= 6 && std::abs(kb % 96 - (k % cols) * 4) <= 6, "upscaled peak moved");
        CHECK(awpu.targets().empty(), "MIMO has no targets");
        CHECK(awpu.stop(MIMO) && !awpu.stop(MIMO), "stop(MIMO) once");
        src.stop();
        producer.join();
        std::printf("4. processing unit: heatmap peak at (%d,%d), upscaled (%d,%d)\n", k / cols, k % cols, kb / 96, kb % 96);
    }
    // ---- 5. the live path: wire datagrams over UDP (loopback) -> PipelineHip -> device ring -> sweep.
    // The sender plays the FPGA / udpreplay: one datagram per sample time, 24-bit samples of the plane
    // wave in wire order (src/fpga/receiver.h:24-30; every other group of 8 columns mirrored).
    {
        const int port = 21000 + (int) (getpid() % 4000);
        std::vector<float> tau(64);
        oracle_steering_delays_f32(xyz.data(), 64, theta, phi, tau.data());
        std::atomic<bool> sending{true};
        std::thread sender([&] {
            const int tx = socket(AF_INET, SOCK_DGRAM, IPPROTO_UDP);
            struct sockaddr_in to;
            std::memset(&to, 0, sizeof(to));
            to.sin_family = AF_INET;
            to.sin_port = htons((uint16_t) port);
            to.sin_addr.s_addr = inet_addr("127.0.0.1");
            WireMessage msg{};
            msg.frequency = 48828;
            msg.n_arrays = 1;
            msg.version = 2;
            for (uint32_t p = 0; sending.load(); p++) {
                msg.counter = p;
                for (int sensor = 0; sensor < 64; sensor++) {
                    const bool inverted = ((sensor / 8) % 2) == 0;
                    const int wire = inverted ? 8 * (1 + sensor / 8) - 1 - sensor % 8 : sensor;
                    const double v = 1e-2 * std::sin(2.0 * M_PI * 9e3 * ((double) p + tau[sensor]) / 48828.0);
                    msg.stream[wire] = (int32_t) std::lround(v * 8388608.0);
                }
                (void) sendto(tx, &msg, sizeof(msg), 0, (struct sockaddr *) &to, sizeof(to));
                if (p % 64 == 63) std::this_thread::sleep_for(std::chrono::microseconds(700));  // ~2x real time
            }
            close(tx);
        });
        PipelineHip pipeline("127.0.0.1", port);
        CHECK(pipeline.connect() == 0, "connect");
        CHECK(pipeline.get_n_sensors() == 64, "n_sensors %d", pipeline.get_n_sensors());
        {
            AWProcessingUnitHip awpu(&pipeline, 180.f, rows, /*verbose=*/0);  // calibrates on a device ring of its own
            CHECK(awpu.calibration_status() == AWPU_OK, "calibrate: %d", awpu.calibration_status());
            CHECK(awpu.usable() == 64, "calibrate kept %d mics", awpu.usable());
            CHECK(awpu.start(MIMO), "start(MIMO): %s", awpu_hip_last_error());
            awpu.resume();
            const int seen = pipeline.mostRecent();
            for (int spin = 0; spin < 400 && pipeline.mostRecent() < seen + 6; spin++)
                std::this_thread::sleep_for(std::chrono::milliseconds(5));  // a full ring of blocks after the attach
            CHECK(pipeline.mostRecent() >= seen + 6, "no blocks arrived");
            std::this_thread::sleep_for(std::chrono::milliseconds(30));
            std::vector<uint8_t> img(rows * cols, 0);
            awpu.draw_heatmap(img.data());
            const int k = (int) (std::max_element(img.begin(), img.end()) - img.begin());
            CHECK(awpu.status() == AWPU_OK && pipeline.last_status() == AWPU_OK, "status %d / %d: %s", awpu.status(),
                  pipeline.last_status(), awpu_hip_last_error());
            CHECK(img[k] == 255 && k / cols == 8 && k % cols == 10, "live heatmap peak at (%d,%d), expected (8,10)", k / cols, k % cols);
            std::printf("5. UDP -> device ring -> sweep: heatmap peak at (%d,%d) after %d blocks\n", k / cols, k % cols,
                        pipeline.mostRecent());
            // stop the worker while blocks are still arriving: the pipeline must let go of the engine before it
            // is destroyed (the producer ingests into it from its own thread), and keep running afterwards
            const int before = pipeline.mostRecent();
            CHECK(awpu.stop(MIMO) && !awpu.stop(MIMO), "stop(MIMO) once, while the sender is running");
            for (int spin = 0; spin < 400 && pipeline.mostRecent() < before + 4; spin++)
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            CHECK(pipeline.mostRecent() >= before + 4 && pipeline.last_status() == AWPU_OK,
                  "the pipeline stalled after the worker was stopped (status %d)", pipeline.last_status());
            // and a second worker can be started on the same pipeline
            CHECK(awpu.start(MIMO), "restart(MIMO): %s", awpu_hip_last_error());
            std::this_thread::sleep_for(std::chrono::milliseconds(60));
            CHECK(awpu.status() == AWPU_OK, "restarted worker status %d", awpu.status());
        }  // ~AWProcessingUnitHip with the sender still running: same hand-over in the destructor
        sending = false;
        sender.join();
        CHECK(pipeline.disconnect() == 0, "disconnect");
    }
    // ---- 6. a device group behind the same worker: the grid's rows over two engines (both on this box's one GPU)
    {
        SyntheticSource src(xyz.data(), 64, theta, phi);
        AntennaView ant{xyz.data(), 64, 64, all.data()};
        for (int b = 0; b < 5; b++) src.publish_block();
        MIMOWorkerHip one(&src, ant, &run, rows, cols, 180.f, 0, false);
        MIMOWorkerHip two(&src, ant, &run, rows, cols, 180.f, 0, false, AWPU_MATH_F32_EXACT, {0, 0});
        CHECK(two.status() == AWPU_OK, "group create: %s", awpu_hip_last_error());
        one.update();
        two.update();
        CHECK(two.status() == AWPU_OK && one.power() == two.power(), "device group differs from the single device");
        std::printf("6. device group {0, 0}: identical heatmap\n");
    }
    // ---- 7. caller-built tables (the drop-in keeps the reference's own computeDelayLUT and hands its bits over): a table that differs
    // from the Eigen-free restatement -- every tau one ulp down, an integer boundary crossed where tau sat on it (mimo.cpp:46-54) --
    // goes in through setDelayLUT, and the worker's heatmap equals the oracle's ON THAT TABLE within 1e-5 on every pixel
    {
        SyntheticSource src(xyz.data(), 64, theta, phi);
        AntennaView ant{xyz.data(), 64, 64, all.data()};
        for (int b = 0; b < 5; b++) src.publish_block();
        MIMOWorkerHip w(&src, ant, &run, rows, cols, 180.f, 0, false);
        std::vector<int32_t> off = w.offsets();
        std::vector<float> frac = w.fractions();
        for (size_t i = 0; i < off.size(); i++) {  // tau = (256 - off) + frac, one ulp of the fraction down
            if (frac[i] > 0.0f) frac[i] = std::nextafter(frac[i], 0.0f);
            else if (off[i] < 256) { off[i] += 1; frac[i] = std::nextafter(1.0f, 0.0f); }
        }
        CHECK(w.setDelayLUT(off.data(), frac.data()) == AWPU_OK, "setDelayLUT: %s", awpu_hip_last_error());
        CHECK(off == w.offsets() && frac == w.fractions(), "the worker does not hold the caller's tables");
        w.update();
        std::vector<float> snap((size_t) 64 * 1024), want(rows * cols);
        for (int s = 0; s < 64; s++) src.read_stream(s, &snap[(size_t) s * 1024]);
        oracle_das_f32(snap.data(), 1024, off.data(), frac.data(), rows * cols, 64, all.data(), 64, want.data(), nullptr);
        const double e = rel_err(w.power(), want);
        CHECK(w.status() == AWPU_OK && e < 1e-5, "caller-built tables: power rel err %.3e (status %d)", e, w.status());
        std::printf("7. caller-built tables through setDelayLUT: rel err %.2e\n", e);
    }
    std::printf(failures ? "FAILED\n" : "OK\n");
    return failures ? 1 : 0;
}
