/*
 * ref_mimo_driver.cpp -- drives the REFERENCE's own delay() (compiled in place
 * from /root/reference/src/dsp/delay.cpp by oracle/Makefile) through a
 * restatement of the loop nest of MIMOWorker::update, src/dsp/mimo.cpp:121-151.
 *
 * TEST INFRASTRUCTURE ONLY (see das_oracle.h).  mimo.cpp itself cannot be
 * compiled here (Eigen, OpenCV absent), so the loop nest is restated; the
 * inner kernel -- >95 % of the frame time and all of the interpolation
 * arithmetic -- is the reference object code.
 *
 * Output goes to oracle/_ref/ only (git-ignored; travels to the GPU box).
 */
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#ifndef N_SAMPLES
#define N_SAMPLES 256
#endif

/* the reference symbol, src/dsp/delay.h:31 */
void delay(float *out, const float *signal, const float fraction);

extern "C" {

/* 1 = AVX2 linear interpolation (delay.cpp:16-26), 2 = 8-tap FIR (delay.cpp:31-40) */
int ref_variant(void) {
#if defined(__AVX2__)
    return 1;
#else
    return 2;
#endif
}

void ref_delay(float *out, const float *signal, float fraction) { delay(out, signal, fraction); }

/* src/dsp/mimo.cpp:121-151 around the reference delay(). */
void ref_das(const float *X, int hist, const int32_t *off, const float *frac, int P, int lut_stride,
             const int32_t *index, int usable, float *power, float *out_dbg) {
    for (int m = 0; m < P; m++) {
        float out[N_SAMPLES] = {0.0};
        int count = 0;
        for (int s = 0; s < usable; s++) {
            int i = index[s];
            float fraction = frac[(size_t) m * lut_stride + i];
            int offset = off[(size_t) m * lut_stride + i];
            delay(&out[0], &X[(size_t) i * hist + offset], fraction);
            count++;
        }
        if (out_dbg) memcpy(out_dbg + (size_t) m * N_SAMPLES, out, sizeof(out));
        float p = 0.0;
        for (int i = 1; i < N_SAMPLES - 1; i++) {
            float MA = out[i] * 0.5f - 0.25f * (out[i + 1] + out[i - 1]);
            p += MA * MA;
        }
        p /= static_cast<float>(N_SAMPLES * count);
        power[m] = p;
    }
}

/* The same loop nest with the pixels dealt to `threads` host threads (contiguous slabs): NOT something the
 * reference does -- its MIMO worker is one thread (src/dsp/mimo.cpp:12) -- but the honest "what could the
 * host do" figure next to the one-thread number.  Whole frames until `min_seconds` have passed. */
double ref_das_bench_mt(const float *X, int hist, const int32_t *off, const float *frac, int P, int lut_stride,
                        const int32_t *index, int usable, float *power, double min_seconds, int threads,
                        int *frames_done) {
    using clk = std::chrono::steady_clock;
    if (threads < 1) threads = 1;
    const auto t0 = clk::now();
    int frames = 0;
    double el = 0.0;
    do {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) {
            const int a = (int) ((long long) P * t / threads), b = (int) ((long long) P * (t + 1) / threads);
            pool.emplace_back([=] {
                ref_das(X, hist, off + (size_t) a * lut_stride, frac + (size_t) a * lut_stride, b - a, lut_stride, index,
                        usable, power + a, nullptr);
            });
        }
        for (auto &th : pool) th.join();
        frames++;
        el = std::chrono::duration<double>(clk::now() - t0).count();
    } while (el < min_seconds);
    if (frames_done) *frames_done = frames;
    return frames / el;
}

/* src/dsp/particle.cpp:51-82 / :88-103 (Particle::beam, Particle::das) around the reference delay() */
void ref_particle_beams(const float *X, int hist, const int32_t *off, const float *frac, int n_dir, int lut_stride,
                        const int32_t *index, int usable, float *power, float *beams) {
    for (int m = 0; m < n_dir; m++) {
        float out[N_SAMPLES] = {0.0};
        for (int s = 0; s < usable; s++) {
            int i = index[s];
            delay(&out[0], &X[(size_t) i * hist + off[(size_t) m * lut_stride + i]], frac[(size_t) m * lut_stride + i]);
        }
        float power_accumulator = 0.0f;
        for (int i = 1; i < N_SAMPLES - 1; i++) {
            float MA = out[i] * 0.5f - 0.25f * (out[i + 1] + out[i - 1]);
            power_accumulator += MA * MA;
        }
        power_accumulator /= static_cast<float>(N_SAMPLES);
        if (power) power[m] = power_accumulator;
        if (beams) memcpy(beams + (size_t) m * N_SAMPLES, out, sizeof(out));
    }
}

/* frames/s of ref_das on one thread: runs whole frames until `min_seconds`
 * elapsed (at least one), returns frames / seconds. */
double ref_das_bench(const float *X, int hist, const int32_t *off, const float *frac, int P,
                     int lut_stride, const int32_t *index, int usable, float *power,
                     double min_seconds, int *frames_done) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    int frames = 0;
    double el = 0.0;
    do {
        ref_das(X, hist, off, frac, P, lut_stride, index, usable, power, nullptr);
        frames++;
        el = std::chrono::duration<double>(clk::now() - t0).count();
    } while (el < min_seconds);
    if (frames_done) *frames_done = frames;
    return frames / el;
}
}
