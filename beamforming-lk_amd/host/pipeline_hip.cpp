// pipeline_hip.cpp -- see pipeline_hip.h.
#include "pipeline_hip.h"

#include <arpa/inet.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

namespace awpu_host {

int init_receiver(const char *address, const int port) {
    int socket_desc = socket(AF_INET, SOCK_DGRAM, IPPROTO_UDP);
    if (socket_desc < 0) {
        std::fprintf(stderr, "Error creating socket\n");
        return -1;
    }
    // an exposure is 264 KB: leave room for a few so that a busy consumer does not drop datagrams
    int rcvbuf = 8 << 20;
    (void) setsockopt(socket_desc, SOL_SOCKET, SO_RCVBUF, &rcvbuf, sizeof(rcvbuf));
    struct timeval tv = {0, 200000};  // so that disconnect() can stop a producer that hears nothing
    (void) setsockopt(socket_desc, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    struct sockaddr_in server_addr;
    std::memset(&server_addr, 0, sizeof(server_addr));
    server_addr.sin_family = AF_INET;
    server_addr.sin_port = htons((uint16_t) port);
    server_addr.sin_addr.s_addr = inet_addr(address);
    if (bind(socket_desc, (struct sockaddr *) &server_addr, sizeof(server_addr)) < 0) {
        std::fprintf(stderr, "Couldn't bind socket to the port\n");
        close(socket_desc);
        return -1;
    }
    return socket_desc;
}

int receive_message(int socket_desc, WireMessage *msg) {
    return recv(socket_desc, msg, sizeof(WireMessage), 0) == (ssize_t) sizeof(WireMessage) ? 0 : -1;
}

PipelineHip::PipelineHip(const char *address, const int port, bool verbose)
    : address(address), port(port), verbose(verbose), exposure(AWPU_N_SAMPLES) {}

PipelineHip::~PipelineHip() {
    if (connected) disconnect();
}

int PipelineHip::connect() {
    if (connected) {
        std::fprintf(stderr, "Beamformer is already connected\n");
        return -1;
    }
    socket_desc = init_receiver(address, port);
    if (socket_desc == -1) {
        std::fprintf(stderr, "Unable to establish a connection to antenna\n");
        return -1;
    }
    // pipeline.cpp:62-71: the first datagram tells how many arrays are on the wire
    int tries = 50;  // 10 s of the receive timeout
    while (receive_message(socket_desc, &first) < 0)
        if (--tries == 0) {
            std::fprintf(stderr, "Unable to receive message\n");
            close(socket_desc);
            socket_desc = -1;
            return -1;
        }
    have_first = true;
    n_sensors = first.n_arrays * AWPU_ELEMENTS;
    if (n_sensors < AWPU_ELEMENTS || n_sensors > 256) {
        std::fprintf(stderr, "Unsupported number of arrays on the wire: %d\n", (int) first.n_arrays);
        close(socket_desc);
        socket_desc = -1;
        return -1;
    }
    ring.assign((size_t) n_sensors * AWPU_HIST, 0.0f);
    if (verbose) std::printf("Connected: %d arrays, %d sensors\n", (int) first.n_arrays, n_sensors);
    connected = 1;
    receiver_thread = std::thread(&PipelineHip::producer, this);
    return 0;
}

int PipelineHip::disconnect() {
    if (!connected) {
        std::fprintf(stderr, "Beamformer is not connected\n");
        return -1;
    }
    {
        std::unique_lock<std::mutex> lock(pool_mutex);
        connected = 0;
    }
    receiver_thread.join();
    release_barrier();
    close(socket_desc);
    socket_desc = -1;
    return 0;
}

int PipelineHip::isRunning() {
    std::unique_lock<std::mutex> lock(pool_mutex);
    return connected;
}

int PipelineHip::mostRecent() {
    std::unique_lock<std::mutex> lock(barrier_mutex);
    return modified;
}

void PipelineHip::barrier() {
    std::unique_lock<std::mutex> lock(barrier_mutex);
    barrier_count++;
    barrier_condition.wait(lock, [&] { return barrier_count == 0; });
}

void PipelineHip::release_barrier() {
    std::unique_lock<std::mutex> lock(barrier_mutex);
    barrier_count = 0;
    modified++;
    barrier_condition.notify_all();
}

void PipelineHip::attach(awpu_hip_t *engine, std::mutex *guard) {
    std::unique_lock<std::mutex> lock(ring_mutex);
    engines.push_back(Attached{engine, guard});
}

void PipelineHip::detach(awpu_hip_t *engine) {
    // receive_exposure() holds ring_mutex for its whole ingest loop, so once we have it no ingest into
    // `engine` is running and none will start
    std::unique_lock<std::mutex> lock(ring_mutex);
    for (auto it = engines.begin(); it != engines.end();)
        it = it->engine == engine ? engines.erase(it) : it + 1;
}

void PipelineHip::producer() {
    while (isRunning()) {
        receive_exposure();
        release_barrier();
    }
}

void PipelineHip::receive_exposure() {
    for (int i = 0; i < AWPU_N_SAMPLES; i++) {
        if (have_first) {  // the datagram connect() looked at is the first sample of the first exposure
            exposure[i] = first;
            have_first = false;
            continue;
        }
        while (receive_message(socket_desc, &exposure[i]) < 0)
            if (!isRunning()) return;  // timeout while shutting down
    }
    std::unique_lock<std::mutex> lock(ring_mutex);
    // host copy: pipeline.cpp:277-290 (flip every other group of 8 columns, normalise by 2^23), then
    // Streams::write_stream + forward (streams.hpp:103-105,136-139): the block replaces the oldest 256
    for (int sensor_index = 0; sensor_index < n_sensors; sensor_index++) {
        const bool inverted = ((sensor_index / 8) % 2) == 0;
        const int index = inverted ? 8 * (1 + sensor_index / 8) - 1 - sensor_index % 8 : sensor_index;
        float *row = &ring[(size_t) sensor_index * AWPU_HIST];
        for (int i = 0; i < AWPU_N_SAMPLES; i++)
            row[(position + i) % AWPU_HIST] = static_cast<float>(exposure[i].stream[index]) / 8388608.0f;
    }
    position = (position + AWPU_N_SAMPLES) % AWPU_HIST;
    // device rings: the raw datagrams, unpacked on the GPU
    for (auto &a : engines) {
        if (a.guard) a.guard->lock();
        const int rc = awpu_hip_ingest_block(a.engine, exposure.data(), (int32_t) sizeof(WireMessage));
        if (a.guard) a.guard->unlock();
        if (rc != AWPU_OK) status = rc;
    }
}

void PipelineHip::read_stream(unsigned index, float *data) {
    std::unique_lock<std::mutex> lock(ring_mutex);
    const float *row = &ring[(size_t) index * AWPU_HIST];
    for (int i = 0; i < AWPU_HIST; i++) data[i] = row[(position + i) % AWPU_HIST];
}

}  // namespace awpu_host
