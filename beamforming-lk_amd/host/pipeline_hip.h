// pipeline_hip.h -- C++ mirror of the reference's receive side for the heatmap path: Pipeline
// (src/fpga/pipeline.h:40-125, pipeline.cpp:38-103,160-297) + the UDP receiver (src/fpga/receiver.{h,cpp}),
// i.e. what stands between the FPGA (or udpreplay) and MIMOWorker::update.  Same roles: connect() binds
// the socket and learns the number of arrays from the first datagram, a producer thread receives one
// exposure (256 datagrams) at a time and releases the workers' barrier.
//
// What differs: an exposure is kept as the 256 raw datagrams and handed, unconverted, to every attached
// engine (awpu_hip_ingest_block: the int32 -> float conversion, the daisy-chain column flip and the ring
// write happen on the GPU, include/awpu_hip.h), so a worker sweeps with awpu_hip_process_ring and no
// sample crosses the host twice.  A host copy of the ring (the reference's Streams, streams.hpp:103-139,
// without the double page mapping) is kept as well: calibration and host-snapshot workers read it.
#pragma once

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "mimo_worker_hip.h"

namespace awpu_host {

#pragma pack(push, 1)
struct WireMessage {  // src/fpga/receiver.h:24-30
    uint16_t frequency;
    uint8_t n_arrays;
    uint8_t version;
    uint32_t counter;
    int32_t stream[256];
};
#pragma pack(pop)
static_assert(sizeof(WireMessage) == AWPU_DATAGRAM_BYTES, "wire format");

int init_receiver(const char *address, const int port);   // receiver.cpp:29-52
int receive_message(int socket_desc, WireMessage *msg);   // receiver.cpp:54-61

class PipelineHip : public FrameSource {
public:
    PipelineHip(const char *address, const int port, bool verbose = false);  // pipeline.h:42
    ~PipelineHip() override;

    int connect();     // pipeline.cpp:38-103 (UDP branch): bind, first datagram -> n_sensors, start the producer
    int disconnect();  // pipeline.cpp:162-188
    int isRunning() override;
    int mostRecent();  // pipeline.cpp:201-204
    void barrier() override;
    int get_n_sensors() override { return n_sensors; }
    void read_stream(unsigned index, float *data) override;  // streams.hpp:113-116 on the host copy

    // every exposure from now on also goes, raw, into this engine's device ring; `guard` (may be null) is
    // held around the call -- the engine handle is not thread-safe and its worker sweeps from another thread
    void attach(awpu_hip_t *engine, std::mutex *guard) override;
    void detach(awpu_hip_t *engine) override;
    bool feeds_device_ring() override { return true; }
    int last_status() const { return status; }

private:
    void producer();          // pipeline.cpp:246-258
    void receive_exposure();  // pipeline.cpp:260-297
    void release_barrier();   // pipeline.cpp:236-241

    const char *address;
    int port;
    bool verbose;
    int socket_desc = -1;
    int connected = 0;
    int n_sensors = 0;
    int status = AWPU_OK;
    std::thread receiver_thread;
    std::mutex pool_mutex, barrier_mutex, ring_mutex;
    std::condition_variable barrier_condition;
    int barrier_count = 0, modified = 0;

    std::vector<WireMessage> exposure;  // the 256 datagrams of the block being received
    WireMessage first{};                // the datagram connect() consumed
    bool have_first = false;
    std::vector<float> ring;            // [n_sensors][1024] host copy, position = oldest sample
    int position = 0;
    struct Attached {
        awpu_hip_t *engine;
        std::mutex *guard;
    };
    std::vector<Attached> engines;
};

}  // namespace awpu_host
