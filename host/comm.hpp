// comm.hpp -- the exchange steps of the sharded job behind one small interface, for a C++ host.
// The single-process reference has no counterpart (image_stitching.cpp main() is one process); what shards here is
//   features  -> all-gather of {counts, keypoints, descriptors}          (before :653, the matcher needs every frame's features)
//   pairs     -> sum of the n x n confidence matrix (disjoint supports)  (before :661, the pruning)
//   blend     -> all-to-all of pyramid rectangles by column strips, all-gather of the finished strips  (:1218, :1225)
// Two implementations:
//   RcclComm  RCCL called directly (rccl.h): ncclAllGather, grouped ncclSend / ncclRecv, ncclAllReduce, all enqueued on the HIP
//             stream the caller names (the job's main / compose streams): one process per GPU over xGMI, no torch;
//   HostComm  the same exchanges staged through POSIX shared memory (device -> host -> shared file -> host -> device): for
//             rehearsals of the N > 1 path with several processes on ONE GPU (RCCL refuses two ranks on one device).
// Both rendezvous through a small control file in /dev/shm named by `session` (created by the launcher before the ranks start).
#pragma once
#include <cstddef>
#include <memory>
#include <string>

namespace mis {

class Communicator {
public:
    virtual ~Communicator() {}
    virtual int rank() const = 0;
    virtual int world() const = 0;
    virtual const char* name() const = 0;
    // Device buffers, `stream` = the hipStream_t whose earlier work produced `send` and whose later work reads `recv`.
    // recv holds world * bytes: rank r's block at r * bytes.
    virtual void all_gather(const void* send, void* recv, size_t bytes, void* stream) = 0;
    // send[k] (send_bytes[k]) goes to rank k; recv[r] (recv_bytes[r]) comes from rank r; sizes are known to both sides.
    virtual void all_to_all(const void* const* send, const size_t* send_bytes, void* const* recv, const size_t* recv_bytes, void* stream) = 0;
    // small host-side records (feature counts, the confidence matrix): blocking
    virtual void all_gather_host(const void* send, void* recv, size_t bytes) = 0;
    virtual void all_reduce_sum_host(double* v, size_t n) = 0;      // sums in rank order
    virtual void barrier() = 0;
};

// `session`: name of the control file under /dev/shm (comm_session_create by the launcher, comm_session_destroy when the ranks are gone)
void comm_session_create(const std::string& session, int world);
void comm_session_destroy(const std::string& session, int world);
std::unique_ptr<Communicator> make_host_comm(const std::string& session, int rank, int world);
std::unique_ptr<Communicator> make_rccl_comm(const std::string& session, int rank, int world);      // the current HIP device is the rank's GPU
std::unique_ptr<Communicator> make_solo_comm();      // world = 1 without any machinery (identity collectives)

}  // namespace mis
