// comm.cpp -- see comm.hpp.  POSIX shared memory for the rendezvous (both implementations) and for the host-staged exchanges.
#include "comm.hpp"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace mis {
namespace {

constexpr int MAX_RANKS = 16;
constexpr double WAIT_LIMIT_S = 180.0;      // a peer that died must not leave the others spinning for ever

struct Control {        // /dev/shm/<session>.ctl, zero-filled by comm_session_create
    std::atomic<int> arrive, generation;
    std::atomic<int> id_ready;
    char nccl_id[128];
    std::atomic<unsigned long long> file_bytes[MAX_RANKS];                  // size of rank r's data file
    unsigned long long a2a_off[MAX_RANKS][MAX_RANKS], a2a_bytes[MAX_RANKS][MAX_RANKS];   // [src][dst], written by src before the barrier
};
static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId does not fit the control block");

std::string shm_path(const std::string& session, const std::string& suffix) { return "/dev/shm/" + session + suffix; }
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void fail(const std::string& what) { throw std::runtime_error("mis::Communicator: " + what); }
void hipchk(hipError_t e, const char* what) { if (e != hipSuccess) fail(std::string(what) + ": " + hipGetErrorString(e)); }
void ncclchk(ncclResult_t r, const char* what) { if (r != ncclSuccess) fail(std::string(what) + ": " + ncclGetErrorString(r)); }

struct Mapping {
    void* p = nullptr;
    size_t bytes = 0;
    void unmap() { if (p) munmap(p, bytes); p = nullptr; bytes = 0; }
};

Control* map_control(const std::string& session) {
    const int fd = open(shm_path(session, ".ctl").c_str(), O_RDWR);
    if (fd < 0) fail("no control file for session '" + session + "' (comm_session_create runs in the launcher before the ranks)");
    void* p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) fail("mmap of the control file failed");
    return static_cast<Control*>(p);
}

class ShmBase : public Communicator {
public:
    ShmBase(const std::string& session, int rank, int world) : session_(session), rank_(rank), world_(world) {
        if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) fail("bad rank / world size");
        ctl_ = map_control(session);
    }
    ~ShmBase() override {
        for (auto& m : views_) m.unmap();
        own_.unmap();
        if (ctl_) munmap(ctl_, sizeof(Control));
    }
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    void barrier() override {
        const int g = ctl_->generation.load();
        if (ctl_->arrive.fetch_add(1) + 1 == world_) { ctl_->arrive.store(0); ctl_->generation.store(g + 1); return; }
        const double t0 = now();
        for (int spins = 0; ctl_->generation.load() == g; spins++) {
            if (spins > 200) usleep(50); else sched_yield();
            if ((spins & 1023) == 1023 && now() - t0 > WAIT_LIMIT_S) fail("barrier timed out (a peer rank is gone)");
        }
    }
    // host records through the ranks' data files (both implementations use it for the small, blocking collectives)
    void all_gather_host(const void* send, void* recv, size_t bytes) override {
        std::memcpy(own(bytes), send, bytes);
        barrier();
        for (int r = 0; r < world_; r++) std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes, view(r, bytes), bytes);
        barrier();
    }
    void all_reduce_sum_host(double* v, size_t n) override {
        std::vector<double> all((size_t)world_ * n);
        all_gather_host(v, all.data(), n * sizeof(double));
        for (size_t i = 0; i < n; i++) {
            double s = 0.0;
            for (int r = 0; r < world_; r++) s += all[(size_t)r * n + i];     // rank order on every rank: identical sums
            v[i] = s;
        }
    }

protected:
    // this rank's data file, at least `bytes` long (grow-only), mapped
    char* own(size_t bytes) {
        if (own_.bytes < bytes) {
            const size_t want = std::max(bytes, own_.bytes * 2);
            own_.unmap();
            const int fd = open(shm_path(session_, ".r" + std::to_string(rank_)).c_str(), O_RDWR | O_CREAT, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)want) != 0) { if (fd >= 0) close(fd); fail("cannot size this rank's shared data file"); }
            own_.p = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (own_.p == MAP_FAILED) { own_.p = nullptr; fail("mmap of this rank's data file failed"); }
            own_.bytes = want;
            ctl_->file_bytes[rank_].store(want);
        }
        return static_cast<char*>(own_.p);
    }
    // rank r's data file (after a barrier behind r's writes), at least `bytes` long
    const char* view(int r, size_t bytes) {
        if (r == rank_) return own(bytes);
        if (views_.empty()) views_.resize(world_);
        Mapping& m = views_[r];
        const size_t have = (size_t)ctl_->file_bytes[r].load();
        if (have < bytes) fail("a peer's data file is shorter than the exchange needs");
        if (m.bytes != have) {
            m.unmap();
            const int fd = open(shm_path(session_, ".r" + std::to_string(r)).c_str(), O_RDONLY);
            if (fd < 0) fail("cannot open a peer's data file");
            m.p = mmap(nullptr, have, PROT_READ, MAP_SHARED, fd, 0);
            close(fd);
            if (m.p == MAP_FAILED) { m.p = nullptr; fail("mmap of a peer's data file failed"); }
            m.bytes = have;
        }
        return static_cast<const char*>(m.p);
    }
    std::string session_;
    int rank_, world_;
    Control* ctl_ = nullptr;
    Mapping own_;
    std::vector<Mapping> views_;
};

// ---------------------------------------------------------------- host-staged exchanges ---------
class HostComm : public ShmBase {
public:
    using ShmBase::ShmBase;
    const char* name() const override { return "host-staged (POSIX shared memory)"; }
    void all_gather(const void* send, void* recv, size_t bytes, void* stream) override {
        hipchk(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
        hipchk(hipMemcpy(own(bytes), send, bytes, hipMemcpyDeviceToHost), "hipMemcpy (device -> shared file)");
        barrier();
        for (int r = 0; r < world_; r++)
            hipchk(hipMemcpy(static_cast<char*>(recv) + (size_t)r * bytes, view(r, bytes), bytes, hipMemcpyHostToDevice), "hipMemcpy (shared file -> device)");
        barrier();
    }
    void all_to_all(const void* const* send, const size_t* send_bytes, void* const* recv, const size_t* recv_bytes, void* stream) override {
        hipchk(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
        size_t total = 0;
        for (int k = 0; k < world_; k++) { ctl_->a2a_off[rank_][k] = total; ctl_->a2a_bytes[rank_][k] = send_bytes[k]; total += (send_bytes[k] + 255) & ~(size_t)255; }
        char* base = own(std::max<size_t>(total, 256));
        for (int k = 0; k < world_; k++)
            if (send_bytes[k]) hipchk(hipMemcpy(base + ctl_->a2a_off[rank_][k], send[k], send_bytes[k], hipMemcpyDeviceToHost), "hipMemcpy (device -> shared file)");
        barrier();
        for (int r = 0; r < world_; r++) {
            if (ctl_->a2a_bytes[r][rank_] != recv_bytes[r]) fail("all_to_all: the two sides disagree about a buffer's size");
            if (recv_bytes[r])
                hipchk(hipMemcpy(recv[r], view(r, ctl_->a2a_off[r][rank_] + recv_bytes[r]) + ctl_->a2a_off[r][rank_], recv_bytes[r], hipMemcpyHostToDevice), "hipMemcpy (shared file -> device)");
        }
        barrier();
    }
};

// ---------------------------------------------------------------- RCCL --------------------------
// One communicator per process; the collectives are enqueued on the caller's stream (no host synchronisation: the next
// kernel of that stream reads what they delivered).  Rendezvous of the ncclUniqueId through the session's control file.
class RcclComm : public ShmBase {
public:
    RcclComm(const std::string& session, int rank, int world) : ShmBase(session, rank, world) {
        ncclUniqueId id;
        if (rank == 0) {
            ncclchk(ncclGetUniqueId(&id), "ncclGetUniqueId");
            std::memcpy(ctl_->nccl_id, &id, sizeof(id));
            ctl_->id_ready.store(1);
        } else {
            const double t0 = now();
            while (!ctl_->id_ready.load()) { usleep(200); if (now() - t0 > WAIT_LIMIT_S) fail("rank 0 never published the RCCL id"); }
            std::memcpy(&id, ctl_->nccl_id, sizeof(id));
        }
        ncclchk(ncclCommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
    }
    ~RcclComm() override { if (comm_) ncclCommDestroy(comm_); }
    const char* name() const override { return "RCCL (ncclAllGather / grouped ncclSend + ncclRecv over xGMI)"; }
    void all_gather(const void* send, void* recv, size_t bytes, void* stream) override {
        ncclchk(ncclAllGather(send, recv, bytes, ncclUint8, comm_, (hipStream_t)stream), "ncclAllGather");
    }
    void all_to_all(const void* const* send, const size_t* send_bytes, void* const* recv, const size_t* recv_bytes, void* stream) override {
        // every pair of ranks exchanges one buffer each way: point-to-point operations of one group (each of a GPU's seven xGMI links
        // carries its own pair)
        ncclchk(ncclGroupStart(), "ncclGroupStart");
        for (int k = 0; k < world_; k++) {
            if (send_bytes[k]) ncclchk(ncclSend(send[k], send_bytes[k], ncclUint8, k, comm_, (hipStream_t)stream), "ncclSend");
            if (recv_bytes[k]) ncclchk(ncclRecv(recv[k], recv_bytes[k], ncclUint8, k, comm_, (hipStream_t)stream), "ncclRecv");
        }
        ncclchk(ncclGroupEnd(), "ncclGroupEnd");
    }

private:
    ncclComm_t comm_ = nullptr;
};

class SoloComm : public Communicator {
public:
    int rank() const override { return 0; }
    int world() const override { return 1; }
    const char* name() const override { return "single rank"; }
    void all_gather(const void* send, void* recv, size_t bytes, void* stream) override {
        if (send != recv) hipchk(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync");
    }
    void all_to_all(const void* const* send, const size_t* send_bytes, void* const* recv, const size_t* recv_bytes, void* stream) override {
        if (send_bytes[0] != recv_bytes[0]) fail("all_to_all: size mismatch");
        if (send_bytes[0]) hipchk(hipMemcpyAsync(recv[0], send[0], send_bytes[0], hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync");
    }
    void all_gather_host(const void* send, void* recv, size_t bytes) override { if (send != recv) std::memcpy(recv, send, bytes); }
    void all_reduce_sum_host(double*, size_t) override {}
    void barrier() override {}
};

}  // namespace

void comm_session_create(const std::string& session, int world) {
    comm_session_destroy(session, world);
    const int fd = open(shm_path(session, ".ctl").c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Control)) != 0) { if (fd >= 0) close(fd); fail("cannot create the control file of session '" + session + "'"); }
    close(fd);      // a fresh shared-memory file reads as zeros: counters, flags and sizes start at 0
}

void comm_session_destroy(const std::string& session, int world) {
    unlink(shm_path(session, ".ctl").c_str());
    for (int r = 0; r < std::max(world, 0) && r < MAX_RANKS; r++) unlink(shm_path(session, ".r" + std::to_string(r)).c_str());
}

std::unique_ptr<Communicator> make_host_comm(const std::string& session, int rank, int world) { return std::unique_ptr<Communicator>(new HostComm(session, rank, world)); }
std::unique_ptr<Communicator> make_rccl_comm(const std::string& session, int rank, int world) { return std::unique_ptr<Communicator>(new RcclComm(session, rank, world)); }
std::unique_ptr<Communicator> make_solo_comm() { return std::unique_ptr<Communicator>(new SoloComm()); }

}  // namespace mis
