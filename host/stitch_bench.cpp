// stitch_bench.cpp -- the job of bench.py (4K frames stitched per second, frames resident in HBM) driven from C++:
// mis::StitchJob over the C ABI, synthetic frames rendered into HBM by synth/libmissynth_gpu.so.
//   stitch_bench <cams.txt> [--steps K] [--warmup W] [--dump prefix] [--ranks N] [--comm host|rccl] [--one-gpu]
// --ranks N > 1: the SHARDED job (mis::ShardedJob, host/sharded_job.hpp): this process never touches the GPU -- it creates the ranks'
// rendezvous file, starts N child processes of itself (rank r on GPU r; --one-gpu: all on GPU 0, a rehearsal), relays rank 0's line,
// and when a rank fails it ends the others and exits non-zero.  --comm rccl (default for N > 1): RCCL called directly; --comm host:
// exchanges staged through shared memory (several ranks on ONE GPU).  --ranks 1 --comm rccl runs the sharded flow on a one-rank
// RCCL communicator (the RCCL calls themselves on a box with one GPU).
// cams.txt (written by bench.py / the tests): "n width height" then per frame "f cx cy gain r0 ... r8" (repr doubles).
// Prints ONE JSON line: {"host": "c++", "value": frames/s, "ms_per_step": ..., ...}.  --dump writes the last run's panorama
// (<prefix>.pano.s16, tight rows), mask (<prefix>.mask.u8) and "<prefix>.txt" (indices, sizes, bands) for the parity test.
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "job.hpp"
#include "sharded_job.hpp"
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>
extern char** environ;

extern "C" {
typedef struct { int width, height; double f, cx, cy; double R[9]; double gain; } SyCamera;     // synth/scene.h
int synth_render_frame_gpu(const SyCamera* cam, void* dev_bgr, size_t stride, void* stream);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)


struct Args {
    std::string cams_path, dump, comm, session;
    int steps = 20, warmup = 5, ranks = 1, child_rank = -1;
    bool one_gpu = false;
};

static bool read_cams(const std::string& path, int* n, int* W, int* H, std::vector<SyCamera>* sy, std::vector<mis::CameraParams>* cams) {
    std::ifstream in(path);
    in >> *n >> *W >> *H;
    if (!in || *n < 2) { std::fprintf(stderr, "bad camera file\n"); return false; }
    sy->resize(*n); cams->resize(*n);
    for (int i = 0; i < *n; i++) {
        std::string tok[13];
        for (auto& t : tok) in >> t;
        if (!in) { std::fprintf(stderr, "bad camera file (frame %d)\n", i); return false; }
        SyCamera& c = (*sy)[i];
        c.width = *W; c.height = *H;
        c.f = std::strtod(tok[0].c_str(), nullptr); c.cx = std::strtod(tok[1].c_str(), nullptr); c.cy = std::strtod(tok[2].c_str(), nullptr);
        c.gain = std::strtod(tok[3].c_str(), nullptr);
        for (int k = 0; k < 9; k++) c.R[k] = std::strtod(tok[4 + k].c_str(), nullptr);
        mis::CameraParams& p = (*cams)[i];
        p.focal = c.f; p.aspect = 1; p.ppx = c.cx; p.ppy = c.cy;
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) p.R(r, cc) = c.R[r * 3 + cc];
    }
    return true;
}

static int write_dump(const std::string& dump, const MisImage& pano_d, const MisImage& mask_d, int pw, int ph, int bands, const std::vector<int>& indices,
                      const std::vector<int>& nfeat, const std::vector<double>& conf) {
    std::vector<int16_t> pano((size_t)pw * ph * 3);
    std::vector<uint8_t> mask((size_t)pw * ph);
    HIPCHK(hipMemcpy2D(pano.data(), (size_t)pw * 6, pano_d.data, pano_d.stride, (size_t)pw * 6, ph, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy2D(mask.data(), (size_t)pw, mask_d.data, mask_d.stride, (size_t)pw, ph, hipMemcpyDeviceToHost));
    std::ofstream(dump + ".pano.s16", std::ios::binary).write((const char*)pano.data(), pano.size() * 2);
    std::ofstream(dump + ".mask.u8", std::ios::binary).write((const char*)mask.data(), mask.size());
    std::ofstream t(dump + ".txt");
    t << pw << " " << ph << " " << bands << "\n";
    for (int i : indices) t << i << " ";
    t << "\n";
    for (int v : nfeat) t << v << " ";
    t << "\n";
    t.precision(17);
    for (double c : conf) t << c << " ";
    t << "\n";
    return 0;
}

// the unsharded job on one GPU (mis::StitchJob)
static int run_single(const Args& a) {
    int n = 0, W = 0, H = 0;
    std::vector<SyCamera> sy;
    std::vector<mis::CameraParams> cams;
    if (!read_cams(a.cams_path, &n, &W, &H, &sy, &cams)) return 2;
    try {
        mis::StitchJob job(0, W, H, cams);
        // synthetic frames straight into HBM (rows of 3 W bytes, as bench.py's torch tensors)
        std::vector<MisImage> frames(n);
        for (int i = 0; i < n; i++) {
            void* p = nullptr;
            HIPCHK(hipMalloc(&p, (size_t)W * H * 3));
            if (synth_render_frame_gpu(&sy[i], p, (size_t)W * 3, nullptr) != 0) { std::fprintf(stderr, "render failed\n"); return 1; }
            frames[i] = MisImage{p, W, H, 3, (size_t)W * 3, MIS_U8, MIS_MEM_DEVICE};
        }
        HIPCHK(hipDeviceSynchronize());
        mis::JobOutput out;
        for (int i = 0; i < a.warmup; i++) out = job.run(frames);
        job.synchronize();
        HIPCHK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < a.steps; i++) out = job.run(frames);
        job.synchronize();
        HIPCHK(hipDeviceSynchronize());
        const double dt = now() - t0;
        if (!a.dump.empty() && write_dump(a.dump, out.pano, out.mask, out.pano_width, out.pano_height, out.num_bands, out.indices, out.num_features, out.confidence)) return 1;
        std::printf("{\"host\": \"c++ (host/stitch_bench: mis::StitchJob over the C ABI)\", \"metric\": \"4K frames stitched/sec\", \"value\": %.3f, \"unit\": \"frames/s\", "
                    "\"n_gpus\": 1, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.3f, \"frames\": %d, \"frame_size\": [%d, %d], \"pano_size\": [%d, %d], "
                    "\"num_bands\": %d, \"kept\": %d, \"speculation_kept\": %s}\n",
                    n * a.steps / dt, a.steps, a.warmup, dt / a.steps * 1e3, n, W, H, out.pano_width, out.pano_height, out.num_bands, (int)out.indices.size(),
                    out.speculation_kept ? "true" : "false");
        for (auto& f : frames) (void)hipFree(f.data);
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}

// one rank of the sharded job (mis::ShardedJob); rank 0 prints the result line
static int run_rank(const Args& a, int rank) {
    int n = 0, W = 0, H = 0;
    std::vector<SyCamera> sy;
    std::vector<mis::CameraParams> cams;
    if (!read_cams(a.cams_path, &n, &W, &H, &sy, &cams)) return 2;
    try {
        const int device = a.one_gpu ? 0 : rank;
        HIPCHK(hipSetDevice(device));
        std::unique_ptr<mis::Communicator> comm = a.comm == "host" ? mis::make_host_comm(a.session, rank, a.ranks) : mis::make_rccl_comm(a.session, rank, a.ranks);
        mis::ShardedJob job(device, W, H, cams, *comm);
        std::vector<MisImage> frames;
        for (int i : job.my_frames()) {
            void* p = nullptr;
            HIPCHK(hipMalloc(&p, (size_t)W * H * 3));
            if (synth_render_frame_gpu(&sy[i], p, (size_t)W * 3, nullptr) != 0) { std::fprintf(stderr, "render failed\n"); return 1; }
            frames.push_back(MisImage{p, W, H, 3, (size_t)W * 3, MIS_U8, MIS_MEM_DEVICE});
        }
        HIPCHK(hipDeviceSynchronize());
        mis::ShardedOutput out;
        for (int i = 0; i < a.warmup; i++) out = job.run(frames);
        job.synchronize();
        HIPCHK(hipDeviceSynchronize());
        comm->barrier();
        const double t0 = now();
        for (int i = 0; i < a.steps; i++) out = job.run(frames);
        job.synchronize();
        HIPCHK(hipDeviceSynchronize());
        comm->barrier();
        double dt = now() - t0;
        std::vector<double> all(a.ranks);
        comm->all_gather_host(&dt, all.data(), sizeof(double));
        for (double v : all) dt = std::max(dt, v);      // the slowest rank's clock
        if (rank == 0) {
            if (!a.dump.empty() && write_dump(a.dump, out.pano, out.mask, out.pano_width, out.pano_height, out.num_bands, out.indices, out.num_features, out.confidence)) return 1;
            std::printf("{\"host\": \"c++ (host/stitch_bench: mis::ShardedJob, %d rank%s, %s)\", \"metric\": \"4K frames stitched/sec\", \"value\": %.3f, \"unit\": \"frames/s\", "
                        "\"n_gpus\": %d, \"one_gpu_rehearsal\": %s, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.3f, \"frames\": %d, \"frame_size\": [%d, %d], \"pano_size\": [%d, %d], "
                        "\"num_bands\": %d, \"kept\": %d, \"speculation_kept\": %s}\n",
                        a.ranks, a.ranks == 1 ? "" : "s", comm->name(), n * a.steps / dt, a.one_gpu ? 1 : a.ranks, a.one_gpu ? "true" : "false", a.steps, a.warmup,
                        dt / a.steps * 1e3, n, W, H, out.pano_width, out.pano_height, out.num_bands, (int)out.indices.size(), out.speculation_kept ? "true" : "false");
            std::fflush(stdout);
        }
        comm->barrier();
        for (auto& f : frames) (void)hipFree(f.data);
    } catch (const std::exception& e) {
        std::printf("error (rank %d): %s\n", rank, e.what());
        std::fflush(stdout);
        return 1;
    }
    return 0;
}

// N ranks as child processes of a parent that never touches the GPU
static int launch(const Args& a, int argc, char** argv) {
    const std::string session = "mis_bench_" + std::to_string((long long)getpid());
    try { mis::comm_session_create(session, a.ranks); } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    std::vector<pid_t> pids(a.ranks, -1);
    int rc = 0;
    for (int r = 0; r < a.ranks && rc == 0; r++) {
        std::vector<std::string> av(argv, argv + argc);
        av.push_back("--rank-child"); av.push_back(std::to_string(r));
        av.push_back("--session"); av.push_back(session);
        std::vector<char*> cav;
        for (auto& s : av) cav.push_back(const_cast<char*>(s.c_str()));
        cav.push_back(nullptr);
        if (posix_spawn(&pids[r], "/proc/self/exe", nullptr, nullptr, cav.data(), environ) != 0) { std::fprintf(stderr, "cannot start rank %d\n", r); pids[r] = -1; rc = 1; }
    }
    int left = 0;
    for (pid_t p : pids) left += p > 0;
    while (left > 0) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p <= 0) break;
        for (auto& q : pids) if (q == p) { q = -1; left--; }
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (code != 0 && rc == 0) {
            rc = code;
            std::fprintf(stderr, "stitch_bench: a rank exited with code %d: ending the others\n", code);
            for (pid_t q : pids) if (q > 0) kill(q, SIGKILL);       // exactly the processes started above
        }
    }
    mis::comm_session_destroy(session, a.ranks);
    return rc;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: stitch_bench cams.txt [--steps K] [--warmup W] [--dump prefix] [--ranks N] [--comm host|rccl] [--one-gpu]\n"); return 2; }
    Args a;
    a.cams_path = argv[1];
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) a.steps = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc) a.warmup = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) a.dump = argv[++i];
        else if (!std::strcmp(argv[i], "--ranks") && i + 1 < argc) a.ranks = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--comm") && i + 1 < argc) a.comm = argv[++i];
        else if (!std::strcmp(argv[i], "--one-gpu")) a.one_gpu = true;
        else if (!std::strcmp(argv[i], "--rank-child") && i + 1 < argc) a.child_rank = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--session") && i + 1 < argc) a.session = argv[++i];
    }
    if (a.ranks < 1 || a.ranks > 16) { std::fprintf(stderr, "--ranks 1..16\n"); return 2; }
    if (a.comm.empty() && a.ranks > 1) a.comm = "rccl";
    if (!a.comm.empty() && a.comm != "host" && a.comm != "rccl") { std::fprintf(stderr, "--comm host | rccl\n"); return 2; }
    if (a.child_rank >= 0) return run_rank(a, a.child_rank);
    if (a.ranks > 1) return launch(a, argc, argv);
    if (!a.comm.empty()) {          // one rank through the sharded flow (the RCCL / host-staged calls themselves): in this process
        a.session = "mis_bench_" + std::to_string((long long)getpid());
        try { mis::comm_session_create(a.session, 1); } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
        const int rc = run_rank(a, 0);
        mis::comm_session_destroy(a.session, 1);
        return rc;
    }
    return run_single(a);
}
