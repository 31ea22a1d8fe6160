// stitch_bench.cpp -- the job of bench.py (4K frames stitched per second, frames resident in HBM) driven from C++:
// mis::StitchJob over the C ABI, synthetic frames rendered into HBM by synth/libmissynth_gpu.so.
//   stitch_bench <cams.txt> [--steps K] [--warmup W] [--dump prefix]
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

extern "C" {
typedef struct { int width, height; double f, cx, cy; double R[9]; double gain; } SyCamera;     // synth/scene.h
int synth_render_frame_gpu(const SyCamera* cam, void* dev_bgr, size_t stride, void* stream);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: stitch_bench cams.txt [--steps K] [--warmup W] [--dump prefix]\n"); return 2; }
    int steps = 20, warmup = 5;
    std::string dump;
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
    }
    std::ifstream in(argv[1]);
    int n = 0, W = 0, H = 0;
    in >> n >> W >> H;
    if (!in || n < 2) { std::fprintf(stderr, "bad camera file\n"); return 2; }
    std::vector<SyCamera> sy(n);
    std::vector<mis::CameraParams> cams(n);
    for (int i = 0; i < n; i++) {
        std::string tok[13];
        for (auto& t : tok) in >> t;
        if (!in) { std::fprintf(stderr, "bad camera file (frame %d)\n", i); return 2; }
        sy[i].width = W; sy[i].height = H;
        sy[i].f = std::strtod(tok[0].c_str(), nullptr); sy[i].cx = std::strtod(tok[1].c_str(), nullptr); sy[i].cy = std::strtod(tok[2].c_str(), nullptr);
        sy[i].gain = std::strtod(tok[3].c_str(), nullptr);
        for (int k = 0; k < 9; k++) sy[i].R[k] = std::strtod(tok[4 + k].c_str(), nullptr);
        cams[i].focal = sy[i].f; cams[i].aspect = 1; cams[i].ppx = sy[i].cx; cams[i].ppy = sy[i].cy;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) cams[i].R(r, c) = sy[i].R[r * 3 + c];
    }
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
        for (int i = 0; i < warmup; i++) out = job.run(frames);
        job.synchronize();
        HIPCHK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < steps; i++) out = job.run(frames);
        job.synchronize();
        HIPCHK(hipDeviceSynchronize());
        const double dt = now() - t0;
        if (!dump.empty()) {
            std::vector<int16_t> pano((size_t)out.pano_width * out.pano_height * 3);
            std::vector<uint8_t> mask((size_t)out.pano_width * out.pano_height);
            HIPCHK(hipMemcpy2D(pano.data(), (size_t)out.pano_width * 6, out.pano.data, out.pano.stride, (size_t)out.pano_width * 6, out.pano_height, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy2D(mask.data(), (size_t)out.pano_width, out.mask.data, out.mask.stride, (size_t)out.pano_width, out.pano_height, hipMemcpyDeviceToHost));
            std::ofstream(dump + ".pano.s16", std::ios::binary).write((const char*)pano.data(), pano.size() * 2);
            std::ofstream(dump + ".mask.u8", std::ios::binary).write((const char*)mask.data(), mask.size());
            std::ofstream t(dump + ".txt");
            t << out.pano_width << " " << out.pano_height << " " << out.num_bands << "\n";
            for (int i : out.indices) t << i << " ";
            t << "\n";
            for (int v : out.num_features) t << v << " ";
            t << "\n";
            t.precision(17);
            for (double c : out.confidence) t << c << " ";
            t << "\n";
        }
        std::printf("{\"host\": \"c++ (host/stitch_bench: mis::StitchJob over the C ABI)\", \"metric\": \"4K frames stitched/sec\", \"value\": %.3f, \"unit\": \"frames/s\", "
                    "\"n_gpus\": 1, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.3f, \"frames\": %d, \"frame_size\": [%d, %d], \"pano_size\": [%d, %d], "
                    "\"num_bands\": %d, \"kept\": %d, \"speculation_kept\": %s}\n",
                    n * steps / dt, steps, warmup, dt / steps * 1e3, n, W, H, out.pano_width, out.pano_height, out.num_bands, (int)out.indices.size(),
                    out.speculation_kept ? "true" : "false");
        for (auto& f : frames) (void)hipFree(f.data);
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
