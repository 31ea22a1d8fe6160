// job.hpp -- the hot path of the reference's main() (image_stitching/image_stitching.cpp:567-1228) as ONE job over frames that
// are already resident in HBM, in C++ over the C ABI of libmistitch: the flow bench.py times through
// image_stitching_amd/distributed.py (StitchJob.run on one rank), re-authored for a C++ host.
//
//   blender sizing + zeroing on the compose stream (cameras only)                      :1119-1140, :1175-1192
//   ORB of all frames in one batched call                                              :567-622
//   all-pairs matching; from the matcher's own hook (mis_match_on_enqueued: the calling thread is idle there)
//     the composition of ALL frames is enqueued on the compose stream behind the 2-NN pass (mis_match_knn_fence):
//     batched fused warp + feed (mis_compose_frames) and the collapse (mis_blender_blend)       :647-653, :1086-1228
//   pruning (myLeaveBiggestComponent); when a frame was dropped the composition is redone for the kept set      :215-278
// The result stays in HBM (MisImage with mem = MIS_MEM_DEVICE).
#pragma once
#include <vector>
#include "stitcher.hpp"

namespace mis {

struct JobOutput {
    std::vector<int> indices;         // frames kept by the pruning
    std::vector<double> confidence;   // n x n
    std::vector<int> num_features;
    MisImage pano{}, mask{};          // device: 16SC3 panorama and 8U mask, owned by the job (valid until its next run)
    int num_bands = 0, pano_width = 0, pano_height = 0;
    bool speculation_kept = false;    // the composition enqueued under the matcher was the final one
    std::vector<MisMatchesInfo> matches;   // n x n (host arrays owned by the library; released by the job's next run)
};

class StitchJob {
public:
    StitchJob(int device, int width, int height, const std::vector<CameraParams>& cameras, const StitchConfig& cfg = StitchConfig());
    ~StitchJob();
    StitchJob(const StitchJob&) = delete;
    StitchJob& operator=(const StitchJob&) = delete;
    // frames: n device-resident 8UC3 images of the job's size, complete on the main context's stream (or synchronised)
    JobOutput run(const std::vector<MisImage>& frames);
    void synchronize();
    MisContext* context() const { return ctx_; }           // features + matcher
    MisContext* compose_context() const { return cctx_; }  // warp + blend (own stream)

private:
    struct Compose { int type = 0, bands = 0; float sharp = 0; MisRect pano{}; };
    void check(MisContext* c, int rc, const char* what) const;
    Compose prepare(const std::vector<int>& idx);
    void compose(const std::vector<MisImage>& frames, const std::vector<int>& idx);
    void finalize();
    static void hook(void* self);
    static void prep_hook(void* self);      // the finder's hook (mis_orb_on_enqueued): prepare() under the feature stage

    int w_, h_, n_;
    std::vector<CameraParams> cams_;
    StitchConfig cfg_;
    MisContext* ctx_ = nullptr;
    MisContext* cctx_ = nullptr;
    void* cstream_ = nullptr;
    MisOrb* orb_ = nullptr;
    MisBlender* blender_ = nullptr;
    Compose key_{};
    std::vector<float> Ks_, Rs_;      // n x 9 each (float, as main() hands them to the warper)
    std::vector<MisRect> rois_;       // of the frames of the current composition
    MisImage pano_{}, mask_{};
    std::vector<MisMatchesInfo> pairwise_;
    // state of the hook
    const std::vector<MisImage>* hook_frames_ = nullptr;
    bool hook_ran_ = false;
    std::string hook_error_;
    bool prep_ran_ = false;
    std::string prep_error_;
};

}  // namespace mis
