// sharded_job.hpp -- ONE panorama job spread over N ranks (one process per GPU) from a C++ host: the hot path of the reference's
// main() (image_stitching/image_stitching.cpp:567-1228) with the three exchange steps of SURVEY section 8(e), RCCL called directly.
//
//   stage                      partition                          exchange (mis::Communicator: RCCL, or host-staged for rehearsals)
//   detect + describe (:613)   frames, a contiguous block / rank  --
//   match + RANSAC (:653)      pairs dealt round-robin            all-gather of {counts, keypoints, descriptors} before it
//   pruning (:215-278)         replicated (n <= 64)               sum of the n x n confidence matrix (every pair has one owner)
//   warp + feed (:1154-1218)   frames (the same blocks)           --
//   blend (:1225)              column strips of the panorama      all-to-all of pyramid rectangles (rank -> strip owner), then an
//                                                                 all-gather of the finished strips
// The flow, the strip plan and the order of the f32 additions are those of image_stitching_amd/distributed.py (StitchJob with
// world_size > 1): the two hosts produce the same panorama byte for byte (tests/test_host_cpp.py).  The composition of a rank's own
// frames is speculated under the matcher (all frames kept is the rule) and redone for the kept set when the pruning drops one.
#pragma once
#include <memory>
#include <vector>
#include "comm.hpp"
#include "stitcher.hpp"

namespace mis {

struct ShardedOutput {
    std::vector<int> indices;          // frames kept by the pruning (every rank)
    std::vector<double> confidence;    // n x n, summed over the ranks (every rank)
    std::vector<int> num_features;     // all n frames (every rank)
    MisImage pano{}, mask{};           // device: the assembled 16SC3 panorama and 8U mask (every rank; owned by the job until its next run)
    int num_bands = 0, pano_width = 0, pano_height = 0;
    bool speculation_kept = false;
};

std::vector<int> frame_block(int n, int rank, int world);     // contiguous block of frame indices owned by `rank`

class ShardedJob {
public:
    // device: this rank's GPU; cameras: all n; comm: rank / world of the job
    ShardedJob(int device, int width, int height, const std::vector<CameraParams>& cameras, Communicator& comm, const StitchConfig& cfg = StitchConfig());
    ~ShardedJob();
    ShardedJob(const ShardedJob&) = delete;
    ShardedJob& operator=(const ShardedJob&) = delete;
    const std::vector<int>& my_frames() const { return mine_; }
    // frames: this rank's block (my_frames() order), device-resident 8UC3 of the job's size, complete on the main stream (or synchronised)
    ShardedOutput run(const std::vector<MisImage>& frames);
    void synchronize();

private:
    struct Compose { int type = 0, bands = 0; float sharp = 0; MisRect pano{}; };
    struct DevBuf { void* p = nullptr; size_t bytes = 0; };
    struct Rect { int level, x0, y0, x1, y1; unsigned long long offset; };
    void check(MisContext* c, int rc, const char* what) const;
    void* reserve(DevBuf& b, size_t bytes);
    Compose prepare(const std::vector<int>& idx);
    void compose_mine(const std::vector<MisImage>& frames, const std::vector<int>& idx);
    void exchange_finalize(const std::vector<int>& idx);
    static void hook(void* self);
    static void prep_hook(void* self);

    int device_, w_, h_, n_;
    std::vector<CameraParams> cams_;
    StitchConfig cfg_;
    Communicator& comm_;
    std::vector<int> mine_;
    void* mstream_ = nullptr;         // main stream: features, matcher, the feature all-gather
    void* cstream_ = nullptr;         // compose stream: warp, feed, the blend exchange, the strip all-gather
    MisContext* ctx_ = nullptr;
    MisContext* cctx_ = nullptr;
    MisOrb* orb_ = nullptr;
    MisBlender* blender_ = nullptr;
    Compose key_{};
    std::vector<float> Ks_, Rs_;
    std::vector<MisRect> rois_;       // of the frames of the current composition (position in idx)
    std::vector<MisMatchesInfo> pairwise_;
    DevBuf kps_send_, desc_send_, kps_all_, desc_all_, strip_mine_, strips_all_, pano_buf_, mask_buf_;
    std::vector<DevBuf> send_, recv_;
    MisImage pano_{}, mask_{};
    const std::vector<MisImage>* hook_frames_ = nullptr;
    bool hook_ran_ = false, prep_ran_ = false;
    std::string hook_error_, prep_error_;
};

}  // namespace mis
