// sharded_job.cpp -- see sharded_job.hpp.
#include "sharded_job.hpp"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstring>
#include <numeric>

namespace mis {

std::vector<int> frame_block(int n, int rank, int world) {
    const int base = n / world, rem = n % world;
    const int lo = rank * base + std::min(rank, rem), cnt = base + (rank < rem ? 1 : 0);
    std::vector<int> out(cnt);
    std::iota(out.begin(), out.end(), lo);
    return out;
}

static float median_focal(const std::vector<CameraParams>& cams, const std::vector<int>& idx) {
    // image_stitching.cpp:884-895: median of the kept cameras' focals (mean of the middle two for an even count), as float
    std::vector<double> f;
    for (int i : idx) f.push_back(cams[i].focal);
    std::sort(f.begin(), f.end());
    return f.size() % 2 == 1 ? static_cast<float>(f[f.size() / 2]) : static_cast<float>(f[f.size() / 2 - 1] + f[f.size() / 2]) * 0.5f;
}

void ShardedJob::check(MisContext* c, int rc, const char* what) const {
    if (rc != MIS_OK) throw std::runtime_error(std::string(what) + " failed on rank " + std::to_string(comm_.rank()) + ": " + mis_last_error(c));
}

void* ShardedJob::reserve(DevBuf& b, size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    if (b.bytes < bytes) {
        if (b.p) (void)hipFree(b.p);      // (synchronises the device: nothing still reads the old block)
        b.p = nullptr; b.bytes = 0;
        if (hipMalloc(&b.p, bytes) != hipSuccess) throw std::runtime_error("hipMalloc of an exchange buffer failed");
        b.bytes = bytes;
    }
    return b.p;
}

ShardedJob::ShardedJob(int device, int width, int height, const std::vector<CameraParams>& cameras, Communicator& comm, const StitchConfig& cfg)
    : device_(device), w_(width), h_(height), n_((int)cameras.size()), cams_(cameras), cfg_(cfg), comm_(comm) {
    if (cfg_.features_type != "orb" || cfg_.ba_cost_func != "no" || cfg_.expos_comp_type != "no" || cfg_.seam_find_type != "no")
        throw std::runtime_error("mis::ShardedJob runs the hot path (ORB, supplied cameras, no seam-scale step)");
    if (n_ % comm_.world() != 0) throw std::runtime_error("the frame count must divide evenly over the ranks");
    mine_ = frame_block(n_, comm_.rank(), comm_.world());
    if (mis_stream_create(device, 0, &mstream_) != MIS_OK || mis_stream_create(device, 0, &cstream_) != MIS_OK)
        throw std::runtime_error("mis_stream_create failed: no HIP device (there is no CPU fallback)");
    if (mis_context_create(device, mstream_, &ctx_) != MIS_OK || mis_context_create(device, cstream_, &cctx_) != MIS_OK)
        throw std::runtime_error("mis_context_create failed");
    MisOrbParams op;
    mis_orb_default_params(&op);
    check(ctx_, mis_orb_create(ctx_, &op, w_, h_, &orb_), "mis_orb_create");
    Ks_.resize((size_t)n_ * 9); Rs_.resize((size_t)n_ * 9);
    for (int i = 0; i < n_; i++) {
        const Mat3<float> K = cams_[i].K().cast<float>(), R = cams_[i].R.cast<float>();
        std::copy(K.m.begin(), K.m.end(), Ks_.begin() + 9 * i);
        std::copy(R.m.begin(), R.m.end(), Rs_.begin() + 9 * i);
    }
    send_.resize(comm_.world()); recv_.resize(comm_.world());
}

ShardedJob::~ShardedJob() {
    if (cctx_) mis_context_synchronize(cctx_);
    if (ctx_) mis_context_synchronize(ctx_);
    if (!pairwise_.empty()) mis_matches_free(pairwise_.data(), (int)pairwise_.size());
    if (blender_) mis_blender_destroy(blender_);
    if (orb_) mis_orb_destroy(orb_);
    for (DevBuf* b : {&kps_send_, &desc_send_, &kps_all_, &desc_all_, &strip_mine_, &strips_all_, &pano_buf_, &mask_buf_}) if (b->p) (void)hipFree(b->p);
    for (auto& b : send_) if (b.p) (void)hipFree(b.p);
    for (auto& b : recv_) if (b.p) (void)hipFree(b.p);
    if (cctx_) mis_context_destroy(cctx_);
    if (ctx_) mis_context_destroy(ctx_);
    if (cstream_) mis_stream_destroy(cstream_);
    if (mstream_) mis_stream_destroy(mstream_);
}

void ShardedJob::synchronize() {
    check(ctx_, mis_context_synchronize(ctx_), "mis_context_synchronize");
    check(cctx_, mis_context_synchronize(cctx_), "mis_context_synchronize (compose)");
}

// warpRoi of the kept frames at the scale of that set, panorama roi, blender sizing + prepare (every rank, all kept frames)
ShardedJob::Compose ShardedJob::prepare(const std::vector<int>& idx) {
    const int m = (int)idx.size();
    const float scale = median_focal(cams_, idx);
    std::vector<float> Ks((size_t)m * 9), Rs((size_t)m * 9);
    for (int k = 0; k < m; k++) {
        std::copy(Ks_.begin() + 9 * idx[k], Ks_.begin() + 9 * idx[k] + 9, Ks.begin() + 9 * k);
        std::copy(Rs_.begin() + 9 * idx[k], Rs_.begin() + 9 * idx[k] + 9, Rs.begin() + 9 * k);
    }
    rois_.assign(m, MisRect{});
    check(cctx_, mis_warp_roi_batch(cctx_, scale, w_, h_, m, Ks.data(), Rs.data(), rois_.data()), "mis_warp_roi_batch");
    std::vector<MisPoint> corners(m);
    std::vector<MisSize> sizes(m);
    for (int k = 0; k < m; k++) { corners[k] = {rois_[k].x, rois_[k].y}; sizes[k] = {rois_[k].width, rois_[k].height}; }
    Compose c;
    check(cctx_, mis_result_roi(corners.data(), sizes.data(), m, &c.pano), "mis_result_roi");
    check(cctx_, mis_blend_config(cfg_.blend_type, cfg_.blend_strength, c.pano.width, c.pano.height, &c.type, &c.bands, &c.sharp), "mis_blend_config");
    if (c.type != MIS_BLEND_MULTI_BAND) throw std::runtime_error("mis::ShardedJob exchanges the multi-band blender's pyramids");
    if (!blender_ || c.type != key_.type || c.bands != key_.bands || c.sharp != key_.sharp) {
        if (blender_) { mis_blender_destroy(blender_); blender_ = nullptr; }
        check(cctx_, mis_blender_create(cctx_, c.type, c.bands, c.sharp, &blender_), "mis_blender_create");
    }
    key_ = c;
    check(cctx_, mis_blender_prepare(blender_, corners.data(), sizes.data(), m), "mis_blender_prepare");
    return c;
}

// batched fused warp + feed of this rank's frames among the kept ones (rois_ from prepare(idx))
void ShardedJob::compose_mine(const std::vector<MisImage>& frames, const std::vector<int>& idx) {
    std::vector<MisImage> fr;
    std::vector<float> Ks, Rs;
    std::vector<MisRect> rois;
    for (size_t q = 0; q < mine_.size(); q++) {
        const auto it = std::find(idx.begin(), idx.end(), mine_[q]);
        if (it == idx.end()) continue;
        const int k = (int)(it - idx.begin());
        fr.push_back(frames[q]);
        Ks.insert(Ks.end(), Ks_.begin() + 9 * mine_[q], Ks_.begin() + 9 * mine_[q] + 9);
        Rs.insert(Rs.end(), Rs_.begin() + 9 * mine_[q], Rs_.begin() + 9 * mine_[q] + 9);
        rois.push_back(rois_[k]);
    }
    if (fr.empty()) return;     // every frame of this rank was pruned: nothing to warp or feed (the exchanges around still run)
    check(cctx_, mis_compose_frames(blender_, fr.data(), (int)fr.size(), median_focal(cams_, idx), Ks.data(), Rs.data(), rois.data()), "mis_compose_frames");
}

// Strip exchange + per-strip finalise + assembly (distributed.py: stage_reduce_finalize): every rank ends with the panorama.
void ShardedJob::exchange_finalize(const std::vector<int>& idx) {
    const int world = comm_.world(), me = comm_.rank(), bands = key_.bands;
    std::vector<int> lw(bands + 1), lh(bands + 1);
    for (int l = 0; l <= bands; l++) check(cctx_, mis_blender_level_info(blender_, l, &lw[l], &lh[l], nullptr, nullptr), "mis_blender_level_info");
    const int w0 = lw[0], h0 = lh[0];
    // column strips [c_k, c_k+1) of the padded panorama, boundaries multiples of 2^bands
    const int q = 1 << bands, nq = w0 / q;
    std::vector<std::pair<int, int>> bounds(world);
    for (int k = 0; k < world; k++) bounds[k] = {(int)((long long)k * nq / world) * q, k + 1 < world ? (int)((long long)(k + 1) * nq / world) * q : w0};
    // need[k][l]: the columns of level l whose summed accumulators the owner of strip k needs (pyrUp's footprint, level by level)
    std::vector<std::vector<std::pair<int, int>>> need(world);
    for (int k = 0; k < world; k++) {
        int lo = bounds[k].first, hi = bounds[k].second;
        need[k].push_back({lo, hi});
        for (int l = 1; l <= bands; l++) { lo = std::max(0, (lo >> 1) - 1); hi = std::min(lw[l], ((hi - 1) >> 1) + 2); need[k].push_back({lo, hi}); }
    }
    // region[r]: the level-0 rectangle rank r's kept frames can touch (frame rois + the blender's 3 * 2^bands margin, aligned)
    int px = 1 << 30, py = 1 << 30;
    for (size_t k = 0; k < idx.size(); k++) { px = std::min(px, rois_[k].x); py = std::min(py, rois_[k].y); }
    struct Region { bool any; int x0, y0, x1, y1; };
    std::vector<Region> region(world);
    const int gap = 3 << bands, a = q - 1;
    for (int r = 0; r < world; r++) {
        Region g{false, 1 << 30, 1 << 30, -(1 << 30), -(1 << 30)};
        for (int i : frame_block(n_, r, world)) {
            const auto it = std::find(idx.begin(), idx.end(), i);
            if (it == idx.end()) continue;
            const MisRect& rc = rois_[it - idx.begin()];
            g.any = true;
            g.x0 = std::min(g.x0, rc.x); g.y0 = std::min(g.y0, rc.y); g.x1 = std::max(g.x1, rc.x + rc.width); g.y1 = std::max(g.y1, rc.y + rc.height);
        }
        if (g.any) {
            g.x0 = std::max(0, g.x0 - px - gap) & ~a; g.y0 = std::max(0, g.y0 - py - gap) & ~a;
            g.x1 = std::min(w0, (g.x1 - px + gap + a) & ~a); g.y1 = std::min(h0, (g.y1 - py + gap + a) & ~a);
        }
        region[r] = g;
    }
    // plan[src][dst]: rectangles of src's region inside dst's need columns, packed: per level the 16SC3 block then the f32 block
    auto rects_of = [&](int src, int dst, std::vector<MisLevelRect>& out) -> size_t {
        out.clear();
        size_t off = 0;
        const Region& g = region[src];
        if (!g.any) return 0;
        for (int l = 0; l <= bands; l++) {
            const int x0 = std::max(std::min(g.x0 >> l, lw[l]), need[dst][l].first), x1 = std::min(std::min(-((-g.x1) >> l), lw[l]), need[dst][l].second);
            const int y0 = std::min(g.y0 >> l, lh[l]), y1 = std::min(-((-g.y1) >> l), lh[l]);
            if (x1 <= x0 || y1 <= y0) continue;
            const size_t m = (size_t)(x1 - x0) * (y1 - y0);
            out.push_back(MisLevelRect{l, x0, y0, x1, y1, (unsigned long long)off});
            off += (m * 6 + 15) / 16 * 16 + (m * 4 + 15) / 16 * 16;
        }
        return off;
    };
    std::vector<MisLevelRect> rects;
    std::vector<const void*> sp(world);
    std::vector<void*> rp(world);
    std::vector<size_t> sb(world), rb(world);
    for (int k = 0; k < world; k++) {
        sb[k] = rects_of(me, k, rects);
        sp[k] = reserve(send_[k], sb[k]);
        if (!rects.empty()) check(cctx_, mis_blender_pack_rects(blender_, rects.data(), (int)rects.size(), send_[k].p, send_[k].bytes), "mis_blender_pack_rects");
        rb[k] = rects_of(k, me, rects);
        rp[k] = reserve(recv_[k], rb[k]);
    }
    comm_.all_to_all(sp.data(), sb.data(), rp.data(), rb.data(), cstream_);
    // the owner zeroes its need ranges and adds the N buffers in rank order (its own included): fixed association of the f32 sums
    std::vector<MisLevelRect> full;
    for (int l = 0; l <= bands; l++)
        if (need[me][l].second > need[me][l].first) full.push_back(MisLevelRect{l, need[me][l].first, 0, need[me][l].second, lh[l], 0ull});
    if (!full.empty()) check(cctx_, mis_blender_zero_rects(blender_, full.data(), (int)full.size()), "mis_blender_zero_rects");
    for (int r = 0; r < world; r++) {
        rects_of(r, me, rects);
        if (!rects.empty()) check(cctx_, mis_blender_add_rects(blender_, rects.data(), (int)rects.size(), recv_[r].p, recv_[r].bytes), "mis_blender_add_rects");
    }
    // this rank's strip, finalised straight into its slot of the all-gather: rows of cap_w pixels (16SC3), then the mask rows
    const int pw = key_.pano.width, ph = key_.pano.height;
    int cap_w = 0;
    for (int k = 0; k < world; k++) cap_w = std::max(cap_w, std::min(bounds[k].second, pw) - bounds[k].first);
    cap_w = std::max(cap_w, 1);
    const size_t strip_bytes = (size_t)ph * cap_w * 7;
    uint8_t* mine = (uint8_t*)reserve(strip_mine_, strip_bytes);
    const int x0 = bounds[me].first, x1 = std::min(bounds[me].second, pw);
    if (x1 > x0) {
        MisImage img{mine, x1 - x0, ph, 3, (size_t)cap_w * 6, MIS_S16, MIS_MEM_DEVICE};
        MisImage msk{mine + (size_t)ph * cap_w * 6, x1 - x0, ph, 1, (size_t)cap_w, MIS_U8, MIS_MEM_DEVICE};
        check(cctx_, mis_blender_blend_columns(blender_, x0, x1, &img, &msk), "mis_blender_blend_columns");
    }
    uint8_t* all = (uint8_t*)reserve(strips_all_, strip_bytes * world);
    comm_.all_gather(mine, all, strip_bytes, cstream_);
    // assembly: every strip's columns into the panorama (tight rows)
    uint8_t* pano = (uint8_t*)reserve(pano_buf_, (size_t)ph * pw * 6);
    uint8_t* mask = (uint8_t*)reserve(mask_buf_, (size_t)ph * pw);
    for (int k = 0; k < world; k++) {
        const int b0 = bounds[k].first, b1 = std::min(bounds[k].second, pw);
        if (b1 <= b0) continue;
        const uint8_t* base = all + (size_t)k * strip_bytes;
        check(cctx_, mis_copy_2d(cctx_, pano + (size_t)b0 * 6, (size_t)pw * 6, base, (size_t)cap_w * 6, (size_t)(b1 - b0) * 6, ph), "mis_copy_2d");
        check(cctx_, mis_copy_2d(cctx_, mask + b0, (size_t)pw, base + (size_t)ph * cap_w * 6, (size_t)cap_w, (size_t)(b1 - b0), ph), "mis_copy_2d");
    }
    pano_ = MisImage{pano, pw, ph, 3, (size_t)pw * 6, MIS_S16, MIS_MEM_DEVICE};
    mask_ = MisImage{mask, pw, ph, 1, (size_t)pw, MIS_U8, MIS_MEM_DEVICE};
}

void ShardedJob::hook(void* self_) {      // inside the matcher call, once its device work is enqueued: this rank's composition
    ShardedJob* self = static_cast<ShardedJob*>(self_);
    self->hook_ran_ = true;
    try {
        const int rc = mis_match_knn_fence(self->ctx_, self->cstream_, mis_match_sequence(self->ctx_), 0);
        if (rc < 0) self->check(self->ctx_, rc, "mis_match_knn_fence");
        std::vector<int> everyone(self->n_);
        std::iota(everyone.begin(), everyone.end(), 0);
        self->compose_mine(*self->hook_frames_, everyone);
    } catch (const std::exception& e) { self->hook_error_ = e.what(); }
}

void ShardedJob::prep_hook(void* self_) {
    ShardedJob* self = static_cast<ShardedJob*>(self_);
    self->prep_ran_ = true;
    try {
        std::vector<int> everyone(self->n_);
        std::iota(everyone.begin(), everyone.end(), 0);
        self->prepare(everyone);
    } catch (const std::exception& e) { self->prep_error_ = e.what(); }
}

ShardedOutput ShardedJob::run(const std::vector<MisImage>& frames) {
    const int m = (int)mine_.size(), world = comm_.world();
    if ((int)frames.size() != m) throw std::runtime_error("ShardedJob::run: one frame per camera of this rank's block");
    ShardedOutput out;
    std::vector<int> everyone(n_);
    std::iota(everyone.begin(), everyone.end(), 0);
    check(cctx_, mis_context_wait(cctx_, ctx_), "mis_context_wait");
    // ---- features of this rank's block (:567-622); the blender is sized from the finder's hook (cameras only) ----
    prep_ran_ = false; prep_error_.clear();
    check(ctx_, mis_orb_on_enqueued(orb_, &ShardedJob::prep_hook, this), "mis_orb_on_enqueued");
    std::vector<MisFeatures> local(m);
    std::memset(local.data(), 0, sizeof(MisFeatures) * m);
    const int rc_f = m > 0 ? mis_orb_detect_batch(orb_, frames.data(), m, local.data()) : MIS_OK;
    mis_orb_on_enqueued(orb_, nullptr, nullptr);
    check(ctx_, rc_f, "mis_orb_detect_batch");
    if (!prep_ran_) prep_hook(this);
    if (!prep_error_.empty()) throw std::runtime_error(prep_error_);
    // ---- feature all-gather: counts (host), then keypoints and descriptors packed to the job's largest count ----
    std::vector<int> cnt_mine(m), cnt_all(n_);
    for (int k = 0; k < m; k++) cnt_mine[k] = local[k].n;
    comm_.all_gather_host(cnt_mine.data(), cnt_all.data(), sizeof(int) * m);
    const int cap = std::max(1, *std::max_element(cnt_all.begin(), cnt_all.end()));
    const size_t kb = (size_t)cap * sizeof(MisKeyPoint), db = (size_t)cap * 32;
    reserve(kps_send_, kb * m); reserve(desc_send_, db * m); reserve(kps_all_, kb * n_); reserve(desc_all_, db * n_);
    if (m > 0) check(ctx_, mis_features_pack(ctx_, local.data(), m, cap, 32, kps_send_.p, desc_send_.p), "mis_features_pack");
    comm_.all_gather(kps_send_.p, kps_all_.p, kb * m, mstream_);
    comm_.all_gather(desc_send_.p, desc_all_.p, db * m, mstream_);
    std::vector<MisFeatures> feats(n_);
    for (int i = 0; i < n_; i++) {
        MisFeatures& f = feats[i];
        f.img_idx = i; f.img_w = w_; f.img_h = h_; f.n = cnt_all[i];
        f.keypoints = reinterpret_cast<MisKeyPoint*>((uint8_t*)kps_all_.p + kb * i);
        f.descriptors = (uint8_t*)desc_all_.p + db * i;
        f.desc_cols = 32; f.desc_dtype = MIS_U8; f.owner_ = nullptr;
    }
    out.num_features = cnt_all;
    // ---- this rank's pairs (:647-653), its composition speculated from the matcher's hook ----
    if (!pairwise_.empty()) { mis_matches_free(pairwise_.data(), (int)pairwise_.size()); pairwise_.clear(); }
    pairwise_.assign((size_t)n_ * n_, MisMatchesInfo{});
    MisMatchParams mp;
    mis_match_default_params(&mp);
    mp.match_conf = cfg_.match_conf;
    hook_frames_ = &frames; hook_ran_ = false; hook_error_.clear();
    check(ctx_, mis_match_on_enqueued(ctx_, &ShardedJob::hook, this), "mis_match_on_enqueued");
    const int rc = mis_match_pairs_sharded(ctx_, feats.data(), n_, &mp, comm_.rank(), world, pairwise_.data());
    mis_match_on_enqueued(ctx_, nullptr, nullptr);
    for (auto& f : local) mis_features_free(ctx_, &f);
    check(ctx_, rc, "mis_match_pairs_sharded");
    if (!hook_ran_) hook(this);
    if (!hook_error_.empty()) throw std::runtime_error(hook_error_);
    // ---- the n x n confidences: every pair has exactly one owner, the sum is a gather ----
    out.confidence.resize((size_t)n_ * n_);
    for (int k = 0; k < n_ * n_; k++) out.confidence[k] = pairwise_[k].confidence;
    comm_.all_reduce_sum_host(out.confidence.data(), out.confidence.size());
    std::vector<int> idx(n_);
    int kept = 0;
    check(ctx_, mis_leave_biggest_component_conf(out.confidence.data(), n_, cfg_.conf_thresh, idx.data(), &kept), "mis_leave_biggest_component_conf");
    idx.resize(kept);
    if (kept < 2) throw std::runtime_error("Need more images");
    out.indices = idx;
    out.speculation_kept = kept == n_;
    if (!out.speculation_kept) {       // a frame was dropped: the kept set's own scale, roi and band count
        prepare(idx);
        compose_mine(frames, idx);
    }
    exchange_finalize(idx);
    check(cctx_, mis_context_synchronize(cctx_), "mis_context_synchronize (compose)");
    out.pano = pano_; out.mask = mask_;
    out.num_bands = key_.bands; out.pano_width = key_.pano.width; out.pano_height = key_.pano.height;
    return out;
}

}  // namespace mis
