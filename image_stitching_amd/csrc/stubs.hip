// stubs.hip -- entry points of include/mistitch.h that are not implemented yet report
// MIS_E_UNSUPPORTED (never a CPU fallback).  Shrinks as kernels land; empty when the ABI is complete.
#include "common.h"
#define MIS_STUB(ctx) return mis_set_error((ctx), MIS_E_UNSUPPORTED, "%s is not implemented yet", __func__)
struct MisOrb { MisContext* ctx; };
extern "C" {
void mis_orb_default_params(MisOrbParams* p) { if (p) *p = MisOrbParams{4000, 1.2f, 8, 1, 0, 2, 0, 40, 20}; }
int mis_orb_create(MisContext* ctx, const MisOrbParams*, int, int, MisOrb**) { MIS_STUB(ctx); }
int mis_orb_destroy(MisOrb*) { return MIS_OK; }
int mis_orb_detect(MisOrb* o, const MisImage*, MisFeatures*) { MIS_STUB(o ? o->ctx : nullptr); }
int mis_orb_detect_batch(MisOrb* o, const MisImage*, int, MisFeatures*) { MIS_STUB(o ? o->ctx : nullptr); }
int mis_features_download(MisContext* ctx, const MisFeatures*, MisKeyPoint*, void*) { MIS_STUB(ctx); }
int mis_features_upload(MisContext* ctx, int, int, int, const MisKeyPoint*, const void*, int, int, MisFeatures*) { MIS_STUB(ctx); }
int mis_features_free(MisContext*, MisFeatures*) { return MIS_OK; }
int mis_orb_debug_level(MisOrb* o, int, int, uint8_t*, int*, int*) { MIS_STUB(o ? o->ctx : nullptr); }
void mis_match_default_params(MisMatchParams* p) { if (p) *p = MisMatchParams{0.32f, 6, 6, 3.0, 2000, 0.995}; }
int mis_match_all_pairs(MisContext* ctx, const MisFeatures*, int, const MisMatchParams*, MisMatchesInfo*) { MIS_STUB(ctx); }
int mis_match_pairs_sharded(MisContext* ctx, const MisFeatures*, int, const MisMatchParams*, int, int, MisMatchesInfo*) { MIS_STUB(ctx); }
int mis_matches_free(MisMatchesInfo*, int) { return MIS_OK; }
int mis_knn2(MisContext* ctx, const MisFeatures*, const MisFeatures*, int*, float*) { MIS_STUB(ctx); }
int mis_find_homography(MisContext* ctx, const float*, const float*, int, double, int, double, double*, uint8_t*, int*) { MIS_STUB(ctx); }
int mis_leave_biggest_component(const MisMatchesInfo*, int, float, int*, int*) { return MIS_E_UNSUPPORTED; }
}
