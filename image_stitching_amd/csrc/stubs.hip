// stubs.hip -- entry points of include/mistitch.h that are not implemented yet report
// MIS_E_UNSUPPORTED (never a CPU fallback).  Shrinks as kernels land; empty when the ABI is complete.
#include "common.h"
#define MIS_STUB(ctx) return mis_set_error((ctx), MIS_E_UNSUPPORTED, "%s is not implemented yet", __func__)
extern "C" {
void mis_match_default_params(MisMatchParams* p) { if (p) *p = MisMatchParams{0.32f, 6, 6, 3.0, 2000, 0.995}; }
int mis_match_all_pairs(MisContext* ctx, const MisFeatures*, int, const MisMatchParams*, MisMatchesInfo*) { MIS_STUB(ctx); }
int mis_match_pairs_sharded(MisContext* ctx, const MisFeatures*, int, const MisMatchParams*, int, int, MisMatchesInfo*) { MIS_STUB(ctx); }
int mis_matches_free(MisMatchesInfo*, int) { return MIS_OK; }
int mis_knn2(MisContext* ctx, const MisFeatures*, const MisFeatures*, int*, float*) { MIS_STUB(ctx); }
int mis_find_homography(MisContext* ctx, const float*, const float*, int, double, int, double, double*, uint8_t*, int*) { MIS_STUB(ctx); }
int mis_leave_biggest_component(const MisMatchesInfo*, int, float, int*, int*) { return MIS_E_UNSUPPORTED; }
}
