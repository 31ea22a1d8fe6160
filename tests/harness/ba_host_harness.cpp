// Test harness (not shipped): the library's bundle-adjustment HOST code (csrc/motion.hip has no kernels)
// compiled for the CPU with the three HIP runtime calls it makes turned into host operations, so that it can be stepped against
// the oracle's restatement without a GPU.   tests/test_motion_host_cpu.py builds and drives it; tools/dbg/ba_cmp.py is the interactive form.
#include <cstdarg>
#include <cstring>
#include "../../image_stitching_amd/csrc/common.h"
#define hipSetDevice(d) hipSuccess
#define hipStreamSynchronize(s) hipSuccess
#define hipMemcpy(d, s, n, k) (std::memcpy((d), (s), (n)), hipSuccess)
#include "../../image_stitching_amd/csrc/motion.hip"
int mis_set_error(MisContext* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    fprintf(stderr, "mis error %d: %s\n", code, buf);
    return code;
}
extern "C" int dbg_bundle_adjust(const MisFeatures* f, const MisMatchesInfo* pm, int n, float conf, const char* mask, MisCameraParams* cams) {
    MisContext ctx;
    return mis_bundle_adjust_reproj(&ctx, f, pm, n, conf, mask, cams);
}
