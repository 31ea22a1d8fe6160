/* synth_gpu.hip -- GPU build of the synthetic scene renderer (test / bench infrastructure: it only
 * manufactures input frames in HBM; it is not part of the stitching hot path). */
#include <hip/hip_runtime.h>
#include "scene.h"
__global__ void synth_render_kernel(SyCamera cam, uint8_t* bgr, size_t stride) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cam.width || y >= cam.height) return;
    uint8_t px[3];
    sy_render_pixel(&cam, x, y, px);
    uint8_t* o = bgr + (size_t)y * stride + 3 * (size_t)x;
    o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
}
extern "C" int synth_render_frame_gpu(const SyCamera* cam, void* dev_bgr, size_t stride, void* stream) {
    dim3 b(64, 4), g((cam->width + 63) / 64, (cam->height + 3) / 4);
    hipLaunchKernelGGL(synth_render_kernel, g, b, 0, (hipStream_t)stream, *cam, (uint8_t*)dev_bgr, stride);
    return (int)hipGetLastError();
}
