/* synth_cpu.c -- CPU build of the synthetic scene renderer (test / bench infrastructure). */
#include "scene.h"
#include <stddef.h>
void synth_render_frame_cpu(const SyCamera* cam, uint8_t* bgr, size_t stride) {
    int y;
#pragma omp parallel for schedule(dynamic, 8)
    for (y = 0; y < cam->height; y++)
        for (int x = 0; x < cam->width; x++) sy_render_pixel(cam, x, y, bgr + (size_t)y * stride + 3 * (size_t)x);
}
