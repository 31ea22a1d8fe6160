/* mo_seam.h -- ORACLE (test infrastructure): DpSeamFinder(COLOR); see mo_seam.c. */
#ifndef MO_SEAM_H
#define MO_SEAM_H
#include <stdint.h>
/* images_bgr[i]: tight 8UC3 (sizes_wh[2i] x sizes_wh[2i+1]); masks[i]: tight 8U of the same size, edited in place;
 * corners_xy: top-left corners.  Returns 0. */
int mo_seam_dp_color(int n, const int* corners_xy, const int* sizes_wh, const uint8_t* const* images_bgr, uint8_t* const* masks);
#endif
