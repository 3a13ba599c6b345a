/*
 * svo_oracle_klt.c -- CPU restatement, part 2: pyramidal KLT feature alignment
 * (a-9).  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED -- see svo_oracle.h.
 *
 * Follows feature_alignment::alignPyr2D / alignPyr2DVec
 *   src/svo_direct/src/feature_alignment.cpp:761-973, 732-758
 * (the scalar branch; the NEON branch is the same integer arithmetic) as driven
 * by FeatureTracker::trackFrameBundle, src/svo_tracker/src/feature_tracker.cpp:52-122.
 * Third-party arithmetic restated: Eigen 3.4 Matrix2f::inverse()
 * (compute_inverse_size2: invdet = 1/det, adjugate * invdet).
 */
#include <math.h>
#include <string.h>

#include "svo_oracle.h"

#define SVO_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

/* returns 1 = converged, 0 = failed; px_cur_level_0 in/out */
int orc_align_pyr_2d(const orc_pyramid* pyr_ref, const orc_pyramid* pyr_cur, int max_level, int min_level,
                     const int32_t* patch_sizes, int n_iter, float min_update_squared,
                     const int32_t px_ref_level_0[2], double px_cur_level_0[2])
{
  uint8_t ref_patch[32 * 32];
  int16_t ref_patch_dx[32 * 32];
  int16_t ref_patch_dy[32 * 32];
  int converged = 0;

  for (int level = max_level; level >= min_level; --level) {
    const int patch_size = patch_sizes[level];
    const int halfpatch_size = patch_size / 2;
    const int scale = (1 << level);
    const orc_image* img_ref = &pyr_ref->level[level];
    const orc_image* img_cur = &pyr_cur->level[level];
    const int width = img_ref->width;
    const int height = img_ref->height;
    const int step = img_ref->pitch;
    const float px_ref_flt[2] = { (float)px_ref_level_0[0] / (float)scale - (float)halfpatch_size,
                                  (float)px_ref_level_0[1] / (float)scale - (float)halfpatch_size };
    const int px_ref[2] = { (int)px_ref_flt[0], (int)px_ref_flt[1] };
    const float px_ref_offset[2] = { px_ref_flt[0] - (float)px_ref[0], px_ref_flt[1] - (float)px_ref[1] };

    if (px_ref[0] < 1 || px_ref[1] < 1 || px_ref[0] >= width - patch_size - 1 || px_ref[1] >= height - patch_size - 1)
      continue;

    /* template, raw central differences (no 1/2), H = sum J J^T in float (exact integers) */
    uint8_t* it_patch = ref_patch;
    int16_t* it_dx = ref_patch_dx;
    int16_t* it_dy = ref_patch_dy;
    float H00 = 0, H01 = 0, H10 = 0, H11 = 0;
    for (int y = 0; y < patch_size; ++y) {
      const uint8_t* it = img_ref->data + (ptrdiff_t)(px_ref[1] + y) * step + px_ref[0];
      for (int x = 0; x < patch_size; ++x, ++it, ++it_patch, ++it_dx, ++it_dy) {
        *it_patch = *it;
        *it_dx = (int16_t)((int16_t)it[1] - it[-1]);
        *it_dy = (int16_t)((int16_t)it[step] - it[-step]);
        const float J0 = *it_dx, J1 = *it_dy;
        H00 += J0 * J0; H01 += J0 * J1; H10 += J1 * J0; H11 += J1 * J1;
      }
    }
    /* Eigen Matrix2f::inverse() */
    const float det = H00 * H11 - H10 * H01;
    const float invdet = 1.0f / det;
    const float Hi00 = H11 * invdet, Hi10 = -H10 * invdet, Hi01 = -H01 * invdet, Hi11 = H00 * invdet;

    float u = (float)(px_cur_level_0[0] / scale - halfpatch_size - px_ref_offset[0]);
    float v = (float)(px_cur_level_0[1] / scale - halfpatch_size - px_ref_offset[1]);
    int go_to_next_level = 0;
    const int SHIFT_BITS = 7;
    converged = 0;
    for (int iter = 0; iter < n_iter; ++iter) {
      if (isnan(u) || isnan(v)) return 0;
      go_to_next_level = 0;
      const int u_r = (int)floorf(u);
      const int v_r = (int)floorf(v);
      if (u_r < 0 || v_r < 0 || u_r >= width - patch_size || v_r >= height - patch_size) {
        go_to_next_level = 1;
        break;
      }
      const float subpix_x = u - u_r;
      const float subpix_y = v - v_r;
      const uint16_t wTL = (uint16_t)((1.0f - subpix_x) * (1.0f - subpix_y) * (1 << SHIFT_BITS));
      const uint16_t wTR = (uint16_t)(subpix_x * (1.0f - subpix_y) * (1 << SHIFT_BITS));
      const uint16_t wBL = (uint16_t)((1.0f - subpix_x) * subpix_y * (1 << SHIFT_BITS));
      const uint16_t wBR = (uint16_t)((1 << SHIFT_BITS) - wTL - wTR - wBL);

      const uint8_t* it_ref = ref_patch;
      const int16_t* it_ref_dx = ref_patch_dx;
      const int16_t* it_ref_dy = ref_patch_dy;
      float Jres0 = 0, Jres1 = 0;
      for (int y = 0; y < patch_size; ++y) {
        const uint8_t* it = img_cur->data + (ptrdiff_t)(v_r + y) * step + u_r;
        for (int x = 0; x < patch_size; ++x, ++it, ++it_ref, ++it_ref_dx, ++it_ref_dy) {
          const uint16_t cur = (uint16_t)SVO_DESCALE(wTL * it[0] + wTR * it[1] + wBL * it[step] + wBR * it[step + 1], SHIFT_BITS);
          const float res = (float)cur - *it_ref;
          Jres0 -= res * (*it_ref_dx);
          Jres1 -= res * (*it_ref_dy);
        }
      }
      /* update = Hinv * Jres * 2.0 */
      const float up0 = (Hi00 * Jres0 + Hi01 * Jres1) * 2.0f;
      const float up1 = (Hi10 * Jres0 + Hi11 * Jres1) * 2.0f;
      u += up0;
      v += up1;
      if (up0 * up0 + up1 * up1 < min_update_squared) {
        converged = 1;
        break;
      }
    }
    px_cur_level_0[0] = (double)((u + halfpatch_size + px_ref_offset[0]) * scale);
    px_cur_level_0[1] = (double)((v + halfpatch_size + px_ref_offset[1]) * scale);
    if (!converged && !go_to_next_level) return 0;
  }
  return converged;
}

/* batch form: the per-track loop of FeatureTracker::trackFrameBundle (feature_tracker.cpp:64-99);
 * ref_pyrs[i] = pyramid of the frame of the track's template observation */
void orc_klt_track_batch(const orc_pyramid* const* ref_pyrs, const orc_pyramid* cur_pyr, int max_level, int min_level,
                         const int32_t* patch_sizes, int n_iter, float min_update_squared, int n_tracks,
                         const int32_t* px_ref, double* px_cur, uint8_t* status)
{
  for (int i = 0; i < n_tracks; ++i) {
    double p[2] = { px_cur[2 * i], px_cur[2 * i + 1] };
    const int ok = orc_align_pyr_2d(ref_pyrs[i], cur_pyr, max_level, min_level, patch_sizes, n_iter,
                                    min_update_squared, &px_ref[2 * i], p);
    status[i] = ok ? 1 : 0;
    px_cur[2 * i] = p[0];
    px_cur[2 * i + 1] = p[1];
  }
}
