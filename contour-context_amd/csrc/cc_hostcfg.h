// Host-side derivation of the kernel constants from the public config, mirroring the reference's
// ContourManager constructor (contour_mng.h:478-498) and hashPointToImage padding (:450-452).
#pragma once
#include "cc_dev.h"

static inline int cc_make_dev_cfg(const cc_manager_cfg_t *m, cc_dev_cfg *c) {
  if (!m || !c) return -1;
  if (m->n_row % 2 != 0 || m->n_col % 2 != 0) return -1;                   // CHECK(cfg_.n_col_ % 2 == 0)
  if (m->n_row <= 0 || m->n_col <= 0 || m->n_row * m->n_col > CC_MAX_CELLS) return -1;
  if (m->n_row > 255 || m->n_col > 127 * 2 + 1) return -1;                  // u8 bbox / 7-bit block columns
  for (int i = 1; i < CC_NLEV; i++)
    if (!(m->lv_grads[i] > m->lv_grads[i - 1])) return -1;                   // nested level sets
  if (m->piv_firsts < 1 || m->piv_firsts > CC_NPIV || m->dist_firsts < 1 || m->dist_firsts > CC_NDIST) return -1;
  if (m->min_cont_cell_cnt < 1) return -1;
  if (!(m->reso_row > 0.f) || !(m->reso_col > 0.f)) return -1;              // also rejects NaN
  const float padding = 1e-2f;
  const float x_min = -(float)(m->n_row / 2) * m->reso_row, x_max = -x_min;
  const float y_min = -(float)(m->n_col / 2) * m->reso_col, y_max = -y_min;
  c->x_lo = x_min + padding;
  c->x_hi = x_max - padding;
  c->y_lo = y_min + padding;
  c->y_hi = y_max - padding;
  c->blind_sq = m->blind_sq;
  c->reso_row = m->reso_row;
  c->reso_col = m->reso_col;
  c->lidar_height = m->lidar_height;
  c->n_row = m->n_row;
  c->n_col = m->n_col;
  c->half_row = m->n_row / 2;
  c->half_col = m->n_col / 2;
  c->n_cell = m->n_row * m->n_col;
  for (int i = 0; i < CC_NLEV; i++) c->lv_grads[i] = m->lv_grads[i];
  c->min_cont_key_cnt = m->min_cont_key_cnt;
  c->min_cont_cell_cnt = m->min_cont_cell_cnt;
  c->piv_firsts = m->piv_firsts;
  c->dist_firsts = m->dist_firsts;
  c->roi_radius = m->roi_radius;
  c->min_cell_cov = m->min_cell_cov;
  c->point_sigma = m->point_sigma;
  c->com_bias_thres = m->com_bias_thres;
  {
    int er = 0, ec = 0;
    const bool p2 = frexpf(m->reso_row, &er) == 0.5f && frexpf(m->reso_col, &ec) == 0.5f && er > -100 && er < 100 && ec > -100 && ec < 100;
    c->reso_pow2 = p2 ? 1 : 0;
    c->inv_row = 1.0f / m->reso_row;
    c->inv_col = 1.0f / m->reso_col;
  }
  return 0;
}
