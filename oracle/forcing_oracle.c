/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of the per-step loops of RDycore's forcing module that feed
 * the SWE operator (external water source, Dirichlet values).  Only tests/ may
 * load this.  Pinning: these loops are copies, one multiply and an argmin; they
 * are pinned by hand-computed known answers in tests/test_forcing_oracle.py
 * (the reference holds no fixture for them that can be read without PETSc).
 */
#include <math.h>

/* RDyForcingGetCurrentData, src/forcing/rdyforcing_dataset.c:32-67 */
void oracle_forcing_current_data(const double *data_ptr, int ndata, double cur_time, int temporally_interpolate, int *cur_data_idx, double *cur_data) {
  int    found  = 0;
  int    stride = 2;
  double time_up = 0, time_dn = 0, data_up = 0, data_dn = 0;
  for (int itime = 0; itime < ndata - 1; itime++) {
    time_dn = data_ptr[itime * stride];
    data_dn = data_ptr[itime * stride + 1];
    time_up = data_ptr[itime * stride + 2];
    data_up = data_ptr[itime * stride + 3];
    if (cur_time >= time_dn && cur_time < time_up) {
      found         = 1;
      *cur_data_idx = itime;
      break;
    }
  }
  if (!found) {
    *cur_data_idx = ndata - 1;
    *cur_data     = data_ptr[ndata * 2 - 1];
  } else if (temporally_interpolate) {
    *cur_data = (cur_time - time_dn) / (time_up - time_dn) * (data_up - data_dn) + data_dn;
  } else {
    *cur_data = data_dn;
  }
}

/* RDyForcingSetRasterData's loop, src/forcing/rdyforcing_dataset.c:303-310 */
void oracle_forcing_set_raster(const double *data_ptr, int offset, const int *data2mesh_idx, int ncells, double *rain) {
  double mm_per_hr_2_m_per_sec = 1.0 / (1000.0 * 3600.0);
  for (int icell = 0; icell < ncells; icell++) rain[icell] = data_ptr[data2mesh_idx[icell] + offset] * mm_per_hr_2_m_per_sec;
}

/* RDyForcingSetUnstructuredData's loop, src/forcing/rdyforcing_dataset.c:357-369 */
void oracle_forcing_set_unstructured(const double *data_ptr, int stride, const int *data2mesh_idx, int nelements, double *values) {
  int offset = 2;
  for (int icell = 0; icell < nelements; icell++) {
    int idx = data2mesh_idx[icell] * stride;
    for (int ii = 0; ii < stride; ii++) values[icell * stride + ii] = data_ptr[idx + ii + offset];
  }
}

/* RDyForcingCreateRasterDatasetMapping, src/forcing/rdyforcing_map.c:111-141 (data2mesh_idx is calloc'ed by the caller) */
void oracle_forcing_raster_map(int ncells, const double *mesh_xc, const double *mesh_yc, int ncols, int nrows, double cellsize, const double *data_xc,
                               const double *data_yc, int *data2mesh_idx) {
  for (int icell = 0; icell < ncells; icell++) {
    double min_dist = ((ncols > nrows ? ncols : nrows) + 1) * cellsize;
    double xc = mesh_xc[icell], yc = mesh_yc[icell];
    int    idx = 0;
    for (int irow = 0; irow < nrows; irow++) {
      for (int icol = 0; icol < ncols; icol++) {
        double dx = xc - data_xc[idx], dy = yc - data_yc[idx];
        double dist = pow(dx * dx + dy * dy, 0.5);
        if (dist < min_dist) {
          min_dist             = dist;
          data2mesh_idx[icell] = idx;
        }
        idx++;
      }
    }
  }
}

/* RDyForcingCreateUnstructuredDatasetMap, src/forcing/rdyforcing_map.c:77-104 */
void oracle_forcing_unstructured_map(int nelements, const double *mesh_xc, const double *mesh_yc, int ndata, const double *data_xc, const double *data_yc,
                                     int *data2mesh_idx) {
  for (int icell = 0; icell < nelements; icell++) {
    double xc = mesh_xc[icell], yc = mesh_yc[icell];
    double min_dist = 0.0;
    for (int kk = 0; kk < ndata; kk++) {
      double dx = xc - data_xc[kk], dy = yc - data_yc[kk];
      double dist = pow(dx * dx + dy * dy, 0.5);
      if (kk == 0 || dist < min_dist) {
        min_dist             = dist;
        data2mesh_idx[icell] = kk;
      }
    }
  }
}
