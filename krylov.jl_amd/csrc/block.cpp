// block.cpp -- block-GMRES entry points (panel kernels live in panel.hip).
#include "khip_internal.hpp"

using namespace khip;

extern "C" {

#define KHIP_TODO(name) \
  set_error(name ": not implemented yet in this build"); \
  return KHIP_ERR_UNSUPPORTED

int khip_panel_from_colmajor(khip_ctx *, int64_t, int, const double *, double *) { KHIP_TODO("panel_from_colmajor"); }
int khip_panel_to_colmajor(khip_ctx *, int64_t, int, const double *, double *) { KHIP_TODO("panel_to_colmajor"); }
int khip_panel_gemm_tn(khip_ctx *, int64_t, int, const double *, const double *, double *) { KHIP_TODO("panel_gemm_tn"); }
int khip_panel_gemm_nn(khip_ctx *, int64_t, int, double, const double *, const double *, double, double *) { KHIP_TODO("panel_gemm_nn"); }
int khip_panel_qr(khip_ctx *, int64_t, int, double *, double *) { KHIP_TODO("panel_qr"); }
int khip_panel_norm(khip_ctx *, int64_t, int, const double *, double *) { KHIP_TODO("panel_norm"); }
int khip_block_gmres_workspace_create(khip_ctx *, int64_t, int64_t, int, int, khip_block_gmres_workspace **) { KHIP_TODO("block_gmres_workspace_create"); }
int khip_block_gmres_workspace_destroy(khip_block_gmres_workspace *) { return KHIP_OK; }
int khip_block_gmres_solve(khip_block_gmres_workspace *, const khip_operator *, const double *, const khip_options *) { KHIP_TODO("block_gmres_solve"); }
int khip_block_gmres_get_X(khip_block_gmres_workspace *, double *) { KHIP_TODO("block_gmres_get_X"); }
const khip_stats *khip_block_gmres_stats(khip_block_gmres_workspace *) { return nullptr; }

}  // extern "C"
