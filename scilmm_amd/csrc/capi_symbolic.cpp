// C-ABI entry points for the host-side symbolic phase (no GPU needed). Declared in include/scilmm_hip.h.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scilmm_hip.h"
#include "handles.h"
#include "host_threads.h"

using scilmm::Symbolic;

extern "C" {

int scilmm_symbolic_create(int32_t n, int32_t K, const int64_t* const* indptr, const int32_t* const* indices,
                           const int32_t* perm_in, const scilmm_options* opts, int32_t ngpus, scilmm_symbolic** out) {
  scilmm::use_host_threads();
  if (!out || n < 0 || K <= 0 || !indptr || !indices || ngpus < 1) return SCILMM_ERR_ARG;
  scilmm::SymbolicOptions o;
  if (opts) {
    o.ordering = opts->ordering;
    if (opts->relax_small >= 0) o.relax_small = opts->relax_small;
    if (opts->relax_w1 >= 0) o.relax_w1 = opts->relax_w1;
    if (opts->relax_w2 >= 0) o.relax_w2 = opts->relax_w2;
    if (opts->relax_z1 >= 0) o.relax_z1 = opts->relax_z1;
    if (opts->relax_z2 >= 0) o.relax_z2 = opts->relax_z2;
    if (opts->relax_z3 >= 0) o.relax_z3 = opts->relax_z3;
    if (opts->amd_dense != 0) o.amd_dense = opts->amd_dense;
    if (opts->max_width != 0) o.max_width = opts->max_width < 0 ? 0 : opts->max_width;
    if (opts->nd_oksep > 0) o.nd_oksep = opts->nd_oksep;
    if (opts->dense_relax != 0) o.dense_relax = o.dense_relax_wide = opts->dense_relax < 0 ? 0.0 : opts->dense_relax;  // explicit: one budget
  }
  if (const char* t = getenv("SCILMM_TUNING"))
    if (t[0] == '1')
      if (const char* e = getenv("SCILMM_TAIL_WIDE")) o.dense_relax_wide = atof(e);  // flop budget of a wide tail
  if (perm_in && !opts) o.ordering = 2;
  scilmm_symbolic* h = new scilmm_symbolic();
  h->S = scilmm::symbolic_analyze(n, K, indptr, indices, perm_in, o);
  *out = h;
  if (!h->S->error.empty()) {
    h->err = h->S->error;
    return SCILMM_ERR_ARG;
  }
  return SCILMM_OK;
}

int scilmm_symbolic_save(const scilmm_symbolic* h, const char* path, uint64_t key) {
  if (!h || !h->S || !path) return SCILMM_ERR_ARG;
  return scilmm::symbolic_save(*h->S, path, key) ? SCILMM_OK : SCILMM_ERR_STATE;
}

int scilmm_symbolic_load(const char* path, uint64_t key, scilmm_symbolic** out) {
  scilmm::use_host_threads();
  if (!path || !out) return SCILMM_ERR_ARG;
  scilmm::Symbolic* S = scilmm::symbolic_load(path, key);
  if (!S) return SCILMM_ERR_STATE;  // no file, another key / build, or a damaged image: the caller analyses afresh
  scilmm_symbolic* h = new scilmm_symbolic();
  h->S = S;
  *out = h;
  return SCILMM_OK;
}

int scilmm_symbolic_release_host_maps(scilmm_symbolic* h) {
  if (!h || !h->S) return SCILMM_ERR_ARG;
  if (!h->device) return SCILMM_ERR_STATE;  // the device plan reads them: call after the first numeric call
  scilmm::Symbolic& S = *h->S;
  std::vector<int64_t>().swap(S.asm_dst);
  std::vector<int32_t>().swap(S.pat_row);
  for (auto& v : S.val_slot) std::vector<int64_t>().swap(v);
  for (auto& v : S.val_src) std::vector<int64_t>().swap(v);
  h->maps_released = true;
  return SCILMM_OK;
}

int scilmm_symbolic_info(const scilmm_symbolic* h, scilmm_info* info) {
  if (!h || !h->S || !info) return SCILMM_ERR_ARG;
  const Symbolic& S = *h->S;
  info->n = S.n;
  info->K = S.K;
  info->nsuper = S.nsuper;
  info->nlevels = S.nlevels;
  info->nnzL = S.nnzL;
  info->nnzL_stored = S.nnzL_stored;
  info->nnz_pattern = S.nnz_pattern;
  info->flops = S.flops;
  info->n_rows_total = (int64_t)S.sn_rows.size();
  info->n_updates = (int64_t)S.upd_src.size();
  info->update_flops = S.update_flops - S.update_flops_pad;  // algorithmic: without the dense-tail padding
  info->update_flops_executed = S.update_flops;
  info->dense_first = S.dense_first;
  info->dense_flops = S.dense_flops;
  info->solve_flops_per_rhs = 4.0 * (double)S.nnzL_stored;
  return SCILMM_OK;
}

#define GET(name, vec)                                                     \
  if (!std::strcmp(what, name)) {                                          \
    if (out) std::memcpy(out, S.vec.data(), S.vec.size() * sizeof(S.vec[0])); \
    *count = (int64_t)S.vec.size();                                        \
    return SCILMM_OK;                                                      \
  }

int scilmm_symbolic_get(const scilmm_symbolic* h, const char* what, void* out, int64_t* count) {
  scilmm::use_host_threads();
  if (!h || !h->S || !what || !count) return SCILMM_ERR_ARG;
  if (!h->S->combos_built && !std::strncmp(what, "combo_", 6)) scilmm::build_tile_combos(h->S, nullptr);
  const Symbolic& S = *h->S;
  if (!std::strcmp(what, "dense_first")) {
    if (out) *(int32_t*)out = S.dense_first;
    *count = 1;
    return SCILMM_OK;
  }
  // value-assembly maps of input matrix k: "val_slot:k" / "val_src:k" (pattern slot <- index into data_k)
  if (!std::strncmp(what, "val_slot:", 9) || !std::strncmp(what, "val_src:", 8)) {
    const bool slot = what[4] == 's' && what[5] == 'l';
    const int k = std::atoi(what + (slot ? 9 : 8));
    if (k < 0 || k >= S.K) return SCILMM_ERR_ARG;
    const std::vector<int64_t>& v = slot ? S.val_slot[k] : S.val_src[k];
    if (out) std::memcpy(out, v.data(), v.size() * sizeof(int64_t));
    *count = (int64_t)v.size();
    return SCILMM_OK;
  }
  GET("perm", perm)
  GET("iperm", iperm)
  GET("parent", parent)
  GET("colcount", colcount)
  GET("sn_start", sn_start)
  GET("sn_parent", sn_parent)
  GET("sn_rowptr", sn_rowptr)
  GET("sn_rows", sn_rows)
  GET("sn_loff", sn_loff)
  GET("sn_level", sn_level)
  GET("level_ptr", level_ptr)
  GET("level_fronts", level_fronts)
  GET("asm_dst", asm_dst)
  GET("diag_dst", diag_dst)
  GET("upd_ptr", upd_ptr)
  GET("upd_src", upd_src)
  GET("upd_p0", upd_p0)
  GET("upd_p1", upd_p1)
  GET("tile_base", tile_base)
  GET("tile_front", tile_front)
  GET("combo_ptr", combo_ptr)
  GET("combo_pair", combo_pair)
  GET("combo_ta", combo_ta)
  GET("combo_tb", combo_tb)
  GET("combo_ip0", combo_ip0)
  GET("upd_jp0", upd_jp0)
  GET("level_tile_ptr", level_tile_ptr)
  GET("level_tiles", level_tiles)
  GET("level_pair_ptr", level_pair_ptr)
  GET("level_pairs", level_pairs)
  GET("pat_colptr", pat_colptr)
  GET("pat_row", pat_row)
  GET("inv_off", inv_off)
  GET("tail_blk_ptr", tail_blk_ptr)
  GET("tail_blk", tail_blk)
  GET("child_ptr", child_ptr)
  GET("child_idx", child_idx)
  return SCILMM_ERR_ARG;
}

int scilmm_order(int32_t n, const int64_t* indptr, const int32_t* indices, int32_t method, int32_t* perm_out) {
  scilmm::use_host_threads();
  if (n < 0 || !indptr || !indices || !perm_out) return SCILMM_ERR_ARG;
  // symmetrised adjacency without the diagonal
  std::vector<int64_t> gptr((size_t)n + 1, 0);
  for (int32_t i = 0; i < n; ++i)
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
      const int32_t j = indices[e];
      if (j < 0 || j >= n || j >= i) continue;
      gptr[i + 1]++;
      gptr[j + 1]++;
    }
  for (int32_t i = 0; i < n; ++i) gptr[i + 1] += gptr[i];
  std::vector<int32_t> gidx((size_t)gptr[n]);
  {
    std::vector<int64_t> fill(gptr.begin(), gptr.end() - 1);
    for (int32_t i = 0; i < n; ++i)
      for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
        const int32_t j = indices[e];
        if (j < 0 || j >= n || j >= i) continue;
        gidx[fill[i]++] = j;
        gidx[fill[j]++] = i;
      }
  }
  if (method == 0) {
    scilmm::amd_order(n, gptr.data(), gidx.data(), perm_out, 10.0);
    return SCILMM_OK;
  }
  if (method == 1 || method == 2) {  // 1: nested dissection with the default acceptance threshold; 2: always dissect
    scilmm::NdOptions o;
    if (method == 2) o.oksep = 1.0;
    scilmm::nd_order(n, gptr.data(), gidx.data(), perm_out, o, nullptr);
    return SCILMM_OK;
  }
  return SCILMM_ERR_ARG;
}

int scilmm_fill_count(int32_t n, const int64_t* indptr, const int32_t* indices, const int32_t* perm, int64_t* nnzL,
                      double* flops, int32_t* max_colcount) {
  scilmm::use_host_threads();
  if (n < 0 || !indptr || !indices) return SCILMM_ERR_ARG;
  scilmm::fill_count(n, indptr, indices, perm, nnzL, flops, max_colcount, nullptr);
  return SCILMM_OK;
}

const char* scilmm_symbolic_error(const scilmm_symbolic* h) { return h ? h->err.c_str() : "null handle"; }

void scilmm_symbolic_free(scilmm_symbolic* h) {
  if (!h) return;
  if (h->device && h->device_free) h->device_free(h->device);
  delete h->S;
  delete h;
}

}  // extern "C"
