// Host side of the numeric phase: device-resident symbolic arrays and A_k values, the level-scheduled
// launch sequences (factorize / solve / L*R / quadratic forms) and the numeric C-ABI entry points of
// include/scilmm_hip.h.  Everything runs on one HIP stream per symbolic handle; host entry points
// synchronise only where they hand data back to the caller.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scilmm_hip.h"
#include "kernels.hip.h"
#include "cellplan.hip.h"
#include "handles.h"
#include "host_threads.h"

using namespace scilmm;

namespace {

struct Dev {
  int device = 0;                  // HIP device the handle was created on; every entry point runs on it (DevGuard)
  hipStream_t stream = nullptr;
  DevSym v{};
  std::vector<void*> allocs;
  int32_t* d_level_tiles = nullptr;
  int32_t* d_level_fronts = nullptr;
  int32_t* d_level_pairs = nullptr;
  int32_t* d_all_fronts = nullptr;  // multi-GPU: Symbolic::level_fronts unfiltered (the forward sweep solves every
                                    // tail block on every rank: the inverse diagonal blocks are replicated)
  std::vector<double*> vals;       // per matrix: pattern-order values or diagonal values
  std::vector<uint8_t> have_vals;
  double* W = nullptr;             // n x RPMAX workspaces (permuted right-hand sides)
  double* X = nullptr;
  double* IO = nullptr;            // staging for host<->device dense transfers
  size_t io_cap = 0;
  double* partial = nullptr;
  int64_t nwaves_quad = 0;
  double* d_out = nullptr;         // RPMAX doubles
  bool use_mfma = true;
  bool trsm_lite = true;           // k_trsm_lite instead of k_trsm<true> (SCILMM_TUNING=1 SCILMM_TRSM_LITE=0: the round-1 kernel)
  hipEvent_t ev[8];
  scilmm_timing timing{};
  bool quad_pending = false;           // a scilmm_quadforms_dev call whose timer has not been read yet
  bool attrs_set = false;
  // update-kernel plan: flattened combo descriptors, per-level work items (split-K), partial slots
  ComboDesc* d_combos = nullptr;
  UpdWork* d_work = nullptr;
  std::vector<int64_t> work_ptr;   // [nlevels+1] LATE items (descendants one level below the target): main stream
  UpdWork* d_work_early = nullptr; // EARLY items (older descendants): side stream, overlaps the previous level
  std::vector<int64_t> early_ptr;  // [nlevels+1]
  int64_t max_slots = 0;           // partial slabs per scratch half (scratch is double-buffered by level parity)
  hipStream_t side = nullptr;
  hipStream_t side2 = nullptr;     // early updates alternate between two side streams (their tails overlap)
  hipStream_t side3 = nullptr;     // optional third one (SCILMM_SIDE_STREAMS=3)
  int nside = 2;
  bool serial_early = false;       // profiling mode 2: every early launch on ONE side stream (launch durations do not overlap)
  std::vector<hipEvent_t> lev_ev;  // 2 per level: [2l] = level l finished, [2l+1] = early update of level l finished
  hipEvent_t ev_asm = nullptr;
  hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;  // main <-> comm stream hand-offs (multi-GPU)
  int32_t* d_tile_pslot = nullptr;   // late partial slabs of a tile (main stream)
  int32_t* d_tile_pnseg = nullptr;
  int32_t* d_tile_pslot_e = nullptr; // early partial slabs of a tile (side stream)
  int32_t* d_tile_pnseg_e = nullptr;
  int32_t* d_red_tiles_e = nullptr;
  std::vector<int64_t> red_ptr_e;
  std::vector<int64_t> lev_cost_e, lev_cost_l;  // per-level dense update cost units (diagnostics)
  double* scratch = nullptr;       // max slots per level * TM*NB doubles
  int32_t* d_red_tiles = nullptr;  // tiles that carry partial slabs, grouped by level
  // cell-wise path for small update pairs: set 0 = early (side stream), set 1 = late (main stream)
  struct CellSet {
    int64_t* dst = nullptr;
    int64_t* grp = nullptr;
    int64_t* srct = nullptr;
    int64_t* srcq = nullptr;
    int32_t* md = nullptr;
    int32_t* wd = nullptr;
    std::vector<int64_t> level_ptr;    // [nlevels+1] over unique target cells
    std::vector<int64_t> level_short;  // [nlevels] short groups (listed first) per level
  } cellset[3];  // 0 = early (side streams), 1 = late (main stream); 2 unused (kept for the device cell plan's key layout)
  int64_t n_dense_combos = 0, n_sparse_combos = 0, n_cells = 0;
  std::vector<int64_t> red_ptr;    // [nlevels+1]
  bool profiling = false;
  int ablate = 0;
  // multi-GPU: the fronts of the dense tail (>= dist_first) are owned 1-D block-cyclically.  A rank STORES the prelude
  // (replicated), its own tail panels and a ring of dist_G slots through which the other ranks' panels pass (fan-out:
  // a received panel is applied to every own target that needs it and then dropped) -- see DistLayout.
  int32_t rank = 0, world = 1;
  int32_t dist_first = 0;               // first distributed front (nsuper: nothing is distributed)
  int32_t dist_Wg = 8, dist_G = 32;     // source-group size of the batched updates; ring slots
  std::vector<int64_t> loff;            // [nsuper+1] rank-local panel offsets (== Symbolic::sn_loff when world == 1)
  int64_t nL_local = 0;                 // doubles of rank-local panel storage (prelude + own tail + ring)
  std::vector<uint8_t> keep_front;      // [nsuper] this rank computes the panel of front s
  std::vector<int32_t> tail_of_level;   // [nlevels] the distributed front of level l, or -1
  // level lists without the tail fronts of other ranks (== the Symbolic's when world == 1)
  std::vector<int32_t> lv_ptr, lv_fronts, lv_tiles, lv_pairs;  // lv_ptr: [nlevels+1] into lv_fronts
  std::vector<int64_t> lv_tile_ptr, lv_tile_mid, lv_pair_ptr;  // lv_tile_mid[l]: first tile of the level's own distributed panel
  int32_t* d_lmul_tiles = nullptr;      // tiles this rank multiplies in L*R (own tail; the prelude on rank 0 only)
  int64_t n_lmul_tiles = 0;
  DenseWork* d_dwork_b = nullptr;       // batch items of the distributed tail
  std::vector<int64_t> dbatch_ptr;      // [ngroups+1]
  std::vector<hipEvent_t> batch_ev;     // [ngroups] batch g applied to all own targets
  std::vector<hipEvent_t> bpev;         // profiling: [2 ngroups] begin / end of batch g on the batches' stream
  std::vector<int32_t> last_own_level;  // [ngroups] level of this rank's last own tail front in group g, or -1
  hipStream_t bstream = nullptr;        // the batches' stream
  hipStream_t comm = nullptr;           // caller-owned stream the collectives are issued on
  std::vector<hipEvent_t> done_ev;      // per level with a distributed front: this rank's kernels of the level finished
  double* ACC = nullptr;                // forward sweep: contributions of this rank's own tail panels, n x RPMAX (dist)
  // selected inverse (scilmm_selected_inverse): column -> front, per-front offset of Y inside the per-level scratch
  int32_t* d_col_front = nullptr;
  int64_t* d_yoff = nullptr;
  double* d_ybuf = nullptr;
  int32_t* d_sinv_pre_tiles = nullptr;        // tiles of the non-tail fronts, by level (k_sinv_w)
  std::vector<int64_t> sinv_pre_ptr;          // [nlevels+1]
  std::vector<int32_t> sinv_tail_front;       // [nlevels] the dense-tail front of the level, or -1
  int32_t* d_sinv_tail_fronts = nullptr;      // the tail fronts, one per entry (k_sinv_zero takes a list)
  SinvWork* d_sinv_work = nullptr;            // items of k_sinv_tail, grouped by tail front
  std::vector<int64_t> sinv_work_ptr;         // [ntail+1]
  bool work_external = false;           // W / X / ACC belong to the caller (scilmm_dist_set_work)
  // dense tail (Symbolic::dense_first): implicit work items of k_dense, early (side streams) and late (main stream)
  // prelude -> tail contributions in descendant coordinates (k_outside; fp64 atomics): one launch between the last
  // prelude level and the first tail level
  bool outside_on = false;
  int32_t tail_level = 0;               // level of the first tail front
  std::vector<uint8_t> outside_desc;    // [nsuper] descendant handled by k_outside
  OutsideWork* d_owork = nullptr;
  int32_t* d_grp_next = nullptr;   // [nsuper] k_outside: next descendant with the same tail rows as this one, or -1
  int32_t* d_grp_t0 = nullptr;     // [nsuper] its first tail row
  int64_t n_owork = 0;
  int32_t* d_tail_front = nullptr;
  uint8_t* d_keep_front = nullptr;
  hipStream_t outside_st = nullptr;
  // PROGRESSIVE k_outside: the items are sorted by the FIRST tail panel they touch and cut into chunks; chunk g is one launch
  // and one event, and whatever touches tail panel f (its early / late updates, its potrf) waits only for the last chunk that
  // holds an item reaching f or an earlier panel -- left-looking updates write nothing but the level's own panel, so the rest
  // of the atomic contributions (to LATER panels only) overlaps with the first levels of the tail, which are chain-bound
  std::vector<int64_t> ochunk_ptr;       // [nchunks + 1] into d_owork
  std::vector<hipEvent_t> out_evs;       // [nchunks]
  std::vector<int32_t> out_wait_chunk;   // [nlevels] chunk the level's tail front waits for, -1: none
  bool dense_on = false;
  int front_bits = 64;                  // 32: dense-tail products on the fp32 matrix pipe (k_dense32), sums in fp64
  double* d_zeros = nullptr;            // 2 KiB of zeros: source of the B k-rows past a descendant's end (k_dense_b)
  DenseWork* d_dwork_e = nullptr;
  DenseWork* d_dwork_l = nullptr;
  std::vector<int64_t> dwork_e_ptr, dwork_l_ptr;  // [nlevels+1]
  // distributed tail, look-ahead split of an own target's late update: items [dwork_l_ptr[l], dwork_l_mid[l]) take the sources
  // that arrived EARLIER (they run while the newest source panel is still being factored / broadcast), items
  // [dwork_l_mid[l], dwork_l_ptr[l+1]) the newest source alone (== dwork_l_ptr[l+1] where nothing is split)
  std::vector<int64_t> dwork_l_mid;               // [nlevels]
  int64_t n_late_split = 0;                       // levels of the last factorization whose late launch was split
  int look_depth = 2;      // "late" = descendants at most this many levels below the target; older ones are "early"
  int rhs_pending = -1;            // mode of the last run_rhs whose events have not been read yet
  // dense-chain sweeps (k_chain): the last chain_T levels are single fronts whose mutual update pairs are contiguous
  int32_t chain_T = 0, chain_l0 = 0;
  int32_t* d_chain = nullptr;
  int32_t* d_colmap = nullptr;     // forward: column -> row maps of the non-contiguous chain pairs
  int32_t* d_cf_ptr = nullptr;     // forward: pairs of chain target i (descendants ascending)
  ChainPair* d_cf = nullptr;
  int32_t* d_cb_ptr = nullptr;     // backward: pairs of chain descendant i (targets descending)
  ChainPair* d_cb = nullptr;
  int64_t* d_cg_ptr = nullptr;     // backward: pairs (chain target, non-chain descendant) grouped by descendant
  int32_t* d_cg_pairs = nullptr;
  int64_t chain_groups = 0;
  // long groups are cut into row slices that write partial sums; k_push_fold adds them up in fixed order
  int32_t* d_cg_slot = nullptr;      // per work item: partial slot or -1 (subtract straight from X)
  int32_t* d_fold = nullptr;         // triples (descendant, first slot, slices)
  int64_t n_fold = 0;
  double* d_push_partial = nullptr;  // [slots][NB][RPMAX]
  int32_t* d_chain_flags = nullptr;  // [chain_T * RPMAX/CW] epoch stamps
  int32_t* d_chain_err = nullptr;    // [0] error flag, [1] progress beacon, [2] ticket counter of the running sweep
  int32_t* h_chain_err = nullptr;    // pinned mirror of [0], refreshed by a queued copy after every solve
  int32_t chain_epoch = 0;
  std::vector<hipEvent_t> pev;     // 4 events per level when profiling
};

// Schedule / tuning switches (SCILMM_LOOK_DEPTH, SCILMM_CELL_LIMIT, ...) are honoured only when SCILMM_TUNING=1 is set
// as well, so that a stray variable in a production environment cannot change the schedule.  Every value of every
// such switch gives the same factor to rounding (parity-tested); switches that would change RESULTS (the timing
// ablations) exist only in builds with -DSCILMM_DIAG.  SCILMM_VERBOSE / SCILMM_LEVEL_DUMP only print.
inline const char* tune_env(const char* name) {
  const char* t = getenv("SCILMM_TUNING");  // read on every call: tests switch it on and off inside one process
  return (t && t[0] == '1') ? getenv(name) : nullptr;
}

#define HIPCHK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (call);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      sym->err = std::string(#call) + ": " + hipGetErrorString(_e);                                    \
      return SCILMM_ERR_DEVICE;                                                                        \
    }                                                                                                  \
  } while (0)

// Makes the handle's device current for the duration of an entry point and restores the caller's device afterwards
// (a handle may be used from a thread whose current device is a different one).
struct DevGuard {
  int prev = -1;
  bool switched = false;
  explicit DevGuard(const scilmm_symbolic* sym) {
    const Dev* D = sym ? (const Dev*)sym->device : nullptr;
    if (D) enter(D->device);
  }
  explicit DevGuard(int device) { enter(device); }
  void enter(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    if (!switched) (void)hipGetLastError();  // never leave a failed hipSetDevice behind as the thread's "last error"
  }
  ~DevGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

template <typename T>
int upload(scilmm_symbolic* sym, Dev* D, const std::vector<T>& h, const T** out) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc(&p, bytes));
  D->allocs.push_back(p);
  if (!h.empty()) HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T*)p;
  return SCILMM_OK;
}

void dev_free(void* p) {
  Dev* D = (Dev*)p;
  if (!D) return;
  for (void* a : D->allocs) (void)hipFree(a);
  for (double* v : D->vals)
    if (v) (void)hipFree(v);
  if (D->W && !D->work_external) (void)hipFree(D->W);
  if (D->X && !D->work_external) (void)hipFree(D->X);
  if (D->IO) (void)hipFree(D->IO);
  if (D->partial) (void)hipFree(D->partial);
  if (D->d_out) (void)hipFree(D->d_out);
  for (auto& e : D->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : D->pev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : D->lev_ev)
    if (e) (void)hipEventDestroy(e);
  if (D->ev_asm) (void)hipEventDestroy(D->ev_asm);
  if (D->ev_x0) (void)hipEventDestroy(D->ev_x0);
  if (D->ev_x1) (void)hipEventDestroy(D->ev_x1);
  if (D->side) (void)hipStreamDestroy(D->side);
  if (D->side2) (void)hipStreamDestroy(D->side2);
  if (D->side3) (void)hipStreamDestroy(D->side3);
  if (D->outside_st) (void)hipStreamDestroy(D->outside_st);
  for (auto& e : D->out_evs)
    if (e) (void)hipEventDestroy(e);
  if (D->h_chain_err) (void)hipHostFree(D->h_chain_err);
  if (D->stream) (void)hipStreamDestroy(D->stream);
  for (auto& e : D->done_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : D->batch_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : D->bpev)
    if (e) (void)hipEventDestroy(e);
  if (D->bstream) (void)hipStreamDestroy(D->bstream);
  delete D;
}

// Expand the small combos to the cell lists of k_sparse_cells on the device (see cellplan.hip.h).
// (ccparts: the small combos as the classification threads produced them, in tile order; they are uploaded part by
// part -- concatenating 17 GB of them on the host first cost seconds of every first evaluation at the 1M config)
int build_cells_device(scilmm_symbolic* sym, Dev* D, const std::vector<const std::vector<CellCombo>*>& ccparts, int32_t NL,
                       int64_t* ngroups_total, int64_t* n_early) {
  const Symbolic& S = *sym->S;
  int64_t ncc = 0;
  for (auto* pv : ccparts) ncc += (int64_t)pv->size();
  std::vector<int64_t> off((size_t)ncc + 1, 0);
  {
    int64_t c = 0;
    for (auto* pv : ccparts)
      for (const CellCombo& q : *pv) {
        off[(size_t)c + 1] = off[(size_t)c] + (int64_t)q.nt * q.nq;
        ++c;
      }
  }
  const int64_t total = off[(size_t)ncc];
  auto dmalloc = [&](void** p, size_t bytes) -> int {
    HIPCHK(hipMalloc(p, std::max<size_t>(bytes, 8)));
    return SCILMM_OK;
  };
  int st;
  std::vector<void*> tmp;  // freed on exit
  auto tmalloc = [&](void** p, size_t bytes) -> int {
    int r = dmalloc(p, bytes);
    if (r == SCILMM_OK) tmp.push_back(*p);
    return r;
  };
  struct Cleanup {
    std::vector<void*>& v;
    ~Cleanup() { for (void* p : v) (void)hipFree(p); }
  } cleanup{tmp};
  for (int c = 0; c < 3; ++c) {
    D->cellset[c].level_ptr.assign(S.nlevels + 1, 0);
    D->cellset[c].level_short.assign(std::max<int32_t>(NL, 1), 0);
  }
  *ngroups_total = 0;
  *n_early = 0;
  D->n_cells = 0;
  if (total == 0) {
    void* d8 = nullptr;
    if ((st = dmalloc(&d8, 64)) != SCILMM_OK) return st;
    D->allocs.push_back(d8);
    HIPCHK(hipMemset(d8, 0, 64));
    for (int c = 0; c < 3; ++c) {
      Dev::CellSet& CS = D->cellset[c];
      CS.dst = CS.grp = CS.srct = CS.srcq = (int64_t*)d8;
      CS.md = CS.wd = (int32_t*)d8;
    }
    return SCILMM_OK;
  }
  CellCombo* d_cc = nullptr; int64_t* d_off = nullptr;
  unsigned long long *key = nullptr, *skey = nullptr, *d_ninv = nullptr;
  uint32_t *idx = nullptr, *sidx = nullptr;
  int64_t *cst = nullptr, *csq = nullptr; int32_t *cmd = nullptr, *cwd = nullptr;
  if ((st = tmalloc((void**)&d_cc, sizeof(CellCombo) * (size_t)ncc)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&d_off, sizeof(int64_t) * (size_t)(ncc + 1))) != SCILMM_OK) return st;
  {
    size_t at = 0;
    for (auto* pv : ccparts) {
      if (!pv->empty()) HIPCHK(hipMemcpy(d_cc + at, pv->data(), sizeof(CellCombo) * pv->size(), hipMemcpyHostToDevice));
      at += pv->size();
    }
  }
  HIPCHK(hipMemcpy(d_off, off.data(), sizeof(int64_t) * (size_t)(ncc + 1), hipMemcpyHostToDevice));
  if ((st = tmalloc((void**)&key, 8 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&skey, 8 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&idx, 4 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&sidx, 4 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&cst, 8 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&csq, 8 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&cmd, 4 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&cwd, 4 * (size_t)total)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&d_ninv, 8)) != SCILMM_OK) return st;
  HIPCHK(hipMemset(d_ninv, 0, 8));
  hipStream_t s0 = D->stream;
  hipLaunchKernelGGL(k_emit_cells, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, s0, total, ncc,
                     (const CellCombo*)d_cc, (const int64_t*)d_off, D->v.sn_rows, key, idx, cst, csq, cmd, cwd, d_ninv);
  void* cubtmp = nullptr;
  size_t cubbytes = 0, need = 0;
  auto ensure_tmp = [&](size_t bytes) -> int {
    if (bytes <= cubbytes) return SCILMM_OK;
    if ((st = tmalloc(&cubtmp, bytes)) != SCILMM_OK) return st;  // the smaller one is freed at exit as well
    cubbytes = bytes;
    return SCILMM_OK;
  };
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, key, skey, idx, sidx, total, 0, 62, s0));
  if ((st = ensure_tmp(need)) != SCILMM_OK) return st;
  need = cubbytes;
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(cubtmp, need, key, skey, idx, sidx, total, 0, 62, s0));
  unsigned long long ninv = 0;
  HIPCHK(hipMemcpyAsync(&ninv, d_ninv, 8, hipMemcpyDeviceToHost, s0));
  HIPCHK(hipStreamSynchronize(s0));
  const int64_t nvalid = total - (int64_t)ninv;
  D->n_cells = nvalid;
  // groups of equal key (= equal class, level, target address)
  unsigned long long* ukey = nullptr; int64_t* ucnt = nullptr; int64_t* ustart = nullptr; int64_t* d_ng = nullptr;
  if ((st = tmalloc((void**)&ukey, 8 * (size_t)std::max<int64_t>(nvalid, 1))) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&ucnt, 8 * (size_t)std::max<int64_t>(nvalid, 1))) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&d_ng, 8)) != SCILMM_OK) return st;
  HIPCHK(hipMemset(d_ng, 0, 8));
  int64_t ng = 0;
  if (nvalid > 0) {
    need = 0;
    HIPCHK(hipcub::DeviceRunLengthEncode::Encode(nullptr, need, skey, ukey, ucnt, d_ng, (int)nvalid, s0));
    if ((st = ensure_tmp(need)) != SCILMM_OK) return st;
    need = cubbytes;
    HIPCHK(hipcub::DeviceRunLengthEncode::Encode(cubtmp, need, skey, ukey, ucnt, d_ng, (int)nvalid, s0));
    HIPCHK(hipMemcpyAsync(&ng, d_ng, 8, hipMemcpyDeviceToHost, s0));
    HIPCHK(hipStreamSynchronize(s0));
  }
  *ngroups_total = ng;
  // final arrays (kept): group targets, entry offsets, entries
  int64_t *udst = nullptr, *grp2 = nullptr, *ost = nullptr, *osq = nullptr; int32_t *omd = nullptr, *owd = nullptr;
  if ((st = dmalloc((void**)&udst, 8 * (size_t)std::max<int64_t>(ng, 1))) != SCILMM_OK) return st; D->allocs.push_back(udst);
  if ((st = dmalloc((void**)&grp2, 8 * (size_t)(ng + 1))) != SCILMM_OK) return st; D->allocs.push_back(grp2);
  if ((st = dmalloc((void**)&ost, 8 * (size_t)std::max<int64_t>(nvalid, 1))) != SCILMM_OK) return st; D->allocs.push_back(ost);
  if ((st = dmalloc((void**)&osq, 8 * (size_t)std::max<int64_t>(nvalid, 1))) != SCILMM_OK) return st; D->allocs.push_back(osq);
  if ((st = dmalloc((void**)&omd, 4 * (size_t)std::max<int64_t>(nvalid, 1))) != SCILMM_OK) return st; D->allocs.push_back(omd);
  if ((st = dmalloc((void**)&owd, 4 * (size_t)std::max<int64_t>(nvalid, 1))) != SCILMM_OK) return st; D->allocs.push_back(owd);
  std::vector<unsigned int> counters((size_t)3 * NL * 2, 0u);
  if (ng > 0) {
    if ((st = tmalloc((void**)&ustart, 8 * (size_t)ng)) != SCILMM_OK) return st;
    need = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, ucnt, ustart, (int)ng, s0));
    if ((st = ensure_tmp(need)) != SCILMM_OK) return st;
    need = cubbytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(cubtmp, need, ucnt, ustart, (int)ng, s0));
    unsigned long long *gkey = nullptr, *gkey_s = nullptr; uint32_t *gidx = nullptr, *order = nullptr; int64_t* cnt2 = nullptr;
    unsigned int* d_counters = nullptr;
    if ((st = tmalloc((void**)&gkey, 8 * (size_t)ng)) != SCILMM_OK) return st;
    if ((st = tmalloc((void**)&gkey_s, 8 * (size_t)ng)) != SCILMM_OK) return st;
    if ((st = tmalloc((void**)&gidx, 4 * (size_t)ng)) != SCILMM_OK) return st;
    if ((st = tmalloc((void**)&order, 4 * (size_t)ng)) != SCILMM_OK) return st;
    if ((st = tmalloc((void**)&cnt2, 8 * (size_t)ng)) != SCILMM_OK) return st;
    if ((st = tmalloc((void**)&d_counters, 4 * counters.size())) != SCILMM_OK) return st;
    HIPCHK(hipMemsetAsync(d_counters, 0, 4 * counters.size(), s0));
    const unsigned gb = (unsigned)((ng + 255) / 256);
    hipLaunchKernelGGL(k_group_keys, dim3(gb), dim3(256), 0, s0, ng, (const unsigned long long*)ukey, (const int64_t*)ucnt,
                       (int64_t)16, gkey, gidx);
    need = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, gkey, gkey_s, gidx, order, ng, 0, 62, s0));
    if ((st = ensure_tmp(need)) != SCILMM_OK) return st;
    need = cubbytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(cubtmp, need, gkey, gkey_s, gidx, order, ng, 0, 62, s0));
    hipLaunchKernelGGL(k_gather_counts, dim3(gb), dim3(256), 0, s0, ng, (const uint32_t*)order, (const int64_t*)ucnt, cnt2);
    need = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, cnt2, grp2, (int)ng, s0));
    if ((st = ensure_tmp(need)) != SCILMM_OK) return st;
    need = cubbytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(cubtmp, need, cnt2, grp2, (int)ng, s0));
    HIPCHK(hipMemcpy(grp2 + ng, &nvalid, 8, hipMemcpyHostToDevice));  // one element past what the scan writes
    hipLaunchKernelGGL(k_finish_groups, dim3(gb), dim3(256), 0, s0, ng, (const unsigned long long*)gkey_s, udst);
    hipLaunchKernelGGL(k_bucket_counts, dim3((unsigned)((counters.size() + 255) / 256)), dim3(256), 0, s0, (int32_t)counters.size(), NL,
                       ng, (const unsigned long long*)gkey_s, d_counters);
    hipLaunchKernelGGL(k_gather_entries, dim3((unsigned)std::min<int64_t>((nvalid + 255) / 256, 1 << 20)), dim3(256), 0, s0, nvalid, ng,
                       (const int64_t*)grp2, (const uint32_t*)order, (const int64_t*)ustart, (const uint32_t*)sidx,
                       (const int64_t*)cst, (const int64_t*)csq, (const int32_t*)cmd, (const int32_t*)cwd, ost, osq, omd, owd);
    HIPCHK(hipMemcpyAsync(counters.data(), d_counters, 4 * counters.size(), hipMemcpyDeviceToHost, s0));
    HIPCHK(hipStreamSynchronize(s0));
  } else {
    const int64_t zero = 0;
    HIPCHK(hipMemcpy(grp2, &zero, 8, hipMemcpyHostToDevice));
  }
  HIPCHK(hipGetLastError());
  int64_t gbase = 0, ebase_unused = 0;
  (void)ebase_unused;
  for (int c = 0; c < 3; ++c) {
    Dev::CellSet& CS = D->cellset[c];
    CS.dst = udst + gbase;
    CS.grp = grp2 + gbase;
    CS.srct = ost;
    CS.srcq = osq;
    CS.md = omd;
    CS.wd = owd;
    int64_t run = 0;
    for (int32_t l = 0; l < NL; ++l) {
      const int64_t ns = counters[((size_t)c * NL + l) * 2], nl = counters[((size_t)c * NL + l) * 2 + 1];
      if (l < S.nlevels) {
        CS.level_short[l] = ns;
        run += ns + nl;
        CS.level_ptr[l + 1] = run;
      }
    }
    if (c == 0) *n_early = run;  // groups of the early class (diagnostic)
    gbase += run;
  }
  return SCILMM_OK;
}

// Rank-local storage of a distributed factor (world > 1).  Tail front dense_first + jj belongs to rank jj % world.
//   [ prelude panels, as in the Symbolic | own tail panels, packed | ring: G slots of the largest tail panel ]
// A panel of another rank lives in slot jj % G from its broadcast until every own target has consumed it: the batch of
// its source group g = jj / Wg (applied when the group is complete) and the late updates of the own targets of groups
// g and g + 1 -- so a slot is free again well before panel jj + G arrives (G = 4 Wg; the level loop still orders the
// re-use with events).  Per rank: nnz(L_tail) / world + G panels instead of the whole factor.
struct DistLayout {
  int32_t first = 0, Wg = 8, G = 32;
  std::vector<int64_t> loff;
  int64_t nL = 0, ring_base = 0, slot = 0;
};
void dist_layout(const Symbolic& S, int32_t rank, int32_t world, DistLayout* o) {
  o->loff.assign(S.sn_loff.begin(), S.sn_loff.end());
  o->nL = std::max<int64_t>(S.nnzL_stored, 1);
  o->first = S.nsuper;
  if (world <= 1 || S.dense_first >= S.nsuper) return;
  o->first = S.dense_first;
  int32_t wg = world;
  while (wg < 8) wg += world;
  if (const char* e = tune_env("SCILMM_DIST_GROUP")) wg = std::max(world, atoi(e) / world * world);
  o->Wg = wg;
  o->G = 4 * wg;
  const int32_t nT = S.nsuper - o->first;
  int64_t at = S.sn_loff[o->first];
  for (int32_t jj = 0; jj < nT; ++jj) {
    const int32_t f = o->first + jj;
    const int64_t sz = S.sn_loff[f + 1] - S.sn_loff[f];
    o->slot = std::max(o->slot, (sz + 1) & ~(int64_t)1);
    if (jj % world == rank) {
      o->loff[f] = at;
      at += (sz + 1) & ~(int64_t)1;
    }
  }
  o->ring_base = at;
  const int32_t nslots = std::min(o->G, nT);
  for (int32_t jj = 0; jj < nT; ++jj)
    if (jj % world != rank) o->loff[o->first + jj] = o->ring_base + (int64_t)(jj % o->G) * o->slot;
  o->nL = o->ring_base + (int64_t)nslots * o->slot;
  o->loff[S.nsuper] = o->nL;
}

int ensure_device(scilmm_symbolic* sym, Dev** out) {
  if (sym->device) {
    *out = (Dev*)sym->device;
    return SCILMM_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    sym->err = "no HIP device available (the numeric phase has no CPU fallback)";
    return SCILMM_ERR_DEVICE;
  }
  {
    // a stale "last error" of this host thread (left by any earlier runtime call, ours or the caller's) would be
    // reported by the first library that polls hipGetLastError() -- hipCUB does, inside the plan construction
    const hipError_t stale = hipGetLastError();
    if (stale != hipSuccess && getenv("SCILMM_VERBOSE"))
      fprintf(stderr, "[scilmm plan] cleared a stale HIP error of this thread: %s\n", hipGetErrorString(stale));
  }
  Dev* D = new Dev();
  sym->device = D;
  sym->device_free = dev_free;
  if (hipGetDevice(&D->device) != hipSuccess) D->device = 0;  // the handle binds to the caller's current device
  const bool pverb = getenv("SCILMM_VERBOSE") != nullptr;
  auto ptl = std::chrono::steady_clock::now();
  auto plap = [&](const char* what) {
    auto now = std::chrono::steady_clock::now();
    if (pverb) fprintf(stderr, "[scilmm plan] %-30s %8.3f s\n", what, std::chrono::duration<double>(now - ptl).count());
    ptl = now;
  };
  for (auto& e : D->ev) e = nullptr;
  const Symbolic& S = *sym->S;
  {
    // the main stream carries the latency-bound per-level chain: give it dispatch priority over the side
    // stream that streams the look-ahead updates
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&D->stream, hipStreamNonBlocking, hi));
    // SCILMM_RESERVE_CUS = r > 0: the look-ahead side streams are created with a CU mask that leaves r CUs per
    // XCD-group free, so the main stream's single-workgroup kernels never queue behind resident update items.
    const char* er = tune_env("SCILMM_RESERVE_CUS");
    const int reserve = er ? atoi(er) : 0;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, D->device));
    const int ncu = prop.multiProcessorCount;
    if (reserve > 0 && reserve < ncu) {
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      // keep every (ncu / reserve)-th CU out of the mask so the reserved CUs are spread over the XCDs
      const int stride = std::max(1, ncu / reserve);
      int kept_out = 0;
      for (int c = 0; c < ncu; ++c) {
        const bool out = (c % stride == stride - 1) && kept_out < reserve;
        if (out) { kept_out++; continue; }
        mask[c / 32] |= (1u << (c % 32));
      }
      HIPCHK(hipExtStreamCreateWithCUMask(&D->side, (uint32_t)mask.size(), mask.data()));
      HIPCHK(hipExtStreamCreateWithCUMask(&D->side2, (uint32_t)mask.size(), mask.data()));
    } else {
      HIPCHK(hipStreamCreateWithPriority(&D->side, hipStreamNonBlocking, lo));
      HIPCHK(hipStreamCreateWithPriority(&D->side2, hipStreamNonBlocking, lo));
      const char* ens3 = tune_env("SCILMM_SIDE_STREAMS");
      if (ens3 && atoi(ens3) == 3) {
        HIPCHK(hipStreamCreateWithPriority(&D->side3, hipStreamNonBlocking, lo));
        D->nside = 3;
      } else if (ens3 && atoi(ens3) == 1) {
        D->nside = 1;  // early updates strictly one after the other (their launch durations then do not overlap)
      }
    }
  }
  HIPCHK(hipEventCreateWithFlags(&D->ev_asm, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&D->ev_x0, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&D->ev_x1, hipEventDisableTiming));
  D->lev_ev.assign((size_t)2 * std::max(S.nlevels, 1), nullptr);
  for (auto& e : D->lev_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : D->ev) HIPCHK(hipEventCreate(&e));
  for (int32_t b = 0; b < S.nsuper; ++b)
    if (S.sn_start[b + 1] - S.sn_start[b] > NB) {
      sym->err = "symbolic analysis has supernode blocks wider than the kernels' block width (max_width > NB)";
      return SCILMM_ERR_ARG;
    }
  const char* nm = tune_env("SCILMM_NO_MFMA");
  D->use_mfma = !(nm && nm[0] == '1');
  const char* etl = tune_env("SCILMM_TRSM_LITE");
  D->trsm_lite = !(etl && etl[0] == '0');
#ifdef SCILMM_DIAG
  const char* ab = getenv("SCILMM_ABLATE");  // timing ablations (WRONG numbers): diagnostic builds only
  D->ablate = ab ? atoi(ab) : 0;
#endif
  // ---- multi-GPU ownership and rank-local storage
  D->rank = sym->rank;
  D->world = std::max<int32_t>(1, sym->world);
  D->comm = (hipStream_t)sym->comm_stream;
  D->keep_front.assign((size_t)std::max(S.nsuper, 1), 1);
  D->tail_of_level.assign((size_t)std::max(S.nlevels, 1), -1);
  {
    DistLayout lay;
    dist_layout(S, D->rank, D->world, &lay);
    D->dist_first = lay.first;
    D->dist_Wg = lay.Wg;
    D->dist_G = lay.G;
    D->loff.swap(lay.loff);
    D->nL_local = lay.nL;
  }
  if (D->world > 1 && D->dist_first < S.nsuper) {
    const int32_t nT = S.nsuper - D->dist_first, ngroups = (nT + D->dist_Wg - 1) / D->dist_Wg;
    for (int32_t f = D->dist_first; f < S.nsuper; ++f) {
      D->keep_front[f] = ((f - D->dist_first) % D->world) == D->rank ? 1 : 0;
      if (D->tail_of_level[S.sn_level[f]] >= 0) {
        sym->err = "multi-GPU: two fronts of the dense tail share a level (the tail is expected to be a chain)";
        return SCILMM_ERR_ARG;
      }
      D->tail_of_level[S.sn_level[f]] = f;
    }
    D->done_ev.assign((size_t)std::max(S.nlevels, 1), nullptr);
    for (int32_t l = 0; l < S.nlevels; ++l)
      if (D->tail_of_level[l] >= 0) HIPCHK(hipEventCreateWithFlags(&D->done_ev[l], hipEventDisableTiming));
    D->batch_ev.assign((size_t)ngroups, nullptr);
    for (auto& e : D->batch_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    D->last_own_level.assign((size_t)ngroups, -1);
    for (int32_t f = D->dist_first; f < S.nsuper; ++f)
      if (D->keep_front[f]) D->last_own_level[(size_t)((f - D->dist_first) / D->dist_Wg)] = S.sn_level[f];
    int lo3 = 0, hi3 = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo3, &hi3));
    HIPCHK(hipStreamCreateWithPriority(&D->bstream, hipStreamNonBlocking, lo3));
    if (pverb)
      fprintf(stderr, "[scilmm plan] rank %d of %d: %d tail panels distributed (every %d-th one mine), groups of %d, ring of %d slots; "
              "local panel storage %.2f GB of %.2f GB\n", D->rank, D->world, nT, D->world, D->dist_Wg, D->dist_G,
              8e-9 * (double)D->nL_local, 8e-9 * (double)S.nnzL_stored);
  }
  // level lists of this rank: everything except the tail fronts of other ranks
  {
    D->lv_ptr.assign(1, 0);
    D->lv_tile_ptr.assign(1, 0);
    D->lv_pair_ptr.assign(1, 0);
    for (int32_t l = 0; l < S.nlevels; ++l) {
      for (int32_t q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q)
        if (D->keep_front[S.level_fronts[q]]) D->lv_fronts.push_back(S.level_fronts[q]);
      // (tiles of an own distributed panel last: the forward sweep pushes them into a different accumulator)
      for (int64_t q = S.level_tile_ptr[l]; q < S.level_tile_ptr[l + 1]; ++q)
        if (S.tile_front[S.level_tiles[q]] < D->dist_first) D->lv_tiles.push_back(S.level_tiles[q]);
      D->lv_tile_mid.push_back((int64_t)D->lv_tiles.size());
      for (int64_t q = S.level_tile_ptr[l]; q < S.level_tile_ptr[l + 1]; ++q)
        if (S.tile_front[S.level_tiles[q]] >= D->dist_first && D->keep_front[S.tile_front[S.level_tiles[q]]]) D->lv_tiles.push_back(S.level_tiles[q]);
      // backward pushes (target in level l -> descendant d): this rank needs the panel of d
      for (int64_t q = S.level_pair_ptr[l]; q < S.level_pair_ptr[l + 1]; ++q)
        if (D->keep_front[S.upd_src[S.level_pairs[q]]]) D->lv_pairs.push_back(S.level_pairs[q]);
      D->lv_ptr.push_back((int32_t)D->lv_fronts.size());
      D->lv_tile_ptr.push_back((int64_t)D->lv_tiles.size());
      D->lv_pair_ptr.push_back((int64_t)D->lv_pairs.size());
    }
  }
  const std::vector<int64_t>& LOFF = D->loff;
  {
    // The dense-tail path (k_dense_b + k_outside) serves every tail of 8192+ columns.  Round 2 kept the 100k config (15.7k
    // columns, 123 panels) on the explicit path (its one-workgroup-per-CU items balanced worse: 66 -> 72 ms); with k_dense_b,
    // k_outside and SHORT launches of ~128 items fitted to whole rounds of workgroups it is the faster one there too:
    // 65.3 -> 58.6 ms (items 64 / 96 / 128 / 192 / 256 / 512: 60.0 / 59.2 / 58.6 / 60.6 / 60.9 / 60.4; without k_outside 66.3;
    // without the fitting 62.7).  SCILMM_DENSE=1 / 0 forces it.
    const char* edn = tune_env("SCILMM_DENSE");
    const int32_t tail_w = S.dense_first < S.nsuper ? S.n - S.sn_start[S.dense_first] : 0;
    // (k_dense_b has no scalar form: with SCILMM_NO_MFMA=1 the tail goes through the explicit items of k_update2<false>)
    D->dense_on = S.dense_first < S.nsuper && D->use_mfma && (edn ? edn[0] != '0' : tail_w >= 8192);
    // a distributed tail is always updated by the implicit items (the batches have no explicit-combo form)
    if (D->world > 1 && D->dist_first < S.nsuper) D->dense_on = true;
    if (D->dense_on && !D->d_zeros) {
      HIPCHK(hipMalloc((void**)&D->d_zeros, 2048));
      HIPCHK(hipMemset(D->d_zeros, 0, 2048));
    }
  }
  {
    // k_outside takes over the update pairs (tail target, prelude descendant below the tail's first level) unless the
    // caller asks for the bitwise-reproducible schedule (SCILMM_DETERMINISTIC=1) or the combos were already built
    const char* edet = getenv("SCILMM_DETERMINISTIC");
    const char* eout = tune_env("SCILMM_OUTSIDE");
    int st = SCILMM_OK;
    D->outside_desc.assign((size_t)std::max(S.nsuper, 1), 0);
    // (switched on with the dense-tail path, by the width of the tail -- SCILMM_OUTSIDE=1 / 0 forces it)
    const int32_t tail_w2 = S.dense_first < S.nsuper ? S.n - S.sn_start[S.dense_first] : 0;
    D->outside_on = S.dense_first < S.nsuper && !(edet && edet[0] == '1') && !sym->S->combos_built &&
                    (eout ? eout[0] != '0' : tail_w2 >= 8192);
    if (D->outside_on) {
      D->tail_level = S.sn_level[S.dense_first];
      const int32_t c0_tail = S.sn_start[S.dense_first];
      std::vector<OutsideWork> ow;
      // descendants with IDENTICAL tail rows (the 128-column blocks of one wide supernode) form a group: one set of items for
      // the group's leader, the kernel sums the members' products in its registers before the one atomic scatter
      std::vector<int32_t> grp_next((size_t)std::max(S.nsuper, 1), -1), grp_t0((size_t)std::max(S.nsuper, 1), 0);
      std::vector<int32_t> grp_width((size_t)std::max(S.nsuper, 1), 0);  // leader -> columns of the whole group
      {
        const char* egm = tune_env("SCILMM_OUTSIDE_MERGE");
        const bool merge = !(egm && egm[0] == '0');
        std::vector<std::pair<uint64_t, int32_t>> keyed;  // (hash of the tail rows, descendant)
        for (int32_t d = 0; d < S.dense_first; ++d) {
          if (S.sn_level[d] >= D->tail_level) continue;  // finished too late for the launches before the tail
          const int32_t* rd = S.sn_rows.data() + S.sn_rowptr[d];
          const int32_t md = (int32_t)(S.sn_rowptr[d + 1] - S.sn_rowptr[d]);
          const int32_t t0 = (int32_t)(std::lower_bound(rd, rd + md, c0_tail) - rd);
          if (t0 >= md) continue;
          D->outside_desc[d] = 1;
          grp_t0[(size_t)d] = t0;
          uint64_t h = 1469598103934665603ull ^ (uint64_t)(md - t0);
          for (int32_t t = t0; t < md; ++t) h = (h ^ (uint64_t)(uint32_t)rd[t]) * 1099511628211ull;
          keyed.push_back({merge ? h : (uint64_t)d, d});
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, int32_t>& a, const std::pair<uint64_t, int32_t>& b) { return a.first < b.first; });
        auto same_rows = [&](int32_t a, int32_t b) -> bool {
          const int64_t na = S.sn_rowptr[a + 1] - S.sn_rowptr[a] - grp_t0[(size_t)a], nb = S.sn_rowptr[b + 1] - S.sn_rowptr[b] - grp_t0[(size_t)b];
          return na == nb && std::memcmp(S.sn_rows.data() + S.sn_rowptr[a] + grp_t0[(size_t)a], S.sn_rows.data() + S.sn_rowptr[b] + grp_t0[(size_t)b],
                                         sizeof(int32_t) * (size_t)na) == 0;
        };
        std::vector<int32_t> leaders;
        for (size_t i = 0; i < keyed.size();) {
          // members of one hash bucket, split into runs of truly identical row lists (a collision must not merge anything)
          size_t j = i;
          while (j < keyed.size() && keyed[j].first == keyed[i].first) ++j;
          std::vector<uint8_t> used(j - i, 0);
          for (size_t a = i; a < j; ++a) {
            if (used[a - i]) continue;
            const int32_t lead = keyed[a].second;
            leaders.push_back(lead);
            int32_t last = lead;
            grp_width[(size_t)lead] = S.sn_start[lead + 1] - S.sn_start[lead];
            for (size_t b = a + 1; b < j; ++b) {
              if (used[b - i] || !merge || !same_rows(lead, keyed[b].second)) continue;
              used[b - i] = 1;
              grp_next[(size_t)last] = keyed[b].second;
              last = keyed[b].second;
              grp_width[(size_t)lead] += S.sn_start[last + 1] - S.sn_start[last];
            }
          }
          i = j;
        }
        std::sort(leaders.begin(), leaders.end());
        // multi-GPU: a block pair adds into the panels of the columns of its block bj only; a rank keeps the pairs that reach
        // a panel it owns (a 128-row block of a tall front spans a few panels: at 8 ranks most pairs are somebody else's --
        // until round 4 every rank multiplied all of them and threw 7/8 of the products away in the epilogue)
        const bool own_only = D->world > 1 && D->dist_first < S.nsuper;
        int64_t pairs_all = 0;
        for (int32_t d : leaders) {
          const int32_t md = (int32_t)(S.sn_rowptr[d + 1] - S.sn_rowptr[d]), t0 = grp_t0[(size_t)d];
          const int32_t* rd = S.sn_rows.data() + S.sn_rowptr[d];
          const int32_t nb = (md - t0 + TM - 1) / TM;
          std::vector<uint8_t> col_mine((size_t)nb, 1);
          if (own_only)
            for (int32_t bj = 0; bj < nb; ++bj) {
              // panels of the block's first and last column label (sorted rows: everything in between lies between them)
              const int32_t r_lo = rd[t0 + NB * bj], r_hi = rd[std::min(md, t0 + NB * (bj + 1)) - 1];
              int32_t f_lo = (int32_t)(std::upper_bound(S.sn_start.begin() + S.dense_first, S.sn_start.begin() + S.nsuper + 1, r_lo) - S.sn_start.begin()) - 1;
              int32_t f_hi = (int32_t)(std::upper_bound(S.sn_start.begin() + S.dense_first, S.sn_start.begin() + S.nsuper + 1, r_hi) - S.sn_start.begin()) - 1;
              uint8_t mine = 0;
              for (int32_t f = f_lo; f <= f_hi && !mine; ++f) mine = D->keep_front[(size_t)f];
              col_mine[(size_t)bj] = mine;
            }
          for (int32_t bi = 0; bi < nb; ++bi)
            for (int32_t bj = 0; bj <= bi; ++bj) {
              ++pairs_all;
              if (col_mine[(size_t)bj]) ow.push_back(OutsideWork{d, t0, bi, bj});
            }
        }
        if (pverb)
          fprintf(stderr, "[scilmm plan] k_outside: %zu descendants in %zu groups of identical tail rows; %lld of %lld block pairs reach a panel of this rank\n",
                  keyed.size(), leaders.size(), (long long)ow.size(), (long long)pairs_all);
      }
      D->n_owork = (int64_t)ow.size();
      if (D->n_owork == 0 || D->tail_level == 0) {
        D->outside_on = false;
        std::fill(D->outside_desc.begin(), D->outside_desc.end(), 0);
      } else {
        std::vector<int32_t> tf((size_t)(S.n - c0_tail));
        for (int32_t f = S.dense_first; f < S.nsuper; ++f)
          for (int32_t c = S.sn_start[f]; c < S.sn_start[f + 1]; ++c) tf[(size_t)(c - c0_tail)] = f;
        // first tail panel an item touches = the panel of its smallest column label (first row of block bj)
        auto first_front = [&](const OutsideWork& w) -> int32_t {
          return tf[(size_t)(S.sn_rows[S.sn_rowptr[w.d] + w.t0 + NB * w.bj] - c0_tail)];
        };
        const char* ech = tune_env("SCILMM_OUTSIDE_CHUNKS");
        // (100k / 300k factorization, ms: 1 chunk 57.8 / 1357; 4 / 8 / 16 chunks on a low-priority stream 55.5 / 1344, - / 1339, 56.1 / 1335)
        // ... and the count follows the size: one chunk per ~16k block pairs, 4 .. 32 (1M: 2 / 8 / 32 chunks 26.64 / 26.63 / 26.51 s)
        const int32_t want_chunks = std::max(1, ech ? atoi(ech) : (int32_t)std::min<int64_t>(32, std::max<int64_t>(4, (int64_t)ow.size() / 16384)));
        std::vector<int32_t> ffront(ow.size());
        for (size_t i = 0; i < ow.size(); ++i) ffront[i] = first_front(ow[i]);
        std::vector<size_t> ord(ow.size());
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return ffront[a] < ffront[b]; });
        // chunk boundaries where the first panel changes, about equal item counts
        D->ochunk_ptr.assign(1, 0);
        std::vector<int32_t> chunk_lo;  // first panel of the chunk's first item
        {
          const size_t N = ord.size(), per = std::max<size_t>(1, (N + want_chunks - 1) / want_chunks);
          size_t i = 0;
          while (i < N) {
            chunk_lo.push_back(ffront[ord[i]]);
            size_t e = std::min(N, i + per);
            while (e < N && ffront[ord[e]] == ffront[ord[e - 1]]) ++e;
            D->ochunk_ptr.push_back((int64_t)e);
            i = e;
          }
        }
        const int32_t nch = (int32_t)chunk_lo.size();
        // inside a chunk: widest descendants first (all items are 128 x 128 x w_d: the long ones start early)
        {
          std::vector<OutsideWork> sorted(ow.size());
          for (int32_t g = 0; g < nch; ++g) {
            std::stable_sort(ord.begin() + D->ochunk_ptr[g], ord.begin() + D->ochunk_ptr[g + 1], [&](size_t a, size_t b) {
              return grp_width[(size_t)ow[a].d] > grp_width[(size_t)ow[b].d];
            });
          }
          for (size_t i = 0; i < ord.size(); ++i) sorted[i] = ow[ord[i]];
          ow.swap(sorted);
        }
        // per level: the chunk its tail front waits for = the last chunk whose first item starts at that panel or before it
        D->out_wait_chunk.assign((size_t)std::max(S.nlevels, 1), -1);
        for (int32_t f = S.dense_first; f < S.nsuper; ++f) {
          const int32_t g = (int32_t)(std::upper_bound(chunk_lo.begin(), chunk_lo.end(), f) - chunk_lo.begin()) - 1;
          int32_t& w = D->out_wait_chunk[(size_t)S.sn_level[f]];
          w = std::max(w, g);
        }
        // (a level at or above the tail's first one without a tail front of its own waits like the level before it)
        for (int32_t l = D->tail_level + 1; l < S.nlevels; ++l)
          D->out_wait_chunk[(size_t)l] = std::max(D->out_wait_chunk[(size_t)l], D->out_wait_chunk[(size_t)l - 1]);
        D->out_evs.assign((size_t)nch, nullptr);
        for (auto& e : D->out_evs) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const OutsideWork* dow;
        if ((st = upload(sym, D, ow, &dow)) != SCILMM_OK) return st;
        D->d_owork = (OutsideWork*)dow;
        {
          const int32_t* dg;
          if ((st = upload(sym, D, grp_next, &dg)) != SCILMM_OK) return st;
          D->d_grp_next = (int32_t*)dg;
          if ((st = upload(sym, D, grp_t0, &dg)) != SCILMM_OK) return st;
          D->d_grp_t0 = (int32_t*)dg;
        }
        const int32_t* dtf;
        if ((st = upload(sym, D, tf, &dtf)) != SCILMM_OK) return st;
        D->d_tail_front = (int32_t*)dtf;
        const uint8_t* dkf;
        if ((st = upload(sym, D, D->keep_front, &dkf)) != SCILMM_OK) return st;
        D->d_keep_front = (uint8_t*)dkf;
        int lo4 = 0, hi4 = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo4, &hi4));
        {
          const char* epr = tune_env("SCILMM_OUTSIDE_PRIO");  // 1: the chain's priority, 0: the look-ahead streams'
          // (low: the chunks that later panels wait for fill the gaps of the chain-bound first tail levels instead of taking
          //  the chain's CU slots -- with the chain's priority the overlap gains nothing)
          HIPCHK(hipStreamCreateWithPriority(&D->outside_st, hipStreamNonBlocking, (epr && epr[0] == '1') ? hi4 : lo4));
        }
        if (pverb) {
          fprintf(stderr, "[scilmm plan] k_outside: %lld block-pair items of prelude fronts below level %d (tail starts at column %d), %d chunks by first panel:",
                  (long long)D->n_owork, D->tail_level, c0_tail, nch);
          for (int32_t g = 0; g < nch; ++g) fprintf(stderr, " [%d..: %lld]", chunk_lo[g] - S.dense_first, (long long)(D->ochunk_ptr[g + 1] - D->ochunk_ptr[g]));
          fprintf(stderr, "\n");
        }
      }
    }
  }
  if (!sym->S->combos_built) {
    // (a handle analysed through scilmm_symbolic_get("combo_*") carries the full lists: then the dense path stays off)
    scilmm::build_tile_combos(sym->S, D->world > 1 ? D->keep_front.data() : nullptr, D->dense_on,
                              D->outside_on ? D->outside_desc.data() : nullptr);
    plap("tile combos");
  } else {
    D->dense_on = false;
  }
  D->v.n = S.n;
  D->v.nsuper = S.nsuper;
  int st;
#define UP(field, vec)                                          \
  if ((st = upload(sym, D, S.vec, &D->v.field)) != SCILMM_OK) return st;
  UP(sn_start, sn_start)
  UP(sn_rowptr, sn_rowptr)
  UP(sn_rows, sn_rows)
  if ((st = upload(sym, D, D->loff, &D->v.sn_loff)) != SCILMM_OK) return st;
  UP(inv_off, inv_off)
  UP(upd_src, upd_src)
  UP(upd_p0, upd_p0)
  UP(upd_p1, upd_p1)
  UP(tile_front, tile_front)
  UP(tile_base, tile_base)
  // (the per-tile combo arrays stay on the host: the kernels read the flattened descriptors of the plan below)
  D->v.combo_ptr = nullptr;
  D->v.combo_pair = nullptr;
  D->v.combo_ta = nullptr;
  D->v.combo_tb = nullptr;
  if (D->world > 1 && D->dist_first < S.nsuper) {
    // value-assembly maps in rank-local offsets; entries of other ranks' tail panels are dropped (-1)
    std::vector<int64_t> ad(S.asm_dst.size()), dd(S.diag_dst.size());
    const int nth = std::max(1, std::min(16, scilmm::host_threads()));
    std::vector<std::thread> pool;
    auto part = [&](int q) {
      for (int32_t f = q; f < S.nsuper; f += nth) {
        const bool keep = D->keep_front[f] != 0;
        const int64_t delta = LOFF[f] - S.sn_loff[f];
        for (int32_t j = S.sn_start[f]; j < S.sn_start[f + 1]; ++j) {
          dd[(size_t)j] = keep ? S.diag_dst[(size_t)j] + delta : -1;
          for (int64_t e = S.pat_colptr[j]; e < S.pat_colptr[j + 1]; ++e) ad[(size_t)e] = keep ? S.asm_dst[(size_t)e] + delta : -1;
        }
      }
    };
    for (int q = 1; q < nth; ++q) pool.emplace_back(part, q);
    part(0);
    for (auto& th : pool) th.join();
    if ((st = upload(sym, D, ad, &D->v.asm_dst)) != SCILMM_OK) return st;
    if ((st = upload(sym, D, dd, &D->v.diag_dst)) != SCILMM_OK) return st;
  } else {
    UP(asm_dst, asm_dst)
    UP(diag_dst, diag_dst)
  }
  UP(pat_colptr, pat_colptr)
  UP(pat_row, pat_row)
  UP(perm, perm)
#undef UP
  const int32_t* tmp;
  if ((st = upload(sym, D, D->lv_tiles, &tmp)) != SCILMM_OK) return st;
  D->d_level_tiles = (int32_t*)tmp;
  if ((st = upload(sym, D, D->lv_fronts, &tmp)) != SCILMM_OK) return st;
  D->d_level_fronts = (int32_t*)tmp;
  if ((st = upload(sym, D, D->lv_pairs, &tmp)) != SCILMM_OK) return st;
  D->d_level_pairs = (int32_t*)tmp;
  if (D->world > 1) {
    // L*R: every panel is multiplied by exactly one rank (own tail panels; the replicated prelude by rank 0), then summed
    std::vector<int32_t> lt;
    for (int32_t g : D->lv_tiles)
      if (S.tile_front[g] >= D->dist_first || D->rank == 0) lt.push_back(g);
    D->n_lmul_tiles = (int64_t)lt.size();
    if ((st = upload(sym, D, lt, &tmp)) != SCILMM_OK) return st;
    D->d_lmul_tiles = (int32_t*)tmp;
    if ((st = upload(sym, D, S.level_fronts, &tmp)) != SCILMM_OK) return st;
    D->d_all_fronts = (int32_t*)tmp;
  }
  D->vals.assign(S.K, nullptr);
  D->have_vals.assign(S.K, 0);
  HIPCHK(hipMalloc((void**)&D->d_out, sizeof(double) * RPMAX));
  plap("symbolic arrays -> device");
  // ---- update-kernel plan
  {
    const int64_t nc = (int64_t)S.combo_pair.size();
    const int64_t ntiles0 = (int64_t)S.tile_front.size();
    const char* ecs = tune_env("SCILMM_CELL_LIMIT");
    double cell_limit = ecs ? atof(ecs) : 4096.0;  // pairs with cells*width below this take the cell-wise path
    if (!ecs) {
      // very large patterns: keep the expanded cell plan below ~1.5e9 cells (32-bit counts in the device sort;
      // 32 B per cell) by lowering the limit -- the 1 M-individual config ends at 64
      const double cand[5] = {4096.0, 1024.0, 256.0, 64.0, 16.0};
      double cells_at[5] = {0, 0, 0, 0, 0};
      for (int64_t c = 0; c < nc; ++c) {
        const int32_t e = S.combo_pair[c];
        const int32_t d = S.upd_src[e];
        const double cellsn = (double)(S.combo_tb[c] - S.combo_ta[c]) * (double)(S.upd_p1[e] - S.upd_p0[e]);
        const double vol = cellsn * (double)(S.sn_start[d + 1] - S.sn_start[d]);
        for (int k = 0; k < 5; ++k)
          if (vol <= cand[k]) cells_at[k] += cellsn;
      }
      int pick = 0;
      while (pick < 4 && cells_at[pick] > 1.5e9) ++pick;
      cell_limit = cand[pick];
      if (getenv("SCILMM_VERBOSE") && pick > 0)
        fprintf(stderr, "[scilmm plan] cell limit lowered to %.0f (%.3e cells)\n", cell_limit, cells_at[pick]);
    }
    std::vector<ComboDesc> cd;                 // dense combos only, grouped by tile
    std::vector<int64_t> dptr((size_t)ntiles0 + 1, 0), dmid((size_t)ntiles0 + 1, 0);
    const char* ela = tune_env("SCILMM_NO_LOOKAHEAD");
    const bool lookahead = !(ela && ela[0] == '1');
    {
      const char* eld = tune_env("SCILMM_LOOK_DEPTH");
      D->look_depth = eld ? std::max(1, atoi(eld)) : 2;  // measured at 100k: depth 1 81.4 ms, 2 77.6 ms, 3 78.4 ms
    }
    const int32_t depth = D->look_depth;
    // "late" = on the main stream, right before the target's potrf: the descendant finished at most `depth` levels below
    // the target.  A DISTRIBUTED tail target takes all its explicit items late: its panel is read-modify-written by the
    // batches on their own stream until its late update starts, so nothing else may touch it ahead of time.
    auto is_late = [&](int32_t d, int32_t sfr) -> bool {
      return !lookahead || S.sn_level[d] + depth >= S.sn_level[sfr] || (D->world > 1 && sfr >= D->dist_first);
    };
    struct Cell { int64_t dst, st, sq; int32_t md, wd, level, late; };  // late: 0 early (side streams), 1 late (main stream)
    std::vector<Cell> cells;
    // The tiles are classified by a few host threads over contiguous tile ranges of about equal combo counts; the
    // per-range outputs are concatenated in tile order, so the plan does not depend on the thread count.
    struct Part {
      std::vector<ComboDesc> cd;
      std::vector<int64_t> dend, dmidv;  // per tile: end of its dense list, its early|late split
      std::vector<Cell> cells;
      std::vector<CellCombo> cellcombos;  // device-built cell plan: the small combos themselves
      int64_t n_sparse = 0;
    };
    // The cell lists are built on the device from the small combos (cellplan.hip.h); SCILMM_HOST_CELLS=1 keeps the
    // host enumeration (same lists up to the order of the contributions inside a group).
    const char* ehc = tune_env("SCILMM_HOST_CELLS");
    const bool gpu_cells = !(ehc && ehc[0] == '1') && S.nnzL_stored < ((int64_t)1 << 38);
    std::vector<std::vector<CellCombo>> cellparts;  // the small combos, one vector per classification thread (tile order)
    std::vector<uint8_t> cd_cost;                   // per dense-path combo: 1 + K chunks (what the work-item cuts need)
    int64_t n_dense_total = 0;
    auto process_range = [&](int64_t gbeg, int64_t gend, Part& Pt) {
    std::vector<ComboDesc>& cd = Pt.cd;
    std::vector<Cell>& cells = Pt.cells;
    std::vector<ComboDesc> late_tmp;
    for (int64_t g = gbeg; g < gend; ++g) {
      const int32_t sfr = S.tile_front[g];
      const int32_t ti = (int32_t)(g - S.tile_base[sfr]);
      const int32_t c0s = S.sn_start[sfr];
      const int64_t ms = S.sn_rowptr[sfr + 1] - S.sn_rowptr[sfr];
      const int32_t* rs = S.sn_rows.data() + S.sn_rowptr[sfr];
      const int64_t R0 = (int64_t)ti * TM;
      const int64_t tile_end = std::min<int64_t>(R0 + TM, ms);
      for (int64_t c = S.combo_ptr[g]; c < S.combo_ptr[g + 1]; ++c) {
        const int32_t e = S.combo_pair[c];
        const int32_t d = S.upd_src[e];
        ComboDesc x;
        x.loff = LOFF[d];
        x.rowoff = S.sn_rowptr[d];
        x.md = (int32_t)(S.sn_rowptr[d + 1] - S.sn_rowptr[d]);
        x.wd = S.sn_start[d + 1] - S.sn_start[d];
        x.ta = S.combo_ta[c];
        x.nt = S.combo_tb[c] - S.combo_ta[c];
        x.p0 = S.upd_p0[e];
        x.nq = S.upd_p1[e] - S.upd_p0[e];
        x.ip0 = S.combo_ip0[c];
        x.jp0 = S.upd_jp0[e];
        const bool fake_contig = D->ablate == 3;  // diagnostic: pretend every combo is contiguous (wrong numbers, timing only)
        {
          // spans in target coordinates: rows and columns of a descendant are sorted, so first/last suffice
          const int32_t* rdx = S.sn_rows.data() + x.rowoff;
          const int32_t* lo0 = rs + R0;
          x.ilo = (int32_t)(std::lower_bound(lo0, rs + tile_end, rdx[x.ta]) - lo0);
          x.ihi = (int32_t)(std::lower_bound(lo0, rs + tile_end, rdx[x.ta + x.nt - 1]) - lo0);
          x.jlo = rdx[x.p0] - c0s;
          x.jhi = rdx[x.p0 + x.nq - 1] - c0s;
        }
        if (fake_contig) {
          if (x.ip0 < 0) x.ip0 = std::min<int32_t>(x.ilo, TM - x.nt);
          if (x.jp0 < 0) x.jp0 = std::min<int32_t>(x.jlo, NB - x.nq);
        }
        if ((double)x.nt * (double)x.nq * (double)x.wd > cell_limit) {
          // "late" = the descendant sits one level below the target (finished only just before this level)
          const bool late = is_late(d, sfr);
          if (late) late_tmp.push_back(x); else cd.push_back(x);
          continue;
        }
        Pt.n_sparse++;
        if (gpu_cells) {
          Pt.cellcombos.push_back(CellCombo{x.loff, x.rowoff, LOFF[sfr], S.sn_rowptr[sfr] + R0, x.md, x.wd, x.ta, x.nt, x.p0, x.nq,
                                            x.ip0, (int32_t)ms, (int32_t)R0, (int32_t)(tile_end - R0), c0s, S.sn_level[sfr],
                                            is_late(d, sfr) ? 1 : 0});
          continue;
        }
        const int32_t* rd = S.sn_rows.data() + x.rowoff;
        const int32_t* lo = rs + R0;
        for (int32_t t = x.ta; t < x.ta + x.nt; ++t) {
          const int64_t R = (x.ip0 >= 0) ? R0 + x.ip0 + (t - x.ta) : (std::lower_bound(lo, rs + tile_end, rd[t]) - rs);
          for (int32_t q = x.p0; q < x.p0 + x.nq; ++q) {
            const int64_t j = rd[q] - c0s;
            if (R < j) continue;  // strict upper part of the diagonal block is never referenced
            cells.push_back(Cell{LOFF[sfr] + j * ms + R, x.loff + t, x.loff + q, x.md, x.wd, S.sn_level[sfr],
                                 is_late(d, sfr) ? 1 : 0});
          }
        }
      }
      Pt.dmidv.push_back((int64_t)cd.size());
      cd.insert(cd.end(), late_tmp.begin(), late_tmp.end());
      late_tmp.clear();
      Pt.dend.push_back((int64_t)cd.size());
    }
    };
    {
      const unsigned nth = (unsigned)std::max<int64_t>(
          1, std::min<int64_t>((nc > 50000000 ? 3 : 1) * scilmm::host_threads(), ntiles0));  // static shares: finer = better balanced
      std::vector<int64_t> cut(nth + 1, ntiles0);
      cut[0] = 0;
      for (unsigned k = 1; k < nth; ++k) {
        const int64_t want = nc * (int64_t)k / nth;  // first tile whose combos start at or after this share
        cut[k] = std::lower_bound(S.combo_ptr.begin(), S.combo_ptr.begin() + ntiles0, want) - S.combo_ptr.begin();
        cut[k] = std::max(cut[k], cut[k - 1]);
      }
      std::vector<Part> parts(nth);
      std::vector<std::thread> pool;
      for (unsigned k = 1; k < nth; ++k) pool.emplace_back([&, k]() { process_range(cut[k], cut[k + 1], parts[k]); });
      process_range(cut[0], cut[1], parts[0]);
      for (auto& th : pool) th.join();
      size_t ncd = 0, ncell = 0;
      for (auto& Pt : parts) { ncd += Pt.cd.size(); ncell += Pt.cells.size(); }
      cells.reserve(ncell);
      // The dense-path descriptors (18 GB at the 1M config) are NOT concatenated on the host: every part goes straight
      // to its place in the device array, and the host keeps one byte per combo (its cost) for the work-item cuts.
      {
        void* pdc = nullptr;
        HIPCHK(hipMalloc(&pdc, sizeof(ComboDesc) * (ncd + 1)));
        D->allocs.push_back(pdc);
        D->d_combos = (ComboDesc*)pdc;
      }
      cd_cost.resize(ncd);
      std::vector<int64_t> dbases(nth + 1, 0);
      for (unsigned k = 0; k < nth; ++k) dbases[k + 1] = dbases[k] + (int64_t)parts[k].cd.size();
      {
        std::vector<std::thread> pool2;
        auto fill_cost = [&](unsigned k) {
          const std::vector<ComboDesc>& v = parts[k].cd;
          uint8_t* dst = cd_cost.data() + dbases[k];
          for (size_t c = 0; c < v.size(); ++c) dst[c] = (uint8_t)(1 + (v[c].wd + KC - 1) / KC);
        };
        for (unsigned k = 1; k < nth; ++k) pool2.emplace_back(fill_cost, k);
        fill_cost(0);
        for (auto& th : pool2) th.join();
      }
      for (unsigned k = 0; k < nth; ++k) {
        Part& Pt = parts[k];
        const int64_t dbase = dbases[k];
        for (int64_t g = cut[k]; g < cut[k + 1]; ++g) {
          dmid[g] = dbase + Pt.dmidv[(size_t)(g - cut[k])];
          dptr[g + 1] = dbase + Pt.dend[(size_t)(g - cut[k])];
        }
        if (!Pt.cd.empty())
          HIPCHK(hipMemcpy(D->d_combos + dbase, Pt.cd.data(), sizeof(ComboDesc) * Pt.cd.size(), hipMemcpyHostToDevice));
        cells.insert(cells.end(), Pt.cells.begin(), Pt.cells.end());
        D->n_sparse_combos += Pt.n_sparse;
        Part().cd.swap(Pt.cd);
        std::vector<Cell>().swap(Pt.cells);
      }
      n_dense_total = (int64_t)ncd;
      for (unsigned k = 0; k < nth; ++k) cellparts.push_back(std::move(parts[k].cellcombos));
    }
    D->n_dense_combos = n_dense_total;

    D->n_cells = (int64_t)cells.size();
    plap("classify combos, list cells");
    size_t split = 0;
    int64_t ngroups_total = 0;
    if (gpu_cells) {
      int64_t potential = 0;
      std::vector<const std::vector<CellCombo>*> ccparts;
      for (auto& pv : cellparts) {
        ccparts.push_back(&pv);
        for (const CellCombo& q : pv) potential += (int64_t)q.nt * q.nq;
      }
      if (potential >= ((int64_t)1 << 31)) {
        sym->err = "cell plan: more than 2^31 cells (raise SCILMM_CELL_LIMIT granularity or set SCILMM_HOST_CELLS=1)";
        return SCILMM_ERR_ARG;
      }
      int64_t n_early_groups = 0;
      if ((st = build_cells_device(sym, D, ccparts, std::max(S.nlevels, 1), &ngroups_total, &n_early_groups)) != SCILMM_OK) return st;
      split = (size_t)n_early_groups;
      std::vector<std::vector<CellCombo>>().swap(cellparts);
    } else {
      // Cells are ordered by (late class, level, dst, st, sq): counting sort on (class, level), then every bucket is
      // sorted, cut into groups of equal target address (short groups first) and written to the upload arrays
      // independently on a few host threads (one global std::sort of 27 M cells cost 7 s of every first evaluation).
      const size_t NL = (size_t)std::max(S.nlevels, 1), nbk = 3 * NL;
      std::vector<size_t> bptr(nbk + 1, 0);
      for (const Cell& c : cells) bptr[(size_t)c.late * NL + c.level + 1]++;
      for (size_t k = 0; k < nbk; ++k) bptr[k + 1] += bptr[k];
      split = bptr[NL];
      {
        std::vector<Cell> sorted(cells.size());
        std::vector<size_t> fill(bptr.begin(), bptr.end() - 1);
        for (const Cell& c : cells) sorted[fill[(size_t)c.late * NL + c.level]++] = c;
        cells.swap(sorted);
      }
      const unsigned nth = (unsigned)std::max(1, std::min(16, scilmm::host_threads()));
      auto parallel_buckets = [&](const std::function<void(size_t)>& fn) {
        std::atomic<size_t> next{0};
        auto worker = [&]() {
          for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= nbk) break;
            fn(k);
          }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nth; ++t) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
      };
      const int64_t long_limit = 16;
      std::vector<int64_t> g_short(nbk, 0), g_long(nbk, 0), e_short(nbk, 0);
      parallel_buckets([&](size_t k) {
        std::sort(cells.begin() + bptr[k], cells.begin() + bptr[k + 1], [](const Cell& a, const Cell& b) {
          if (a.dst != b.dst) return a.dst < b.dst;
          if (a.st != b.st) return a.st < b.st;
          return a.sq < b.sq;
        });
        for (size_t i = bptr[k]; i < bptr[k + 1];) {
          size_t j = i + 1;
          while (j < bptr[k + 1] && cells[j].dst == cells[i].dst) ++j;
          if ((int64_t)(j - i) <= long_limit) { g_short[k]++; e_short[k] += (int64_t)(j - i); } else g_long[k]++;
          i = j;
        }
      });
      for (int which = 0; which < 3; ++which) {
        Dev::CellSet& CS = D->cellset[which];
        CS.level_ptr.assign(S.nlevels + 1, 0);
        CS.level_short.assign(NL, 0);
        std::vector<int64_t> gbase(NL + 1, 0), ebase(NL + 1, 0);
        for (size_t l = 0; l < NL; ++l) {
          const size_t k = (size_t)which * NL + l;
          gbase[l + 1] = gbase[l] + g_short[k] + g_long[k];
          ebase[l + 1] = ebase[l] + (int64_t)(bptr[k + 1] - bptr[k]);
          if ((int32_t)l < S.nlevels) {
            CS.level_ptr[l + 1] = gbase[l + 1];
            CS.level_short[l] = g_short[k];
          }
        }
        const int64_t ng = gbase[NL], ne = ebase[NL];
        std::vector<int64_t> udst((size_t)ng), grp((size_t)ng + 1), st_((size_t)ne), sq_((size_t)ne);
        std::vector<int32_t> md_((size_t)ne), wd_((size_t)ne);
        grp[(size_t)ng] = ne;
        parallel_buckets([&](size_t k) {
          if (k / NL != (size_t)which) return;
          const size_t l = k - (size_t)which * NL;
          // short groups first, then the long ones; both in address order
          int64_t gs = gbase[l], gl = gbase[l] + g_short[k];
          int64_t es = ebase[l], el = ebase[l] + e_short[k];
          for (size_t i = bptr[k]; i < bptr[k + 1];) {
            size_t j = i + 1;
            while (j < bptr[k + 1] && cells[j].dst == cells[i].dst) ++j;
            const bool shortg = (int64_t)(j - i) <= long_limit;
            int64_t& gi = shortg ? gs : gl;
            int64_t& ei = shortg ? es : el;
            udst[(size_t)gi] = cells[i].dst;
            grp[(size_t)gi] = ei;
            ++gi;
            for (size_t c = i; c < j; ++c, ++ei) {
              st_[(size_t)ei] = cells[c].st; sq_[(size_t)ei] = cells[c].sq; md_[(size_t)ei] = cells[c].md; wd_[(size_t)ei] = cells[c].wd;
            }
            i = j;
          }
        });
        ngroups_total += ng;
        if (udst.empty()) udst.push_back(0);
        if (st_.empty()) { st_.push_back(0); sq_.push_back(0); md_.push_back(0); wd_.push_back(0); }
        const int64_t* t64; const int32_t* t32;
        if ((st = upload(sym, D, udst, &t64)) != SCILMM_OK) return st; CS.dst = (int64_t*)t64;
        if ((st = upload(sym, D, grp, &t64)) != SCILMM_OK) return st; CS.grp = (int64_t*)t64;
        if ((st = upload(sym, D, st_, &t64)) != SCILMM_OK) return st; CS.srct = (int64_t*)t64;
        if ((st = upload(sym, D, sq_, &t64)) != SCILMM_OK) return st; CS.srcq = (int64_t*)t64;
        if ((st = upload(sym, D, md_, &t32)) != SCILMM_OK) return st; CS.md = (int32_t*)t32;
        if ((st = upload(sym, D, wd_, &t32)) != SCILMM_OK) return st; CS.wd = (int32_t*)t32;
      }
    }
    {
      if (getenv("SCILMM_VERBOSE"))
        fprintf(stderr, "[scilmm plan] dense combos %lld, cell-path combos %lld, cells %lld (early %lld) in %lld target groups\n",
                (long long)D->n_dense_combos, (long long)D->n_sparse_combos, (long long)D->n_cells, (long long)split,
                (long long)ngroups_total);
      std::vector<Cell>().swap(cells);
      plap("sort/group/upload cells");
    }
    const int64_t ntiles = (int64_t)S.tile_front.size();
    std::vector<int32_t> pslot((size_t)std::max<int64_t>(ntiles, 1), 0), pnseg((size_t)std::max<int64_t>(ntiles, 1), 0);
    std::vector<int32_t> pslot_e(pslot.size(), 0), pnseg_e(pslot.size(), 0), red_tiles_e;
    D->red_ptr_e.assign(S.nlevels + 1, 0);
    std::vector<UpdWork> work, work_early;
    std::vector<DenseWork> dwork_e, dwork_l;
    D->dwork_e_ptr.assign(S.nlevels + 1, 0);
    D->dwork_l_ptr.assign(S.nlevels + 1, 0);
    D->dwork_l_mid.assign((size_t)std::max(S.nlevels, 1), 0);
    D->work_ptr.assign(S.nlevels + 1, 0);
    D->early_ptr.assign(S.nlevels + 1, 0);
    D->red_ptr.assign(S.nlevels + 1, 0);
    std::vector<int32_t> red_tiles;
    int64_t max_slots = 0;
    const char* ens = tune_env("SCILMM_NO_SPLITK");
    const bool allow_split = !(ens && ens[0] == '1');
    // Cost model: a combo costs one fixed unit plus one unit per K-chunk it streams.  Each launch (the early
    // and the late part of a level) is cut into about 4 work items per CU of equal cost, so that one launch
    // fills the chip once with balanced items (late levels of a dense chain: few tiles, long combo lists).
    auto combo_cost = [&](int64_t c) -> int64_t { return cd_cost[(size_t)c]; };
    // ... but an item never exceeds max_item units (~0.5 ms): the main stream's kernels start in the slots that
    // retiring update items free, so long items starve the per-level chain (300k probe: 2.3 ms per trsm launch)
    const char* emi = tune_env("SCILMM_MAX_ITEM");
    const char* eti = tune_env("SCILMM_TARGET_ITEMS");
    const char* emn = tune_env("SCILMM_MIN_ITEM");
    const char* edi = tune_env("SCILMM_DENSE_ITEMS");
    const char* edf = tune_env("SCILMM_DENSE_FILL");
    const int64_t dense_fill = edf ? atoll(edf) : 256;  // workgroups per round the dense item counts are fitted to (0: no fitting)
    // k_dense_b items per launch (target): long tails want launches of several rounds of workgroups (300k: 384 / 512 / 768 /
    // 1024 / 2048 / 3072 = 1373 / 1369 / 1365 / 1357 / 1384 / 1407 ms, 1M: 1024 vs 2048 = 27.0 vs 27.35 s), the short chain of the
    // 100k config short ones (see dense_on above)
    const int32_t tail_w_items = S.dense_first < S.nsuper ? S.n - S.sn_start[S.dense_first] : 0;
    const int64_t dense_items = std::max<int64_t>(64, edi ? atoll(edi) : tail_w_items < 24576 ? 128 : tail_w_items < 32768 ? 512 : 1024);
    const int64_t target_items = eti ? atoll(eti) : 1024, min_item = emn ? atoll(emn) : 24, max_item = std::max<int64_t>(min_item, emi ? atoll(emi) : 96);
    // cut [cb,ce) into segments; returns the number of items appended to `out` (slot = 0 placeholder)
    auto cut = [&](int32_t g, int64_t cb, int64_t ce, int64_t per_item, std::vector<UpdWork>& out) -> int64_t {
      if (ce <= cb) return 0;
      int64_t tcost = 0;
      for (int64_t c = cb; c < ce; ++c) tcost += combo_cost(c);
      const int64_t nseg = std::min<int64_t>(64, std::max<int64_t>(1, (tcost + per_item / 2) / per_item));
      const int64_t seg_cost = (tcost + nseg - 1) / nseg;
      const size_t first = out.size();
      int64_t a = cb, acc = 0;
      for (int64_t c = cb; c < ce; ++c) {
        acc += combo_cost(c);
        if (nseg > 1 && acc >= seg_cost && c + 1 < ce) {
          out.push_back(UpdWork{g, 0, a, c + 1});
          a = c + 1;
          acc = 0;
        }
      }
      out.push_back(UpdWork{g, 0, a, ce});
      return (int64_t)(out.size() - first);
    };
    // dense tail, block pattern: tail_src[j] = the tail fronts (relative index, ascending) whose TRUE row lists reach
    // the columns of tail front j -- the others hold only padding there and are left out of j's dense items
    std::vector<std::vector<int32_t>> tail_src;
    if (D->dense_on && !S.tail_blk_ptr.empty()) {
      const int32_t nT = S.nsuper - S.dense_first;
      tail_src.resize((size_t)nT);
      for (int32_t d = 0; d < nT; ++d)
        for (int64_t e = S.tail_blk_ptr[d]; e < S.tail_blk_ptr[d + 1]; ++e) tail_src[(size_t)S.tail_blk[e]].push_back(d);
    }
    std::vector<int32_t> tail_col_front;  // label - first tail column -> relative tail front
    if (!tail_src.empty()) {
      const int32_t c0t = S.sn_start[S.dense_first];
      tail_col_front.resize((size_t)(S.n - c0t));
      for (int32_t f = S.dense_first; f < S.nsuper; ++f)
        for (int32_t c = S.sn_start[f]; c < S.sn_start[f + 1]; ++c) tail_col_front[(size_t)(c - c0t)] = f - S.dense_first;
    }
    // runs of ACTIVE descendants (those whose true structure reaches target jj) inside [lo, hi)
    auto active_runs = [&](int32_t jj, int32_t lo, int32_t hi, std::vector<std::pair<int32_t, int32_t>>& runs) -> int64_t {
      runs.clear();
      if (hi <= lo) return 0;
      if (tail_src.empty()) {
        runs.push_back({lo, hi});
      } else {
        const std::vector<int32_t>& src = tail_src[(size_t)jj];
        auto it = std::lower_bound(src.begin(), src.end(), lo);
        for (; it != src.end() && *it < hi; ++it) {
          if (!runs.empty() && runs.back().second == *it) runs.back().second = *it + 1;
          else runs.push_back({*it, *it + 1});
        }
        if (runs.size() > 16) runs = {{runs.front().first, runs.back().second}};  // too fragmented: take the hull
      }
      int64_t total = 0;
      for (auto& r : runs) total += r.second - r.first;
      return total;
    };
    // distributed tail: the far part of an own target's update arrives as one BATCH per source group (see the level
    // loop of run_factorize); the per-target tile-pair masks are kept for the batch items built after this loop
    const bool dist = D->world > 1 && D->dist_first < S.nsuper;
    const int32_t Wg = D->dist_Wg;
    std::vector<std::vector<uint8_t>> own_pair_on;  // [own tail front (relative)] -> mask, dist mode only
    if (dist) own_pair_on.resize((size_t)(S.nsuper - S.dense_first));
    int64_t dense_pairs_all = 0, dense_pairs_kept = 0, dense_tiles_all = 0, dense_tiles_kept = 0;
    for (int32_t l = 0; l < S.nlevels; ++l) {
      int64_t total_e = 0, total_l = 0;
      for (int64_t i = S.level_tile_ptr[l]; i < S.level_tile_ptr[l + 1]; ++i) {
        const int32_t g = S.level_tiles[i];
        for (int64_t c = dptr[g]; c < dmid[g]; ++c) total_e += combo_cost(c);
        for (int64_t c = dmid[g]; c < dptr[g + 1]; ++c) total_l += combo_cost(c);
      }
      // dense tail: the level's (single) front j = dense_first + jj receives every earlier tail front; the last
      // look_depth of them are "late", the others "early" -- implicit items, one per (pair of tiles, K segment).
      // Distributed tail: late = the sources of the target's own group and of the group before it (they arrive while
      // the chain advances); everything older is applied by the per-group batches.
      int32_t dj = -1, dcnt_e = 0, dcnt_l = 0;
      std::vector<std::pair<int32_t, int32_t>> segs_e, segs_l;  // descendant ranges of the level's dense items
      std::vector<uint8_t> pair_on;                             // per tile pair of the dense target: does it get items
      const int64_t dunit = 1 + (NB + KC - 1) / KC;  // cost units of one tail descendant on one tile
      int32_t nseg_older = -1;  // look-ahead split: K segments of the level's late dense items that do NOT read the newest source
      int32_t dfr = -1;  // the tail fronts lie on a chain: at most one of them per level
      if (D->dense_on)
        for (int32_t q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q)
          if (S.level_fronts[q] >= S.dense_first) dfr = S.level_fronts[q];
      if (dfr >= 0) {
        const int32_t fr = dfr;
        if (D->keep_front[fr]) {
          dj = fr;
          const int32_t jj = fr - S.dense_first;
          if (dist) {
            const int32_t lo = std::max(0, (jj / Wg - 1) * Wg);
            dcnt_l = jj - lo;
            dcnt_e = 0;  // (the batches)
          } else {
            dcnt_l = lookahead ? std::min<int32_t>(depth, jj) : jj;
            dcnt_e = jj - dcnt_l;
          }
          const int64_t ntl = S.tile_base[fr + 1] - S.tile_base[fr];
          // K segments = contiguous ranges of ACTIVE descendants (the same for every tile of the front), about
          // dense_items items per launch: every item writes two 128 KB slabs that k_reduce reads back, so few long
          // items beat many short ones as long as the launch still fills the chip a few times over
          const int64_t npairs = (ntl + 1) / 2;
          // rows of the target that NO active descendant reaches receive nothing but padding: their tile pairs get no
          // items (below the dense region the fronts of one side branch do not reach the columns of the others, nor
          // the part of the region sorted to its start)
          pair_on.assign((size_t)npairs, 1);
          if (!tail_src.empty() && jj > 0) {
            const std::vector<int32_t>& src = tail_src[(size_t)jj];
            const auto a_end = std::lower_bound(src.begin(), src.end(), jj);
            const int32_t c0t = S.sn_start[S.dense_first], c0j = S.sn_start[fr];
            for (int64_t pq = 0; pq < npairs; ++pq) {
              const int64_t lo = (int64_t)c0j + 2 * TM * pq, hi = std::min<int64_t>(lo + 2 * TM, S.n);
              const int32_t f_lo = tail_col_front[(size_t)(lo - c0t)], f_hi = tail_col_front[(size_t)(hi - 1 - c0t)];
              bool need = f_lo <= jj;  // the target's own columns
              for (int32_t f = std::max(f_lo, jj + 1); f <= f_hi && !need; ++f) {
                const std::vector<int32_t>& sf = tail_src[(size_t)f];
                auto x = src.begin();
                auto y = sf.begin();
                while (x != a_end && y != sf.end()) {
                  if (*x < *y) ++x;
                  else if (*y < *x) ++y;
                  else { need = true; break; }
                }
              }
              pair_on[(size_t)pq] = need ? 1 : 0;
            }
          }
          int64_t np_on = 0;
          for (uint8_t v : pair_on) np_on += v;
          // K segments: about dense_items items per launch, and -- when the plan may choose (dense_fill) -- a count that
          // fills the last round of workgroups: the items of a launch last about equally long, so I items on 256 CUs take
          // ceil(I / 256) rounds whatever I is (1M config: 1100 items = 4.3 rounds paid as 5)
          const int64_t want = std::max<int64_t>(1, (dense_items + std::max<int64_t>(1, np_on) / 2) / std::max<int64_t>(1, np_on));
          std::vector<std::pair<int32_t, int32_t>> runs;
          auto cut_runs = [&](int64_t total, int64_t nseg, std::vector<std::pair<int32_t, int32_t>>* out) -> int64_t {
            int64_t cnt = 0;
            for (auto& r : runs) {
              const int64_t len = r.second - r.first;
              const int64_t ns_r = std::max<int64_t>(1, std::min<int64_t>(len, (nseg * len + total / 2) / total));
              for (int64_t q = 0; q < ns_r; ++q) {
                const int32_t a = r.first + (int32_t)(len * q / ns_r), b = r.first + (int32_t)(len * (q + 1) / ns_r);
                if (b > a) {
                  ++cnt;
                  if (out) out->push_back({a, b});
                }
              }
            }
            return cnt;
          };
          bool taper = false;
          auto build = [&](int32_t lo, int32_t hi, std::vector<std::pair<int32_t, int32_t>>& out) -> int64_t {
            out.clear();
            const int64_t total = active_runs(jj, lo, hi, runs);
            if (total == 0) return 0;
            const int64_t cap = std::min<int64_t>(64, total);
            int64_t nseg = std::min(cap, want);
            if (dense_fill && np_on > 0) {
              double best = -1.0;
              for (int64_t ns = std::max<int64_t>(1, want * 2 / 3); ns <= std::min(cap, want * 3 / 2 + 1); ++ns) {
                const int64_t items = np_on * cut_runs(total, ns, nullptr);
                const int64_t rounds = (items + dense_fill - 1) / dense_fill;
                const double score = (double)items / (double)(rounds * dense_fill) - 0.02 * std::fabs((double)(ns - want)) / (double)want;
                if (score > best) { best = score; nseg = ns; }
              }
            }
            cut_runs(total, nseg, &out);
            // TAPER (long launches only): the items of a launch start in list order, K-segment major, and last about as long as
            // their segment is deep -- with equal segments the chip idles at the end of a launch while the last round of
            // workgroups finishes (measured ~8 % of a serialised 9.8 ms launch at 1M).  The second-to-last segment is therefore
            // cut in two and the last one in four: the launch ends on quarter-length items (a few more partial slabs per tile).
            if (taper && out.size() >= 3) {
              std::vector<std::pair<int32_t, int32_t>> tp(out.begin(), out.end() - 2);
              auto split = [&](std::pair<int32_t, int32_t> sgm, int parts) {
                const int32_t len = sgm.second - sgm.first;
                for (int q = 0; q < parts; ++q) {
                  const int32_t a = sgm.first + (int32_t)((int64_t)len * q / parts), b = sgm.first + (int32_t)((int64_t)len * (q + 1) / parts);
                  if (b > a) tp.push_back({a, b});
                }
              };
              split(out[out.size() - 2], 2);
              split(out[out.size() - 1], 4);
              out.swap(tp);
            }
            return total;
          };
          int64_t act_l;
          if (dist && dcnt_l >= 2 && !(tune_env("SCILMM_DIST_NOSPLIT") && tune_env("SCILMM_DIST_NOSPLIT")[0] == '1')) {
            // look-ahead split (multi-GPU critical path): the NEWEST source, panel jj - 1, gets K segments of its own, listed
            // last -- the level loop launches the segments of the older sources before it waits for that panel's broadcast
            std::vector<std::pair<int32_t, int32_t>> newest;
            act_l = build(jj - dcnt_l, jj - 1, segs_l);
            nseg_older = (int32_t)segs_l.size();
            act_l += build(jj - 1, jj, newest);
            segs_l.insert(segs_l.end(), newest.begin(), newest.end());
            if (newest.empty()) nseg_older = -1;  // (the newest source does not reach this target: nothing to wait for separately)
          } else {
            act_l = build(jj - dcnt_l, jj, segs_l);
          }
          // (the early launch of a long tail: 1024-item launches, several rounds of workgroups)
          const char* etp = tune_env("SCILMM_DENSE_TAPER");
          taper = etp ? etp[0] == '1' : dense_items >= 1024;
          const int64_t act_e = dist ? 0 : build(0, dcnt_e, segs_e);
          taper = false;
          dense_pairs_all += dist ? dcnt_l : jj;
          dense_pairs_kept += act_e + act_l;
          for (uint8_t v : pair_on) { dense_tiles_all += 1; dense_tiles_kept += v; }
          if (dist) own_pair_on[(size_t)jj] = pair_on;
          total_e += ntl * dunit * act_e;
          total_l += ntl * dunit * act_l;
        }
      }
      const int64_t big = (int64_t)1 << 60;
      // (at most ~8192 items per launch: the slabs of a level must stay a few GB on the largest patterns)
      const int64_t cap_e = std::max<int64_t>(max_item, total_e / 8192), cap_l = std::max<int64_t>(max_item, total_l / 8192);
      const int64_t per_e = allow_split ? std::min(cap_e, std::max<int64_t>(min_item, (total_e + target_items - 1) / target_items)) : big;
      const int64_t per_l = allow_split ? std::min(cap_l, std::max<int64_t>(min_item, (total_l + target_items - 1) / target_items)) : big;
      int64_t slots = 0;
      const int64_t nde = (int64_t)segs_e.size(), ndl = (int64_t)segs_l.size();
      std::vector<int32_t> dbase_e, dbase_l;  // per tile of front dj: first dense slab, -1 = subtract directly
      if (dj >= 0) {
        dbase_e.assign((size_t)(S.tile_base[dj + 1] - S.tile_base[dj]), -1);
        dbase_l.assign(dbase_e.size(), -1);
      }
      std::vector<int32_t> order(S.level_tiles.begin() + S.level_tile_ptr[l], S.level_tiles.begin() + S.level_tile_ptr[l + 1]);
      // heaviest tiles (most combos) first: the long items of a launch start early
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return (S.combo_ptr[a + 1] - S.combo_ptr[a]) > (S.combo_ptr[b + 1] - S.combo_ptr[b]);
      });
      for (size_t oi = 0; oi < order.size(); ++oi) {
        const int32_t g = order[oi];
        const size_t fe = work_early.size(), fl = work.size();
        const int64_t ne = cut(g, dptr[g], dmid[g], per_e, work_early);
        const int64_t nl = cut(g, dmid[g], dptr[g + 1], per_l, work);
        // a single dense item of a launch subtracts straight into the panel (the early and the late launch of a
        // level never overlap in time); two or more go through partial slabs
        // (the implicit dense-tail items of the tile count like explicit ones: dte / dtl of them)
        const bool dtile = dj >= 0 && S.tile_front[g] == dj && pair_on[(size_t)((g - S.tile_base[dj]) / 2)];
        const int64_t dte = dtile ? nde : 0, dtl = dtile ? ndl : 0;
        const int64_t pe = (ne + dte) >= 2 ? ne : 0, pl = (nl + dtl) >= 2 ? nl : 0;
        const int64_t pde = (ne + dte) >= 2 ? dte : 0, pdl = (nl + dtl) >= 2 ? dtl : 0;
        if (ne == 1 && pe == 0) work_early[fe].slot = -1;
        if (nl == 1 && pl == 0) work[fl].slot = -1;
        if (pe + pde > 0) {
          pslot_e[g] = (int32_t)slots;
          pnseg_e[g] = (int32_t)(pe + pde);
          red_tiles_e.push_back(g);
          for (int64_t k = 0; k < pe; ++k) work_early[fe + k].slot = (int32_t)(slots + k);
          if (pde > 0) dbase_e[(size_t)(g - S.tile_base[dj])] = (int32_t)(slots + pe);
          slots += pe + pde;
        }
        if (pl + pdl > 0) {
          pslot[g] = (int32_t)slots;
          pnseg[g] = (int32_t)(pl + pdl);
          red_tiles.push_back(g);
          for (int64_t k = 0; k < pl; ++k) work[fl + k].slot = (int32_t)(slots + k);
          if (pdl > 0) dbase_l[(size_t)(g - S.tile_base[dj])] = (int32_t)(slots + pl);
          slots += pl + pdl;
        }
      }
      // Launch order = K-segment major, tile minor: the workgroups resident at any moment then work on the SAME few
      // descendant panels (their target-column rows -- the B operand -- are shared by every tile of the level), so that
      // operand comes out of the L2s / the infinity cache instead of HBM once per tile.  (Slots were assigned above:
      // the partial slabs of a tile stay contiguous whatever the launch order.)
      {
        auto seg_major = [&](std::vector<UpdWork>& v, size_t first) {
          if (v.size() - first < 2) return;
          std::vector<std::pair<int32_t, int32_t>> key(v.size() - first);  // (segment index within its tile, position)
          int32_t seg = 0;
          for (size_t k = first; k < v.size(); ++k) {
            seg = (k > first && v[k].tile == v[k - 1].tile) ? seg + 1 : 0;
            key[k - first] = {seg, (int32_t)(k - first)};
          }
          std::stable_sort(key.begin(), key.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) { return a.first < b.first; });
          std::vector<UpdWork> tmp(v.begin() + first, v.end());
          for (size_t k = 0; k < key.size(); ++k) v[first + k] = tmp[(size_t)key[k].second];
        };
        seg_major(work_early, (size_t)D->early_ptr[l]);
        seg_major(work, (size_t)D->work_ptr[l]);
      }
      if (dj >= 0) {
        // K-segment major, tile-pair minor (same reason as above); a pair = two vertically adjacent tiles of the front
        const int32_t ntl = (int32_t)(S.tile_base[dj + 1] - S.tile_base[dj]);
        auto emit = [&](std::vector<DenseWork>& out, const std::vector<std::pair<int32_t, int32_t>>& segs, const std::vector<int32_t>& base) {
          for (size_t sg = 0; sg < segs.size(); ++sg) {
            const int32_t k0 = segs[sg].first, k1 = segs[sg].second;
            for (int32_t q = 0; q < ntl; q += 2) {
              if (!pair_on[(size_t)(q / 2)]) continue;
              const int32_t nt2 = std::min<int32_t>(2, ntl - q);
              DenseWork wk{dj, q, nt2, k0, k1, base[(size_t)q] < 0 ? -1 : base[(size_t)q] + (int32_t)sg,
                           (nt2 == 2 && base[(size_t)q + 1] >= 0) ? base[(size_t)q + 1] + (int32_t)sg : -1, 0};
              out.push_back(wk);
            }
          }
        };
        emit(dwork_e, segs_e, dbase_e);
        emit(dwork_l, segs_l, dbase_l);
      }
      D->dwork_e_ptr[l + 1] = (int64_t)dwork_e.size();
      D->dwork_l_ptr[l + 1] = (int64_t)dwork_l.size();
      {
        // (items are K-segment major: the first nseg_older segments x the active tile pairs are the older sources' items)
        int64_t np_on_l = 0;
        if (dj >= 0 && nseg_older >= 0)
          for (uint8_t v : pair_on) np_on_l += v;
        D->dwork_l_mid[(size_t)l] = (dj >= 0 && nseg_older >= 0) ? D->dwork_l_ptr[l] + (int64_t)nseg_older * np_on_l : D->dwork_l_ptr[l + 1];
      }
      max_slots = std::max(max_slots, slots);
      D->lev_cost_e.push_back(total_e);
      D->lev_cost_l.push_back(total_l);
      D->work_ptr[l + 1] = (int64_t)work.size();
      D->early_ptr[l + 1] = (int64_t)work_early.size();
      D->red_ptr[l + 1] = (int64_t)red_tiles.size();
      D->red_ptr_e[l + 1] = (int64_t)red_tiles_e.size();
    }
    D->max_slots = std::max<int64_t>(max_slots, 1);
    if (dist) {
      // ---- batches of the distributed tail: batch g = the contribution of source group g (tail fronts
      //      [g Wg, (g+1) Wg)) to every own target that lies at least two groups later -- one item per (target, tile
      //      pair), K = the hull of the group's active sources, subtracted straight from the panel (batches run one
      //      after the other on one stream, and a target's late update waits for its last batch)
      const int32_t nT = S.nsuper - S.dense_first;
      const int32_t ngroups = (nT + Wg - 1) / Wg;
      std::vector<DenseWork> dwork_b;
      D->dbatch_ptr.assign((size_t)ngroups + 1, 0);
      std::vector<std::pair<int32_t, int32_t>> runs;
      for (int32_t g = 0; g < ngroups; ++g) {
        for (int32_t jj = (g + 2) * Wg; jj < nT; ++jj) {
          const int32_t fr = S.dense_first + jj;
          if (!D->keep_front[fr]) continue;
          if (active_runs(jj, g * Wg, std::min((g + 1) * Wg, nT), runs) == 0) continue;
          const int32_t k0 = runs.front().first, k1 = runs.back().second;
          const int32_t ntl = (int32_t)(S.tile_base[fr + 1] - S.tile_base[fr]);
          const std::vector<uint8_t>& pon = own_pair_on[(size_t)jj];
          for (int32_t q = 0; q < ntl; q += 2) {
            if (!pon.empty() && !pon[(size_t)(q / 2)]) continue;
            dwork_b.push_back(DenseWork{fr, q, std::min<int32_t>(2, ntl - q), k0, k1, -1, -1, 0});
          }
        }
        D->dbatch_ptr[(size_t)g + 1] = (int64_t)dwork_b.size();
      }
      if (dwork_b.empty()) dwork_b.push_back(DenseWork{});
      const DenseWork* ddb;
      if ((st = upload(sym, D, dwork_b, &ddb)) != SCILMM_OK) return st;
      D->d_dwork_b = (DenseWork*)ddb;
      if (getenv("SCILMM_VERBOSE"))
        fprintf(stderr, "[scilmm plan] rank %d: %lld batch items in %d source groups of %d tail panels\n", D->rank,
                (long long)D->dbatch_ptr[(size_t)ngroups], ngroups, Wg);
    }
    {
      if (dwork_e.empty()) dwork_e.push_back(DenseWork{});
      if (dwork_l.empty()) dwork_l.push_back(DenseWork{});
      const DenseWork* ddw;
      if ((st = upload(sym, D, dwork_e, &ddw)) != SCILMM_OK) return st;
      D->d_dwork_e = (DenseWork*)ddw;
      if ((st = upload(sym, D, dwork_l, &ddw)) != SCILMM_OK) return st;
      D->d_dwork_l = (DenseWork*)ddw;
      if (getenv("SCILMM_VERBOSE") && dense_pairs_all > 0)
        fprintf(stderr, "[scilmm plan] dense tail: %lld of %lld (target, descendant) panel pairs carry true entries, %lld of %lld target tile pairs are reached by a descendant (the others are padding only and skipped)\n",
                (long long)dense_pairs_kept, (long long)dense_pairs_all, (long long)dense_tiles_kept, (long long)dense_tiles_all);
      if (getenv("SCILMM_VERBOSE"))
        fprintf(stderr, "[scilmm plan] dense tail: fronts %d..%d (%d wide), %lld early + %lld late implicit items (k_dense_b)\n",
                S.dense_first, S.nsuper - 1, S.dense_first < S.nsuper ? S.n - S.sn_start[S.dense_first] : 0,
                (long long)D->dwork_e_ptr[S.nlevels], (long long)D->dwork_l_ptr[S.nlevels]);
    }
    if (work_early.empty()) work_early.push_back(UpdWork{0, -1, 0, 0});
    {
      const UpdWork* dwe;
      if ((st = upload(sym, D, work_early, &dwe)) != SCILMM_OK) return st;
      D->d_work_early = (UpdWork*)dwe;
    }
    if (red_tiles_e.empty()) red_tiles_e.push_back(0);
    if ((st = upload(sym, D, red_tiles_e, &tmp)) != SCILMM_OK) return st;
    D->d_red_tiles_e = (int32_t*)tmp;
    if ((st = upload(sym, D, pslot_e, &tmp)) != SCILMM_OK) return st;
    D->d_tile_pslot_e = (int32_t*)tmp;
    if ((st = upload(sym, D, pnseg_e, &tmp)) != SCILMM_OK) return st;
    D->d_tile_pnseg_e = (int32_t*)tmp;
    if (red_tiles.empty()) red_tiles.push_back(0);
    if ((st = upload(sym, D, red_tiles, &tmp)) != SCILMM_OK) return st;
    D->d_red_tiles = (int32_t*)tmp;
    if (work.empty()) work.push_back(UpdWork{0, -1, 0, 0});
    const UpdWork* dw;
    if ((st = upload(sym, D, work, &dw)) != SCILMM_OK) return st;
    D->d_work = (UpdWork*)dw;
    if ((st = upload(sym, D, pslot, &tmp)) != SCILMM_OK) return st;
    D->d_tile_pslot = (int32_t*)tmp;
    if ((st = upload(sym, D, pnseg, &tmp)) != SCILMM_OK) return st;
    D->d_tile_pnseg = (int32_t*)tmp;
    void* sc = nullptr;
    // three regions: early slabs by level parity (two side streams), late slabs (main stream)
    HIPCHK(hipMalloc(&sc, sizeof(double) * (size_t)4 * (size_t)D->max_slots * TM * NB));
    D->allocs.push_back(sc);
    D->scratch = (double*)sc;
  }
  plap("work items, slabs");
  // ---- dense-chain plan for the triangular sweeps
  {
    const int32_t ns = S.nsuper;
    // the sweep set: all fronts of the top levels, as many levels as fit the cap (the dense chain and the few
    // wide fronts just below it; every dependency of a member is either a member or finished by the level kernels)
    // (a level joins while it has at most `wide` fronts: a pull step costs ~8 us whatever its size, so the many
    // small fronts of the lower levels stay with the level kernels -- measured optimum at the 100k pedigree)
    const char* ecap = tune_env("SCILMM_CHAIN_CAP");
    const char* ewide = tune_env("SCILMM_CHAIN_WIDE");
    const int32_t cap = ecap ? atoi(ecap) : 2048, wide = ewide ? atoi(ewide) : 12;
    int32_t l0 = S.nlevels;
    while (l0 > 0 && S.level_ptr[l0] - S.level_ptr[l0 - 1] <= wide && S.level_ptr[S.nlevels] - S.level_ptr[l0 - 1] <= cap) --l0;
    const char* enc = tune_env("SCILMM_NO_CHAIN");
    int32_t T = l0 < S.nlevels ? S.level_ptr[S.nlevels] - S.level_ptr[l0] : 0;
    if (S.nlevels - l0 < 4 || (enc && enc[0] == '1')) T = 0;
    if (D->world > 1) T = 0;  // a distributed factor is swept level by level with a collective per tail block (run_rhs)
    D->chain_T = T;
    D->chain_l0 = l0;
    if (T > 0) {
      std::vector<int32_t> chain(T), pos(ns, -1);
      for (int32_t i = 0; i < T; ++i) {
        chain[i] = S.level_fronts[S.level_ptr[l0] + i];  // level order = a topological order of the update pairs
        pos[chain[i]] = i;
      }
      std::vector<std::vector<ChainPair>> fw(T), bw(T);
      std::vector<int32_t> colmap;  // forward, non-contiguous pairs: target column -> row of the pair (or -1)
      std::vector<std::pair<int32_t, int32_t>> outside;  // (descendant, pair id): chain target, descendant below the chain
      for (int32_t i = 0; i < T; ++i) {
        const int32_t t = chain[i];
        for (int64_t e = S.upd_ptr[t]; e < S.upd_ptr[t + 1]; ++e) {
          const int32_t d = S.upd_src[e];
          if (pos[d] >= 0) {
            int32_t moff = 0;
            if (S.upd_jp0[e] < 0) {
              moff = (int32_t)colmap.size();
              colmap.resize(colmap.size() + NB, -1);
              const int32_t* rd = S.sn_rows.data() + S.sn_rowptr[d];
              for (int32_t q = S.upd_p0[e]; q < S.upd_p1[e]; ++q) colmap[(size_t)moff + (rd[q] - S.sn_start[t])] = q - S.upd_p0[e];
            }
            fw[i].push_back(ChainPair{pos[d], S.upd_p0[e], S.upd_p1[e] - S.upd_p0[e], S.upd_jp0[e], moff});
            bw[pos[d]].push_back(ChainPair{i, S.upd_p0[e], S.upd_p1[e] - S.upd_p0[e], S.upd_jp0[e], 0});
          } else {
            outside.push_back({d, (int32_t)e});
          }
        }
      }
      std::vector<int32_t> fptr(T + 1, 0), bptr(T + 1, 0);
      std::vector<ChainPair> fl, bl;
      for (int32_t i = 0; i < T; ++i) {
        std::sort(fw[i].begin(), fw[i].end(), [](const ChainPair& a, const ChainPair& b) { return a.other < b.other; });
        std::sort(bw[i].begin(), bw[i].end(), [](const ChainPair& a, const ChainPair& b) { return a.other > b.other; });
        fl.insert(fl.end(), fw[i].begin(), fw[i].end());
        bl.insert(bl.end(), bw[i].begin(), bw[i].end());
        fptr[i + 1] = (int32_t)fl.size();
        bptr[i + 1] = (int32_t)bl.size();
      }
      // group by descendant, order by first row, merge adjacent row ranges (rows of consecutive chain blocks)
      std::sort(outside.begin(), outside.end(), [&](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) {
        if (a.first != b.first) return a.first < b.first;
        return S.upd_p0[a.second] < S.upd_p0[b.second];
      });
      std::vector<int64_t> gptr;
      std::vector<int32_t> gpairs;  // triples (descendant, p0, p1)
      for (size_t k = 0; k < outside.size(); ++k) {
        const int32_t d = outside[k].first, e = outside[k].second;
        const bool newgrp = k == 0 || d != outside[k - 1].first;
        if (newgrp) gptr.push_back((int64_t)gpairs.size() / 3);
        if (!newgrp && gpairs.back() == S.upd_p0[e]) {
          gpairs.back() = S.upd_p1[e];
        } else {
          gpairs.push_back(d);
          gpairs.push_back(S.upd_p0[e]);
          gpairs.push_back(S.upd_p1[e]);
        }
      }
      gptr.push_back((int64_t)gpairs.size() / 3);
      std::vector<int32_t> gslot, fold;
      {
        // one workgroup sweeps its rows 32 at a time (~3.5 us per step): a descendant with 10^4 rows in the chain
        // would take a millisecond alone, so long groups are cut into slices of <= slice_rows rows
        const char* esr = tune_env("SCILMM_PUSH_SLICE");
        const int64_t slice_rows = std::max<int64_t>(256, esr ? atoll(esr) : 512);
        std::vector<int64_t> gptr2;
        std::vector<int32_t> gp2;
        int32_t nslots = 0;
        for (size_t g = 0; g + 1 < gptr.size(); ++g) {
          int64_t rows = 0;
          for (int64_t q = gptr[g]; q < gptr[g + 1]; ++q) rows += gpairs[3 * q + 2] - gpairs[3 * q + 1];
          const int64_t nsl = (rows + slice_rows - 1) / slice_rows;
          if (nsl <= 1) {
            gptr2.push_back((int64_t)gp2.size() / 3);
            gp2.insert(gp2.end(), gpairs.begin() + 3 * gptr[g], gpairs.begin() + 3 * gptr[g + 1]);
            gslot.push_back(-1);
            continue;
          }
          const int64_t per = (rows + nsl - 1) / nsl;
          fold.push_back(gpairs[3 * gptr[g]]);
          fold.push_back(nslots);
          int32_t made = 0;
          int64_t acc = 0;
          gptr2.push_back((int64_t)gp2.size() / 3);
          gslot.push_back(nslots + made);
          ++made;
          for (int64_t q = gptr[g]; q < gptr[g + 1]; ++q) {
            int32_t a = gpairs[3 * q + 1];
            const int32_t b = gpairs[3 * q + 2];
            while (a < b) {
              if (acc == per) {  // start the next slice
                gptr2.push_back((int64_t)gp2.size() / 3);
                gslot.push_back(nslots + made);
                ++made;
                acc = 0;
              }
              const int32_t take = (int32_t)std::min<int64_t>(b - a, per - acc);
              gp2.push_back(gpairs[3 * q]);
              gp2.push_back(a);
              gp2.push_back(a + take);
              a += take;
              acc += take;
            }
          }
          fold.push_back(made);
          nslots += made;
        }
        gptr2.push_back((int64_t)gp2.size() / 3);
        gptr.swap(gptr2);
        gpairs.swap(gp2);
        D->n_fold = (int64_t)fold.size() / 3;
        if (nslots > 0) {
          void* pp = nullptr;
          HIPCHK(hipMalloc(&pp, sizeof(double) * (size_t)nslots * NB * RPMAX));
          D->allocs.push_back(pp);
          D->d_push_partial = (double*)pp;
        }
        if (fold.empty()) fold.assign(3, 0);
        if (gslot.empty()) gslot.push_back(-1);
      }
      D->chain_groups = (int64_t)gptr.size() - 1;
      const int32_t* t32; const int64_t* t64; const ChainPair* tcp;
      if (fl.empty()) fl.push_back(ChainPair{0, 0, 0, 0, 0});
      if (bl.empty()) bl.push_back(ChainPair{0, 0, 0, 0, 0});
      if (colmap.empty()) colmap.push_back(-1);
      if ((st = upload(sym, D, colmap, &t32)) != SCILMM_OK) return st; D->d_colmap = (int32_t*)t32;
      if (gpairs.empty()) gpairs.assign(3, 0);
      const int64_t n_ranges = (int64_t)gpairs.size() / 3;
      if ((st = upload(sym, D, chain, &t32)) != SCILMM_OK) return st; D->d_chain = (int32_t*)t32;
      if ((st = upload(sym, D, fptr, &t32)) != SCILMM_OK) return st; D->d_cf_ptr = (int32_t*)t32;
      if ((st = upload(sym, D, bptr, &t32)) != SCILMM_OK) return st; D->d_cb_ptr = (int32_t*)t32;
      if ((st = upload(sym, D, fl, &tcp)) != SCILMM_OK) return st; D->d_cf = (ChainPair*)tcp;
      if ((st = upload(sym, D, bl, &tcp)) != SCILMM_OK) return st; D->d_cb = (ChainPair*)tcp;
      if ((st = upload(sym, D, gptr, &t64)) != SCILMM_OK) return st; D->d_cg_ptr = (int64_t*)t64;
      if ((st = upload(sym, D, gpairs, &t32)) != SCILMM_OK) return st; D->d_cg_pairs = (int32_t*)t32;
      if ((st = upload(sym, D, gslot, &t32)) != SCILMM_OK) return st; D->d_cg_slot = (int32_t*)t32;
      if ((st = upload(sym, D, fold, &t32)) != SCILMM_OK) return st; D->d_fold = (int32_t*)t32;
      std::vector<int32_t> zeros((size_t)T * (RPMAX / CW) + 4, 0);
      if ((st = upload(sym, D, zeros, &t32)) != SCILMM_OK) return st;
      D->d_chain_flags = (int32_t*)t32;
      D->d_chain_err = D->d_chain_flags + (size_t)T * (RPMAX / CW);
      HIPCHK(hipHostMalloc((void**)&D->h_chain_err, sizeof(int32_t), hipHostMallocDefault));
      *D->h_chain_err = 0;
      if (getenv("SCILMM_VERBOSE"))
        fprintf(stderr, "[scilmm plan] chain sweep: %d fronts (levels %d..%d), %lld inner pairs (%lld column maps), %lld outside pairs in %lld groups\n",
                T, l0, S.nlevels - 1, (long long)fl.size(), (long long)(colmap.size() / NB), (long long)outside.size(), (long long)D->chain_groups);
      (void)n_ranges;
    }
  }
  plap("chain sweep plan");
  *out = D;
  return SCILMM_OK;
}

int set_attrs(scilmm_symbolic* sym, Dev* D) {
  if (D->attrs_set) return SCILMM_OK;
  const int big = 150 * 1024;
  HIPCHK(hipFuncSetAttribute((const void*)k_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_fwd<true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_fwd<true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_fwd<true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_fwd<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_fwd<false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_fwd<false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dense32, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dense_h, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dense_b, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_update2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_update2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  D->attrs_set = true;
  return SCILMM_OK;
}

int ensure_work(scilmm_symbolic* sym, Dev* D) {
  const Symbolic& S = *sym->S;
  size_t bytes = (size_t)std::max(S.n, 1) * RPMAX * sizeof(double);
  if (D->world > 1 && !D->work_external) {
    sym->err = "multi-GPU: scilmm_dist_set_work has not been called (the sweeps' buffers must be addressable by the communication layer)";
    return SCILMM_ERR_STATE;
  }
  if (!D->W) HIPCHK(hipMalloc((void**)&D->W, bytes));
  if (!D->X) HIPCHK(hipMalloc((void**)&D->X, bytes));
  return set_attrs(sym, D);
}

int ensure_io(scilmm_symbolic* sym, Dev* D, size_t doubles) {
  if (D->io_cap >= doubles) return SCILMM_OK;
  if (D->IO) (void)hipFree(D->IO);
  D->IO = nullptr;
  D->io_cap = 0;
  HIPCHK(hipMalloc((void**)&D->IO, std::max<size_t>(doubles, 1) * sizeof(double)));
  D->io_cap = doubles;
  return SCILMM_OK;
}

inline int rp_of(int rc) { return (rc + 15) & ~15; }
inline int ldy_of(int rp) { return (rp % 32 == 16) ? rp : rp + 16; }

}  // namespace

struct scilmm_factor {
  scilmm_symbolic* sym = nullptr;
  double* L = nullptr;
  double* invD = nullptr;
  double* logd = nullptr;
  int32_t* status = nullptr;
  bool valid = false;
  int device = -1;            // device of the owning handle (kept here: the factor may outlive its symbolic handle's Dev)
  bool external = false;      // L / invD / logd belong to the caller (scilmm_factor_create_external)
  bool pending = false;       // a factorization has been queued (scilmm_refactorize_async) and not yet waited for
  bool inverted = false;      // L has been replaced by the selected inverse (scilmm_selected_inverse): no solves until refactorized
  float* L32 = nullptr;       // front precision 32: fp32 shadow of the dense-tail panels (k_shadow / k_dense_h), else null
  int64_t base32 = 0;         // index in L of the shadow's first entry (= sn_loff[dense_first])
  int32_t* h_status = nullptr;  // pinned host copy of *status, filled by the queued copy
};

namespace {

int finish_factorize(scilmm_factor* fac, int32_t* bad_col);

int run_factorize(scilmm_factor* fac, const double* sigma2, int32_t* bad_col, bool wait = true) {
  scilmm_symbolic* sym = fac->sym;
  if (fac->pending) {
    int stp = finish_factorize(fac, nullptr);
    if (stp != SCILMM_OK && stp != SCILMM_ERR_NOT_PD) return stp;
  }
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  for (int k = 0; k < S.K; ++k)
    if (!D->have_vals[k]) {
      sym->err = "scilmm_values_upload has not been called for every matrix";
      return SCILMM_ERR_STATE;
    }
  fac->valid = false;
  fac->inverted = false;
  if (D->world > 1 && (!sym->comm_fn || !D->comm)) {
    sym->err = "multi-GPU: scilmm_dist_init was called without a communication callback / stream";
    return SCILMM_ERR_STATE;
  }
  if (D->world > 1 && !fac->external) {
    sym->err = "multi-GPU: the factor must live in caller-owned storage (scilmm_factor_create_external)";
    return SCILMM_ERR_STATE;
  }
  {
    int stc = set_attrs(sym, D);
    if (stc != SCILMM_OK) return stc;
  }
  const bool dist = D->world > 1 && D->dist_first < S.nsuper;
  const size_t sm_upd = sizeof(double) * (size_t)(2 * KC * LDA + 2 * KC * LDB) + sizeof(int32_t) * TM;
  const size_t sm_potrf = sizeof(double) * (size_t)((NB / 2) * (NB + 1) + NJB * 16 * 17 + 32);
  hipStream_t st = D->stream;
  int64_t launches = 0;
  {
    // front precision 32: the dense-tail kernel reads an fp32 shadow of the finished tail panels (k_dense_h) -- of the rank's own
    // panels and of its ring slots when the tail is distributed (same rank-local offsets as L, minus the prelude).  It is
    // allocated with the first such factorization and kept (zeroed once: the kernel may read a few entries past a panel, into
    // the next panel's or the slack's, which must be finite); a device without room for it keeps k_dense32 (fp64 operands
    // rounded while they are staged).  SCILMM_TUNING=1 SCILMM_SHADOW=0: k_dense32.
    const char* esh = tune_env("SCILMM_SHADOW");
    const bool want = D->front_bits == 32 && D->dense_on && !(esh && esh[0] == '0');
    if (want && !fac->L32) {
      fac->base32 = S.sn_loff[S.dense_first];  // (= the prelude's size: the prelude is replicated and stored first on every rank)
      const size_t cnt32 = (size_t)(D->nL_local - fac->base32) + 16384;
      if (hipMalloc((void**)&fac->L32, sizeof(float) * cnt32) != hipSuccess) {
        (void)hipGetLastError();
        fac->L32 = nullptr;
      } else {
        HIPCHK(hipMemsetAsync(fac->L32, 0, sizeof(float) * cnt32, st));
      }
    } else if (!want && fac->L32) {
      HIPCHK(hipStreamSynchronize(st));
      (void)hipFree(fac->L32);
      fac->L32 = nullptr;
    }
  }
  HIPCHK(hipEventRecord(D->ev[0], st));
  HIPCHK(hipMemsetAsync(fac->L, 0, sizeof(double) * (size_t)std::max<int64_t>(D->nL_local, 1), st));
  int32_t big = 0x7fffffff;
  HIPCHK(hipMemcpyAsync(fac->status, &big, sizeof(int32_t), hipMemcpyHostToDevice, st));
  ValPtrs gen{}, dia{};
  for (int k = 0; k < S.K; ++k) {
    ValPtrs& t = S.is_diag[k] ? dia : gen;
    if (t.count >= 8) {
      sym->err = "more than 8 matrices of one kind";
      return SCILMM_ERR_ARG;
    }
    t.v[t.count] = D->vals[k];
    t.s2[t.count] = sigma2[k];
    t.count++;
  }
  if (S.nnz_pattern > 0) {
    int blocks = (int)std::min<int64_t>((S.nnz_pattern + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(256), 0, st, S.nnz_pattern, D->v.asm_dst, gen, fac->L);
    launches++;
  }
  if (dia.count > 0 && S.n > 0) {
    hipLaunchKernelGGL(k_add_diag, dim3((S.n + 255) / 256), dim3(256), 0, st, S.n, D->v.diag_dst, dia, fac->L);
    launches++;
  }
  HIPCHK(hipEventRecord(D->ev[1], st));
  HIPCHK(hipEventRecord(D->ev_asm, st));
  HIPCHK(hipStreamWaitEvent(D->side, D->ev_asm, 0));
  HIPCHK(hipStreamWaitEvent(D->side2, D->ev_asm, 0));
  if (D->side3) HIPCHK(hipStreamWaitEvent(D->side3, D->ev_asm, 0));
  if (D->outside_st) HIPCHK(hipStreamWaitEvent(D->outside_st, D->ev_asm, 0));
  if (dist) {
    // nothing may land in the ring (or be read-modify-written by a batch) before the storage is cleared and assembled
    HIPCHK(hipStreamWaitEvent(D->bstream, D->ev_asm, 0));
    HIPCHK(hipStreamWaitEvent(D->comm, D->ev_asm, 0));
  }
  const bool prof = D->profiling;
  constexpr int PE = 14;  // profiling events per level
  D->n_late_split = 0;
  if (prof && D->pev.size() < (size_t)PE * S.nlevels) {
    size_t old = D->pev.size();
    D->pev.resize((size_t)PE * S.nlevels, nullptr);
    for (size_t i = old; i < D->pev.size(); ++i) HIPCHK(hipEventCreate(&D->pev[i]));
  }
  if (prof && dist && D->bpev.size() < 2 * D->batch_ev.size()) {
    size_t old = D->bpev.size();
    D->bpev.resize(2 * D->batch_ev.size(), nullptr);
    for (size_t i = old; i < D->bpev.size(); ++i) HIPCHK(hipEventCreate(&D->bpev[i]));
  }
  // A dispatch counts WORK-ITEMS in 32 bits: cnt workgroups of `threads` threads must stay below 2^32 of them.
  // Every launch helper below cuts its grid at max_groups(threads) workgroups (k_outside was the only
  // chunked launch; the 1M plan has single launches of several million workgroups).
  auto max_groups = [](unsigned threads) -> int64_t { return (int64_t)(((uint64_t)1 << 32) / threads) - 1; };
  auto launch_update = [&](hipStream_t stream, const UpdWork* work, int64_t cnt, double* scratch_half) {
#ifdef SCILMM_DIAG
    if (D->ablate == 4) return;  // timing ablation: no explicit MFMA update items at all (WRONG numbers)
#endif
    for (int64_t o = 0; o < cnt; o += max_groups(UPD_THREADS)) {
      const unsigned c = (unsigned)std::min<int64_t>(cnt - o, max_groups(UPD_THREADS));
      if (D->use_mfma)
        hipLaunchKernelGGL((k_update2<true>), dim3(c), dim3(UPD_THREADS), sm_upd, stream, D->v, work + o, D->d_combos, fac->L, scratch_half);
      else
        hipLaunchKernelGGL((k_update2<false>), dim3(c), dim3(UPD_THREADS), sm_upd, stream, D->v, work + o, D->d_combos, fac->L, scratch_half);
      launches++;
    }
  };
  auto launch_dense = [&](hipStream_t stream, const DenseWork* dw, int64_t cnt, double* scratch_half) {
    for (int64_t o = 0; o < cnt; o += max_groups(1024)) {
      const unsigned c = (unsigned)std::min<int64_t>(cnt - o, max_groups(1024));  // (k_dense_h launches 2 c workgroups of 512)
      if (D->front_bits == 32 && fac->L32)
        // (two workgroups per work item: one per 128-row tile)
        hipLaunchKernelGGL(k_dense_h, dim3(2 * c), dim3(512), dense_h_lds, stream, D->v, S.dense_first, dw + o, fac->L, (const float*)fac->L32,
                           fac->base32, scratch_half, (const float*)D->d_zeros);
      else if (D->front_bits == 32)
        hipLaunchKernelGGL(k_dense32, dim3(c), dim3(512), sizeof(float) * (size_t)(2 * KC * LDA2F + 2 * KC * LDBF), stream, D->v,
                           S.dense_first, dw + o, fac->L, scratch_half);
      else
        hipLaunchKernelGGL(k_dense_b, dim3(c), dim3(512), sizeof(double) * (size_t)(2 * KBA * LDB), stream, D->v, S.dense_first, dw + o,
                           fac->L, scratch_half, (const double*)D->d_zeros);
      launches++;
    }
  };
  const size_t half = (size_t)D->max_slots * TM * NB;
  auto launch_cells = [&](hipStream_t stream, int which, int32_t l) {
    const Dev::CellSet& CS = D->cellset[which];
    const int64_t u0 = CS.level_ptr[l], u1 = CS.level_ptr[l + 1];
    if (u1 <= u0) return;
#ifdef SCILMM_DIAG
    if (D->ablate == 5) return;  // timing ablation: no cell-path updates (WRONG numbers)
#endif
    const int64_t nshort = CS.level_short[l], nlong = (u1 - u0) - nshort;
    const int64_t nblk = (nshort + 255) / 256 + (nlong + 3) / 4;
    if (nblk > max_groups(256)) {  // 2^24 workgroups = 4e9 cells of one level: beyond any plan the cell limit admits
      sym->err = "cell plan: one level has more target cells than a dispatch can hold";
      return;
    }
    hipLaunchKernelGGL(k_sparse_cells, dim3((unsigned)nblk), dim3(256), 0, stream, u0, nshort, u1 - u0, (const int64_t*)CS.dst,
                       (const int64_t*)CS.grp, (const int64_t*)CS.srct, (const int64_t*)CS.srcq, (const int32_t*)CS.md,
                       (const int32_t*)CS.wd, fac->L);
    launches++;
  };
  auto launch_reduce = [&](hipStream_t stream, const int32_t* tiles, int64_t cnt, const int32_t* pslot, const int32_t* pnseg,
                           const double* slabs) {
    const int64_t per = max_groups(128) / (NB / 4);  // tiles per dispatch
    for (int64_t o = 0; o < cnt; o += per) {
      const int64_t c = std::min(cnt - o, per);
      hipLaunchKernelGGL(k_reduce, dim3((unsigned)((NB / 4) * c)), dim3(128), 0, stream, D->v, tiles + o, pslot, pnseg, slabs, fac->L);
      launches++;
    }
  };
  auto has_early = [&](int32_t l) -> bool {
    return D->early_ptr[l + 1] > D->early_ptr[l] || D->cellset[0].level_ptr[l + 1] > D->cellset[0].level_ptr[l] ||
           D->dwork_e_ptr[l + 1] > D->dwork_e_ptr[l];
  };
  // Look-ahead: the EARLY part of level l+1 (descendants finished at levels <= l-1) runs on the side stream
  // while the main stream works through level l's latency-bound tail (late update, reduce, cells, potrf, trsm).
  auto launch_early = [&](int32_t l) -> int {
    const int64_t e0 = D->early_ptr[l], e1 = D->early_ptr[l + 1];
    if (!has_early(l)) return SCILMM_OK;
    const int sidx = D->serial_early ? 0 : l % D->nside;
    hipStream_t sd = sidx == 0 ? D->side : (sidx == 1 ? D->side2 : D->side3);
    // its youngest descendants sit look_depth + 1 levels below
    if (l >= D->look_depth + 1) HIPCHK(hipStreamWaitEvent(sd, D->lev_ev[2 * (l - D->look_depth - 1)], 0));
    // tail targets: the atomic contributions of the prelude (k_outside) must have landed before anything else
    // reads-modifies-writes a tail panel (the event has been recorded: these launches are deferred until it is)
    if (D->outside_on && l >= D->tail_level && D->out_wait_chunk[(size_t)l] >= 0)
      HIPCHK(hipStreamWaitEvent(sd, D->out_evs[(size_t)D->out_wait_chunk[(size_t)l]], 0));
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 5], sd));
    if (e1 > e0) launch_update(sd, D->d_work_early + e0, e1 - e0, D->scratch + (size_t)sidx * half);
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 8], sd));
    launch_dense(sd, D->d_dwork_e + D->dwork_e_ptr[l], D->dwork_e_ptr[l + 1] - D->dwork_e_ptr[l], D->scratch + (size_t)sidx * half);
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 9], sd));
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 6], sd));
    // fold the early partial slabs on the side stream as well: the main stream keeps only its own (rare) ones
    launch_reduce(sd, D->d_red_tiles_e + D->red_ptr_e[l], D->red_ptr_e[l + 1] - D->red_ptr_e[l], D->d_tile_pslot_e, D->d_tile_pnseg_e,
                  (const double*)(D->scratch + (size_t)sidx * half));
    launch_cells(sd, 0, l);  // early cells: same stream, after the early MFMA update of the same panels
    HIPCHK(hipEventRecord(D->lev_ev[2 * l + 1], sd));
    return SCILMM_OK;
  };
  // (early launches of tail levels are deferred until the k_outside launch has been queued: see below)
  auto deferred = [&](int32_t le) -> bool { return D->outside_on && le >= D->tail_level; };
  for (int32_t l = 1; l <= D->look_depth && l < S.nlevels; ++l) {
    if (deferred(l)) continue;
    int rc = launch_early(l);
    if (rc != SCILMM_OK) return rc;
  }
  const int32_t nT = S.nsuper - D->dist_first, Wg = D->dist_Wg;
  int32_t last_tail_level = -1;
  for (int32_t l = 0; l < S.nlevels; ++l) {
    const int64_t t0 = D->lv_tile_ptr[l], t1 = D->lv_tile_ptr[l + 1];
    const int32_t f0 = D->lv_ptr[l], f1 = D->lv_ptr[l + 1];
    double* sh = D->scratch + (size_t)3 * half;
    // early(l + depth) may start as soon as level l-1 is finished: issue it before this level's own kernels
    if (l >= 1 && l + D->look_depth < S.nlevels && !(l < D->tail_level && deferred(l + D->look_depth))) {
      int rc = launch_early(l + D->look_depth);
      if (rc != SCILMM_OK) return rc;
    }
    // (progressive k_outside: this level's tail panel needs the chunks that reach it, not the whole launch sequence)
    if (D->outside_on && l >= D->tail_level && D->out_wait_chunk[(size_t)l] >= 0 &&
        (l == D->tail_level || D->out_wait_chunk[(size_t)l] != D->out_wait_chunk[(size_t)l - 1]))
      HIPCHK(hipStreamWaitEvent(st, D->out_evs[(size_t)D->out_wait_chunk[(size_t)l]], 0));
    const int64_t w0 = D->work_ptr[l], w1 = D->work_ptr[l + 1];
    const int32_t tf = dist ? D->tail_of_level[l] : -1;  // the distributed front of this level
    const int32_t jj = tf >= 0 ? tf - D->dist_first : -1, grp = tf >= 0 ? jj / Wg : -1;
    if (has_early(l)) HIPCHK(hipStreamWaitEvent(st, D->lev_ev[2 * l + 1], 0));
    if (tf >= 0 && D->keep_front[tf]) {
      // own tail panel: every batch up to group grp - 2 has been applied to it; its late sources -- the panels of the
      // group before and of its own group so far -- have arrived (own ones: their level event)
      if (grp >= 2) HIPCHK(hipStreamWaitEvent(st, D->batch_ev[grp - 2], 0));
      // (look-ahead split: the wait for the NEWEST source, panel jj - 1, comes after the launch of the older sources' items)
      const int32_t q_end = D->dwork_l_mid[(size_t)l] < D->dwork_l_ptr[l + 1] ? jj - 1 : jj;
      for (int32_t q = std::max(0, (grp - 1) * Wg); q < q_end; ++q) HIPCHK(hipStreamWaitEvent(st, D->lev_ev[2 * S.sn_level[D->dist_first + q]], 0));
    }
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 0], st));
    if (w1 > w0) launch_update(st, D->d_work + w0, w1 - w0, sh);
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 10], st));
    const bool split_late = tf >= 0 && D->keep_front[tf] && D->dwork_l_mid[(size_t)l] < D->dwork_l_ptr[l + 1];
    launch_dense(st, D->d_dwork_l + D->dwork_l_ptr[l], (split_late ? D->dwork_l_mid[(size_t)l] : D->dwork_l_ptr[l + 1]) - D->dwork_l_ptr[l], sh);
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 11], st));
    if (split_late) {
      // the older sources' items are queued (they run while panel jj - 1 is still on its way); now its arrival, then its items
      HIPCHK(hipStreamWaitEvent(st, D->lev_ev[2 * S.sn_level[D->dist_first + jj - 1]], 0));
      if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 12], st));
      launch_dense(st, D->d_dwork_l + D->dwork_l_mid[(size_t)l], D->dwork_l_ptr[l + 1] - D->dwork_l_mid[(size_t)l], sh);
      if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 13], st));
      D->n_late_split++;
    }
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 1], st));
    // (the late slabs are folded by k_potrf / k_trsm on load: no k_reduce launch on the main stream's chain)
    launch_cells(st, 1, l);
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 2], st));
    if (f1 > f0) {
      hipLaunchKernelGGL(k_potrf, dim3((unsigned)(f1 - f0)), dim3(256), sm_potrf, st, D->v, D->d_level_fronts + f0, fac->L,
                         fac->invD, fac->logd, fac->status, (const int32_t*)D->d_tile_pslot, (const int32_t*)D->d_tile_pnseg,
                         (const double*)sh);
      launches++;
    }
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 3], st));
    if (t1 > t0) {
      if (D->use_mfma && D->trsm_lite && (t1 - t0) < ((int64_t)1 << 29))
        // (four workgroups of 32 rows per tile, register-light and LDS-free: starts beside the resident update kernel)
        hipLaunchKernelGGL(k_trsm_lite, dim3((unsigned)(4 * (t1 - t0))), dim3(256), 0, st, D->v, D->d_level_tiles + t0, fac->L, fac->invD,
                           (const int32_t*)D->d_tile_pslot, (const int32_t*)D->d_tile_pnseg, (const double*)sh);
      else if (D->use_mfma)
        hipLaunchKernelGGL(k_trsm<true>, dim3((unsigned)(t1 - t0)), dim3(256), 0, st, D->v, D->d_level_tiles + t0, fac->L, fac->invD,
                           (const int32_t*)D->d_tile_pslot, (const int32_t*)D->d_tile_pnseg, (const double*)sh);
      else
        hipLaunchKernelGGL(k_trsm<false>, dim3((unsigned)(t1 - t0)), dim3(256), 0, st, D->v, D->d_level_tiles + t0, fac->L, fac->invD,
                           (const int32_t*)D->d_tile_pslot, (const int32_t*)D->d_tile_pnseg, (const double*)sh);
      launches++;
    }
    if (fac->L32) {
      // front precision 32: the level's finished tail panel -> its fp32 shadow, before the level's event (the look-ahead
      // launches of later targets read the shadow only)
      for (int32_t q = f0; q < f1; ++q) {
        const int32_t fr = D->lv_fronts[(size_t)q];  // (the level's fronts THIS rank holds: another rank's panel is shadowed when it arrives, below)
        if (fr < S.dense_first || !D->keep_front[fr]) continue;
        const int64_t cnt = S.sn_loff[fr + 1] - S.sn_loff[fr];
        hipLaunchKernelGGL(k_shadow, dim3((unsigned)std::min<int64_t>((cnt + 255) / 256, 8192)), dim3(256), 0, st,
                           (const double*)(fac->L + D->loff[fr]), fac->L32 + (D->loff[fr] - fac->base32), cnt);
        launches++;
      }
    }
    if (prof) HIPCHK(hipEventRecord(D->pev[PE * l + 4], st));
    if (tf < 0) {
      HIPCHK(hipEventRecord(D->lev_ev[2 * l], st));
    } else {
      // ---- level with a distributed tail panel: the owner's panel (+ inverse diagonal block + log-sum) goes to every
      //      rank -- into the receiver's ring slot.  The level is complete (lev_ev) when the panel has arrived AND this
      //      rank's own kernels of the level (its prelude fronts) have finished.
      const int32_t root = jj % D->world;
      HIPCHK(hipEventRecord(D->done_ev[l], st));
      HIPCHK(hipStreamWaitEvent(D->comm, D->done_ev[l], 0));
      if (!D->keep_front[tf]) {
        // ring re-use: the slot's previous occupant (panel jj - G, group gd) must have been consumed -- by its batch and by
        // the late updates of this rank's targets of groups gd and gd + 1
        const int32_t gd = (jj - D->dist_G) / Wg;
        if (jj >= D->dist_G) {
          HIPCHK(hipStreamWaitEvent(D->comm, D->batch_ev[gd], 0));
          for (int32_t q = std::min<int32_t>(gd + 1, (int32_t)D->last_own_level.size() - 1); q >= 0; --q)
            if (D->last_own_level[q] >= 0) {
              HIPCHK(hipStreamWaitEvent(D->comm, D->done_ev[D->last_own_level[q]], 0));
              break;
            }
        }
      }
      const int64_t pm = S.sn_rowptr[tf + 1] - S.sn_rowptr[tf], pw = S.sn_start[tf + 1] - S.sn_start[tf];
      int rcm = sym->comm_fn(sym->comm_ctx, 0, 0, D->loff[tf], pm * pw, root);
      if (rcm == 0) rcm = sym->comm_fn(sym->comm_ctx, 0, 1, S.inv_off[tf], pw * pw, root);
      if (rcm == 0) rcm = sym->comm_fn(sym->comm_ctx, 0, 2, tf, 1, root);
      if (rcm != 0) {
        sym->err = "multi-GPU: the communication callback failed";
        return SCILMM_ERR_DEVICE;
      }
      if (fac->L32 && !D->keep_front[tf]) {
        // front precision 32: the arrived panel's fp32 shadow, on the communication stream behind the broadcast
        hipLaunchKernelGGL(k_shadow, dim3((unsigned)std::min<int64_t>((pm * pw + 255) / 256, 8192)), dim3(256), 0, D->comm,
                           (const double*)(fac->L + D->loff[tf]), fac->L32 + (D->loff[tf] - fac->base32), pm * pw);
        launches++;
      }
      HIPCHK(hipEventRecord(D->lev_ev[2 * l], D->comm));
      last_tail_level = l;
      if ((jj + 1) % Wg == 0 || jj == nT - 1) {
        // ---- source group grp is complete on this rank: its batch (every own target at least two groups ahead)
        hipStream_t bs = D->bstream;
        for (int32_t q = grp * Wg; q <= jj; ++q) HIPCHK(hipStreamWaitEvent(bs, D->lev_ev[2 * S.sn_level[D->dist_first + q]], 0));
        if (D->outside_on) HIPCHK(hipStreamWaitEvent(bs, D->out_evs.back(), 0));  // (a batch writes targets up to the last panel)
        if (prof) HIPCHK(hipEventRecord(D->bpev[2 * (size_t)grp], bs));
        launch_dense(bs, D->d_dwork_b + D->dbatch_ptr[grp], D->dbatch_ptr[grp + 1] - D->dbatch_ptr[grp], nullptr);
        if (prof) HIPCHK(hipEventRecord(D->bpev[2 * (size_t)grp + 1], bs));
        HIPCHK(hipEventRecord(D->batch_ev[grp], bs));
      }
    }
    if (D->outside_on && l == D->tail_level - 1) {
      // ---- every prelude front below the tail's first level is final: its contribution to the tail, in ITS
      //      coordinates, with atomic subtraction (k_outside); nothing else touches a tail panel meanwhile
      HIPCHK(hipStreamWaitEvent(D->outside_st, D->lev_ev[2 * l], 0));
      for (size_t og = 0; og + 1 < D->ochunk_ptr.size(); ++og) {
      for (int64_t o0 = D->ochunk_ptr[og]; o0 < D->ochunk_ptr[og + 1]; o0 += max_groups(256)) {
        const unsigned cnt = (unsigned)std::min<int64_t>(max_groups(256), D->ochunk_ptr[og + 1] - o0);
        const OutsideWork* ow = (const OutsideWork*)D->d_owork + o0;
#ifdef SCILMM_DIAG
        if (D->ablate == 6)  // timing ablations: no scatter / plain stores (WRONG numbers)
          hipLaunchKernelGGL((k_outside<true, 1>), dim3(cnt), dim3(256), 0, D->outside_st, D->v, S.dense_first, ow,
                             (const int32_t*)D->d_tail_front, (const uint8_t*)D->d_keep_front, (const int32_t*)D->d_grp_next,
                             (const int32_t*)D->d_grp_t0, fac->L);
        else if (D->ablate == 7)
          hipLaunchKernelGGL((k_outside<true, 2>), dim3(cnt), dim3(256), 0, D->outside_st, D->v, S.dense_first, ow,
                             (const int32_t*)D->d_tail_front, (const uint8_t*)D->d_keep_front, (const int32_t*)D->d_grp_next,
                             (const int32_t*)D->d_grp_t0, fac->L);
        else
#endif
        if (D->use_mfma)
          hipLaunchKernelGGL(k_outside<true>, dim3(cnt), dim3(256), 0, D->outside_st, D->v, S.dense_first, ow,
                             (const int32_t*)D->d_tail_front, (const uint8_t*)D->d_keep_front, (const int32_t*)D->d_grp_next,
                             (const int32_t*)D->d_grp_t0, fac->L);
        else
          hipLaunchKernelGGL(k_outside<false>, dim3(cnt), dim3(256), 0, D->outside_st, D->v, S.dense_first, ow,
                             (const int32_t*)D->d_tail_front, (const uint8_t*)D->d_keep_front, (const int32_t*)D->d_grp_next,
                             (const int32_t*)D->d_grp_t0, fac->L);
        HIPCHK(hipGetLastError());
        launches++;
      }
      HIPCHK(hipEventRecord(D->out_evs[og], D->outside_st));
      }
      for (int32_t le = D->tail_level; le <= l + D->look_depth && le < S.nlevels; ++le) {
        if (le < 1) continue;
        int rc = launch_early(le);
        if (rc != SCILMM_OK) return rc;
      }
    }
  }
  if (!sym->err.empty() && sym->err.rfind("cell plan:", 0) == 0) return SCILMM_ERR_ARG;
  if (dist) {
    if (last_tail_level >= 0) HIPCHK(hipStreamWaitEvent(st, D->lev_ev[2 * last_tail_level], 0));
    if (!D->batch_ev.empty()) HIPCHK(hipStreamWaitEvent(st, D->batch_ev.back(), 0));
    // the status word (first non-positive pivot, or "none") is written by the owner of the failing panel only: every
    // rank takes the minimum, so that all of them report SCILMM_ERR_NOT_PD together
    hipLaunchKernelGGL(k_status_pack, dim3(1), dim3(1), 0, st, (const int32_t*)fac->status, fac->logd + S.nsuper);
    HIPCHK(hipEventRecord(D->ev_x0, st));
    HIPCHK(hipStreamWaitEvent(D->comm, D->ev_x0, 0));
    if (sym->comm_fn(sym->comm_ctx, 2, 2, S.nsuper, 1, 0) != 0) {
      sym->err = "multi-GPU: the communication callback failed";
      return SCILMM_ERR_DEVICE;
    }
    HIPCHK(hipEventRecord(D->ev_x1, D->comm));
    HIPCHK(hipStreamWaitEvent(st, D->ev_x1, 0));
    hipLaunchKernelGGL(k_status_unpack, dim3(1), dim3(1), 0, st, (const double*)(fac->logd + S.nsuper), fac->status);
  }
  HIPCHK(hipEventRecord(D->ev[2], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(fac->h_status, fac->status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  D->timing.n_launches = launches;
  fac->pending = true;
  if (!wait) return SCILMM_OK;
  return finish_factorize(fac, bad_col);
}

// second half of a factorization: wait for the queued work, read the event timings and the status word
int finish_factorize(scilmm_factor* fac, int32_t* bad_col) {
  scilmm_symbolic* sym = fac->sym;
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  hipStream_t st = D->stream;
  const bool prof = D->profiling;
  constexpr int PE = 14;
  fac->pending = false;
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipStreamSynchronize(D->side));
  HIPCHK(hipStreamSynchronize(D->side2));
  if (D->side3) HIPCHK(hipStreamSynchronize(D->side3));
  if (D->bstream) HIPCHK(hipStreamSynchronize(D->bstream));
  if (D->world > 1 && D->comm) HIPCHK(hipStreamSynchronize(D->comm));
  if (D->outside_st) HIPCHK(hipStreamSynchronize(D->outside_st));
  float a = 0, f = 0;
  HIPCHK(hipEventElapsedTime(&a, D->ev[0], D->ev[1]));
  HIPCHK(hipEventElapsedTime(&f, D->ev[1], D->ev[2]));
  D->timing.assemble_ms = a;
  D->timing.factor_ms = f;
  const int32_t status = *fac->h_status;
  if (prof) {
    double tu = 0, tp = 0, tt = 0, tmid = 0, td = 0;
    int64_t nu = 0, nd = 0;
    for (int32_t l = 0; l < S.nlevels; ++l) {
      float x = 0;
      if (D->work_ptr[l + 1] > D->work_ptr[l] || D->dwork_l_ptr[l + 1] > D->dwork_l_ptr[l]) {
        HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 0], D->pev[PE * l + 1]));
        tu += x;
        nu += (D->work_ptr[l + 1] > D->work_ptr[l]) + (D->dwork_l_ptr[l + 1] > D->dwork_l_ptr[l]);
      }
      if (D->early_ptr[l + 1] > D->early_ptr[l] || D->dwork_e_ptr[l + 1] > D->dwork_e_ptr[l]) {
        HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 5], D->pev[PE * l + 6]));
        tu += x;
        nu += (D->early_ptr[l + 1] > D->early_ptr[l]) + (D->dwork_e_ptr[l + 1] > D->dwork_e_ptr[l]);
      }
      if (D->dwork_e_ptr[l + 1] > D->dwork_e_ptr[l]) {
        HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 8], D->pev[PE * l + 9]));
        td += x;
        nd++;
      }
      if (D->dwork_l_ptr[l + 1] > D->dwork_l_ptr[l]) {
        const bool split = D->world > 1 && D->dwork_l_mid[(size_t)l] < D->dwork_l_ptr[l + 1] && D->tail_of_level[l] >= 0 &&
                           D->keep_front[D->tail_of_level[l]];
        if (!split || D->dwork_l_mid[(size_t)l] > D->dwork_l_ptr[l]) {
          HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 10], D->pev[PE * l + 11]));
          td += x;
          nd++;
        }
        if (split) {  // (the newest source's items: their own bracket, so that the wait for its broadcast is not counted as kernel time)
          HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 12], D->pev[PE * l + 13]));
          td += x;
          nd++;
        }
      }
      HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 1], D->pev[PE * l + 2]));
      tmid += x;
      HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 2], D->pev[PE * l + 3]));
      tp += x;
      HIPCHK(hipEventElapsedTime(&x, D->pev[PE * l + 3], D->pev[PE * l + 4]));
      tt += x;
    }
    if (D->world > 1 && D->dist_first < S.nsuper && D->bpev.size() >= 2 * D->batch_ev.size()) {
      // distributed tail: the batches (one launch sequence per complete source group, on their own stream) carry the bulk of
      // this rank's dense-tail flops; the per-level "late" launches above the rest
      for (size_t g = 0; g < D->batch_ev.size(); ++g) {
        if (D->dbatch_ptr[g + 1] <= D->dbatch_ptr[g]) continue;
        float x = 0;
        HIPCHK(hipEventElapsedTime(&x, D->bpev[2 * g], D->bpev[2 * g + 1]));
        td += x;
        tu += x;
        nd++;
        nu++;
      }
    }
    D->timing.update_ms = tu;       // sum of k_update launch durations (early + late; they overlap other kernels)
    D->timing.potrf_ms = tp;
    D->timing.trsm_ms = tt;
    D->timing.reduce_cells_ms = tmid;
    D->timing.n_update_launches = nu;
    D->timing.dense_ms = td;          // k_dense launches alone (early + late; they overlap each other on the two side streams)
    D->timing.n_dense_launches = nd;
    {
      // union of the update launches' [start, end] intervals, measured from the end of the assembly
      std::vector<std::pair<float, float>> iv;
      for (int32_t l = 0; l < S.nlevels; ++l) {
        float a0 = 0, a1 = 0;
        if (D->work_ptr[l + 1] > D->work_ptr[l] || D->dwork_l_ptr[l + 1] > D->dwork_l_ptr[l]) {
          HIPCHK(hipEventElapsedTime(&a0, D->ev[1], D->pev[PE * l + 0]));
          HIPCHK(hipEventElapsedTime(&a1, D->ev[1], D->pev[PE * l + 1]));
          iv.push_back({a0, a1});
        }
        if (D->early_ptr[l + 1] > D->early_ptr[l] || D->dwork_e_ptr[l + 1] > D->dwork_e_ptr[l]) {
          HIPCHK(hipEventElapsedTime(&a0, D->ev[1], D->pev[PE * l + 5]));
          HIPCHK(hipEventElapsedTime(&a1, D->ev[1], D->pev[PE * l + 6]));
          iv.push_back({a0, a1});
        }
      }
      std::sort(iv.begin(), iv.end());
      double uni = 0.0;
      float cur0 = 0, cur1 = -1;
      for (auto& q : iv) {
        if (cur1 < cur0 || q.first > cur1) {
          if (cur1 >= cur0) uni += cur1 - cur0;
          cur0 = q.first;
          cur1 = q.second;
        } else {
          cur1 = std::max(cur1, q.second);
        }
      }
      if (cur1 >= cur0) uni += cur1 - cur0;
      D->timing.update_union_ms = uni;
    }
    if (const char* dump = getenv("SCILMM_LEVEL_DUMP")) {
      // diagnostic: one line per level (durations in ms; cost = combos + K-chunks of the level's dense work)
      if (FILE* fp = fopen(dump, "w")) {
        fprintf(fp, "level,fronts,tiles,items_early,items_late,cost_early,cost_late,early_ms,late_ms,mid_ms,potrf_ms,trsm_ms,t_potrf_ms,t_early0_ms,t_early1_ms,dense_early_ms,dense_late_ms\n");
        for (int32_t l = 0; l < S.nlevels; ++l) {
          float xe = 0, xl = 0, xm = 0, xp = 0, xt = 0;
          if (D->work_ptr[l + 1] > D->work_ptr[l]) HIPCHK(hipEventElapsedTime(&xl, D->pev[PE * l + 0], D->pev[PE * l + 1]));
          if (D->early_ptr[l + 1] > D->early_ptr[l]) HIPCHK(hipEventElapsedTime(&xe, D->pev[PE * l + 5], D->pev[PE * l + 6]));
          HIPCHK(hipEventElapsedTime(&xm, D->pev[PE * l + 1], D->pev[PE * l + 2]));
          HIPCHK(hipEventElapsedTime(&xp, D->pev[PE * l + 2], D->pev[PE * l + 3]));
          HIPCHK(hipEventElapsedTime(&xt, D->pev[PE * l + 3], D->pev[PE * l + 4]));
          // absolute times since the start of the factorization: this level's potrf, its early dense launch (begin, end)
          float tp = 0, te0 = 0, te1 = 0, de = 0, dl = 0;
          HIPCHK(hipEventElapsedTime(&tp, D->ev[1], D->pev[PE * l + 2]));
          if (D->dwork_e_ptr[l + 1] > D->dwork_e_ptr[l]) {
            HIPCHK(hipEventElapsedTime(&te0, D->ev[1], D->pev[PE * l + 8]));
            HIPCHK(hipEventElapsedTime(&te1, D->ev[1], D->pev[PE * l + 9]));
            de = te1 - te0;
          }
          if (D->dwork_l_ptr[l + 1] > D->dwork_l_ptr[l]) HIPCHK(hipEventElapsedTime(&dl, D->pev[PE * l + 10], D->pev[PE * l + 11]));
          fprintf(fp, "%d,%d,%lld,%lld,%lld,%lld,%lld,%.4f,%.4f,%.4f,%.4f,%.4f,%.3f,%.3f,%.3f,%.4f,%.4f\n", l, S.level_ptr[l + 1] - S.level_ptr[l],
                  (long long)(S.level_tile_ptr[l + 1] - S.level_tile_ptr[l]), (long long)(D->early_ptr[l + 1] - D->early_ptr[l]),
                  (long long)(D->work_ptr[l + 1] - D->work_ptr[l]), (long long)D->lev_cost_e[l], (long long)D->lev_cost_l[l], xe, xl,
                  xm, xp, xt, tp, te0, te1, de, dl);
        }
        fclose(fp);
      }
    }
  }
  if (status != 0x7fffffff) {
    if (bad_col) *bad_col = status;
    return SCILMM_ERR_NOT_PD;
  }
  fac->valid = true;
  return SCILMM_OK;
}

// dB/dX: device, row-major n x r, ORIGINAL row order.  mode 0: X = V^-1 B.  mode 1: X = P^T L B.
int run_rhs(scilmm_factor* fac, const double* dB, int32_t r, double* dX, int mode) {
  scilmm_symbolic* sym = fac->sym;
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  if (fac->pending) {
    int stp = finish_factorize(fac, nullptr);
    if (stp != SCILMM_OK) return stp;
  }
  if (!fac->valid) {
    sym->err = "factor is not valid";
    return SCILMM_ERR_STATE;
  }
  int stc = ensure_work(sym, D);
  if (stc != SCILMM_OK) return stc;
  if (D->h_chain_err && *D->h_chain_err != 0) {
    // an earlier (already completed) chain sweep timed out: report it before queueing more work on top of it
    HIPCHK(hipStreamSynchronize(D->stream));
    HIPCHK(hipMemset(D->d_chain_err, 0, sizeof(int32_t)));
    *D->h_chain_err = 0;
    sym->err = "chain sweep: a workgroup timed out waiting for its predecessor (previous solve)";
    return SCILMM_ERR_DEVICE;
  }
  hipStream_t st = D->stream;
  const int64_t ntiles_all = (int64_t)S.level_tiles.size();
  const bool mf = D->use_mfma;
  HIPCHK(hipEventRecord(D->ev[3], st));
  bool mid_recorded = false;
  for (int32_t cbeg = 0; cbeg < r; cbeg += RPMAX) {
    const int rc = std::min<int>(RPMAX, r - cbeg);
    const int rp = rp_of(rc);
    const unsigned gy = (unsigned)((rp + CW - 1) / CW);
    // the chain sweeps pick their own window width: 64 columns for long chains (every window streams the whole dense
    // tail once), 32 for short ones (twice the workgroups on the latency-bound chain)
    // (read per call: the parity tests force each width on small chains)
    const char* ecw = tune_env("SCILMM_CHAIN_WIDE_T");
    const char* ecf = tune_env("SCILMM_CHAIN_FULL_T");
    const int chain_wide_T = ecw ? atoi(ecw) : 256, chain_full_T = ecf ? atoi(ecf) : 768;
    auto launch_chain = [&](bool bwd) -> int {
      const bool wide = D->chain_T >= chain_wide_T;
      const bool full = D->chain_T >= chain_full_T && rp > 64;  // every column in one 112-wide window
      const int32_t cwc = full ? 112 : wide ? 64 : 32;
      const int32_t gyc = (int32_t)((rp + cwc - 1) / cwc);
      const unsigned grid = (unsigned)D->chain_T * (unsigned)gyc;
      const int32_t ep = ++D->chain_epoch;
      HIPCHK(hipMemsetAsync(D->d_chain_err + 2, 0, sizeof(int32_t), st));
      const int32_t* cptr = bwd ? (const int32_t*)D->d_cb_ptr : (const int32_t*)D->d_cf_ptr;
      const ChainPair* cpairs = bwd ? (const ChainPair*)D->d_cb : (const ChainPair*)D->d_cf;
#define SCILMM_CHAIN_LAUNCH(MF, BW, NC)                                                                                                  \
  hipLaunchKernelGGL((k_chain<MF, BW, NC>), dim3(grid), dim3(512), 0, st, D->v, D->chain_T, (const int32_t*)D->d_chain, cptr, cpairs, \
                     (const int32_t*)D->d_colmap, (const double*)fac->L, (const double*)fac->invD, (const double*)D->W, D->X, rp, gyc,  \
                     D->d_chain_flags, ep, D->d_chain_err, D->d_chain_err + 2)
      if (mf) {
        if (bwd) { if (full) SCILMM_CHAIN_LAUNCH(true, true, 7); else if (wide) SCILMM_CHAIN_LAUNCH(true, true, 4); else SCILMM_CHAIN_LAUNCH(true, true, 2); }
        else { if (full) SCILMM_CHAIN_LAUNCH(true, false, 7); else if (wide) SCILMM_CHAIN_LAUNCH(true, false, 4); else SCILMM_CHAIN_LAUNCH(true, false, 2); }
      } else {
        if (bwd) SCILMM_CHAIN_LAUNCH(false, true, 2); else SCILMM_CHAIN_LAUNCH(false, false, 2);
      }
#undef SCILMM_CHAIN_LAUNCH
      return SCILMM_OK;
    };
    const size_t sm_fwd = sizeof(double) * (size_t)(NB * LDW + KCS * LDA);
    const int64_t tot = (int64_t)S.n * rp;
    const unsigned pb = (unsigned)((tot + 255) / 256);
    if (D->world > 1) {
      // ---- distributed factor.  The prelude is replicated: every rank sweeps it alike.  A tail panel lives on its owner:
      //   forward : the owner pushes x_f through its panel into ACC (its private sum of tail contributions); when block f
      //             is due, the ranks ALL-REDUCE the 128 rows of ACC that belong to it, add them to W (which carries the
      //             right-hand side and the prelude's contributions, identical everywhere) and every rank solves the block
      //             with the replicated inverse diagonal block: x_f is known everywhere without a broadcast;
      //   backward: the owner of panel f has received every push into X[f] (a push (target t, descendant f) needs the
      //             panel of f): it solves the block and BROADCASTS x_f; then every rank pushes x_f into the descendants
      //             it holds (the prelude: all ranks; tail panels: their owners).
      //   L * R   : every panel is multiplied where it lives (the prelude on rank 0), one all-reduce of the product.
      // One collective per tail block and direction, issued on the communication stream between two event hand-offs.
      const int64_t nW = (int64_t)S.n * RPMAX;  // W | X | ACC inside the caller's work buffer (scilmm_dist_set_work)
      auto handoff = [&](int32_t op, int64_t off, int64_t cnt, int32_t root) -> int {
        HIPCHK(hipEventRecord(D->ev_x0, st));
        HIPCHK(hipStreamWaitEvent(D->comm, D->ev_x0, 0));
        if (sym->comm_fn(sym->comm_ctx, op, 3, off, cnt, root) != 0) {
          sym->err = "multi-GPU: the communication callback failed";
          return SCILMM_ERR_DEVICE;
        }
        HIPCHK(hipEventRecord(D->ev_x1, D->comm));
        HIPCHK(hipStreamWaitEvent(st, D->ev_x1, 0));
        return SCILMM_OK;
      };
      int rcx;
      if (mode == 1) {
        hipLaunchKernelGGL(k_perm_in, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, (const int32_t*)nullptr, dB, D->W);
        HIPCHK(hipMemsetAsync(D->X, 0, sizeof(double) * (size_t)tot, st));
        if (D->n_lmul_tiles > 0) {
          if (mf)
            hipLaunchKernelGGL((k_fwd<true, 1, true>), dim3((unsigned)D->n_lmul_tiles, gy), dim3(256), sm_fwd, st, D->v, D->d_lmul_tiles,
                               fac->L, (const double*)D->W, D->X, rp);
          else
            hipLaunchKernelGGL((k_fwd<false, 1, true>), dim3((unsigned)D->n_lmul_tiles, gy), dim3(256), sm_fwd, st, D->v, D->d_lmul_tiles,
                               fac->L, (const double*)D->W, D->X, rp);
        }
        if ((rcx = handoff(1, nW, tot, 0)) != SCILMM_OK) return rcx;
        hipLaunchKernelGGL(k_perm_out, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, D->v.perm, D->X, dX);
        continue;
      }
      hipLaunchKernelGGL(k_perm_in, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, D->v.perm, dB, D->W);
      HIPCHK(hipMemsetAsync(D->ACC, 0, sizeof(double) * (size_t)tot, st));
      for (int32_t l = 0; l < S.nlevels; ++l) {
        const int32_t tf = D->tail_of_level[l];
        const int64_t t0 = D->lv_tile_ptr[l], tm = D->lv_tile_mid[l], t1 = D->lv_tile_ptr[l + 1];
        const int32_t f0 = S.level_ptr[l], f1 = S.level_ptr[l + 1];  // ALL fronts of the level (replicated diagonal solves)
        if (f1 == f0) continue;
        if (tf >= 0) {
          const int64_t c0 = S.sn_start[tf], wf = S.sn_start[tf + 1] - c0;
          if ((rcx = handoff(1, 2 * nW + c0 * rp, wf * rp, 0)) != SCILMM_OK) return rcx;
          hipLaunchKernelGGL(k_add_rows, dim3((unsigned)((wf * rp + 255) / 256)), dim3(256), 0, st, wf * rp, (const double*)(D->ACC + c0 * rp),
                             D->W + c0 * rp);
        }
        if (mf)
          hipLaunchKernelGGL((k_diag_solve<true, false>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v, D->d_all_fronts + f0,
                             fac->invD, (const double*)D->W, D->X, rp);
        else
          hipLaunchKernelGGL((k_diag_solve<false, false>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v, D->d_all_fronts + f0,
                             fac->invD, (const double*)D->W, D->X, rp);
        // pushes of the prelude fronts of the level go to W (atomic: they may share rows), of an own tail panel to ACC
        if (tm > t0) {
          if (mf)
            hipLaunchKernelGGL((k_fwd<true, 0, true>), dim3((unsigned)(tm - t0), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + t0,
                               fac->L, (const double*)D->X, D->W, rp);
          else
            hipLaunchKernelGGL((k_fwd<false, 0, true>), dim3((unsigned)(tm - t0), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + t0,
                               fac->L, (const double*)D->X, D->W, rp);
        }
        if (t1 > tm) {
          if (mf)
            hipLaunchKernelGGL((k_fwd<true, 0, false>), dim3((unsigned)(t1 - tm), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + tm,
                               fac->L, (const double*)D->X, D->ACC, rp);
          else
            hipLaunchKernelGGL((k_fwd<false, 0, false>), dim3((unsigned)(t1 - tm), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + tm,
                               fac->L, (const double*)D->X, D->ACC, rp);
        }
      }
      if (!mid_recorded) {
        HIPCHK(hipEventRecord(D->ev[4], st));
        mid_recorded = true;
      }
      for (int32_t l = S.nlevels - 1; l >= 0; --l) {
        const int32_t tf = D->tail_of_level[l];
        const int32_t f0 = D->lv_ptr[l], f1 = D->lv_ptr[l + 1];  // the fronts this rank holds
        const int64_t p0 = D->lv_pair_ptr[l], p1 = D->lv_pair_ptr[l + 1];
        if (f1 > f0) {
          if (mf)
            hipLaunchKernelGGL((k_diag_solve<true, true>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v, D->d_level_fronts + f0,
                               fac->invD, (const double*)D->X, D->X, rp);
          else
            hipLaunchKernelGGL((k_diag_solve<false, true>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v, D->d_level_fronts + f0,
                               fac->invD, (const double*)D->X, D->X, rp);
        }
        if (tf >= 0) {
          const int64_t c0 = S.sn_start[tf], wf = S.sn_start[tf + 1] - c0;
          if ((rcx = handoff(0, nW + c0 * rp, wf * rp, (tf - D->dist_first) % D->world)) != SCILMM_OK) return rcx;
        }
        if (p1 > p0) {
          if (mf)
            hipLaunchKernelGGL(k_bwd_push<true>, dim3((unsigned)(p1 - p0), gy), dim3(256), 0, st, D->v, D->d_level_pairs + p0,
                               (const int64_t*)nullptr, fac->L, D->X, rp, (const int32_t*)nullptr, (double*)nullptr);
          else
            hipLaunchKernelGGL(k_bwd_push<false>, dim3((unsigned)(p1 - p0), gy), dim3(256), 0, st, D->v, D->d_level_pairs + p0,
                               (const int64_t*)nullptr, fac->L, D->X, rp, (const int32_t*)nullptr, (double*)nullptr);
        }
      }
      hipLaunchKernelGGL(k_perm_out, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, D->v.perm, D->X, dX);
      continue;
    }
    if (mode == 0) {
      hipLaunchKernelGGL(k_perm_in, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, D->v.perm, dB, D->W);
      const int32_t lend = D->chain_T > 0 ? D->chain_l0 : S.nlevels;  // the chain levels are swept by k_chain
      for (int32_t l = 0; l < lend; ++l) {
        const int64_t t0 = S.level_tile_ptr[l], t1 = S.level_tile_ptr[l + 1];
        const int32_t f0 = S.level_ptr[l], f1 = S.level_ptr[l + 1];
        if (f1 == f0) continue;
        // x_s = invL_s * W[c0:c1] -> X rows c0..c1 (final), then push to the rows below
        if (mf)
          hipLaunchKernelGGL((k_diag_solve<true, false>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v,
                             D->d_level_fronts + f0, fac->invD, (const double*)D->W, D->X, rp);
        else
          hipLaunchKernelGGL((k_diag_solve<false, false>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v,
                             D->d_level_fronts + f0, fac->invD, (const double*)D->W, D->X, rp);
        if (t1 == t0) continue;
        const bool atomic = (f1 - f0) > 1;
        if (mf && atomic)
          hipLaunchKernelGGL((k_fwd<true, 0, true>), dim3((unsigned)(t1 - t0), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + t0,
                             fac->L, (const double*)D->X, D->W, rp);
        else if (mf)
          hipLaunchKernelGGL((k_fwd<true, 0, false>), dim3((unsigned)(t1 - t0), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + t0,
                             fac->L, (const double*)D->X, D->W, rp);
        else if (atomic)
          hipLaunchKernelGGL((k_fwd<false, 0, true>), dim3((unsigned)(t1 - t0), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + t0,
                             fac->L, (const double*)D->X, D->W, rp);
        else
          hipLaunchKernelGGL((k_fwd<false, 0, false>), dim3((unsigned)(t1 - t0), gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles + t0,
                             fac->L, (const double*)D->X, D->W, rp);
      }
      if (D->chain_T > 0) {
        if (int e = launch_chain(false)) return e;
      }
      if (!mid_recorded) {
        HIPCHK(hipEventRecord(D->ev[4], st));
        mid_recorded = true;
      }
      if (D->chain_T > 0) {
        if (int e = launch_chain(true)) return e;
        // descendants below the chain: all their chain targets are final now, one read-modify-write each
        if (D->chain_groups > 0) {
          if (mf)
            hipLaunchKernelGGL(k_bwd_push<true>, dim3((unsigned)D->chain_groups, gy), dim3(256), 0, st, D->v,
                               (const int32_t*)D->d_cg_pairs, (const int64_t*)D->d_cg_ptr, fac->L, D->X, rp,
                               (const int32_t*)D->d_cg_slot, D->d_push_partial);
          else
            hipLaunchKernelGGL(k_bwd_push<false>, dim3((unsigned)D->chain_groups, gy), dim3(256), 0, st, D->v,
                               (const int32_t*)D->d_cg_pairs, (const int64_t*)D->d_cg_ptr, fac->L, D->X, rp,
                               (const int32_t*)D->d_cg_slot, D->d_push_partial);
          if (D->n_fold > 0)
            hipLaunchKernelGGL(k_push_fold, dim3((unsigned)D->n_fold, gy), dim3(256), 0, st, D->v, (const int32_t*)D->d_fold,
                               (const double*)D->d_push_partial, D->X, rp);
        }
      }
      for (int32_t l = lend - 1; l >= 0; --l) {
        const int32_t f0 = S.level_ptr[l], f1 = S.level_ptr[l + 1];
        const int64_t p0 = S.level_pair_ptr[l], p1 = S.level_pair_ptr[l + 1];
        if (f1 > f0) {
          if (mf)
            hipLaunchKernelGGL((k_diag_solve<true, true>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v,
                               D->d_level_fronts + f0, fac->invD, (const double*)D->X, D->X, rp);
          else
            hipLaunchKernelGGL((k_diag_solve<false, true>), dim3((unsigned)(f1 - f0), gy), dim3(256), 0, st, D->v,
                               D->d_level_fronts + f0, fac->invD, (const double*)D->X, D->X, rp);
        }
        if (p1 > p0) {
          if (mf)
            hipLaunchKernelGGL(k_bwd_push<true>, dim3((unsigned)(p1 - p0), gy), dim3(256), 0, st, D->v, D->d_level_pairs + p0,
                               (const int64_t*)nullptr, fac->L, D->X, rp, (const int32_t*)nullptr, (double*)nullptr);
          else
            hipLaunchKernelGGL(k_bwd_push<false>, dim3((unsigned)(p1 - p0), gy), dim3(256), 0, st, D->v, D->d_level_pairs + p0,
                               (const int64_t*)nullptr, fac->L, D->X, rp, (const int32_t*)nullptr, (double*)nullptr);
        }
      }
      hipLaunchKernelGGL(k_perm_out, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, D->v.perm, D->X, dX);
    } else {
      // Z = P^T (L R): R is NOT permuted on the way in (SparseCholesky.py:50-51)
      hipLaunchKernelGGL(k_perm_in, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, (const int32_t*)nullptr, dB, D->W);
      HIPCHK(hipMemsetAsync(D->X, 0, sizeof(double) * (size_t)tot, st));
      if (ntiles_all > 0) {
        if (mf)
          hipLaunchKernelGGL((k_fwd<true, 1, true>), dim3((unsigned)ntiles_all, gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles, fac->L,
                             (const double*)D->W, D->X, rp);
        else
          hipLaunchKernelGGL((k_fwd<false, 1, true>), dim3((unsigned)ntiles_all, gy), dim3(256), sm_fwd, st, D->v, D->d_level_tiles, fac->L,
                             (const double*)D->W, D->X, rp);
      }
      hipLaunchKernelGGL(k_perm_out, dim3(pb), dim3(256), 0, st, S.n, r, rp, cbeg, D->v.perm, D->X, dX);
    }
  }
  if (!mid_recorded) HIPCHK(hipEventRecord(D->ev[4], st));
  if (D->h_chain_err && mode == 0) HIPCHK(hipMemcpyAsync(D->h_chain_err, D->d_chain_err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(D->ev[5], st));
  HIPCHK(hipGetLastError());
  D->rhs_pending = mode;
  return SCILMM_OK;
}

int finish_rhs_timing(scilmm_symbolic* sym, Dev* D, int mode) {
  D->rhs_pending = -1;
  if (D->chain_T > 0 && mode == 0) {
    int32_t cerr = 0;
    HIPCHK(hipMemcpy(&cerr, D->d_chain_err, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (cerr != 0) {
      HIPCHK(hipMemset(D->d_chain_err, 0, sizeof(int32_t)));
      if (D->h_chain_err) *D->h_chain_err = 0;
      sym->err = "chain sweep: a workgroup timed out waiting for its predecessor";
      return SCILMM_ERR_DEVICE;
    }
  }
  float a = 0, b = 0;
  HIPCHK(hipEventElapsedTime(&a, D->ev[3], D->ev[4]));
  HIPCHK(hipEventElapsedTime(&b, D->ev[4], D->ev[5]));
  if (mode == 0) {
    D->timing.solve_fwd_ms = a;
    D->timing.solve_bwd_ms = b;
  } else {
    D->timing.lmul_ms = a + b;
  }
  return SCILMM_OK;
}

int run_quad(scilmm_symbolic* sym, Dev* D, int32_t k, const double* dU, int32_t r, double* d_out) {
  const Symbolic& S = *sym->S;
  if (k < 0 || k >= S.K || !D->have_vals[k]) {
    sym->err = "quadforms: matrix index invalid or values not uploaded";
    return SCILMM_ERR_STATE;
  }
  int stc = ensure_work(sym, D);
  if (stc != SCILMM_OK) return stc;
  hipStream_t st = D->stream;
  const int64_t SPW = 512;
  const int64_t nw_gen = ((S.nnz_pattern + SPW - 1) / SPW + 3) / 4 * 4;
  const int64_t nw_dia = 1024;
  const int64_t nw = std::max(nw_gen, nw_dia);
  if (D->nwaves_quad < nw) {
    if (D->partial) (void)hipFree(D->partial);
    D->partial = nullptr;
    HIPCHK(hipMalloc((void**)&D->partial, sizeof(double) * (size_t)nw * RPMAX));
    D->nwaves_quad = nw;
  }
  HIPCHK(hipEventRecord(D->ev[6], st));
  for (int32_t cbeg = 0; cbeg < r; cbeg += RPMAX) {
    const int rc = std::min<int>(RPMAX, r - cbeg);
    const int rp = rp_of(rc);
    const int64_t tot = (int64_t)S.n * rp;
    hipLaunchKernelGGL(k_perm_in, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, S.n, r, rp, cbeg, D->v.perm, dU, D->W);
    int64_t nwaves;
    if (S.is_diag[k]) {
      nwaves = nw_dia;
      hipLaunchKernelGGL(k_quad_diag, dim3((unsigned)(nwaves / 4)), dim3(256), 0, st, S.n, (const double*)D->vals[k],
                         (const double*)D->W, rp, D->partial);
    } else {
      nwaves = nw_gen;
      hipLaunchKernelGGL(k_quad, dim3((unsigned)(nwaves / 4)), dim3(256), 0, st, D->v, S.nnz_pattern, SPW,
                         (const double*)D->vals[k], (const double*)D->W, rp, D->partial);
    }
    hipLaunchKernelGGL(k_quad_reduce, dim3(1), dim3(RPMAX), 0, st, nwaves, (const double*)D->partial, rp, D->d_out, 1.0, 0);
    HIPCHK(hipMemcpyAsync(d_out + cbeg, D->d_out, sizeof(double) * rc, hipMemcpyDeviceToDevice, st));
  }
  HIPCHK(hipEventRecord(D->ev[7], st));
  HIPCHK(hipGetLastError());
  return SCILMM_OK;
}

}  // namespace

extern "C" {

int scilmm_values_upload(scilmm_symbolic* sym, int32_t k, const double* data_k) {
  if (!sym || !sym->S || !data_k) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  const Symbolic& S = *sym->S;
  if (k < 0 || k >= S.K) return SCILMM_ERR_ARG;
  if (sym->maps_released) {
    sym->err = "scilmm_values_upload: the value-assembly maps of this handle were released";
    return SCILMM_ERR_STATE;
  }
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  // general matrix: values permuted into pattern-slot order; diagonal-only matrix: one value per
  // permuted row.  Either way h[val_slot] = data[val_src].
  std::vector<double> h(S.is_diag[k] ? (size_t)S.n : (size_t)S.nnz_pattern, 0.0);
  {
    const auto& slot = S.val_slot[k];
    const auto& src = S.val_src[k];
    // every pattern slot is written by exactly one entry: the permutation is split over a few host threads
    const size_t cnt = slot.size();
    const unsigned nth = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)std::min(16, scilmm::host_threads()), cnt / (1 << 20) + 1));
    auto part = [&](unsigned q) {
      const size_t a = cnt * q / nth, b = cnt * (q + 1) / nth;
      for (size_t t = a; t < b; ++t) h[slot[t]] = data_k[src[t]];
    };
    std::vector<std::thread> pool;
    for (unsigned q = 1; q < nth; ++q) pool.emplace_back(part, q);
    part(0);
    for (auto& th : pool) th.join();
  }
  if (!D->vals[k]) HIPCHK(hipMalloc((void**)&D->vals[k], std::max<size_t>(h.size(), 1) * sizeof(double)));
  if (!h.empty()) HIPCHK(hipMemcpy(D->vals[k], h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  D->have_vals[k] = 1;
  return SCILMM_OK;
}

// slack behind L and invD: the chain sweeps read whole 4-deep k-steps of a panel / an inverse block and discard the
// lanes past its last column (at most 3 columns of the tallest panel resp. of an NB-wide block)
static void factor_sizes(const scilmm_symbolic* sym, size_t* nL, size_t* padL, size_t* nI, size_t* padI) {
  const Symbolic& S = *sym->S;
  int64_t max_m = 0;
  for (int32_t q = 0; q < S.nsuper; ++q) max_m = std::max<int64_t>(max_m, S.sn_rowptr[q + 1] - S.sn_rowptr[q]);
  *padL = (size_t)(4 * max_m + 4 * NB);
  *padI = (size_t)(5 * NB);
  *nL = (size_t)std::max<int64_t>(S.nnzL_stored, 1);
  if (sym->world > 1) {  // rank-local storage: prelude + own tail panels + ring (DistLayout)
    DistLayout lay;
    dist_layout(S, sym->rank, sym->world, &lay);
    *nL = (size_t)std::max<int64_t>(lay.nL, 1);
  }
  *nI = (size_t)std::max<int64_t>(S.inv_off[S.nsuper], 1);
}

int scilmm_factor_sizes(const scilmm_symbolic* sym, int64_t* L_doubles, int64_t* invD_doubles, int64_t* logd_doubles) {
  if (!sym || !sym->S) return SCILMM_ERR_ARG;
  size_t nL, padL, nI, padI;
  factor_sizes(sym, &nL, &padL, &nI, &padI);
  if (L_doubles) *L_doubles = (int64_t)(nL + padL);
  if (invD_doubles) *invD_doubles = (int64_t)(nI + padI);
  if (logd_doubles) *logd_doubles = (int64_t)sym->S->nsuper + 1;  // + one slot for the status word's trip through the comm layer
  return SCILMM_OK;
}

int scilmm_dist_layout(const scilmm_symbolic* sym, int32_t rank, int32_t world, int32_t* owner, int64_t* loff, int32_t* params) {
  if (!sym || !sym->S || world < 1 || rank < 0 || rank >= world) return SCILMM_ERR_ARG;
  const Symbolic& S = *sym->S;
  DistLayout lay;
  dist_layout(S, rank, world, &lay);
  if (owner)
    for (int32_t f = 0; f < S.nsuper; ++f) owner[f] = (world > 1 && f >= lay.first) ? (f - lay.first) % world : -1;
  if (loff) {
    std::memcpy(loff, lay.loff.data(), sizeof(int64_t) * (size_t)S.nsuper);
    loff[S.nsuper] = lay.nL;
  }
  if (params) {
    params[0] = lay.first;
    params[1] = lay.Wg;
    params[2] = lay.G;
  }
  return SCILMM_OK;
}

int scilmm_dist_work_size(const scilmm_symbolic* sym, int64_t* doubles) {
  if (!sym || !sym->S || !doubles) return SCILMM_ERR_ARG;
  *doubles = (int64_t)3 * std::max(sym->S->n, 1) * RPMAX;
  return SCILMM_OK;
}

int scilmm_dist_set_work(scilmm_symbolic* sym, double* work) {
  if (!sym || !sym->S || !work) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  if ((D->W || D->X) && !D->work_external) {
    sym->err = "scilmm_dist_set_work must precede the first solve on this handle";
    return SCILMM_ERR_STATE;
  }
  const size_t nW = (size_t)std::max(sym->S->n, 1) * RPMAX;
  D->W = work;
  D->X = work + nW;
  D->ACC = work + 2 * nW;
  D->work_external = true;
  return SCILMM_OK;
}

static int factor_create(scilmm_symbolic* sym, double* L_ext, double* invD_ext, double* logd_ext, scilmm_factor** out) {
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  const Symbolic& S = *sym->S;
  scilmm_factor* f = new scilmm_factor();
  f->sym = sym;
  f->device = D->device;
  *out = f;
  size_t nL, padL, nI, padI;
  factor_sizes(sym, &nL, &padL, &nI, &padI);
  if (L_ext) {
    f->external = true;
    f->L = L_ext;
    f->invD = invD_ext;
    f->logd = logd_ext;
  } else {
    HIPCHK(hipMalloc((void**)&f->L, sizeof(double) * (nL + padL)));
    HIPCHK(hipMalloc((void**)&f->invD, sizeof(double) * (nI + padI)));
    HIPCHK(hipMalloc((void**)&f->logd, sizeof(double) * ((size_t)S.nsuper + 1)));
  }
  HIPCHK(hipMemset(f->L + nL, 0, sizeof(double) * padL));
  // (all of invD: k_potrf only ever writes the lower triangles, the upper ones must read as zero)
  HIPCHK(hipMemset(f->invD, 0, sizeof(double) * (nI + padI)));
  HIPCHK(hipMalloc((void**)&f->status, sizeof(int32_t)));
  HIPCHK(hipHostMalloc((void**)&f->h_status, sizeof(int32_t), hipHostMallocDefault));
  return SCILMM_OK;
}

int scilmm_factorize(scilmm_symbolic* sym, const double* sigma2, scilmm_factor** out, int32_t* bad_col) {
  if (!sym || !sym->S || !sigma2 || !out) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  int st = factor_create(sym, nullptr, nullptr, nullptr, out);
  if (st != SCILMM_OK) return st;
  return run_factorize(*out, sigma2, bad_col);
}

int scilmm_factor_create_external(scilmm_symbolic* sym, double* L, double* invD, double* logd, scilmm_factor** out) {
  if (!sym || !sym->S || !L || !invD || !logd || !out) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  return factor_create(sym, L, invD, logd, out);
}

int scilmm_dist_init(scilmm_symbolic* sym, int32_t rank, int32_t world, void* comm_stream, scilmm_comm_fn fn, void* ctx) {
  if (!sym || !sym->S || world < 1 || rank < 0 || rank >= world) return SCILMM_ERR_ARG;
  if (sym->device) {
    sym->err = "scilmm_dist_init must precede the first numeric call on this handle (the device plan depends on it)";
    return SCILMM_ERR_STATE;
  }
  if (world > 1 && (!fn || !comm_stream)) return SCILMM_ERR_ARG;
  sym->rank = rank;
  sym->world = world;
  sym->comm_stream = comm_stream;
  sym->comm_fn = fn;
  sym->comm_ctx = ctx;
  return SCILMM_OK;
}

int scilmm_refactorize(scilmm_factor* fac, const double* sigma2, int32_t* bad_col) {
  if (!fac || !fac->sym || !sigma2) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  return run_factorize(fac, sigma2, bad_col);
}

int scilmm_refactorize_async(scilmm_factor* fac, const double* sigma2) {
  if (!fac || !fac->sym || !sigma2) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  return run_factorize(fac, sigma2, nullptr, false);
}

int scilmm_factor_wait(scilmm_factor* fac, int32_t* bad_col) {
  if (!fac || !fac->sym) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  if (!fac->pending) return fac->valid ? SCILMM_OK : SCILMM_ERR_STATE;
  return finish_factorize(fac, bad_col);
}

void scilmm_factor_free(scilmm_factor* fac) {
  if (!fac) return;
  DevGuard guard(fac->device);
  if (fac->pending) (void)finish_factorize(fac, nullptr);
  if (fac->h_status) (void)hipHostFree(fac->h_status);
  if (!fac->external) {
    if (fac->L) (void)hipFree(fac->L);
    if (fac->invD) (void)hipFree(fac->invD);
    if (fac->logd) (void)hipFree(fac->logd);
  }
  if (fac->status) (void)hipFree(fac->status);
  if (fac->L32) (void)hipFree(fac->L32);
  delete fac;
}

int scilmm_logdet(scilmm_factor* fac, double* out) {
  if (!fac || !fac->sym || !out) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  scilmm_symbolic* sym = fac->sym;
  if (fac->pending) {
    int stp = finish_factorize(fac, nullptr);
    if (stp != SCILMM_OK) return stp;
  }
  if (!fac->valid) return SCILMM_ERR_STATE;
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  std::vector<double> h(S.nsuper);
  HIPCHK(hipMemcpyAsync(h.data(), fac->logd, sizeof(double) * S.nsuper, hipMemcpyDeviceToHost, D->stream));
  HIPCHK(hipStreamSynchronize(D->stream));
  // fixed-order pairwise-free summation: deterministic
  long double s = 0.0L;
  for (double v : h) s += v;
  *out = (double)(2.0L * s);
  return SCILMM_OK;
}

static int host_rhs(scilmm_factor* fac, const double* B, int32_t r, double* X, int mode) {
  if (!fac || !fac->sym || !B || !X || r <= 0) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  scilmm_symbolic* sym = fac->sym;
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  const size_t cnt = (size_t)S.n * (size_t)r;
  int st = ensure_io(sym, D, 2 * cnt);
  if (st != SCILMM_OK) return st;
  double* dB = D->IO;
  double* dX = D->IO + cnt;
  HIPCHK(hipMemcpyAsync(dB, B, cnt * sizeof(double), hipMemcpyHostToDevice, D->stream));
  st = run_rhs(fac, dB, r, dX, mode);
  if (st != SCILMM_OK) return st;
  HIPCHK(hipMemcpyAsync(X, dX, cnt * sizeof(double), hipMemcpyDeviceToHost, D->stream));
  HIPCHK(hipStreamSynchronize(D->stream));
  return finish_rhs_timing(sym, D, mode);
}

int scilmm_solve(scilmm_factor* fac, const double* B, int32_t r, double* X) { return host_rhs(fac, B, r, X, 0); }
int scilmm_lmul(scilmm_factor* fac, const double* R, int32_t r, double* Z) { return host_rhs(fac, R, r, Z, 1); }

int scilmm_solve_dev(scilmm_factor* fac, const double* dB, int32_t r, double* dX) {
  if (!fac || !fac->sym || !dB || !dX || r <= 0) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  return run_rhs(fac, dB, r, dX, 0);
}
int scilmm_lmul_dev(scilmm_factor* fac, const double* dR, int32_t r, double* dZ) {
  if (!fac || !fac->sym || !dR || !dZ || r <= 0) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  return run_rhs(fac, dR, r, dZ, 1);
}

int scilmm_quadforms_dev(scilmm_symbolic* sym, int32_t k, const double* dU, int32_t r, double* d_out) {
  if (!sym || !sym->S || !dU || !d_out || r <= 0) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  st = run_quad(sym, D, k, dU, r, d_out);
  D->quad_pending = st == SCILMM_OK;  // (asynchronous: scilmm_last_timing reads the events once they have happened)
  return st;
}

int scilmm_quadforms(scilmm_symbolic* sym, int32_t k, const double* U, int32_t r, double* out) {
  if (!sym || !sym->S || !U || !out || r <= 0) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  const Symbolic& S = *sym->S;
  const size_t cnt = (size_t)S.n * (size_t)r;
  st = ensure_io(sym, D, cnt + (size_t)r);
  if (st != SCILMM_OK) return st;
  HIPCHK(hipMemcpyAsync(D->IO, U, cnt * sizeof(double), hipMemcpyHostToDevice, D->stream));
  st = run_quad(sym, D, k, D->IO, r, D->IO + cnt);
  if (st != SCILMM_OK) return st;
  HIPCHK(hipMemcpyAsync(out, D->IO + cnt, sizeof(double) * r, hipMemcpyDeviceToHost, D->stream));
  HIPCHK(hipStreamSynchronize(D->stream));
  float q = 0;
  HIPCHK(hipEventElapsedTime(&q, D->ev[6], D->ev[7]));
  D->timing.quad_ms = q;
  return SCILMM_OK;
}

int scilmm_he_moments(scilmm_symbolic* sym, int32_t k1, int32_t k2, double* frob, double* diag_dot) {
  if (!sym || !sym->S || !frob || !diag_dot) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  const Symbolic& S = *sym->S;
  if (k1 < 0 || k1 >= S.K || k2 < 0 || k2 >= S.K || !D->have_vals[k1] || !D->have_vals[k2]) return SCILMM_ERR_STATE;
  constexpr int NBLK = 1024;
  st = ensure_io(sym, D, 2 * NBLK);
  if (st != SCILMM_OK) return st;
  hipStream_t s0 = D->stream;
  double* part = D->IO;
  HIPCHK(hipMemsetAsync(part, 0, sizeof(double) * 2 * NBLK, s0));
  const bool d1 = S.is_diag[k1], d2 = S.is_diag[k2];
  // sum over the FULL symmetric matrices = 2 * (sum over the stored lower-triangle slots) - (diagonal part); a
  // diagonal-only matrix meets any other matrix on the diagonal only
  if (!d1 && !d2 && S.nnz_pattern > 0)
    hipLaunchKernelGGL(k_dot_slots, dim3(NBLK), dim3(256), 0, s0, S.nnz_pattern, (const double*)D->vals[k1], (const double*)D->vals[k2], part);
  if (S.n > 0)
    hipLaunchKernelGGL(k_dot_diag, dim3(NBLK), dim3(256), 0, s0, S.n, D->v.pat_colptr, (const double*)D->vals[k1], (const double*)D->vals[k2],
                       d1 ? 1 : 0, d2 ? 1 : 0, part + NBLK);
  std::vector<double> h(2 * NBLK);
  HIPCHK(hipMemcpyAsync(h.data(), part, sizeof(double) * 2 * NBLK, hipMemcpyDeviceToHost, s0));
  HIPCHK(hipStreamSynchronize(s0));
  HIPCHK(hipGetLastError());
  long double all = 0.0L, dg = 0.0L;
  for (int b = 0; b < NBLK; ++b) { all += h[b]; dg += h[NBLK + b]; }
  *diag_dot = (double)dg;
  *frob = (d1 || d2) ? (double)dg : (double)(2.0L * all - dg);
  return SCILMM_OK;
}

int scilmm_spmm(scilmm_symbolic* sym, int32_t k, const double* X, int32_t r, double* Y) {
  if (!sym || !sym->S || !X || !Y || r <= 0) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  const Symbolic& S = *sym->S;
  if (k < 0 || k >= S.K || !D->have_vals[k]) return SCILMM_ERR_STATE;
  st = ensure_work(sym, D);
  if (st != SCILMM_OK) return st;
  const size_t cnt = (size_t)S.n * (size_t)r;
  st = ensure_io(sym, D, 2 * cnt);
  if (st != SCILMM_OK) return st;
  hipStream_t s = D->stream;
  HIPCHK(hipMemcpyAsync(D->IO, X, cnt * sizeof(double), hipMemcpyHostToDevice, s));
  for (int32_t cbeg = 0; cbeg < r; cbeg += RPMAX) {
    const int rc = std::min<int>(RPMAX, r - cbeg);
    const int rp = rp_of(rc);
    const int64_t tot = (int64_t)S.n * rp;
    const unsigned pb = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_perm_in, dim3(pb), dim3(256), 0, s, S.n, r, rp, cbeg, D->v.perm, (const double*)D->IO, D->W);
    HIPCHK(hipMemsetAsync(D->X, 0, sizeof(double) * (size_t)tot, s));
    if (S.is_diag[k])
      hipLaunchKernelGGL(k_spmm_diag, dim3(pb), dim3(256), 0, s, S.n, (const double*)D->vals[k], (const double*)D->W, rp, D->X);
    else
    {
      // (a wave per 256 pattern slots; lanes = right-hand-side columns)
      const int64_t spw = 256, nwav = (S.nnz_pattern + spw - 1) / spw;
      hipLaunchKernelGGL(k_spmm_w, dim3((unsigned)((nwav + 3) / 4)), dim3(256), 0, s, D->v, S.nnz_pattern, spw,
                         (const double*)D->vals[k], (const double*)D->W, rp, D->X);
    }
    hipLaunchKernelGGL(k_perm_out, dim3(pb), dim3(256), 0, s, S.n, r, rp, cbeg, D->v.perm, (const double*)D->X, D->IO + cnt);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(Y, D->IO + cnt, cnt * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return SCILMM_OK;
}

int scilmm_spmm_dev(scilmm_symbolic* sym, int32_t k, const double* dX, int32_t r, double* dY) {
  // device-pointer form of scilmm_spmm (same [n][r] layout): asynchronous on the engine's stream, like the other _dev calls
  if (!sym || !sym->S || !dX || !dY || r <= 0) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  const Symbolic& S = *sym->S;
  if (k < 0 || k >= S.K || !D->have_vals[k]) return SCILMM_ERR_STATE;
  st = ensure_work(sym, D);
  if (st != SCILMM_OK) return st;
  hipStream_t s = D->stream;
  for (int32_t cbeg = 0; cbeg < r; cbeg += RPMAX) {
    const int rc = std::min<int>(RPMAX, r - cbeg);
    const int rp = rp_of(rc);
    const int64_t tot = (int64_t)S.n * rp;
    const unsigned pb = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_perm_in, dim3(pb), dim3(256), 0, s, S.n, r, rp, cbeg, D->v.perm, dX, D->W);
    HIPCHK(hipMemsetAsync(D->X, 0, sizeof(double) * (size_t)tot, s));
    if (S.is_diag[k])
      hipLaunchKernelGGL(k_spmm_diag, dim3(pb), dim3(256), 0, s, S.n, (const double*)D->vals[k], (const double*)D->W, rp, D->X);
    else
    {
      // (a wave per 256 pattern slots; lanes = right-hand-side columns)
      const int64_t spw = 256, nwav = (S.nnz_pattern + spw - 1) / spw;
      hipLaunchKernelGGL(k_spmm_w, dim3((unsigned)((nwav + 3) / 4)), dim3(256), 0, s, D->v, S.nnz_pattern, spw,
                         (const double*)D->vals[k], (const double*)D->W, rp, D->X);
    }
    hipLaunchKernelGGL(k_perm_out, dim3(pb), dim3(256), 0, s, S.n, r, rp, cbeg, D->v.perm, (const double*)D->X, dY);
  }
  HIPCHK(hipGetLastError());
  return SCILMM_OK;
}

int scilmm_export_L(scilmm_factor* fac, int64_t* colptr, int32_t* rowidx, double* vals, int64_t* nnz) {
  if (!fac || !fac->sym || !nnz) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  scilmm_symbolic* sym = fac->sym;
  const Symbolic& S = *sym->S;
  int64_t total = 0;
  for (int32_t s = 0; s < S.nsuper; ++s) {
    int64_t m = S.sn_rowptr[s + 1] - S.sn_rowptr[s];
    int64_t w = S.sn_start[s + 1] - S.sn_start[s];
    total += m * w - w * (w - 1) / 2;
  }
  *nnz = total;
  if (!vals) return SCILMM_OK;
  if (!colptr || !rowidx) return SCILMM_ERR_ARG;
  if (fac->pending) {
    int stp = finish_factorize(fac, nullptr);
    if (stp != SCILMM_OK) return stp;
  }
  if (!fac->valid && !fac->inverted) return SCILMM_ERR_STATE;  // (after scilmm_selected_inverse: the entries of Z on L's pattern)
  Dev* D = (Dev*)sym->device;
  if (D->world > 1) {
    sym->err = "Factor.L(): a distributed factor is not gathered (every rank holds its own tail panels only)";
    return SCILMM_ERR_STATE;
  }
  std::vector<double> h((size_t)std::max<int64_t>(S.nnzL_stored, 1));
  HIPCHK(hipMemcpyAsync(h.data(), fac->L, sizeof(double) * (size_t)S.nnzL_stored, hipMemcpyDeviceToHost, D->stream));
  HIPCHK(hipStreamSynchronize(D->stream));
  int64_t p = 0;
  for (int32_t s = 0; s < S.nsuper; ++s) {
    const int32_t* rs = S.sn_rows.data() + S.sn_rowptr[s];
    int64_t m = S.sn_rowptr[s + 1] - S.sn_rowptr[s];
    int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
    const double* P = h.data() + S.sn_loff[s];
    for (int32_t j = 0; j < w; ++j) {
      colptr[c0 + j] = p;
      for (int64_t t = j; t < m; ++t) {
        rowidx[p] = rs[t];
        vals[p] = P[(int64_t)j * m + t];
        ++p;
      }
    }
  }
  colptr[S.n] = p;
  return SCILMM_OK;
}

int scilmm_ibd_values_device(scilmm_symbolic* sym, int32_t k, int32_t n, const int32_t* parents) {
  if (!sym || !sym->S || !parents) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  const Symbolic& S = *sym->S;
  if (k < 0 || k >= S.K || n != S.n || S.is_diag[k]) return SCILMM_ERR_ARG;
  if (S.nnz_pattern >= ((int64_t)1 << 32)) {
    sym->err = "scilmm_ibd_values_device: more than 2^32 pattern slots";
    return SCILMM_ERR_ARG;
  }
  // generation (longest path from a founder) of every individual; individuals must be in pedigree order
  std::vector<int32_t> gen((size_t)n, 0);
  int32_t maxgen = 0;
  for (int32_t i = 0; i < n; ++i) {
    int32_t g = 0;
    for (int q = 0; q < 2; ++q) {
      const int32_t p = parents[2 * i + q];
      if (p >= i) {
        sym->err = "scilmm_ibd_values_device: individuals are not in pedigree order (a parent follows its child)";
        return SCILMM_ERR_ARG;
      }
      if (p >= 0) g = std::max(g, gen[p] + 1);
    }
    gen[i] = g;
    maxgen = std::max(maxgen, g);
  }
  if (2 * maxgen > 254) {
    sym->err = "scilmm_ibd_values_device: pedigree deeper than 127 generations";
    return SCILMM_ERR_ARG;
  }
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  hipStream_t s0 = D->stream;
  const int64_t nnz = S.nnz_pattern;
  std::vector<void*> tmp;
  struct Cleanup {
    std::vector<void*>& v;
    ~Cleanup() { for (void* p : v) (void)hipFree(p); }
  } cleanup{tmp};
  auto tmalloc = [&](void** p, size_t bytes) -> int {
    HIPCHK(hipMalloc(p, std::max<size_t>(bytes, 8)));
    tmp.push_back(*p);
    return SCILMM_OK;
  };
  int32_t *d_gen = nullptr, *d_par = nullptr, *d_iperm = nullptr;
  uint8_t *key = nullptr, *skey = nullptr;
  uint32_t *slot = nullptr, *sslot = nullptr;
  int64_t* d_pass = nullptr;
  if ((st = tmalloc((void**)&d_gen, sizeof(int32_t) * (size_t)n)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&d_par, sizeof(int32_t) * 2 * (size_t)n)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&d_iperm, sizeof(int32_t) * (size_t)n)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&key, (size_t)nnz)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&skey, (size_t)nnz)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&slot, sizeof(uint32_t) * (size_t)nnz)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&sslot, sizeof(uint32_t) * (size_t)nnz)) != SCILMM_OK) return st;
  if ((st = tmalloc((void**)&d_pass, sizeof(int64_t) * 256)) != SCILMM_OK) return st;
  if (n > 0) {
    HIPCHK(hipMemcpyAsync(d_gen, gen.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s0));
    HIPCHK(hipMemcpyAsync(d_par, parents, sizeof(int32_t) * 2 * (size_t)n, hipMemcpyHostToDevice, s0));
    HIPCHK(hipMemcpyAsync(d_iperm, S.iperm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s0));
  }
  if (!D->vals[k]) HIPCHK(hipMalloc((void**)&D->vals[k], std::max<size_t>((size_t)nnz, 1) * sizeof(double)));
  if (nnz > 0) {
    hipLaunchKernelGGL(k_ibd_keys, dim3(4096), dim3(256), 0, s0, n, D->v.pat_colptr, D->v.pat_row, D->v.perm, (const int32_t*)d_gen, key, slot);
    size_t need = 0;
    void* cub = nullptr;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, key, skey, slot, sslot, nnz, 0, 8, s0));
    if ((st = tmalloc(&cub, need)) != SCILMM_OK) return st;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(cub, need, key, skey, slot, sslot, nnz, 0, 8, s0));
    HIPCHK(hipMemsetAsync(d_pass, 0xff, sizeof(int64_t) * 256, s0));  // -1 = key absent
    hipLaunchKernelGGL(k_ibd_bounds, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s0, nnz, (const uint8_t*)skey, d_pass);
    std::vector<int64_t> pass(257, -1);
    HIPCHK(hipMemcpyAsync(pass.data(), d_pass, sizeof(int64_t) * 256, hipMemcpyDeviceToHost, s0));
    HIPCHK(hipStreamSynchronize(s0));
    pass[256] = nnz;
    for (int q = 255; q >= 0; --q)
      if (pass[q] < 0) pass[q] = pass[q + 1];
    for (int q = 0; q <= 2 * maxgen; ++q) {
      const int64_t cnt = pass[q + 1] - pass[q];
      if (cnt <= 0) continue;
      hipLaunchKernelGGL(k_ibd_pass, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s0, cnt, (const uint32_t*)(sslot + pass[q]), n,
                         D->v.pat_colptr, D->v.pat_row, D->v.perm, (const int32_t*)d_iperm, (const int32_t*)d_par, D->vals[k]);
    }
  }
  HIPCHK(hipStreamSynchronize(s0));
  HIPCHK(hipGetLastError());
  D->have_vals[k] = 1;
  return SCILMM_OK;
}

int scilmm_dominance_values_device(scilmm_symbolic* sym, int32_t k_dst, int32_t k_src, int32_t n, const int32_t* parents) {
  if (!sym || !sym->S || !parents) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  const Symbolic& S = *sym->S;
  if (k_dst < 0 || k_dst >= S.K || k_src < 0 || k_src >= S.K || k_dst == k_src || n != S.n || S.is_diag[k_dst] || S.is_diag[k_src])
    return SCILMM_ERR_ARG;
  for (int64_t t = 0; t < 2 * (int64_t)n; ++t)
    if (parents[t] >= n) {
      sym->err = "scilmm_dominance_values_device: parent index out of range";
      return SCILMM_ERR_ARG;
    }
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  if (!D->have_vals[k_src]) {
    sym->err = "scilmm_dominance_values_device: the values of the source matrix are not resident";
    return SCILMM_ERR_STATE;
  }
  hipStream_t s0 = D->stream;
  int32_t *d_par = nullptr, *d_iperm = nullptr;
  struct Cleanup {
    int32_t*& a;
    int32_t*& b;
    ~Cleanup() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
  } cleanup{d_par, d_iperm};
  HIPCHK(hipMalloc((void**)&d_par, sizeof(int32_t) * 2 * (size_t)std::max(n, 1)));
  HIPCHK(hipMalloc((void**)&d_iperm, sizeof(int32_t) * (size_t)std::max(n, 1)));
  if (n > 0) {
    HIPCHK(hipMemcpyAsync(d_par, parents, sizeof(int32_t) * 2 * (size_t)n, hipMemcpyHostToDevice, s0));
    HIPCHK(hipMemcpyAsync(d_iperm, S.iperm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s0));
  }
  if (!D->vals[k_dst]) HIPCHK(hipMalloc((void**)&D->vals[k_dst], std::max<size_t>((size_t)S.nnz_pattern, 1) * sizeof(double)));
  if (S.nnz_pattern > 0)
    hipLaunchKernelGGL(k_dom_slots, dim3(4096), dim3(256), 0, s0, n, D->v.pat_colptr, D->v.pat_row, D->v.perm, (const int32_t*)d_iperm,
                       (const int32_t*)d_par, (const double*)D->vals[k_src], D->vals[k_dst]);
  HIPCHK(hipStreamSynchronize(s0));
  HIPCHK(hipGetLastError());
  D->have_vals[k_dst] = 1;
  return SCILMM_OK;
}

int scilmm_values_download(scilmm_symbolic* sym, int32_t k, double* slots_out) {
  if (!sym || !sym->S || !sym->device || !slots_out) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  if (k < 0 || k >= S.K || !D->have_vals[k]) return SCILMM_ERR_STATE;
  const size_t cnt = S.is_diag[k] ? (size_t)S.n : (size_t)S.nnz_pattern;
  HIPCHK(hipMemcpy(slots_out, D->vals[k], cnt * sizeof(double), hipMemcpyDeviceToHost));
  return SCILMM_OK;
}

int scilmm_selected_inverse(scilmm_factor* fac) {
  if (!fac || !fac->sym) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  scilmm_symbolic* sym = fac->sym;
  if (fac->pending) {
    int stp = finish_factorize(fac, nullptr);
    if (stp != SCILMM_OK) return stp;
  }
  if (!fac->valid) {
    sym->err = "scilmm_selected_inverse needs a valid factor";
    return SCILMM_ERR_STATE;
  }
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  if (D->world > 1) {
    sym->err = "scilmm_selected_inverse: not available on a distributed factor";
    return SCILMM_ERR_STATE;
  }
  hipStream_t st = D->stream;
  if (!D->d_col_front) {
    // column -> front, where every front's Y = L21 L11^-1 lives inside the per-level scratch (dense-tail fronts keep it
    // transposed, [u][128]), the non-tail tiles by level, and the items of the dense-tail kernel
    std::vector<int32_t> cf((size_t)std::max(S.n, 1), 0);
    for (int32_t f = 0; f < S.nsuper; ++f)
      for (int32_t c = S.sn_start[f]; c < S.sn_start[f + 1]; ++c) cf[(size_t)c] = f;
    std::vector<int64_t> yo((size_t)std::max(S.nsuper, 1), 0);
    int64_t ymax = 1;
    D->sinv_tail_front.assign((size_t)std::max(S.nlevels, 1), -1);
    D->sinv_pre_ptr.assign((size_t)S.nlevels + 1, 0);
    std::vector<int32_t> pre_tiles;
    for (int32_t l = 0; l < S.nlevels; ++l) {
      int64_t at = 0;
      for (int32_t q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) {
        const int32_t f = S.level_fronts[q];
        const int64_t w = S.sn_start[f + 1] - S.sn_start[f], u = (S.sn_rowptr[f + 1] - S.sn_rowptr[f]) - w;
        yo[(size_t)f] = at;
        at += ((f >= S.dense_first ? u * NB : u * w) + 1) & ~(int64_t)1;
        if (f >= S.dense_first) D->sinv_tail_front[(size_t)l] = f;
      }
      ymax = std::max(ymax, at);
      for (int64_t q = S.level_tile_ptr[l]; q < S.level_tile_ptr[l + 1]; ++q)
        if (S.tile_front[S.level_tiles[q]] < S.dense_first) pre_tiles.push_back(S.level_tiles[q]);
      D->sinv_pre_ptr[(size_t)l + 1] = (int64_t)pre_tiles.size();
    }
    // dense-tail items: (front s, 256 rows of R, a range of later fronts); about 1024 items per front
    std::vector<SinvWork> sw;
    std::vector<int32_t> tails;
    const int32_t nT = S.nsuper - S.dense_first;
    D->sinv_work_ptr.assign((size_t)nT + 1, 0);
    for (int32_t jj = 0; jj < nT; ++jj) {
      const int32_t f = S.dense_first + jj;
      tails.push_back(f);
      const int64_t w = S.sn_start[f + 1] - S.sn_start[f], u = (int64_t)S.n - S.sn_start[f] - w;
      const int32_t count = S.nsuper - 1 - f;  // later fronts
      if (u > 0 && count > 0) {
        const int64_t ntile = (u + 255) / 256;
        // about 1024 items per front, and a count that fills the last round of 256 workgroups: the fronts' launches follow
        // each other on one stream, so I items cost ceil(I / 256) rounds with nothing to fill the gap
        const int64_t base = std::max<int64_t>(1, std::min<int64_t>(count, 1024 / ntile));
        int32_t nseg = (int32_t)base;
        double best = -1.0;
        for (int64_t c = std::max<int64_t>(1, base / 2); c <= std::min<int64_t>(count, 2 * base + 1); ++c) {
          const int64_t items = ntile * c, rounds = (items + 255) / 256;
          const double score = (double)items / (double)(rounds * 256) - 0.02 * std::fabs((double)(c - base)) / (double)base;
          if (score > best) { best = score; nseg = (int32_t)c; }
        }
        for (int32_t sg = 0; sg < nseg; ++sg) {
          const int32_t ka = f + 1 + (int32_t)((int64_t)count * sg / nseg), kb = f + 1 + (int32_t)((int64_t)count * (sg + 1) / nseg);
          if (kb <= ka) continue;
          for (int64_t q = 0; q < ntile; ++q) sw.push_back(SinvWork{f, (int32_t)q, ka, kb});
        }
      }
      D->sinv_work_ptr[(size_t)jj + 1] = (int64_t)sw.size();
    }
    if (sw.empty()) sw.push_back(SinvWork{0, 0, 0, 0});
    if (tails.empty()) tails.push_back(0);
    if (pre_tiles.empty()) pre_tiles.push_back(0);
    const int32_t* t32;
    const int64_t* t64;
    const SinvWork* tsw;
    int stq;
    if ((stq = upload(sym, D, cf, &t32)) != SCILMM_OK) return stq;
    D->d_col_front = (int32_t*)t32;
    if ((stq = upload(sym, D, yo, &t64)) != SCILMM_OK) return stq;
    D->d_yoff = (int64_t*)t64;
    if ((stq = upload(sym, D, pre_tiles, &t32)) != SCILMM_OK) return stq;
    D->d_sinv_pre_tiles = (int32_t*)t32;
    if ((stq = upload(sym, D, tails, &t32)) != SCILMM_OK) return stq;
    D->d_sinv_tail_fronts = (int32_t*)t32;
    if ((stq = upload(sym, D, sw, &tsw)) != SCILMM_OK) return stq;
    D->d_sinv_work = (SinvWork*)tsw;
    void* yb = nullptr;
    HIPCHK(hipMalloc(&yb, sizeof(double) * (size_t)ymax));
    D->allocs.push_back(yb);
    D->d_ybuf = (double*)yb;
    if (!D->d_zeros) {
      HIPCHK(hipMalloc((void**)&D->d_zeros, 2048));
      HIPCHK(hipMemset(D->d_zeros, 0, 2048));
    }
    HIPCHK(hipFuncSetAttribute((const void*)k_sinv_tail, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  }
  HIPCHK(hipEventRecord(D->ev[6], st));
  const char* egen = tune_env("SCILMM_SINV_GENERIC");  // 1: the gather kernel for the dense tail as well (the round's first form)
  const bool tail_kernel = D->use_mfma && !(egen && egen[0] == '1');
  for (int32_t l = S.nlevels - 1; l >= 0; --l) {
    const int64_t t0 = D->lv_tile_ptr[l], t1 = D->lv_tile_ptr[l + 1];
    const int32_t f0 = D->lv_ptr[l], f1 = D->lv_ptr[l + 1];
    if (f1 == f0) continue;
    const unsigned nt = (unsigned)(t1 - t0);
    const int32_t tf = tail_kernel ? D->sinv_tail_front[(size_t)l] : -1;
    // tiles that go through the gather kernel: all of the level's, or the non-tail fronts' only
    const int32_t* wt = tf >= 0 ? D->d_sinv_pre_tiles + D->sinv_pre_ptr[(size_t)l] : D->d_level_tiles + t0;
    const unsigned nwt = tf >= 0 ? (unsigned)(D->sinv_pre_ptr[(size_t)l + 1] - D->sinv_pre_ptr[(size_t)l]) : nt;
    const int32_t ydf = tail_kernel ? S.dense_first : S.nsuper;  // fronts from here on keep Y transposed
    if (nt > 0) {
      if (D->use_mfma)
        hipLaunchKernelGGL(k_sinv_y<true>, dim3(nt), dim3(256), 0, st, D->v, D->d_level_tiles + t0, (const double*)fac->L,
                           (const double*)fac->invD, D->d_ybuf, (const int64_t*)D->d_yoff, ydf);
      else
        hipLaunchKernelGGL(k_sinv_y<false>, dim3(nt), dim3(256), 0, st, D->v, D->d_level_tiles + t0, (const double*)fac->L,
                           (const double*)fac->invD, D->d_ybuf, (const int64_t*)D->d_yoff, ydf);
    }
    hipLaunchKernelGGL(k_sinv_cc0, dim3((unsigned)(f1 - f0)), dim3(256), 0, st, D->v, D->d_level_fronts + f0, fac->L,
                       (const double*)fac->invD);
    if (tf >= 0) {
      const int32_t jj = tf - S.dense_first;
      const int64_t i0 = D->sinv_work_ptr[(size_t)jj], i1 = D->sinv_work_ptr[(size_t)jj + 1];
      if (i1 > i0) {
        const int32_t wtf = S.sn_start[tf + 1] - S.sn_start[tf];
        hipLaunchKernelGGL(k_sinv_zero, dim3(1, (unsigned)wtf), dim3(256), 0, st, D->v, (const int32_t*)(D->d_sinv_tail_fronts + jj), fac->L);
        hipLaunchKernelGGL(k_sinv_tail, dim3((unsigned)(i1 - i0)), dim3(512), sizeof(double) * (size_t)(2 * KBA * LDB), st, D->v,
                           S.dense_first, (const SinvWork*)(D->d_sinv_work + i0), fac->L, (const double*)D->d_ybuf,
                           (const int64_t*)D->d_yoff, (const int32_t*)D->d_col_front, (const double*)D->d_zeros);
      }
    }
    if (nwt > 0) {
      if (D->use_mfma)
        hipLaunchKernelGGL(k_sinv_w<true>, dim3(nwt), dim3(256), 0, st, D->v, wt, fac->L, (const double*)D->d_ybuf,
                           (const int64_t*)D->d_yoff, (const int32_t*)D->d_col_front, S.dense_first);
      else
        hipLaunchKernelGGL(k_sinv_w<false>, dim3(nwt), dim3(256), 0, st, D->v, wt, fac->L, (const double*)D->d_ybuf,
                           (const int64_t*)D->d_yoff, (const int32_t*)D->d_col_front, S.dense_first);
    }
    if (nt > 0) {
      if (D->use_mfma)
        hipLaunchKernelGGL(k_sinv_cc<true>, dim3(nt), dim3(256), 0, st, D->v, D->d_level_tiles + t0, fac->L, (const double*)D->d_ybuf,
                           (const int64_t*)D->d_yoff, ydf);
      else
        hipLaunchKernelGGL(k_sinv_cc<false>, dim3(nt), dim3(256), 0, st, D->v, D->d_level_tiles + t0, fac->L, (const double*)D->d_ybuf,
                           (const int64_t*)D->d_yoff, ydf);
    }
  }
  HIPCHK(hipEventRecord(D->ev[7], st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, D->ev[6], D->ev[7]));
  D->timing.quad_ms = ms;  // (reported through the quad_ms slot: the selected inverse replaces the trace estimator's sweeps)
  fac->valid = false;
  fac->inverted = true;
  return SCILMM_OK;
}

int scilmm_inverse_traces(scilmm_factor* fac, double* out) {
  if (!fac || !fac->sym || !out) return SCILMM_ERR_ARG;
  DevGuard guard(fac->sym);
  scilmm_symbolic* sym = fac->sym;
  if (!fac->inverted) {
    sym->err = "scilmm_inverse_traces: call scilmm_selected_inverse first";
    return SCILMM_ERR_STATE;
  }
  Dev* D = (Dev*)sym->device;
  const Symbolic& S = *sym->S;
  constexpr int NBLK = 1024;
  int st = ensure_io(sym, D, 2 * NBLK);
  if (st != SCILMM_OK) return st;
  hipStream_t s0 = D->stream;
  double* part = D->IO;
  std::vector<double> h(2 * NBLK);
  for (int32_t k = 0; k < S.K; ++k) {
    if (!D->have_vals[k]) return SCILMM_ERR_STATE;
    HIPCHK(hipMemsetAsync(part, 0, sizeof(double) * 2 * NBLK, s0));
    const bool dg = S.is_diag[k];
    if (!dg && S.nnz_pattern > 0)
      hipLaunchKernelGGL(k_sinv_trace, dim3(NBLK), dim3(256), 0, s0, S.nnz_pattern, D->v.asm_dst, (const double*)D->vals[k],
                         (const double*)fac->L, part);
    if (S.n > 0)
      hipLaunchKernelGGL(k_sinv_trace_diag, dim3(NBLK), dim3(256), 0, s0, S.n, D->v.pat_colptr, D->v.diag_dst, (const double*)D->vals[k],
                         dg ? 1 : 0, (const double*)fac->L, part + NBLK);
    HIPCHK(hipMemcpyAsync(h.data(), part, sizeof(double) * 2 * NBLK, hipMemcpyDeviceToHost, s0));
    HIPCHK(hipStreamSynchronize(s0));
    long double all = 0.0L, d1 = 0.0L;
    for (int b = 0; b < NBLK; ++b) { all += h[b]; d1 += h[NBLK + b]; }
    out[k] = dg ? (double)d1 : (double)(2.0L * all - d1);  // every off-diagonal pair counts twice (V^-1 and A_k are symmetric)
  }
  HIPCHK(hipGetLastError());
  return SCILMM_OK;
}

int scilmm_sync(scilmm_symbolic* sym) {
  if (!sym || !sym->device) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D = (Dev*)sym->device;
  HIPCHK(hipStreamSynchronize(D->stream));
  if (D->rhs_pending >= 0) return finish_rhs_timing(sym, D, D->rhs_pending);
  return SCILMM_OK;
}

int scilmm_last_timing(const scilmm_symbolic* sym, scilmm_timing* out) {
  if (!sym || !sym->device || !out) return SCILMM_ERR_ARG;
  Dev* D = (Dev*)sym->device;
  if (D->quad_pending) {
    // the device-pointer form of the quadratic forms returns without waiting: its timer (the LAST call's) is read here
    DevGuard guard(const_cast<scilmm_symbolic*>(sym));
    float q = 0;
    if (hipEventSynchronize(D->ev[7]) == hipSuccess && hipEventElapsedTime(&q, D->ev[6], D->ev[7]) == hipSuccess) D->timing.quad_ms = q;
    D->quad_pending = false;
  }
  D->timing.n_late_split = D->n_late_split;
  *out = D->timing;
  return SCILMM_OK;
}

int scilmm_set_front_precision(scilmm_symbolic* sym, int32_t bits) {
  if (!sym || !sym->S || (bits != 32 && bits != 64)) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  if (bits == 32 && !D->dense_on) {
    // (the handle keeps the precision it had: a refused request must not leave a half-set mode behind)
    sym->err = "fp32 fronts need the dense-tail path (tail narrower than 8192 columns: set SCILMM_TUNING=1 SCILMM_DENSE=1)";
    return SCILMM_ERR_STATE;
  }
  D->front_bits = bits;
  return SCILMM_OK;
}

int scilmm_set_profiling(scilmm_symbolic* sym, int32_t on) {
  if (!sym || !sym->S) return SCILMM_ERR_ARG;
  DevGuard guard(sym);
  Dev* D;
  int st = ensure_device(sym, &D);
  if (st != SCILMM_OK) return st;
  D->profiling = on != 0;
  D->serial_early = on == 2;
  return SCILMM_OK;
}

const char* scilmm_version(void) { return "scilmm_hip 0.1 (gfx950)"; }

}  // extern "C"

#ifdef SCILMM_POTRF_PROF
extern "C" int scilmm_debug_potrf_prof(unsigned long long* out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(scilmm::g_potrf_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -3;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(scilmm::g_potrf_prof), z, sizeof(z)) != hipSuccess) return -3;
  }
  return 0;
}
#endif

#ifdef SCILMM_CHAIN_PROF
extern "C" int scilmm_debug_chain_prof(unsigned long long* out, int n) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(scilmm::g_chain_prof), sizeof(unsigned long long) * 8 * (size_t)n) != hipSuccess) return -3;
  return 0;
}
#endif
