// Host-side symbolic analysis for the supernodal multifrontal Cholesky of V = sum_k s2_k A_k.
// Built once per sparsity pattern (the reference redoes it on every evaluation:
// scilmm/SparseCholesky.py:22-26,92).
#pragma once
#ifndef SCILMM_NB
#define SCILMM_NB 128
#endif
#include <cstdint>
#include <string>
#include <vector>

namespace scilmm {

void amd_order(int32_t n, const int64_t* g_ptr, const int32_t* g_idx, int32_t* perm_out, double dense_factor);

struct NdOptions {
  int64_t leaf_weight = 200;  // subgraphs of at most this many (uncompressed) vertices are ordered by minimum degree
  double oksep = 0.1;         // accept a separator only if its weight is below oksep * subgraph weight (else: AMD)
  double balance = 0.6;       // neither side of the edge bisection may exceed this share of the weight
  int32_t tries = 4;          // greedy-growing starts per bisection
  uint64_t seed = 1;
};
struct NdStats {
  int32_t n_compressed = 0;
  int64_t edges_compressed = 0, n_separators = 0, top_separator = 0;
};
void nd_order(int32_t n, const int64_t* g_ptr, const int32_t* g_idx, int32_t* perm_out, const NdOptions& opt, NdStats* stats);

void fill_count(int32_t n, const int64_t* g_ptr, const int32_t* g_idx, const int32_t* perm, int64_t* nnzL, double* flops,
                int32_t* max_cc, int32_t* colcount_out);

struct SymbolicOptions {
  int32_t ordering = 0;        // 0 = AMD, 1 = natural, 2 = user permutation, 3 = nested dissection, 4 = better of AMD / ND
  double nd_oksep = 0.1;       // nested dissection: separator acceptance threshold (1.0 = always dissect)
  int32_t relax_small = 4;     // always merge a child when the merged width is <= this
  int32_t relax_w1 = 16;       // merged width <= relax_w1 -> allow zero fraction z1
  int32_t relax_w2 = 48;       // merged width <= relax_w2 -> allow zero fraction z2
  double relax_z1 = 0.8, relax_z2 = 0.1, relax_z3 = 0.05;
  double amd_dense = 10.0;     // rows with degree > amd_dense*sqrt(n) are ordered last
  int32_t max_width = SCILMM_NB; // split supernodes wider than this (the kernels' block width; 0 = unlimited)
  int32_t tile_rows = 128;     // rows per target tile of the update kernel
  double dense_relax = 1.10;   // dense tail: padded / true flop ratio accepted when the top of the tree is made dense (0 = off)
  // A tail at least dense_wide_cols wide is updated by the descriptor-free dense kernel (engine: k_dense), 4-5 x faster
  // per flop than the gather path its fronts would otherwise feed: there the tail may grow up to dense_relax_wide
  // (every front it takes still fills >= half of its padded row list, i.e. costs <= 4 x its true flops).
  double dense_relax_wide = 1.25;
  int32_t dense_wide_cols = 32768;
};

// Everything the numeric phase needs.  "Front" s owns columns [sn_start[s], sn_start[s+1]) of the
// permuted matrix, has m = rows.size() rows (the first w are its own columns), stores its
// m x w column-major panel at L offset sn_loff[s] (leading dimension m).
struct Symbolic {
  int32_t n = 0;
  int32_t K = 0;
  std::vector<int32_t> perm;      // perm[new] = old   (factor is of V[perm][:,perm])
  std::vector<int32_t> iperm;     // iperm[old] = new
  std::vector<int32_t> parent;    // column etree (postordered labels)
  std::vector<int32_t> colcount;  // nnz of each column of L (without relaxation zeros)
  int32_t nsuper = 0;
  std::vector<int32_t> sn_start;  // [nsuper+1]
  std::vector<int32_t> sn_parent; // [nsuper]  -1 for roots
  std::vector<int64_t> sn_rowptr; // [nsuper+1] into sn_rows
  std::vector<int32_t> sn_rows;   // row lists (sorted; permuted labels)
  std::vector<int64_t> sn_loff;   // [nsuper+1] offsets of the panels in L storage (doubles)
  std::vector<int32_t> sn_level;  // [nsuper] height above the leaves
  // Fronts [dense_first, nsuper) are the DENSE TAIL: each of them has every later column as a row (m_s = n - start_s;
  // the analysis pads the trailing chain to that, see step 7b), i.e. together they are one dense lower-triangular
  // matrix cut into block columns (the trailing clique of a pedigree factor: 16.6k wide at the 100k config, 170k at
  // 1M).  Updates among them need no index lists.
  int32_t dense_first = 0;
  // children lists
  std::vector<int64_t> child_ptr; // [nsuper+1]
  std::vector<int32_t> child_idx; // children of each front in increasing order
  // left-looking update schedule: target supernode s receives, for every e in
  // [upd_ptr[s], upd_ptr[s+1]), the update  L_d[p0:m_d, :] * L_d[p0:p1, :]^T  from descendant
  // d = upd_src[e], where rows p0..p1-1 of d are exactly the rows of d that fall in the columns of s.
  std::vector<int64_t> upd_ptr;   // [nsuper+1]
  std::vector<int32_t> upd_src;   // descendant supernode
  std::vector<int32_t> upd_p0, upd_p1;
  // row tiles of the target panels (tile_rows rows each) and, per tile, the list of descendant
  // row ranges that contribute to it: combo c of tile g = pair combo_pair[c] restricted to rows
  // [combo_ta[c], combo_tb[c]) of the descendant (all of which land inside the tile).
  int32_t tile_rows = 128;
  std::vector<int64_t> tile_base;   // [nsuper+1] first global tile id of each front
  std::vector<int32_t> tile_front;  // [ntiles] owning front
  std::vector<int64_t> combo_ptr;   // [ntiles+1]
  bool combos_built = false;        // combo_* are filled on demand by build_tile_combos()
  std::vector<int32_t> combo_pair;  // index into upd_src/upd_p0/upd_p1
  std::vector<int32_t> combo_ta, combo_tb;
  std::vector<int32_t> combo_ip0;   // tile position of row ta when rows ta..tb land on consecutive positions, else -1
  std::vector<int32_t> upd_jp0;     // [npairs] target column of row p0 when rows p0..p1 are consecutive columns, else -1
  std::vector<int64_t> level_tile_ptr; // [nlevels+1] tiles of level l are level_tiles[ptr[l]..ptr[l+1])
  std::vector<int32_t> level_tiles;    // global tile ids grouped by level (tile order)
  std::vector<int64_t> level_pair_ptr; // [nlevels+1] update pairs whose TARGET is in level l
  std::vector<int32_t> level_pairs;
  // level schedule: fronts sorted by (level, size class)
  int32_t nlevels = 0;
  std::vector<int32_t> level_ptr; // [nlevels+1] into level_fronts
  std::vector<int32_t> level_fronts;
  // assembly: union lower pattern in permuted CSC order; entry e goes to L[asm_dst[e]]
  int64_t nnz_pattern = 0;        // entries of tril(union pattern)
  std::vector<int64_t> asm_dst;   // [nnz_pattern]
  std::vector<int64_t> diag_dst;  // [n] L offset of each diagonal entry (permuted order)
  std::vector<int64_t> pat_colptr; // [n+1] pattern slots of permuted column j (diagonal first)
  std::vector<int32_t> pat_row;    // [nnz_pattern] permuted row label of each slot
  std::vector<int64_t> inv_off;    // [nsuper+1] offsets of the w x w inverse diagonal blocks
  // per input matrix k: for each stored lower entry (CSR order, j<=i) the pattern slot it lands in
  // (so values_upload can permute data_k into pattern order); empty for diagonal-only matrices
  // dense tail, block pattern of the TRUE structure: tail front dense_first + d reaches (has at least one true row among
  // the columns of) the tail fronts dense_first + tail_blk[tail_blk_ptr[d] .. tail_blk_ptr[d+1]) (ascending, all > d);
  // its panel holds only padding at the columns of the others
  std::vector<int64_t> tail_blk_ptr;
  std::vector<int32_t> tail_blk;
  std::vector<std::vector<int64_t>> val_slot;   // [K][nnz_lower_k]
  std::vector<std::vector<int64_t>> val_src;    // [K][nnz_lower_k] index into data_k
  std::vector<uint8_t> is_diag;                 // [K] matrix k has a diagonal-only pattern
  // statistics
  int64_t nnzL = 0;        // true nonzeros of L (sum colcount)
  int64_t nnzL_stored = 0; // doubles of panel storage (includes relaxation zeros and the upper part of diagonal blocks)
  double flops = 0;        // sum colcount^2
  double update_flops = 0; // flops of all supernodal updates as EXECUTED (lower-triangular count; includes the padding of the dense tail)
  double dense_flops = 0;       // algorithmic update flops among the dense-tail fronts (true structure)
  double update_flops_pad = 0;  // part of update_flops that is padding of the dense tail (executed - algorithmic)
  std::string error;
};

// indptr[k]/indices[k]: CSR of matrix k (both triangles or lower only; only j<=i is read).
// perm_in: optional user permutation (perm_in[new] = old), required when opts.ordering == 2.
Symbolic* symbolic_analyze(int32_t n, int32_t K, const int64_t* const* indptr, const int32_t* const* indices,
                           const int32_t* perm_in, const SymbolicOptions& opts);

// Fills Symbolic::combo_* (step 11b of the analysis).  keep_front: optional [nsuper] mask of the target fronts whose
// tiles are enumerated (multi-GPU: the targets this rank owns); NULL = all.
// skip_dense: leave out the update pairs whose target AND descendant lie in the dense tail (the engine handles those
// with implicit, descriptor-free work items).
// skip_desc (optional, [nsuper]): descendants whose contributions to dense-tail targets are computed elsewhere (k_outside).
void build_tile_combos(Symbolic* S, const uint8_t* keep_front, bool skip_dense = false, const uint8_t* skip_desc = nullptr);

// Binary image of an analysis (everything but the lazily built tile combos); `key` = the caller's hash of the inputs the
// analysis depends on (patterns, permutation, options): symbolic_load returns NULL unless it matches.
bool symbolic_save(const Symbolic& S, const char* path, uint64_t key);
Symbolic* symbolic_load(const char* path, uint64_t key);

}  // namespace scilmm
