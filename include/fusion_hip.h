/*
 * fusion_hip.h -- C ABI of libfusion_hip.so: the MI355X (gfx950) scoring + fusion engine
 * behind the reference's src/retrievers/hybrid.py (Ranker / Aggregator).
 *
 * The reference has no FFI layer: its boundary is a Python API.  These
 * entry points are what a Python binding for that path needs; fusion_amd/_lib.py is
 * that binding (ctypes), and INTEGRATION.md shows the stub a maintainer of the
 * reference would add.  Citations are file:line in the reference tree.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _h (host);
 *  - the caller owns every buffer; nothing is allocated, freed or synchronised inside
 *    (graph-capturable): temporary storage comes from a caller-supplied workspace whose
 *    size fz_*_workspace_bytes() reports;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *    asynchronous on it and re-entrant across streams;
 *  - return value: FZ_OK (0) or a negative fz_status; no exceptions cross the boundary;
 *  - score planes are row-major [rows][ld] with ld >= n; ld*4 should be a multiple of 16 B
 *    for full-rate vector access (any ld is accepted).
 *
 * Data model: the reference's RankedLists (list[Q] of list[<=N] of {'corpus_id','score'},
 * hybrid.py:66-75,93-106) is held per system as dense planes indexed by corpus POSITION:
 *   score[q][j] fp32; rank[q][j] int32 = 0-based position of doc j in the system's list,
 *   -1 if absent (PLAID-pruned ColBERT lists, hybrid.py:137); order[q][r] = doc at rank r;
 *   len[q] = list length.  `idx` of hybrid.py:249,252 is `rank`.
 */
#ifndef FUSION_HIP_H
#define FUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FZ_OK = 0,
    FZ_ERR_ARG = -1,         /* null / negative / inconsistent argument */
    FZ_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for (fails loudly, never falls back) */
    FZ_ERR_HIP = -3,         /* a HIP runtime call failed; fz_last_hip_error() has the code */
    FZ_ERR_WORKSPACE = -4    /* workspace NULL or too small */
} fz_status;

/* Aggregator.transform_scores modes (hybrid.py:235-280) */
typedef enum {
    FZ_NORM_NONE = 0,        /* :280 passthrough */
    FZ_NORM_MINMAX = 1,      /* :254-258 */
    FZ_NORM_ZSCORE = 2,      /* :260-264, unbiased std */
    FZ_NORM_ARCTAN = 3,      /* :266-269 */
    FZ_NORM_PERCENTILE = 4,  /* :271-275 */
    FZ_NORM_NCE = 5          /* :276-277 */
} fz_norm;

typedef enum { FZ_RRF = 0 /* hybrid.py:252 */, FZ_BCF = 1 /* hybrid.py:249 */ } fz_rank_method;

const char* fz_strerror(int status);
int fz_last_hip_error(void);
/* ABI version of this header: bump on any signature change */
int fz_abi_version(void);

/* ---- K1: single-vector scoring (DPR / SPLADE), hybrid.py:101-103 ---------------------- */
/* Y[r] = X[r] / max(||X[r]||_2, 1e-12)   (util.cos_sim's normalisation; splade/base.py:195-196) */
int fz_normalize_rows_f32(const float* X, int rows, int d, int ldx, float* Y, int ldy, void* stream);
/* scores[q][j] = <Qn[q], Dn[j]>  fp32-in/fp32-accumulate MFMA GEMM (torch.mm, splade/base.py:197;
 * util.dot_score, sentence_transformers.py:229).  Qn [Q][ldq], Dn [N][ldd], d = contraction. */
int fz_dot_scores_f32(const float* Qn, int ldq, const float* Dn, int ldd, int Q, int N, int d, float* scores, int lds,
                      void* stream);
/* The same GEMM with the streaming top-k's threshold filter as its epilogue: NO score plane is written; a score that beats its
 * query's threshold tau[q] (or is NaN) is appended, with its document id id_base + corpus row, to the query's candidate list
 * (cand_scores / cand_ids [Q][cap], cand_len [Q] int32, as for fz_topk_filter_append_f32) -- in arrival order, i.e. for
 * fz_topk_fold_f32(unordered = 1).  tau_padded: [round_up(Q, 128)] floats, 16-byte aligned, +inf beyond Q.  Per 1.1 M-document shard
 * the scores would otherwise cross HBM twice (4.5 GB written by the GEMM, read by the filter). */
int fz_dot_scores_filter_f32(const float* Qn, int ldq, const float* Dn, int ldd, int Q, int N, int d, int64_t id_base,
                             const float* tau_padded, float* cand_scores, int64_t* cand_ids, int32_t* cand_len, int cap,
                             int32_t* overflow, void* stream);

/* ---- K2: ColBERT late interaction, hybrid.py:108-137 (exact MaxSim) ------ */
/* scores[q][j] = sum_{i<Lq} max_{t in doc j} <Qtok[q][i], Dtok[t]>.
 * Qtok [Q][Lq][dim] fp16; Dtok packed ragged [sumL][dim] fp16, doc j owns rows [Doff[j], Doff[j+1]);
 * Doff [N+1] int64 (device), sumL = Doff[N] (known to the host: rows of Dtok); max_doc_len = upper bound of the
 * document lengths (the reference's doc_maxlen = 512, hybrid.py:129; tokens beyond it are ignored; <= 16384).
 * dim must be 128 (run_colbert.sh:26); Lq in {32, 64, 128} (64: hybrid.py:129).  Empty documents score 0. */
int fz_maxsim_f16(const void* Qtok, const void* Dtok, const int64_t* Doff, int64_t sumL, int max_doc_len, int Q, int Lq, int N,
                  int dim, float* scores, int lds, void* stream);

/* ---- K5a/K6: stable descending row sort ---------------------------------------------- */
/* Python sorted(..., reverse=True) is stable (bm25.py:104, hybrid.py:306).  For each row:
 * the incoming sequence is keys gathered through init_order (NULL = identity, i.e. ties ->
 * ascending corpus position), of length row_len[row] (NULL = n); it is sorted by key
 * descending, ties keeping incoming order; -0.0 == +0.0; NaN first.
 * key_bits 32 (fp32 keys) or 64 (fp64 keys).  Outputs, each nullable:
 *   order[row][r]       payload (corpus position) at output rank r, r < row_len[row]
 *   sorted_keys[row][r] its key (same type as keys)
 *   rank[row][payload]  = r  (inverse permutation; other entries untouched: pre-fill with -1)
 * Any n.  Rows of up to fz_sort_max_n() (fp32 keys) / fz_sort_max_n_f64() (fp64 keys) elements live in the registers of one
 * workgroup (the fast path: LLeQA's 27,942 articles fit both); longer rows are chunk-sorted and ranked across chunks
 * (exact and stable as well, O(n * chunks * log n); row_stats are not produced there). */
int fz_sort_max_n(void);
int fz_sort_max_n_f64(void);
/* workspace: fz_sort_workspace_bytes(key_bits, rows, n) bytes of device memory (0 for fp32 rows that fit one workgroup).
 * fp64 keys are sorted by their high word (4 radix passes) and repaired in place where equal high words hide a low-word
 * inversion; a row with such a run longer than 17 keys is flagged in the workspace and redone by a generic 8-pass launch. */
size_t fz_sort_workspace_bytes(int key_bits, int rows, int n);
/* fp32 rows of 8,193 .. 28,672 columns (the rows a 1,024-thread workgroup sorts) that hold at least 4,096 keys and whose scores are spread like a ranker's -- few ties -- are ordered
 * without the digit passes: BUCKET RANKING (16,384 buckets whose widths follow the row's own density, a counting sort by bucket, each
 * key's rank = its bucket's first slot + the members below it, a neighbour check on the result; csrc/sort.hip).  Rows it does not
 * suit (heavy ties, a crowd of values 24 binades below the row's largest, an overfull bucket, a check it cannot settle) take the digit
 * passes as before; the output is the same permutation either way, bit for bit.  FZ_SORT_BUCKET_RANK=0 in the environment turns it
 * off (A/B runs).  fz_sort_bucket_rank_rows: counts3[0] = rows ordered that way on the current device since the last reset,
 * counts3[1] = of those, rows that needed a swapped pair put back, counts3[2] = rows that started on it and were handed to the digit
 * passes; reset != 0 clears the counters.  Synchronises the device (tests and tools only). */
int fz_sort_bucket_rank_rows(uint64_t* counts3, int reset);
/* Round 6 (ABI 19): the ranking sort of a LEXICAL system (bm25.py:100-106: BM25 / TF-IDF score every document, and every document that
 * shares no term with the query scores exactly 0.0 -- most of a corpus).  fz_sort_rows_desc(keys, 64, NULL, ...) for float64 rows in the
 * identity sequence, through an instantiation in which a row of 8,193 .. 28,672 columns at least 3/8 of whose >= 4,096 keys are +-0.0 leaves
 * the zeros out of the ordering phases: the non-zero keys are compacted into fewer items per thread, sorted, and written around the block
 * of zeros, which keep their sequence order (csrc/sort.hip, ZC).  Same stable permutation and outputs, bit for bit, whatever the rows hold;
 * a row without zeros costs ~5 % more here than through fz_sort_rows_desc, a row of 60 % zeros 20 % less, of 80 % a third less -- which is why
 * the caller says what its rows are.  Needs the order output (its row doubles as scratch); other shapes take fz_sort_rows_desc's path.
 * FZ_SORT_ZERO_COMPACT=0 in the environment turns the instantiation off (A/B runs).  fz_sort_zero_compact_rows: counts2[0] = rows compacted on
 * the current device since the last reset, counts2[1] = rows that went through it with too few zeros; reset != 0 clears.  Synchronises the
 * device (tests and tools only). */
int fz_sort_rows_desc_lexical(const double* keys, const int32_t* row_len, int rows, int n, int ld, int32_t* order, double* sorted_keys,
                              int32_t* rank, float* row_stats, void* workspace, size_t workspace_bytes, void* stream);
int fz_sort_zero_compact_rows(uint64_t* counts2, int reset);
/* row_stats (nullable, [4][rows] fp32): mean | UNBIASED standard deviation | min | max of each list's values as float32 (fp64 keys
 * rounded first) -- torch.mean / torch.std / torch.min / torch.max of hybrid.py:254-262, a by-product of having the row in
 * registers (min and max of a sorted list are its two ends; a NaN sorts first and makes both NaN): with them min-max and z-score
 * fusion of ranked systems is one flat pass that reduces nothing (fz_fuse_nsf_pstats_f32 takes row_stats + k*rows directly).  An
 * empty list gives NaN, NaN, 0, 0.
 * stats_len (nullable, [rows] int32, needs row_stats; fp32 keys only, FZ_ERR_UNSUPPORTED otherwise): the statistics cover only the
 * first stats_len[row] entries of the SORTED list -- a ranking cut to its top-k (PLAID-style short ColBERT lists, hybrid.py:137;
 * return_topk) normalises over the listed documents only. */
int fz_sort_rows_desc(const void* keys, int key_bits, const int32_t* init_order, const int32_t* row_len, int rows, int n,
                      int ld, int32_t* order, void* sorted_keys, int32_t* rank, float* row_stats, const int32_t* stats_len,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Same sort, incoming sequence given the other way round: init_rank[row][j] = position of column j in the incoming
 * sequence (-1 = not in it); restricted to the columns in the sequence it is a bijection onto [0, row_len[row]).
 * This is what a rank plane IS, so the fused-list ordering (ties keep system 0's order) needs no gather:
 * keys and positions are read coalesced and placed through LDS. */
int fz_sort_rows_desc_placed(const void* keys, int key_bits, const int32_t* init_rank, const int32_t* row_len, int rows, int n,
                             int ld, int32_t* order, void* sorted_keys, int32_t* rank, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Rank fusion (K5b below) AS THE LOAD PHASE of the final ordering -- hybrid.py:248-252 (the per-rank terms), :301-304 (the float64 sums in
 * system order) and :306 (the stable descending sort) in one kernel: what fz_fuse_rank_f64 followed by fz_sort_rows_desc[_placed] on its
 * plane returns, bit for bit (order, the fused float64 scores in sorted_scores, rank), without the [rows][ld] float64 plane existing.
 * ranks_h: HOST array of S device rank planes [rows][ld] int32 (-1 = the system does not list the document); lens [S][rows] int32 (device).
 * Incoming sequence (the fused dict's first-insertion order): init_rank (placed; when it is ranks_h[0] itself -- every list full -- that
 * plane is read once), or init_order (gathered; row_len = the number of listed documents), or neither (columns in order).
 * n <= fz_sort_max_n_f64() (FZ_ERR_UNSUPPORTED beyond: use the two calls).  Workspace: fz_sort_rank_fused_workspace_bytes(rows, n, ld) --
 * one flag per row + one float64 row per row, written only for rows whose repair the fast form hands to the generic eight-pass launch
 * (more than 2,048 keys sharing a high key word: not a ranker's scores; reserved, not moved, otherwise). */
size_t fz_sort_rank_fused_workspace_bytes(int rows, int n, int ld);
int fz_sort_rank_fused_desc(const int32_t* const* ranks_h, const int32_t* lens, int S, int method, const int32_t* init_order,
                            const int32_t* init_rank, const int32_t* row_len, int rows, int n, int ld, int32_t* order,
                            double* sorted_scores, int32_t* rank, void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostic for the call above: out[r] = 1 / (60 + r + 1) (hybrid.py:252) in float64, r < count <= 2^20 -- fast != 0: the division-free
 * sequence its load phase uses, else the IEEE division of fz_fuse_rank_f64.  The two must agree bit for bit (tests). */
int fz_rrf_terms_f64(int count, int fast, double* out, void* stream);

/* Top-k form of the final ordering (what main() reads of the fused lists: predictions(1000), hybrid.py:537).  Per row: the candidates for the
 * first k places of the sort above -- every column with pos >= 0 whose fused score, rounded to float32, is not below the k-th largest such
 * value (the k best and every tie at the k-th place) -- written in no particular order as cand_cols [rows][cap] (column), cand_vals
 * [rows][cap] (its fused score, same type as `fused`), cand_negpos [rows][cap] (-(float)pos: a descending sort of it is ascending insertion
 * order), cand_len [rows].  pos [rows][ld] = first-insertion position of the column (< 0: in no list; NULL: the column index).  Sorting the
 * candidates by cand_negpos and then, stably, by cand_vals (fz_sort_rows_desc twice, rows of cap keys) gives the first k entries of the
 * full sort exactly.  *overflow (device int32, zeroed by the caller) is set when a row has more than cap candidates (a tie run longer
 * than cap - k at the k-th place): sort that batch in full.  n <= 28,672 (one workgroup holds the row), k <= cap. */
int fz_select_topk_f(const void* fused, int key_bits, const int32_t* pos, int rows, int n, int ld, int k, int cap, int32_t* cand_cols,
                     void* cand_vals, float* cand_negpos, int32_t* cand_len, int32_t* overflow, void* stream);

/* ---- K5b: rank-based fusion, hybrid.py:206-211,248-252,301-304 ------------------------- */
/* fused[q][j] = sum over systems s (in the given order, fp64, starting from 0.0) of
 *   rrf: 1/(60+rank+1)     bcf: (len-rank+1)/len      for rank >= 0;  -inf if j is in no list.
 * ranks_h: HOST array of S device pointers, each [Q][ld] int32; lens [S][Q] int32 (device). */
int fz_fuse_rank_f64(const int32_t* const* ranks_h, const int32_t* lens, int S, int Q, int N, int ld, int method,
                     double* fused, void* stream);

/* ---- K3/K4: normalise -> weight -> sum, hybrid.py:212-214,254-280,291,301-304 --------- */
/* per-row statistics over the valid entries (rank NULL = all valid):
 *   FZ_NORM_MINMAX: stat_a=min stat_b=max;  FZ_NORM_ZSCORE: stat_a=mean stat_b=unbiased std
 *   (fp64 accumulation, rounded to fp32; 1-element rows give std=NaN as torch.std does). */
int fz_row_stats_f32(const float* scores, const int32_t* rank, int rows, int N, int ld, int norm, float* stat_a,
                     float* stat_b, void* stream);
/* fused[q][j] = sum_s fl32( t_s(score_s[q][j]) * fl32(w_s) ), fp32, unfused, in system order
 * (NumPy-2 semantics of hybrid.py:291,304), t_s = the normalisation with that row's statistics;
 * docs absent from every system: -inf (so with S = 1 the output is -inf, NOT 0, wherever the system does not list the document:
 * fz_zero_unlisted_f32 turns such a plane into the input fz_gold_ranks_* expects).  One pass over HBM: each plane is read once.
 * planes_h / ranks_h / distr_h: HOST arrays of S device pointers (ranks_h nullable, entries
 * nullable = all docs present; distr_h needed only for PERCENTILE/NCE: ascending fp32 tables of
 * P_h[s] entries, hybrid.py:272).  w_h: HOST fp64 weights (rounded to fp32 inside).
 * norm = FZ_NORM_NONE is rejected here: use fz_fuse_none_f64 (the reference stays in float64). */
/* valid_bits_h (nullable, entries nullable): per system a validity BITMAP [Q][ldb] uint32 (bit j & 31 of word j >> 5 = doc j is in
 * the list) from fz_rank_to_bitmap -- the fusion passes then read 1 bit instead of a 4-byte rank per document; a system with a
 * bitmap needs no ranks_h entry. */
int fz_rank_to_bitmap(const int32_t* rank, int rows, int N, int ld, uint32_t* bits, int ldb, void* stream);
int fz_fuse_nsf_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N,
                    int ld, int norm, const float* const* distr_h, const int32_t* P_h, const uint32_t* const* valid_bits_h, int ldb,
                    float* fused, void* stream);
/* Same result for rows longer than the one-pass kernel holds in registers (N > 32768: fz_fuse_nsf_f32 returns
 * FZ_ERR_UNSUPPORTED): min-max / z-score statistics are supplied by the caller, [S][Q] fp32 each, from fz_row_stats_f32
 * (nullable for the other normalisations).  Two passes over HBM. */
int fz_fuse_nsf_stats_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N,
                          int ld, int norm, const float* const* distr_h, const int32_t* P_h, const float* stat_a,
                          const float* stat_b, const uint32_t* const* valid_bits_h, int ldb, float* fused, void* stream);
/* The same flat pass with ONE statistics pointer PER SYSTEM: stat_a_h / stat_b_h are HOST arrays of S device pointers, each [Q] fp32
 * (min | mean and max | unbiased std of that system's lists) -- e.g. straight into the row_stats block its ranking sort wrote
 * (fz_sort_rows_desc): a fusion call then gathers, concatenates and reduces nothing. */
int fz_fuse_nsf_pstats_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N,
                           int ld, int norm, const float* const* distr_h, const int32_t* P_h, const float* const* stat_a_h,
                           const float* const* stat_b_h, const uint32_t* const* valid_bits_h, int ldb, float* fused, void* stream);
/* ---- percentile-rank / NCE at the table sizes the reference READS: hybrid.py:412,451 (score_distributions_raw_*_28k.csv: |corpus| + 1 =
 * 27,943 quantiles per system) and :374 (_10k: 10,001).  fz_fuse_nsf_f32 keeps all S tables in LDS up to ~1.4 k entries per system at
 * S = 4 and beyond that searches them in global memory (correct, slow).  These entries keep ONE system's table LDS-resident at a time
 * (up to ~38 k entries) and need a prepared workspace: an aligned copy of every table, its bucket table and -- NCE -- the value of every
 * table index.  Same results as fz_fuse_nsf_f32, bit for bit.
 *   fz_nsf_tables_workspace_bytes: bytes for these table lengths (0: a table is too long for LDS or an argument is bad);
 *   fz_nsf_tables_prepare: fills the workspace from the S ascending device tables (one small launch; the workspace stays valid for
 *     any number of fusion calls with the same tables, lengths and norm);
 *   fz_fuse_nsf_tables_f32: fz_fuse_nsf_f32's arguments + the prepared workspace.  FZ_ERR_UNSUPPORTED when a table does not fit LDS or
 *     the planes are not 16-byte aligned with ld % 4 == 0: fz_fuse_nsf_f32 takes those.  Calls whose tables ALL fit LDS at once run
 *     fz_fuse_nsf_f32's table kernel (the workspace is then not read);
 *   fz_nsf_tables_path: which kernel the call above runs for these shapes and pointers (a fz_tables_path, or FZ_ERR_ARG). */
typedef enum {
    FZ_TABLES_PATH_LDS_ALL = 0,   /* every table LDS-resident for the whole launch (fuse_nsf_table_kernel) */
    FZ_TABLES_PATH_LDS_SWAP = 1,  /* one system's table LDS-resident at a time (fuse_nsf_bigtab_kernel) */
    FZ_TABLES_PATH_ROW = 2        /* not taken by fz_fuse_nsf_tables_f32: fz_fuse_nsf_f32's global-memory search */
} fz_tables_path;
size_t fz_nsf_tables_workspace_bytes(int S, const int32_t* P_h, int norm);
/* byte offset, inside a prepared workspace, of system s's 16-byte header {float first entry; float buckets per unit; int32 probes per
 * search; int32 entries in the fullest bucket} -- diagnostics: the search costs `probes` LDS reads per score (log2 of the fullest bucket of
 * an equi-width bucket table; 3..6 for quantile tables of real score distributions, more for tables with long runs of equal entries).
 * (size_t)-1 on a bad argument. */
size_t fz_nsf_tables_header_offset(int S, const int32_t* P_h, int norm, int s);
int fz_nsf_tables_prepare(const float* const* distr_h, const int32_t* P_h, int S, int norm, void* workspace, size_t workspace_bytes,
                          void* stream);
int fz_fuse_nsf_tables_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N,
                           int ld, int norm, const float* const* distr_h, const int32_t* P_h, const uint32_t* const* valid_bits_h,
                           int ldb, float* fused, const void* workspace, size_t workspace_bytes, void* stream);
int fz_nsf_tables_path(const float* const* planes_h, const int32_t* const* ranks_h, int S, int Q, int N, int ld, int norm,
                       const int32_t* P_h, const float* fused);
/* plane[q][j] = 0 where rank[q][j] < 0.  A system adds nothing for a document it does not list (hybrid.py:301-304); the
 * single-system planes of fz_fuse_nsf_f32 hold -inf there (see below), the weight sweep wants 0. */
int fz_zero_unlisted_f32(float* plane, const int32_t* rank, int Q, int N, int ld, void* stream);
/* min / max of RANKED lists without a reduction: a list sorted by score has its maximum first and its minimum last.
 * mn[row] = scores[row][order[row][len-1]], mx[row] = scores[row][order[row][0]], len = lens[row] (NULL = N); an empty list gives
 * 0, 0; a NaN at the head makes both NaN (torch.min / torch.max propagate it, hybrid.py:254-258).  With these,
 * fz_fuse_nsf_stats_f32 is a single flat streaming pass -- the fast form of min-max fusion for systems that come with their
 * order plane. */
int fz_minmax_from_order_f32(const float* scores, const int32_t* order, const int32_t* lens, int rows, int N, int ld,
                             float* mn, float* mx, void* stream);
/* The same for all S systems of a fusion in one launch: planes_h / orders_h HOST arrays of S device planes [Q][ld], lens [S][Q]
 * (device, NULL = N), mn / mx [S][Q] -- the layout fz_fuse_nsf_stats_f32 takes. */
int fz_minmax_from_orders_f32(const float* const* planes_h, const int32_t* const* orders_h, const int32_t* lens, int S, int Q, int N,
                              int ld, float* mn, float* mx, void* stream);
/* 'none' / unknown normalisation: fused[q][j] = sum_s (double)score_s * w_s in fp64 (hybrid.py:280,291,304) */
int fz_fuse_none_f64(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N,
                     int ld, double* fused, void* stream);
/* Weight-and-sum with NumPy's scalar promotion, for what the two entries above do not cover (hybrid.py:291,304):
 *   plane_is_f64_h[s] != 0: planes_h[s] is a float64 plane (BM25's Python-float scores, raw scores of a host list) --
 *     the 'none' passthrough keeps them unrounded;
 *   narrow_h[s] != 0: w_s is a weak (Python float) or float32 weight: fl32 product, and a document's sum stays fl32 until
 *     its first float64 product; narrow_h[s] == 0: w_s is an np.float64 (the tuning grid, hybrid.py:405-409): float64.
 * Inputs for the normalised modes are the per-system transformed planes (fz_fuse_nsf_f32 of one system, weight 1).
 * Both arrays nullable (= all float32 planes, all wide: fz_fuse_none_f64). */
int fz_fuse_wsum_f64(const void* const* planes_h, const int32_t* plane_is_f64_h, const int32_t* const* ranks_h,
                     const double* w_h, const int32_t* narrow_h, int S, int Q, int N, int ld, double* fused, void* stream);

/* ---- first-insertion order of the fused dict, hybrid.py:301-304 (tie-break, SURVEY KAT-1) */
/* ins_order[q][0..U[q]) = docs in the order aggregate_scores first inserts them: system by system,
 * each in its rank order, skipping docs already seen.  orders_h: HOST array of S device pointers
 * [Q][ld]; lens [S][Q].  pos (nullable, [Q][ld], pre-filled with -1 by the caller): the inverse, pos[q][doc] = its first-insertion
 * position -- what fz_gold_ranks_* and fz_select_topk_f take.  Workspace: fz_insertion_order_workspace_bytes(Q, N). */
size_t fz_insertion_order_workspace_bytes(int Q, int N);
int fz_insertion_order(const int32_t* const* orders_h, const int32_t* lens, int S, int Q, int N, int ld, int32_t* ins_order,
                       int32_t* U, int32_t* pos, void* workspace, size_t workspace_bytes, void* stream);

/* ---- top-k + shard merge: sentence_transformers.py:346-364 (chunked score -> topk -> heap) */
/* k best of each row by (score desc, id asc); ids = id_base + column. out [rows][k]; rows with
 * fewer than k columns are padded with (-inf, -1).  k <= fz_topk_max_k(). */
int fz_topk_max_k(void);
size_t fz_topk_workspace_bytes(int rows, int n, int k);
int fz_topk_rows_f32(const float* scores, int rows, int n, int ld, int k, int64_t id_base, float* out_scores,
                     int64_t* out_ids, void* workspace, size_t workspace_bytes, void* stream);
/* streaming form of the same loop: merge one more chunk of scores into the running per-row top-k.  Only scores strictly
 * above a row's current k-th best can enter (chunks arrive in ascending id order, so ties lose), so the chunk is filtered
 * in one streaming pass and only the survivors (+ the running list) are sorted.  run_* [rows][k] -> new_* [rows][k]
 * (distinct buffers, sorted by score desc, id asc; (-inf,-1) padding).  *overflow (device int32, zeroed by the caller)
 * is set when a row had more than `cap` survivors: redo that chunk with fz_topk_rows_f32 + fz_topk_merge. */
size_t fz_topk_update_workspace_bytes(int rows, int k, int cap);
int fz_topk_update_f32(const float* scores, int rows, int n, int ld, int64_t id_base, const float* run_scores,
                       const int64_t* run_ids, int k, int cap, float* new_scores, int64_t* new_ids, int32_t* overflow,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same step in two halves, so that several chunks share ONE sort (the sharded search folds 4 times per 1.1 M-document shard
 * instead of once per chunk): fz_topk_filter_append_f32 appends the chunk's scores above tau[row] (or NaN) to the row's candidate
 * list -- ascending id inside the chunk, chunks fed in ascending id order; fz_topk_fold_f32 merges [running k | candidates] into
 * the new running list (ties: running entries first, then ascending id), writes the new threshold tau_out[row] = k-th best
 * (nullable) and empties the candidate lists.  cand_scores / cand_ids [rows][cap], cand_len [rows] int32 (zero before the first
 * call); *overflow becomes 1 if a row's list would exceed cap (the caller then redoes the search exactly).  k + cap <= 35840.
 * unordered != 0: the candidates are in no particular order (fz_dot_scores_filter_f32 appends them as its waves finish); the fold
 * then also puts every run of equal scores that reaches into the first k into ascending id order -- the same lists as from ordered
 * candidates -- and sets *overflow if such a run is longer than it looks at (64 entries past k, 512 in all). */
int fz_topk_filter_append_f32(const float* scores, int rows, int n, int ld, int64_t id_base, const float* tau, float* cand_scores,
                              int64_t* cand_ids, int32_t* cand_len, int cap, int32_t* overflow, void* stream);
size_t fz_topk_fold_workspace_bytes(int rows, int k, int cap);
int fz_topk_fold_f32(const float* run_scores, const int64_t* run_ids, int rows, int k, const float* cand_scores, const int64_t* cand_ids,
                     int32_t* cand_len, int cap, int unordered, float* new_scores, int64_t* new_ids, float* tau_out, int32_t* overflow,
                     void* workspace, size_t workspace_bytes, void* stream);
/* merge G per-shard lists [G][rows][k] (as all-gathered over RCCL) into the global top-k [rows][k] */
int fz_topk_merge(const float* in_scores, const int64_t* in_ids, int G, int rows, int k, float* out_scores, int64_t* out_ids,
                  void* stream);
/* C1 -- the ONE collective of the corpus-sharded configuration (spec: the heap merge across corpus chunks of
 * sentence_transformers.py:346-364, here across GPUs): ncclAllGather of this rank's [Q][k] (fp32 score, int64 id) lists over
 * `rccl_comm` (an ncclComm_t of `world` ranks, one per GPU) into the workspace, then the local G-way merge -- identical on
 * every rank, ties by ascending global id.  RCCL is resolved at run time (the host process's copy first, then librccl.so);
 * FZ_ERR_UNSUPPORTED when there is none.  Workspace: fz_topk_allgather_workspace_bytes(world, Q, k) device bytes. */
size_t fz_topk_allgather_workspace_bytes(int world, int Q, int k);
int fz_topk_allgather(const float* local_scores, const int64_t* local_ids, int Q, int k, void* rccl_comm, int world,
                      float* out_scores, int64_t* out_ids, void* workspace, size_t workspace_bytes, void* stream);

/* ---- A1: BM25 scoring on device, bm25.py:149-156 -------------------------------------- */
/* scores[q][j] (fp64) = sum over query terms in query order of idf*tf*(k1+1)/(tf+k1*(1-b+b*dl/avgdl)).
 * CSR postings by term (toff [V+1], pdoc, ptf), idf [V] fp64, doc_len [N]; queries as CSR of term
 * ids (qoff [Q+1], qterms; -1 = out of vocabulary).
 * doc_norm (nullable): per-document k1*(1-b+b*dl/avgdl) from fz_bm25_doc_norms_f64 -- the same fp64 bits as the inline
 * sub-expression, computed once per document instead of once per posting. */
int fz_bm25_doc_norms_f64(const int32_t* doc_len, int N, double avgdl, double k1, double b, double* out, void* stream);
/* slice_off (nullable): per-index table [V][NS + 1] int64, NS = ceil(N / fz_bm25_slice_docs()): the first posting of term t whose document
 * is >= s * fz_bm25_slice_docs() (entry NS = toff[t + 1]), from fz_bm25_slice_offsets (a GRAIN of 3,584 documents; a workgroup scores one (query, slice of one or two grains)) -- a workgroup
 * otherwise finds every query term's posting sub-range by two binary searches of ~15 dependent loads each. */
int fz_bm25_slice_docs(void);
int fz_bm25_slice_offsets(const int64_t* toff, const int32_t* pdoc, int V, int N, int64_t* out, void* stream);
int fz_bm25_scores_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf, const int32_t* doc_len,
                       const double* doc_norm, const int64_t* slice_off, double avgdl, double k1, double b, const int64_t* qoff,
                       const int32_t* qterms, int Q, int N, double* scores, int lds, void* stream);
/* The same launch also writing the scores rounded to float32 (scores32 [Q][lds32], nullable) -- the plane the normalisations read
 * (torch.tensor(scores, dtype=float32), hybrid.py:255) -- from the same accumulators: no separate conversion pass over the float64 plane. */
int fz_bm25_scores_f64_f32(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf, const int32_t* doc_len,
                           const double* doc_norm, const int64_t* slice_off, double avgdl, double k1, double b, const int64_t* qoff,
                           const int32_t* qterms, int Q, int N, double* scores, int lds, float* scores32, int lds32, void* stream);

/* Round 6 (ABI 19): the posting-value table.  For a given index and (k1, b) every posting has ONE term value
 *   pval[e] = idf_t * (tf_e * (k1 + 1)) / (tf_e + doc_norm[d_e])      (float64, the expression order of bm25.py:154)
 * -- idf, tf and the length norm are all the index's, nothing in it depends on the query.  fz_bm25_posting_values_f64 tabulates it once
 * (like the idf table; once per (k1, b) of a grid search), fz_bm25_scores_pv_f64_f32 is fz_bm25_scores_f64_f32 whose walk only ADDS the
 * tabulated terms, in query order: the same planes, bit for bit, without a float64 division per (query, posting) -- most of the scoring
 * kernel's time (1024 queries x 27,942 documents: 0.33 -> 0.1x ms).  doc_norm: fz_bm25_doc_norms_f64 (required here). */
int fz_bm25_posting_values_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf, const double* doc_norm,
                               int V, int64_t nnz, double k1, double* out, void* stream);
int fz_bm25_scores_pv_f64_f32(const int64_t* toff, const int32_t* pdoc, const double* pval, const int64_t* slice_off, const int64_t* qoff,
                              const int32_t* qterms, int Q, int N, double* scores, int lds, float* scores32, int lds32, void* stream);
/* TFIDF.score (bm25.py:108-115), the base class of the reference's lexical retrievers: scores[q][j] (fp64) = sum over the query's terms,
 * in query order, of tf(t, d) * idf(t) -- the same posting walk without the length norm.  idf [V] is the caller's table (TFIDF's is
 * log10((N + 1) / (df + 1)), bm25.py:86-88; AtireBM25 hands that table to fz_bm25_scores_f64 instead, bm25.py:170-172).  slice_off,
 * scores32: as above, both nullable.  (ABI 19) */
int fz_tfidf_scores_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf, const int64_t* slice_off,
                        const int64_t* qoff, const int32_t* qterms, int Q, int N, double* scores, int lds, float* scores32, int lds32,
                        void* stream);

/* ---- N1: weight-grid sweep of the linear fusion, hybrid.py:404-426 ------------------------ */
/* Fused ranks of the gold documents for W weight vectors at once, without fusing or sorting:
 * out_ranks[w][q][g] = #{docs that precede gold g in the list Aggregator.fuse(method='nsf') would return with
 * weights[w]} (fused score desc, ties by first-insertion position).  Every metric of run_evaluation is a
 * function of these ranks.  T_h: HOST array of S device planes [Q][ld] holding the NORMALISED scores
 * (fz_fuse_nsf_f32 of one system with weight 1, then -- for a system with a partial list -- fz_zero_unlisted_f32: entries of docs
 * the system does not list MUST be 0, the kernel multiplies them by the weights like any other; fz_fuse_nsf_f32 leaves -inf
 * there); pos [Q][ld] =
 * first-insertion position (-1 = in no list); weights [W][S] fp32; gold [Q][fz_tune_max_gold()] corpus positions
 * (-1 = padding); out_ranks must be zeroed by the caller.  S <= 4. */
int fz_tune_max_gold(void);
int fz_gold_ranks_f32(const float* const* T_h, const int32_t* pos, const float* weights, const int32_t* gold, int S, int W,
                      int Q, int N, int ld, int32_t* out_ranks, void* stream);
/* The same with np.float64 weights [W][S] -- what the reference's own grid holds (np.arange, hybrid.py:405-409): float64
 * products and sums (NumPy promotion of np.float32 * np.float64), ranks by the float64 fused score. */
int fz_gold_ranks_f64w(const float* const* T_h, const int32_t* pos, const double* weights, const int32_t* gold, int S, int W,
                       int Q, int N, int ld, int32_t* out_ranks, void* stream);
/* The metrics of run_evaluation (hybrid.py:24-42 over metrics.py:40-136) of every weight vector, from those ranks, on the device:
 * per query in float64 and in the reference's operation order -- recall@k, MAP@k, MRR@k, nDCG@k (its shifted discount,
 * metrics.py:108), R-precision -- and the mean over the Q queries as an exactly accumulated sum rounded once.
 * ranks [W][Q][G], gold [Q][G], pos [Q][ld] as above (G = fz_tune_max_gold(); a gold document with pos < 0 is in no list and never
 * counts); n_gold [Q] = len(ground_truths), the reference's divisor; idcg [Q] and disc [top+1] (disc[0] = 1, disc[i] = 1/log2(i+1))
 * come from the host, computed as the reference computes them; cuts = the n_recall + n_map + n_mrr + n_ndcg cut-offs in that
 * order.  out [W][M] float64, M = that count + 1 (R-precision last), M <= 24. */
int fz_tune_metrics_f64(const int32_t* ranks, const int32_t* gold, const int32_t* pos, int ld, const int32_t* n_gold, const double* idcg,
                        const double* disc, int top, const int32_t* cuts, int n_recall, int n_map, int n_mrr, int n_ndcg, int W, int Q,
                        double* out, void* stream);

/* ---- A3, sparse form: SPLADE cosine scoring over an inverted index ------------------------------------------------------------------
 * util.semantic_search(query_embeddings, corpus_embeddings, score_function=util.cos_sim) (hybrid.py:103) on SPLADE vectors
 * (splade/splade.py:88-99: a few hundred non-zeros of 32,005) without the dense [N, 32005] matrix: the L2-normalised corpus vectors as
 * postings -- toff [V + 1] (device int64), pdoc / pw [nnz]: (document, weight) of term t in [toff[t], toff[t+1]), documents ascending --
 * and the L2-normalised queries as qoff [Q + 1], qterms / qw: the non-zero terms of query q, ascending.
 *     scores[q][d] = sum over the query's terms t, ascending, of qw * pw        (float32, one rounding per product and per add)
 * = the dense product minus its exact zeros.  slice_off (nullable): fz_sparse_slice_offsets' [V][NS + 1] table, NS = ceil(N /
 * fz_sparse_slice_docs()) -- where every term's postings cross the document slices a workgroup owns.  scores [Q][lds] float32. */
int fz_sparse_slice_docs(void);
int fz_sparse_slice_offsets(const int64_t* toff, const int32_t* pdoc, int V, int N, int64_t* out, void* stream);
int fz_sparse_dot_f32(const int64_t* toff, const int32_t* pdoc, const float* pw, const int64_t* slice_off, const int64_t* qoff,
                      const int32_t* qterms, const float* qw, int Q, int N, float* scores, int lds, void* stream);

/* ---- encoder side: the per-sequence parts of SentenceTransformer.encode (hybrid.py:97-102) on PACKED token rows -- */
/* Self-attention of a BERT/CamemBERT layer for ragged sequences without padding: for every sequence and head,
 * out = softmax(q k^T * scale) v in fp32 (MFMA products, online softmax over 16-key tiles), scale > 0.  qkv [T][ld] = fused
 * projection rows (q | k | v, each H*head_dim wide); strips [n_strips][4] int32 (device): (first row of the sequence, its
 * length, first query row of this strip of <= 32 queries, 0) -- one entry per 32 queries of every sequence, the strips of a
 * sequence adjacent, sequences in any order (longest first balances best).  head_dim must be 64; sequence length <= 16384.
 * out [T][ldo], H*head_dim wide. */
int fz_attn_varlen_f32(const float* qkv, int ld, const int32_t* strips, int n_strips, int H, int head_dim, float scale,
                       float* out, int ldo, void* stream);
/* Inside the model forward of model.encode(...) (hybrid.py:101-102): every BERT sub-layer ends with
 * out = LayerNorm(x + res) * gamma + beta over the last dimension d (biased variance, as torch.nn.LayerNorm);
 * res nullable.  d % 4 == 0, d <= 4096. */
int fz_add_layernorm_f32(const float* x, int ldx, const float* res, int ldr, const float* gamma, const float* beta, float eps,
                         int rows, int d, float* out, int ldo, void* stream);
/* The FFN activation of the same forward (hybrid.py:101-102):
 * y = 0.5 x (1 + erf(x / sqrt 2)) elementwise (the exact "gelu" of BERT / CamemBERT, torch.nn.functional.gelu), count floats,
 * count % 4 == 0, 16-byte aligned; y may alias x. */
int fz_gelu_f32(const float* x, float* y, size_t count, void* stream);
/* The same three steps for a MIXED-PRECISION forward -- colbert-ai wraps every query() / doc() of the ColBERT encoder in torch.cuda.amp.autocast
 * (requirements.txt:15; the repository's own ColBERT runs set 'amp': True, multi_dense_biencoder.py:55, and wrap the model in amp.context(),
 * colbert_ir.py:110,124): the Linears take and return float16, these kernels compute in float32 and hand the next Linear its float16 operand.
 *   fz_attn_varlen_f16:       as fz_attn_varlen_f32 with float16 fused-QKV rows in and float16 context rows out (8-byte aligned, ld / ldo in
 *                             elements); scores, softmax and the weighted sum are float32;
 *   fz_attn_varlen_f16_amp:   the same rows in and out, but q k^T and p v as float16 matmuls with float32 accumulation (v_mfma_f32_16x16x32_f16 /
 *                             16x16x16_f16) around a float32 softmax -- the arithmetic autocast gives the attention of a BERT layer; ld and ldo
 *                             multiples of 8 elements, 16-byte aligned rows;
 *   fz_add_layernorm_x16:     x float16 (a Linear's output), res / out float32 (the residual stream), out16 nullable: a float16 copy of out;
 *   fz_gelu_f16:              float16 in and out (float32 arithmetic, as torch.nn.functional.gelu on a float16 tensor), count % 8 == 0. */
int fz_attn_varlen_f16(const void* qkv /* float16 */, int ld, const int32_t* strips, int n_strips, int H, int head_dim, float scale,
                       void* out /* float16 */, int ldo, void* stream);
int fz_attn_varlen_f16_amp(const void* qkv /* float16 */, int ld, const int32_t* strips, int n_strips, int H, int head_dim, float scale,
                           void* out /* float16 */, int ldo, void* stream);
int fz_add_layernorm_x16(const void* x /* float16 */, int ldx, const float* res, int ldr, const float* gamma, const float* beta, float eps,
                         int rows, int d, float* out, int ldo, void* out16 /* float16, nullable */, int ldo16, void* stream);
int fz_gelu_f16(const void* x /* float16 */, void* y /* float16 */, size_t count, void* stream);
/* The embedding block of the same forward (hybrid.py:101-102) on packed rows: out[t] = LayerNorm(word[ids[t]] + pos[pos_ids[t]] + type0) * gamma + beta.
 * word [V][d], pos [Pmax][d], type0 [d] (token type 0 everywhere); ids / pos_ids [rows] int64 (device), the caller guarantees they
 * index inside the tables.  d % 4 == 0, d <= 4096. */
int fz_embed_layernorm_f32(const float* word, const float* pos, const float* type0, const int64_t* ids, const int64_t* pos_ids,
                           const float* gamma, const float* beta, float eps, int rows, int d, float* out, int ldo, void* stream);
/* mean Pooling (sentence_transformers Pooling(mean)): out[b] = mean of rows [cu_rows[b], cu_rows[b+1]) of x; zeros for an
 * empty sequence.  cu_rows [B+1] int32 (device). */
int fz_segment_mean_f32(const float* x, int ldx, const int32_t* cu_rows, int B, int d, float* out, int ldo, void* stream);
/* SPLADE-max pooling (splade/splade.py:88-99: amax over the tokens of log1p(relu(logits))): out[b] = log1p(relu(max of the
 * rows [cu_rows[b], cu_rows[b+1]) of x)) -- the same value, log1p o relu being monotone; 0 for an empty sequence. */
int fz_segment_splade_max_f32(const float* x, int ldx, const int32_t* cu_rows, int B, int d, float* out, int ldo, void* stream);

/* The SPLADE head with the pooling as its epilogue (splade/splade.py:88-99 over the MLM decoder): pool[s][v] = max over the packed rows
 * t in [cu_rows[s], cu_rows[s+1]) of log1p(relu(<X[t], W[v]> + bias[v])) -- the same value as fz_segment_splade_max_f32 over the decoder's
 * logits, but the [T][V] logits (4.2 GB per 64 x 512-token batch, splade.py:94) are never written: the fp32-MFMA tile stream of
 * fz_dot_scores_f32 with a segment-max / atomicMax epilogue.  X [T][ldx] packed hidden rows (after the head's dense + GELU + LayerNorm), W [V][ldw]
 * the decoder weight, bias [V]; pool [nseq][ldp] MUST be zero on entry (0 = log1p(relu(x)) of every x <= 0 and of an empty sequence).
 * d % 4 == 0, 16-byte aligned rows. */
int fz_splade_head_max_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const int32_t* cu_rows, int nseq, int T, int V,
                           int d, float* pool, int ldp, void* stream);

/* ---- small utilities ------------------------------------------------------------------ */
int fz_fill_i32(int32_t* p, size_t count, int32_t value, void* stream);
int fz_f64_to_f32(const double* src, float* dst, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FUSION_HIP_H */
