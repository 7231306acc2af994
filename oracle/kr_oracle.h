/*
 * kr_oracle.h — C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the per-read query
 * path of krepp v0.8.3 (`krepp dist`), written from the behaviour of the
 * reference sources; every function in kr_oracle.cpp cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (krepp_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - likelihood f(d): pinned bit-for-bit against the reference's own
 *     src/hdhistllh.hpp compiled into oracle/_ref/ (tests/test_oracle_ref.py)
 *   - FASTA/FASTQ record parsing: pinned against reference src/kseq.h
 *   - node-name hashing (MurmurHash3_x86_32): pinned against reference
 *     src/MurmurHash3.cpp
 *   - Newick post-order numbering: pinned against README.md:100-118 edge numbers
 *   - integer front end (encodings, LSH, residual encoding): two independent
 *     formulations (mask/PEXT as the reference builds them, and the closed
 *     form over position lists) checked against each other; the reference's
 *     lshf.cpp/common.hpp need parallel-hashmap (absent) so cannot be built.
 *   - Brent minimiser: PARITY UNPINNED — Boost.Math is an un-vendored
 *     submodule (external/boost, no pinned commit); restated from the
 *     published algorithm (boost/math/tools/minima.hpp).
 */
#ifndef KR_ORACLE_H
#define KR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ko_index ko_index;

typedef struct ko_info {
  uint32_t k, w, h, m;
  uint32_t nlibs;        /* partial libraries loaded                        */
  uint32_t nresidues;    /* residues of rix % m that are served             */
  uint32_t nnodes;       /* tree nodes (se = 1..nnodes)                     */
  uint32_t nleaves;
  uint64_t nkmers;       /* total cmer entries over all libraries           */
  uint64_t nrows;        /* total rows over all libraries                   */
  uint32_t wbackbone;    /* 1 if the index carries a tree-* file            */
  uint32_t pad;
} ko_info;

/* dist parameters: src/krepp.hpp:206-221 defaults, src/krepp.cpp:632-675 */
typedef struct ko_params {
  uint32_t hdist_th;     /* --hdist-th, default 4                            */
  uint32_t tau;          /* place only, default 2                            */
  double chisq;          /* --chisq, default 2.706                           */
  double dist_max;       /* --dist-max, NaN = unset                          */
  uint32_t multi;        /* default 1                                        */
  uint32_t no_filter;    /* default 1 for dist                               */
  uint32_t num_threads;  /* OpenMP threads for ko_dist_batch                 */
  uint32_t collect;      /* bit0: accumulators, bit1: hits, bit2: text, bit3: rows are counted but not returned (timing) */
} ko_params;

/* One (read, strand, leaf) accumulator = the reference's Minfo (src/query.hpp:100-228). */
typedef struct ko_acc {
  uint32_t read;
  uint32_t se;           /* leaf colour id                                   */
  uint32_t strand;       /* 0 = as given, 1 = reverse complement             */
  uint32_t match_count;
  uint32_t hdist_min;
  uint32_t passed;       /* hdist_min <= 2*hdist_filt+1  (src/query.cpp:106) */
  double rho;
  double d_llh;          /* DBL_MAX if not optimised                         */
  double v_llh;          /* NaN if not optimised                             */
  uint32_t hist[17];     /* hist[0..hdist_th]                                */
  uint32_t pad;
} ko_acc;

/* One output row of `krepp dist` (src/query.cpp:158-196). */
typedef struct ko_row {
  uint32_t read;
  uint32_t se;           /* 0 = "NA\tNaN" row                                */
  uint32_t strand;       /* strand whose Minfo was reported                  */
  uint32_t match_count;
  double d_llh;
  double v_llh;
  double chisq;          /* NaN unless filter mode                           */
} ko_row;

typedef struct ko_hit {
  uint32_t read;
  uint32_t strand;
  uint32_t pos;          /* reference pos: i-k (fwd), len-i (rc)             */
  uint32_t kpos;         /* k-mer start index i-k for both strands           */
  uint32_t lib;
  uint32_t hd;
  uint64_t cmer_index;   /* index into that library's cmer array             */
  uint32_t enc;
  uint32_t se;
} ko_hit;

typedef struct ko_readinfo {
  uint32_t onmers;       /* valid k-mer positions (src/query.cpp:66)          */
  uint32_t hdist_filt[2];/* raw per-strand min hd, 0xFFFFFFFF = none          */
  uint32_t nrows;
} ko_readinfo;

/* Exact algorithmic-traffic counters (SURVEY.md §8d formula). */
typedef struct ko_counters {
  uint64_t reads, bases, kmers_valid, lsh_evals, probes, bucket_entries;
  uint64_t hits, pse_reads, rho_reads, accs, brent_runs, llh_evals, rows;
} ko_counters;

/* One placement of `krepp place` (PP_JPLACE_FIELDS, src/query.hpp:202-204). */
typedef struct ko_placement {
  uint32_t read;
  uint32_t edge;         /* Node::get_en = se - 1 of the placement tree           */
  double lwr, d_llh, v_llh, pendant, distal;
} ko_placement;

typedef struct ko_result {
  uint64_t nrows, naccs, nhits;
  ko_row* rows;
  ko_acc* accs;
  ko_hit* hits;
  ko_readinfo* reads;    /* nreads entries                                    */
  ko_placement* placements; /* place mode                                       */
  uint64_t nplacements;
  char* text;            /* report text (collect bit2)                        */
  uint64_t text_len;
  ko_counters counters;
} ko_result;

ko_index* ko_index_load(const char* dir, char* err, int errlen);
void ko_index_free(ko_index*);
int ko_index_replace_table(ko_index*, uint32_t lib, const uint64_t* inc, uint32_t nrows, const uint32_t* cmer,
                           uint64_t nkmers);
void ko_index_info(const ko_index*, ko_info* out);
/* name as printed by the reference (Node::get_name, src/phytree.hpp:134-145) */
const char* ko_node_name(const ko_index*, uint32_t se);
/* 0 = null, 1 = leaf, 2 = internal */
int ko_node_kind(const ko_index*, uint32_t se);
uint32_t ko_node_parent(const ko_index*, uint32_t se);
double ko_node_blen(const ko_index*, uint32_t se);
void ko_lsh_positions(const ko_index*, uint8_t* ppos, uint8_t* npos);

/* Front end for one sequence: every valid k-mer x strand, in reference order
 * (src/query.cpp:40-94).  Arrays must hold 2*len entries.  Returns count. */
uint32_t ko_front_end(const ko_index*, const char* seq, uint64_t len,
                      uint32_t* kpos, uint8_t* strand, uint64_t* enc_bp, uint64_t* enc_lr,
                      uint32_t* rix, uint32_t* enc32, uint8_t* pass);

int ko_dist_batch(const ko_index*, const char* bases, const uint64_t* offsets,
                  const char* const* names, uint32_t nreads, const ko_params* p, ko_result* out);
int ko_dist_summarize(const ko_index*, const char* bases, const uint64_t* offsets, uint32_t nreads, const ko_params* p,
                      ko_result* out); /* out->text = REFERENCE_NAME\tWEIGHTED_COUNT\tSEQUENCE_ABUNDANCE rows */
void ko_result_free(ko_result*);

/* `krepp place` (src/krepp.cpp:434-504, src/query.cpp:198-333).  ko_index_set_placement_tree:
 * nwk == NULL uses the index's own backbone (TargetIndex::ensure_backbone, src/krepp.cpp:48-64),
 * otherwise the index leaves are mapped onto the given tree (Tree::map_to_qtree,
 * src/phytree.cpp:421-450).  ko_index_set_lineage_tree builds the placement tree from a lineage file
 * instead (-l; TargetIndex::read_lineages src/krepp.cpp:37-46, Tree::parse_lineages
 * src/phytree.cpp:320-369).  The text returned by
 * ko_place_batch is the concatenation of the reference's per-batch streams (512-read batches)
 * joined as QueryIndex::place_sequences joins them; ko_place_frame gives header / footer. */
int ko_index_set_placement_tree(ko_index*, const char* nwk_text, char* err, int errlen);
int ko_index_set_lineage_tree(ko_index*, const char* lineage_text, char* err, int errlen);
/* place --summarize: out->text = DISTAL_NODE\tEDGE_NUM\tWEIGHTED_COUNT\tSEQUENCE_ABUNDANCE rows;
 * ko_place_frame(which = 0, tabular = 2) gives its header */
int ko_place_summarize(const ko_index*, const char* bases, const uint64_t* offsets, uint32_t nreads, const ko_params* p,
                       ko_result* out);
int ko_place_batch(const ko_index*, const char* bases, const uint64_t* offsets, const char* const* names,
                   uint32_t nreads, const ko_params* p, int tabular, ko_result* out);
/* which: 0 = text before the batches, 1 = text after them; caller frees with free() */
char* ko_place_frame(const ko_index*, int which, int tabular, const char* invocation, uint64_t total_qseq);

/* `krepp seek` on a single-reference sketch file (src/seek.cpp, src/sketch.cpp; file format
 * SFlatHT::save src/table.cpp:34-40 + BaseLSH::save_configuration src/krepp.cpp:18-29 + rho).
 * ko_seek_batch: out->text = `SEQ_ID\tDIST` / `SEQ_ID\tNaN` rows, out->rows one per read. */
typedef struct ko_sketch ko_sketch;
ko_sketch* ko_sketch_load(const char* path, char* err, int errlen);
void ko_sketch_free(ko_sketch*);
void ko_sketch_info(const ko_sketch*, uint64_t* nkmers, uint32_t* nrows, uint32_t* k, uint32_t* w, uint32_t* h, uint32_t* m,
                    uint32_t* r, uint32_t* frac, double* rho /* scaled as make_rho_partial */);
const uint32_t* ko_sketch_codes(const ko_sketch*);
const uint64_t* ko_sketch_inc(const ko_sketch*);
void ko_sketch_positions(const ko_sketch*, uint8_t* ppos, uint8_t* npos);
int ko_seek_batch(const ko_sketch*, const char* bases, const uint64_t* offsets, const char* const* names, uint32_t nreads,
                  uint32_t hdist_th, ko_result* out);

/* Likelihood and minimiser, callable on their own
 * (src/hdhistllh.hpp:71-89, src/query.cpp:426-433). */
double ko_llh(uint32_t k, uint32_t h, uint32_t th, const double* hist, double uc, double rho, double d);
int ko_brent(uint32_t k, uint32_t h, uint32_t th, const double* hist, double uc, double rho,
             double* d_out, double* v_out);

/* small primitives for unit tests */
uint64_t ko_revcomp_bp64(uint64_t x, uint32_t k);
uint64_t ko_conv_bp64_lr64(uint64_t x);
uint32_t ko_murmur3_x86_32(const void* key, int len, uint32_t seed);
uint64_t ko_name_hash(const char* name);
uint64_t ko_xur64(uint64_t h);

#ifdef __cplusplus
}
#endif
#endif
