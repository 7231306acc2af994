/*
 * krepp_amd.h — C-ABI of the MI355X-native `krepp dist` query path.
 *
 * The reference (bo1929/krepp v0.8.3) has no FFI/plugin layer; its drop-in
 * surfaces are the CLI, the on-disk index, and one internal seam:
 *
 *     IBatch(index, qs, hdist_th, chisq, dist_max, tau, no_filter, multi, summarize)
 *     IBatch::estimate_distances(std::stringstream&)          src/query.hpp:49-63
 *
 * constructed per 512-read batch on the reader thread and run inside an OpenMP
 * task (src/krepp.cpp:365-383).  This header is that seam as a C ABI: plain
 * pointers and sizes, no C++ or torch types.  Each entry point names the
 * reference interface it replaces.  Every function returns 0 on success or a
 * negative kr_status; kr_last_error() returns a per-thread message.
 *
 * Ownership: the caller owns every host buffer it passes in; the library owns
 * all device memory and every buffer it returns (valid until the owning handle
 * is destroyed or, for results, until the next submit on the same stream).
 * A kr_index is immutable after upload and may be shared by any number of
 * kr_stream objects; a kr_stream is used by one host thread at a time.
 *
 * There is NO CPU fallback: without a usable HIP device every device entry
 * point returns KR_ERR_NO_DEVICE.
 */
#ifndef KREPP_AMD_H
#define KREPP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KR_API __attribute__((visibility("default")))

typedef enum kr_status {
  KR_OK = 0,
  KR_ERR_ARG = -1,        /* bad argument / unsupported configuration          */
  KR_ERR_IO = -2,         /* index directory / file problems ([ERROR] paths of  */
                          /* src/index.cpp:51-158, src/krepp.cpp:66-108)        */
  KR_ERR_FORMAT = -3,     /* inconsistent index files, incompatible libraries  */
  KR_ERR_NO_DEVICE = -4,  /* no HIP device / HIP runtime failure               */
  KR_ERR_NOMEM = -5,
  KR_ERR_CAPACITY = -6,   /* a device-side buffer overflowed; resubmit smaller  */
  KR_ERR_STATE = -7,      /* call order (collect without submit, ...)           */
  KR_ERR_UNSUPPORTED = -8 /* this batch cannot be served this way; use the general call (kr_batch_collect_text) */
} kr_status;

#define KR_MAX_HDIST_TH 16u /* k-h <= 16 (src/krepp.hpp:77-79): hd never exceeds 16 */

/* ------------------------------------------------------------------------- */
/* Host-side index: replaces TargetIndex::load_index (src/krepp.cpp:66-108),   */
/* Index::load_partial_index / load_partial_tree / generate_partial_tree /     */
/* make_rho_partial (src/index.cpp:3-158,188-201), FlatHT::load                */
/* (src/table.cpp:65-75), CRecord::load (src/record.cpp:203-211), Tree::load   */
/* and Node::parse / generate_tree (src/phytree.cpp:150-253,394-404).          */
/* ------------------------------------------------------------------------- */
typedef struct kr_host_index kr_host_index;

/* One partial library as laid out on disk (little-endian, no padding). */
typedef struct kr_lib_view {
  const uint64_t* inc;   /* inc-*    payload: cumulative bucket END offsets [nrows]  */
  const uint32_t* cmer;  /* cmer-*   payload: interleaved (enc32, se) [2*nkmers]     */
  const uint32_t* pse;   /* crecord-* se_to_pse: interleaved (first, second) [2*nsubsets] */
  const double* rho;     /* crecord-* se_to_rho [nnodes], already scaled by            */
                         /* make_rho_partial's (#residues present)/m                    */
  uint64_t nkmers;
  uint32_t nrows;
  uint32_t nsubsets;
  uint32_t nnodes;       /* crecord's nnodes = tree nodes + 1                           */
  uint32_t r;            /* metadata-* r                                                */
  uint32_t frac;         /* metadata-* frac                                             */
  uint32_t w;            /* metadata-* w (informational)                                */
} kr_lib_view;

typedef struct kr_index_view {
  uint32_t k, h, m;
  const uint8_t* ppos;       /* [h]   LSH positions, descending (metadata-*)           */
  const uint8_t* npos;       /* [k-h] non-LSH positions, ascending                     */
  uint32_t nlibs;
  const kr_lib_view* libs;
  uint32_t tree_nnodes;      /* Tree::nnodes; colour ids 1..tree_nnodes are tree nodes */
  const uint8_t* node_kind;  /* [tree_nnodes+1] 0 = null (Tree::get_node == nullptr),  */
                             /* 1 = leaf, 2 = internal (src/query.cpp:371-381)         */
  uint32_t wbackbone;        /* Index::check_wbackbone                                 */
} kr_index_view;

KR_API int kr_host_index_load(const char* index_dir, kr_host_index** out);
/* `krepp seek` (src/sketch.cpp:3-39): a single-reference sketch file presented as an index with one library
 * and a one-leaf tree, so that kr_index_upload / kr_stream_* / kr_batch_* serve it unchanged. */
KR_API int kr_host_sketch_load(const char* sketch_path, kr_host_index** out);
KR_API void kr_host_index_free(kr_host_index*);
KR_API int kr_host_index_view(const kr_host_index*, kr_index_view* out);
/* Node::get_name (src/phytree.hpp:134-145): label, or se-1 for unlabelled nodes. */
KR_API const char* kr_host_index_node_name(const kr_host_index*, uint32_t se);
/* the label as written in the Newick / reflist ("" for unlabelled nodes) */
KR_API const char* kr_host_index_node_label(const kr_host_index*, uint32_t se);
KR_API uint32_t kr_host_index_node_parent(const kr_host_index*, uint32_t se);
KR_API double kr_host_index_node_blen(const kr_host_index*, uint32_t se);

/* ------------------------------------------------------------------------- */
/* Device index: the read-only state IBatch borrows from Index                 */
/* (Index::check_partial src/index.hpp:27, Index::bucket_indices               */
/* src/index.cpp:160-168, Index::get_crecord :170, LSHF::compute_hash /        */
/* drop_ppos_lr src/lshf.cpp:62-69).                                           */
/* ------------------------------------------------------------------------- */
typedef struct kr_index kr_index;

#define KR_VIEW_HOST 0u
#define KR_VIEW_DEVICE 1u /* the view's inc/cmer/pse/rho pointers are device pointers */

/* Re-lays the on-disk arrays out for the GPU (DESIGN.md "HBM layout") in HBM of
 * `device`.  With KR_VIEW_DEVICE the big arrays are read from device memory
 * (ppos/npos/node_kind and the kr_lib_view structs themselves stay host). */
KR_API int kr_index_upload(const kr_index_view* view, int device, uint32_t flags, kr_index** out);
KR_API void kr_index_free(kr_index*);

/* Flat device buffers of an uploaded index, for replication to other GPUs by
 * whatever transport the host uses (RCCL broadcast from torch.distributed,
 * hipMemcpyPeer, ...).  `desc` is a small host blob describing sizes and
 * parameters; kr_index_import allocates the same buffers on another device and
 * returns their addresses so the transport can fill them. */
typedef struct kr_index_buffer {
  void* dptr;
  uint64_t bytes;
} kr_index_buffer;
KR_API int kr_index_export(const kr_index*, void* desc, uint64_t* desc_bytes, kr_index_buffer* bufs, uint32_t* nbufs);
KR_API int kr_index_import(const void* desc, uint64_t desc_bytes, int device, kr_index** out, kr_index_buffer* bufs,
                           uint32_t* nbufs);
KR_API uint64_t kr_index_device_bytes(const kr_index*);
/* Words (4 bytes) per slot of the slotted copy of the bucket heads the upload made for this index, 0 when it kept the packed
 * table only (sparse tables; INTEGRATION.md "Memory and tuning knobs").  Says which scan kernel serves the index:
 * kr_scan_pipe_kernel_t (slotted) or kr_scan_kernel_t (packed). */
KR_API uint32_t kr_index_slot_words(const kr_index*);
/* The slot format itself: 0 packed table only; 5 / 6 / 7 / 8 slots of 32 / 64 / 128 / 48 four-byte words (kr_scan_pipe_kernel_t);
 * 9 FILTER slots -- two 128-byte lines of 24-bit codes per row, one line per probe for most buckets, candidates verified by
 * the accumulate kernel (kr_scan_filt_kernel_t; replaces the linear bucket scan of src/query.cpp:361-368). */
KR_API uint32_t kr_index_slot_format(const kr_index*);

/* Replicate an uploaded index into the HBM of `ndev` more devices of this node by RCCL broadcast
 * (ncclBroadcast per flat buffer, one communicator over the root's device and the targets, xGMI):
 * the index crosses PCIe once, whatever the number of GPUs.  Load time only; there is no collective
 * on the query path.  Replaces what the reference does once per process, TargetIndex::load_index
 * (src/krepp.cpp:92-106), for the GPUs after the first.  replicas[i] lives on devices[i] and is freed
 * with kr_index_free; devices are distinct; a device equal to the root's gives a second copy in the
 * same HBM (one-rank communicator).  RCCL (librccl.so.1) is loaded on first use.  RCCL may print a version
 * banner on stdout when the first communicator is made: the library never touches the process's file
 * descriptors, so an application whose stdout carries the report redirects it around this call itself,
 * before its writer threads start. */
KR_API int kr_index_broadcast(const kr_index* root, int ndev, const int* devices, kr_index** replicas);

/* ------------------------------------------------------------------------- */
/* Query: IBatch ctor + IBatch::estimate_distances (src/query.cpp:8-38,        */
/* 141-156): search_mers (:40-94), IMers::add_matching_mer (:352-390),         */
/* Minfo::update_match (src/query.hpp:153-176), summarize_matches              */
/* (src/query.cpp:96-139), Minfo::optimize_likelihood (:426-433) with          */
/* HDistHistLLH (src/hdhistllh.hpp:51-96) and Boost's brent_find_minima, and   */
/* the selection logic of report_distances (:158-196).                         */
/* ------------------------------------------------------------------------- */
typedef struct kr_params {
  uint32_t hdist_th;   /* --hdist-th  [4]      (src/krepp.hpp:210)                */
  uint32_t tau;        /* --tau       [2]      (place only; unused by dist)       */
  double chisq;        /* --chisq     [2.706]                                     */
  double dist_max;     /* --dist-max  [NaN = unset]                               */
  uint32_t multi;      /* --multi     [1]                                         */
  uint32_t no_filter;  /* !--filter   [1 for dist] (src/krepp.cpp:635-644)        */
} kr_params;

KR_API void kr_params_default(kr_params*);

typedef struct kr_stream kr_stream;

/* max_reads / max_bases bound one submitted batch; device buffers are sized once.
 * max_records bounds the (read, strand, leaf) accumulators one batch may emit
 * (0 = default: max_reads * min(16, max(8, tree nodes + 1))); a batch that needs more
 * fails with KR_ERR_CAPACITY and can be resubmitted in smaller pieces. */
/* Kernels of all streams created on one kr_index run one batch after the other in submission order (a per-index
 * event chain); copies to and from the host overlap them.  Use two streams in turn to keep the GPU busy. */
KR_API int kr_stream_create(const kr_index*, const kr_params*, uint32_t max_reads, uint64_t max_bases,
                            uint64_t max_records, kr_stream** out);
KR_API void kr_stream_destroy(kr_stream*);

#define KR_BASES_HOST 0u
#define KR_BASES_DEVICE 1u /* bases/offsets already resident in this device's HBM */
#define KR_TAP_ACCS 2u     /* keep every (read,strand,leaf) histogram: rec_hist of the views, `place` */
#define KR_TAP_HITS 4u     /* record every table hit (debug; slow; the batch runs as one lane) */
#define KR_BASES_PINNED 8u /* host `bases` AND `offsets` are page-locked (hipHostMalloc / hipHostRegister, */
                           /* e.g. kr_host_alloc): copied to the device straight from the caller's       */
                           /* buffers, which must stay valid until the batch has been waited for         */
#define KR_ROWS_ONLY 16u   /* kr_batch_collect brings back only what the `dist` report needs         */
                           /* (read_off/cnt/na, rec_key/sel/d); rec_v, rec_chisq, rec_hist and        */
                           /* read_onmers of the host view are then NULL / undefined, and the kernels */
                           /* do not write rec_v at all (NULL in a device view too) unless the batch  */
                           /* filters (--filter) or taps.  The host view then holds the OUTPUT ROWS   */
                           /* only -- the records report_distances prints (src/query.cpp:158-196),    */
                           /* compacted on the device, 12 bytes a row across PCIe: nrecs == nrows,    */
                           /* every rec_sel is 1, read_off / read_cnt are the reads' row ranges (a    */
                           /* tiled batch still comes back as record slots with flags: same fields,   */
                           /* same meaning, rec_sel 0 where a record is not a row)                    */
#define KR_ROWS_INDEXED 32u /* with KR_ROWS_ONLY: DIST leaves the device as an INDEX -- the likelihood    */
                           /* stage solves every distinct problem of a batch once (one in ~20 records), */
                           /* so a row is (rec_key, rec_dix), 8 bytes across PCIe instead of 12, and    */
                           /* the batch's distinct values come once: DIST of row i is                   */
                           /* dist_list[rec_dix[i]] (bit for bit what rec_d would hold); rec_d of the    */
                           /* host view is NULL then (a formatter may format each distinct value once). */
                           /* A hint: honoured for batches that run as one lane and are not tiled --     */
                           /* otherwise the view holds rec_d as without it (rec_dix == NULL says which). */

/* Queue one batch: `bases` = concatenated ASCII sequences exactly as the FASTX
 * reader delivers them (QSeq::read_next_batch, src/rqseq.cpp:180-197),
 * offsets[nreads+1] = start of each read.  Asynchronous: host buffers must stay
 * valid until kr_batch_collect / kr_batch_wait returns.
 *
 * This replaces the reference's batching (src/rqseq.cpp:180-197 + the task loop src/krepp.cpp:360-387) by
 * pinned-host staging + hipMemcpyAsync: a HOST batch of >= 2 * 65,536 reads is cut into up to KR_LANES (env, default 2)
 * contiguous read ranges, each with its own HIP stream: the H2D copy, the kernels and (in kr_batch_collect) the
 * D2H copy of one range overlap with those of the other.  Results do not depend on the number of lanes.  Across
 * batches, two kr_streams used in turn overlap one batch's copies with the other's kernels (bench.py, the CLI).
 *
 * Long sequences (contigs, genomes): the reference scans a sequence serially (search_mers, src/query.cpp:40-94).  A HOST
 * batch that holds sequences of more than 1,024 k-mer positions is submitted as tiles of 128 positions (neighbours overlap by
 * k - 1 bases), which run on as many waves as there are tiles; the tiles' histograms are added per (reference, strand) and the
 * hdist_filt test is applied with the sequence's minimum, so the results are those of the serial scan (identical records, in
 * the same order).  A tile counts as a read against max_reads: the sequences whose tiles fit the stream are tiled, in order
 * (the others, and every sequence of a batch already in HBM or submitted with KR_TAP_HITS, are one wave's work as before); a
 * tiled batch whose tiles' records overflow the device buffers is run again untiled by kr_batch_wait / kr_batch_collect.  The
 * views always describe the caller's reads. */
KR_API int kr_batch_submit(kr_stream*, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads,
                           uint32_t flags);
KR_API int kr_batch_wait(kr_stream*);

/* One candidate row = one (read, leaf): what report_distances iterates over. */
typedef struct kr_result_view {
  uint32_t nreads;
  uint32_t nrecs;             /* (read, strand, leaf) accumulators that passed the       */
                              /* hdist_filt test (src/query.cpp:106,119)                  */
  const uint32_t* read_off;   /* [nreads] first record of the read                       */
  const uint32_t* read_cnt;   /* [nreads] number of records of the read                  */
  const uint32_t* read_onmers;/* [nreads] valid k-mer positions (src/query.cpp:66)       */
  const uint8_t* read_na;     /* [nreads] 1 = the read gets the "NA\tNaN" row             */
                              /* (src/query.cpp:173-176)                                 */
  const uint32_t* rec_key;    /* [nrecs]  (se << 1) | strand; ascending inside a read    */
  const uint8_t* rec_sel;     /* [nrecs]  1 = this record is an output row of `dist`     */
  const double* rec_d;        /* [nrecs]  d_llh                                          */
  const double* rec_v;        /* [nrecs]  v_llh                                          */
  const double* rec_chisq;    /* [nrecs]  chi-square vs the closest (filter mode), else NaN (NULL in a device view) */
  const uint32_t* rec_hist;   /* with KR_TAP_ACCS, else NULL: hist[x] of record i at      */
                              /* rec_hist[x * rec_hist_stride + i], x = 0..hdist_th       */
  uint64_t rec_hist_stride;
  uint64_t nrows;             /* number of rec_sel == 1                                  */
  const uint32_t* rec_dix;    /* KR_ROWS_INDEXED (host view): [nrecs] index into dist_list, rec_d == NULL; else NULL */
  const double* dist_list;    /* [ndist] distinct d_llh values of the batch (unused positions hold anything)         */
  uint64_t ndist;
} kr_result_view;

/* Waits for the batch and copies results to pinned host memory owned by the stream (lane by lane: a lane's
 * results travel while later lanes still compute).  Host view: records are compact, a read's records are
 * [read_off, read_off + read_cnt) in ascending key order; the order of the reads' record groups in the arrays
 * is not the read order (record slots are handed out to waves in chunks) and may differ from run to run. */
KR_API int kr_batch_collect(kr_stream*, kr_result_view* out);
/* As above but the arrays stay in HBM (device pointers); only counts are read back.  Device view: `nrecs` is
 * the extent of record slots handed out, unused slots (rec_key == 0) included; always go through read_off / read_cnt. */
KR_API int kr_batch_collect_device(kr_stream*, kr_result_view* out);
/* Bytes the last kr_batch_collect of a rows-only batch copied back over PCIe (measurement aid). */
KR_API int kr_debug_last_d2h_bytes(kr_stream*, uint64_t* bytes);
/* Tests: one number as `krepp place` rows print it (std::fixed, 5 decimals; `out` holds 80 bytes); returns its length. */
KR_API int kr_debug_place_fixed5(double v, char* out);

/* Debug taps (parity tests).  Hits: one entry per table entry with hd <= hdist_th. */
typedef struct kr_hit {
  uint32_t read;
  uint32_t kpos;    /* k-mer start index in the read                               */
  uint32_t strand;
  uint32_t lib;
  uint64_t cmer_index;
  uint32_t hd;
  uint32_t se;
} kr_hit;
KR_API int kr_batch_hits(kr_stream*, const kr_hit** hits, uint64_t* nhits);
typedef struct kr_readtap {
  uint32_t hdist_filt[2]; /* per-strand min hd over kept entries, 0xFFFFFFFF = none */
} kr_readtap;
KR_API int kr_batch_readtaps(kr_stream*, const kr_readtap** taps);

/* Experiments (the scan kernel's launch-time levels): give one group of a stream's device buffers a new
 * address -- 0: item list, 1: per-read arrays, 2: counters and cursors, 3: records and de-duplication table -- or its kernels a
 * new HIP stream (4).  Between batches only; results are unaffected. */
KR_API int kr_debug_stream_move(kr_stream*, int which);
/* ... and where they are: item list, counters, cursors, rd_off, rd_it_off, rd_filt, rec_key, de-duplication table. */
KR_API int kr_debug_stream_addrs(kr_stream*, uint64_t* out8);

/* Item-list placement: a stream that is given batches of a million reads or more allocates its item list
 * (the scan kernel's output) up to KR_ITEM_PLACEMENT_TRIALS more times (environment, default 3, 0 = never) during its first
 * batches -- one spare list at a time -- and keeps the allocation the scan kernel ran fastest on; the kernel's launch time depends
 * on the physical pages behind the list (40.5 against 46.5 ms per 8 M reads), which only a new allocation changes.  Results are
 * unaffected.  This call reports what the stream did: allocations tried, how many of them replaced the one before, and the scan
 * time per read on the list it kept. */
KR_API int kr_debug_item_placement(kr_stream*, uint32_t* tried, uint32_t* kept, double* best_ns_per_read);

/* Front-end tap: rix / enc32 / residue test for every (k-mer, strand) of one batch,
 * laid out [read][kpos][strand] with `stride` = max k-mers per read; valid==0 marks
 * positions whose window holds a non-ACGT byte or runs past the read. */
KR_API int kr_debug_front_end(const kr_index*, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads,
                              uint32_t stride, uint32_t* rix, uint32_t* enc32, uint8_t* valid, uint8_t* pass);

/* The host pass of kr_index_upload over a library's colour table (no device needed): the tagged form of every colour id
 * (class in the top two bits: 0 drop, 1 leaf rank, 2 walk through se_to_pse, 3 flat), the se_to_pse table as the device holds it
 * ({part, part} tagged, or for a flat colour {base | list flag << 31, count}) and the leaf lists.  node_kind[se], se <= tree_nnodes:
 * 0 null, 1 leaf, 2 internal (as in kr_index_view).  *nlist: in = capacity of lists_out in entries, out = entries needed.
 * Replaces the per-hit walk of src/query.cpp:369-387 for the colours whose leaves can be named at upload. */
KR_API int kr_debug_colour_classes(const uint32_t* pse_pairs /*[2*nsubsets]*/, uint32_t nsubsets, const uint8_t* node_kind, uint32_t tree_nnodes,
                                   uint32_t* cls_out /*[nsubsets]*/, uint32_t* pse_out /*[2*nsubsets]*/, uint32_t* lists_out, uint64_t* nlist);

/* Likelihood + minimiser on their own (device): one problem per element. */
KR_API int kr_debug_brent(const kr_index*, uint32_t hdist_th, uint32_t n, const uint32_t* hist /*[n*(th+1)]*/,
                          const uint32_t* onmers, const double* rho, double* d_out, double* v_out);

/* Likelihood kernel on arbitrary problems (the per-edge likelihoods of `krepp place`, whose
 * histograms are fractional: Minfo::add, src/query.hpp:139-152).  One problem per element:
 * hist[n*(th+1)] (doubles), uc[n] = mismatch_count, rho[n].  mode 0: Brent minimisation
 * (Minfo::optimize_likelihood, src/query.cpp:426-433) -> d_out, v_out.  mode 1: evaluate f at
 * d_in[n] -> v_out (Minfo::likelihood_ratio's f(d), src/query.cpp:420-424). */
KR_API int kr_llh_batch(const kr_index*, uint32_t hdist_th, uint32_t mode, uint64_t n, const double* hist, const double* uc,
                        const double* rho, const double* d_in, double* d_out, double* v_out);
/* f_{problem pidx[i]}(d_in[i]), i < n: many evaluations of few problems (hist [nprob * (th + 1)], uc, rho [nprob]) --
 * the chi-square tests of `place` (src/query.cpp:276, :420-424): every candidate of a read against its closest leaf. */
KR_API int kr_llh_eval_indexed(const kr_index*, uint32_t th, uint64_t nprob, const double* hist, const double* uc, const double* rho,
                               uint64_t n, const uint32_t* pidx, const double* d_in, double* v_out);

/* Kernel timing of the last collected batch (HIP events on the stream's own stream). */
typedef struct kr_timing {
  float ms_total;    /* first kernel start -> last kernel end                        */
  float ms_scan;     /* kr_scan_kernel: front end + table scan (the dominant kernel) */
  float ms_acc;      /* kr_acc_kernel: colour expansion + histograms + records       */
  float ms_llh;      /* likelihood + selection kernels                               */
  float ms_h2d;      /* host->device copies (0 with KR_BASES_DEVICE)                 */
  uint32_t overflow_reads; /* reads that took the global-memory accumulator path     */
  uint32_t stack_spills;   /* times a colour-expansion stack outgrew the LDS (large clades) */
  uint32_t lanes;          /* read ranges the batch was cut into: with one, ms_scan/acc/llh are the kernels' own  */
                           /* durations; with several they are sums over lanes that share the chip (> ms_total)   */
} kr_timing;
KR_API int kr_batch_timing(kr_stream*, kr_timing* out);

/* ------------------------------------------------------------------------- */
/* Host helpers mirroring the reference's reader and writer.                    */
/* ------------------------------------------------------------------------- */
/* QSeq (src/rqseq.cpp:146-203): gz/FASTA/FASTQ batches of >= 76,800 bases.  Plain (uncompressed)   */
/* regular files of >= 32 MB are cut at record starts and parsed by a thread pool (KR_FASTX_THREADS,   */
/* default min(8, cores/2); 0 = off): a batch is then one chunk of about 2 * min_bases bytes of input. */
/* Block-gzipped files (BGZF: independent members with a "BC" extra field) are inflated member by       */
/* member on the same threads (CRC-checked; a damaged member is KR_ERR_IO); ordinary gzip is one stream  */
/* and is inflated by zlib in the calling thread.                                                        */
/* Records, names and order are those of the sequential reader in every case.                          */
typedef struct kr_fastx kr_fastx;
typedef struct kr_fastx_batch {
  const uint8_t* bases;
  const uint64_t* offsets;     /* [nreads+1] */
  const char* const* names;    /* [nreads]   NUL-terminated, back to back in one buffer, in order */
  uint32_t nreads;
  uint32_t more;               /* 0 once the input is exhausted                    */
} kr_fastx_batch;
KR_API int kr_fastx_open(const char* path, kr_fastx** out);
KR_API int kr_fastx_next(kr_fastx*, uint64_t min_bases, kr_fastx_batch* out);
/* The batch kr_fastx_next just returned changes hands (what QSeq gives IBatch by swap, src/query.cpp:32-33): the kr_fastx_batch's
 * pointers stay valid, nothing is copied, until kr_fastx_release hands the buffers back for reuse -- from any thread, before
 * kr_fastx_close.  For consumers that keep several batches in flight (the CLI's workers). */
typedef struct kr_fastx_held kr_fastx_held;
KR_API int kr_fastx_detach(kr_fastx*, kr_fastx_held** out);
KR_API void kr_fastx_release(kr_fastx*, kr_fastx_held*);
KR_API uint64_t kr_fastx_parallel_chunks(const kr_fastx*); /* chunks taken from the thread pool so far */
/* ordinary gzip input (csrc/kr_pgz.inc): chunks handed out with their records parsed by the pool / chunks whose speculative start
 * the verified stream did not pass through (their range was inflated by the reader) / gaps closed by the reader */
KR_API void kr_fastx_pgz_stats(const kr_fastx*, uint64_t* parsed, uint64_t* discarded, uint64_t* gaps);
KR_API void kr_fastx_close(kr_fastx*);

/* report_distances text (src/query.cpp:158-196; DISTANCE_FIELDS src/query.hpp:210;
 * std::fixed, precision 5 src/query.cpp:152-153).  Appends to a malloc'ed buffer. */
KR_API int kr_format_dist(const kr_host_index*, const kr_result_view*, const char* const* names, char** text,
                          uint64_t* len);
/* `krepp seek` rows `SEQ_ID\tDIST` / `SEQ_ID\tNaN` (SBatch::seek_sequences, src/seek.cpp:22-56) from a batch
 * on a sketch index; stream parameters multi = 1, no_filter = 1, dist_max unset. */
KR_API int kr_format_seek(const kr_host_index*, const kr_index*, const kr_result_view*, uint32_t hdist_th,
                          const char* const* names, char** text, uint64_t* len);
/* The same rows as TEXT written by the GPU (csrc/kr_dev_text.inc): IBatch::report_distances (src/query.cpp:158-196) ran inside the
 * reference's parallel batch task; kr_format_dist runs on host threads at ~100 M rows/s, 45 times under the kernels on a 1000-genome
 * index (33 rows per read).  A stream with text enabled formats its rows-only batches on the device -- SEQ_ID, reference name,
 * "%.5f" of DIST with printf's own rounding (round-half-even of the exact binary value), "SEQ_ID\tNA\tNaN" for a read without rows --
 * into a text buffer in HBM that is copied back as bytes: the host only write()s them, in batch order.
 *   kr_stream_text_enable   once per stream: uploads the index's node names (once per kr_index) and sizes the stream's id and text
 *                           buffers (max_text_bytes of page-locked host memory and as much HBM; max_id_bytes for a batch's ids)
 *   kr_batch_submit_text    kr_batch_submit(flags | KR_ROWS_ONLY) + the reads' ids: `ids` holds them back to back in read order,
 *                           id r = ids[id_off[r] .. id_off[r+1] - id_sep) (id_sep = 1 for NUL-terminated names, as kr_fastx_batch
 *                           delivers them); ids and id_off may be pageable, they are staged before the call returns
 *   kr_batch_collect_text   waits, copies the text back; *text points into the stream's page-locked buffer and stays valid until the
 *                           next submit on this stream.  KR_ERR_CAPACITY: more text (or ids) than the buffers hold -- resubmit in
 *                           smaller pieces; KR_ERR_UNSUPPORTED: the batch could not be formatted on the device (it was tiled: long
 *                           sequences; or a DIST outside [0, 1000)): kr_batch_collect + kr_format_dist still serve it, no resubmit.
 * Byte-identical to kr_format_dist (tests/test_gpu_text.py). */
KR_API int kr_stream_text_enable(kr_stream*, const kr_host_index*, uint64_t max_text_bytes, uint64_t max_id_bytes);
KR_API int kr_batch_submit_text(kr_stream*, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads, uint32_t flags,
                                const char* ids, const uint32_t* id_off, uint32_t id_sep);
KR_API int kr_batch_collect_text(kr_stream*, const char** text, uint64_t* len);
KR_API uint32_t kr_debug_fixed5(double v, char* out); /* tests: the device formatter's "%.5f" (0: outside [0, 1000)) */
KR_API void kr_free(void*);
/* Page-locked host memory for read batches (KR_BASES_PINNED): what a reader fills instead of a std::string. */
KR_API void* kr_host_alloc(uint64_t bytes);
KR_API void kr_host_free(void*);

/* ------------------------------------------------------------------------- */
/* `krepp place`: IBatch::place_sequences / report_placement (src/query.cpp:198-333), */
/* TargetIndex::ensure_backbone (src/krepp.cpp:48-64), Tree::map_to_qtree /           */
/* compute_eff_nchildren (src/phytree.cpp:421-473); lineage trees: read_lineages         */
/* (src/krepp.cpp:37-46), Tree::parse_lineages (src/phytree.cpp:320-369).               */
/* ------------------------------------------------------------------------- */
typedef struct kr_place_tree kr_place_tree;
/* nwk_text == NULL: place on the index's own backbone; otherwise map the index leaves onto the
 * given rooted Newick tree.  kr_place_tree_kinds gives the node_kind array to upload the index
 * with (leaves absent from the placement tree become null nodes, as after map_to_qtree). */
KR_API int kr_place_tree_create(const kr_host_index*, const char* nwk_text, kr_place_tree** out);
/* -l/--lineage-file: the placement tree is the taxonomy of a Greengenes/GTDB style lineage file
 * (`ID <tab> r__Taxon; r__Taxon; ...` per line); its leaves are the reference IDs. */
KR_API int kr_place_tree_create_lineage(const kr_host_index*, const char* lineage_text, kr_place_tree** out);
KR_API uint32_t kr_place_tree_nnodes(const kr_place_tree*);
KR_API void kr_place_tree_free(kr_place_tree*);
KR_API const uint8_t* kr_place_tree_kinds(const kr_place_tree*);

typedef struct kr_placement {
  uint32_t read;
  uint32_t edge;  /* Node::get_en = se - 1 in the placement tree */
  double lwr, d_llh, v_llh, pendant, distal;
} kr_placement;

/* Back end for one collected batch.  `rv` must come from kr_batch_collect of a stream created with
 * multi = 1, no_filter = 1, dist_max unset, submitted with KR_TAP_ACCS (it needs the histograms);
 * `p` carries the place options (tau, chisq, multi, no_filter; filter defaults to on for place,
 * src/krepp.cpp:593-630).  The tree aggregation runs on the host; every likelihood (Brent on the
 * fractional histograms of internal nodes, the chi-square evaluations) runs on the GPU through
 * kr_llh_batch.  `has_previous` carries the jplace comma state across batches (in/out). */
KR_API int kr_place_batch(const kr_host_index*, const kr_index*, const kr_place_tree*, const kr_result_view* rv,
                          const uint64_t* offsets, const char* const* names, const kr_params* p, int tabular,
                          int* has_previous, char** text, uint64_t* len, kr_placement** placements, uint64_t* nplacements);
/* The same for the batch last submitted on `s` (KR_TAP_ACCS; no collect needed): the whole back end of
 * report_placement up to the candidate list -- ancestor accumulation (Minfo::add, src/query.hpp:139-152;
 * src/query.cpp:248-265), candidate listing (:268-272), Brent on the internal candidates (:273-275) and the
 * chi-square of every candidate (:276) -- runs on the device, on the records where they lie; the host receives
 * the candidates and does the last phase (filter, LWR, Jukes-Cantor, text).  Output identical to kr_place_batch.
 * Reads of any weight stay on the device (see kr_place_counters). */
KR_API int kr_place_stream(const kr_host_index*, const kr_index*, const kr_place_tree*, kr_stream* s, uint32_t nreads,
                           const uint64_t* offsets, const char* const* names, const kr_params* p, int tabular,
                           int* has_previous, char** text, uint64_t* len, kr_placement** placements, uint64_t* nplacements);
/* How many kr_place_stream batches of this process ran their back end on the device, and how many were sent whole to the
 * host back end (kr_place_batch: a placement tree that is not numbered in post-order, KR_PLACE_HOST set, or a batch that ran
 * out of candidate slots); heavy_reads: reads with more leaves / distinct ancestors than kr_place_kernel's LDS arrays hold
 * (256 / 1024), which its second launch does with the arrays in global scratch (counted in list chunks of 8: an upper
 * bound, 0 when there was none).  Any pointer may be NULL. */
KR_API void kr_place_counters(uint64_t* device_batches, uint64_t* host_batches, uint64_t* heavy_reads);
/* Ranges of reads whose `place` rows were written on the device / that the host formatted although device text was asked for. */
KR_API void kr_place_text_counters(uint64_t* device_ranges, uint64_t* fallback_ranges);
/* `tabular`: 0 jplace, 1 --tabular, 2 --summarize (no per-read text; feed the placements to
 * kr_place_summary_add).  place --summarize (src/krepp.cpp:466-471,493-497): `wcount` has
 * kr_place_tree_nnodes + 1 doubles, zeroed by the caller before the first batch and indexed by edge + 1. */
KR_API int kr_place_summary_add(const kr_place_tree*, const kr_placement* placements, uint64_t n, double* wcount,
                                double* twcount);
KR_API int kr_place_summary_text(const kr_place_tree*, const double* wcount, double twcount, char** text, uint64_t* len);
/* which = 0: text before the batches (jplace opening / tabular or summary header), 1: after (jplace metadata + tree) */
KR_API int kr_place_frame(const kr_place_tree*, int which, int tabular, const char* invocation, uint64_t total_qseq,
                          char** text, uint64_t* len);

/* CPU-side index construction (`krepp index`, src/krepp.cpp:131-303): stays on the
 * CPU as in the reference; needed to make any index at all. */
typedef struct kr_build_params {
  uint32_t k, w, h, m, r, frac; /* defaults 29, 35, 13, 4, 1, 1 (src/krepp.hpp:47-58) */
  uint32_t num_threads;
  uint32_t seed;
  const uint8_t* ppos;          /* optional explicit LSH positions [h] (descending)    */
  uint32_t gpu_minimizers;      /* 1: leaf stage (window minimizers) on the GPU `device`; */
  int32_t device;               /*    identical output, checked by tests/test_minimizers.py */
} kr_build_params;
KR_API int kr_build_index(const char* input_tsv, const char* nwk_path /*may be NULL*/, const char* out_dir,
                          const kr_build_params*);
/* `krepp sketch` (SketchSingle::create_sketch / save_sketch, src/krepp.cpp:110-129): the minimizers of one
 * FASTA/FASTQ file in the reference's sketch format (defaults there: k 26, w k+6, h k-16, m 4, r 1, frac). */
KR_API int kr_build_sketch(const char* input_path, const char* out_path, const kr_build_params*);

/* The leaf stage of the build on its own: RSeq::extract_mers (src/rqseq.cpp:51-144) + the
 * sort/unique of DynHT::fill_table (src/table.cpp:247-260) for ONE genome held in memory
 * (contigs concatenated, offsets[ncontigs+1]).  keys = (row << 32) | enc32, sorted, unique;
 * n1/n2 = sums over contigs of the HyperLogLog(12) estimates of distinct k-mers / distinct
 * window minimizers (rho = n2 / n1, src/rqseq.hpp:79).  `ppos` must be given.
 * _cpu is the builder's own code path; _device computes the window minima on the GPU
 * (wave ballots for the 2-bit strings, LDS window minimum) and must return identical results. */
typedef struct kr_minimizer_result {
  uint64_t* keys;
  uint64_t nkeys;
  double n1, n2;
} kr_minimizer_result;
KR_API int kr_minimizers_cpu(const kr_build_params*, const uint8_t* bases, const uint64_t* offsets, uint32_t ncontigs,
                             kr_minimizer_result* out);
KR_API int kr_minimizers_device(int device, const kr_build_params*, const uint8_t* bases, const uint64_t* offsets,
                                uint32_t ncontigs, kr_minimizer_result* out);
KR_API void kr_minimizers_free(kr_minimizer_result*);

KR_API const char* kr_last_error(void);
KR_API const char* kr_version(void);

#ifdef __cplusplus
}
#endif
#endif
