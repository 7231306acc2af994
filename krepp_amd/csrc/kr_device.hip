// kr_device.hip — the per-read `krepp dist` hot path as hand-written HIP for gfx950
// (MI355X), and the device half of the C ABI (include/krepp_amd.h).
//
// Pipeline per submitted batch (one HIP stream per kr_stream):
//
//   kr_scan_kernel       one wave64 per read, 4-6 waves per SIMD.  Front end from wave ballots (no LDS,
//                        no rolling state): every k-mer x strand -> LSH row (rix) and residual code
//                        (enc32) [src/query.cpp:40-94, src/common.hpp:177-243, src/lshf.cpp:39-69];
//                        bucket lookup [src/index.cpp:160-168]; bucket scan with lane groups over
//                        16-byte chunks of the bucket (slotted or packed table), Hamming filter
//                        [src/query.cpp:361-368]; output: the read's hit items in HBM.
//   kr_acc_kernel        one wave64 per read: colour gather, colour-DAG expansion from an LDS work
//                        stack [src/query.cpp:369-387]; per-(strand, leaf) accumulation as events /
//                        position bit-planes [Minfo::update_match, src/query.hpp:153-176]; hdist_filt
//                        test [src/query.cpp:101-106,119] and record emission.
//   kr_dedup_kernel      the distinct likelihood problems (leaf, histogram, #k-mers) of the batch, found with
//                        an open-addressing table in HBM; nothing is kept between batches.
//   kr_llh_pre_kernel    the first three objective values of every distinct problem (shared abscissas: the
//                        d-dependent part of the objective is evaluated once per workgroup).
//   kr_llh_kernel        one lane per distinct problem, lanes refilled as their minimisations converge:
//                        Brent minimisation of HDistHistLLH in fp64 [src/hdhistllh.hpp:51-96,
//                        src/query.cpp:426-433, boost::math::tools::brent_find_minima].
//   kr_llh_copy_kernel   (d, v) of its problem to every record.
//   kr_select_kernel     32 lanes per read, one record per lane: strand merge, closest reference, --filter /
//                        --dist-max / --no-multi selection [src/query.cpp:96-139,158-196].
//
// Exactness: Minfo::update_match counts, per read position, only the smallest Hamming
// distance among all hits that reach a leaf.  Here a hit sets bit `pos` in plane `hd` of
// the (strand, leaf) accumulator with an atomic OR; at the end
//     hist[x] = popcount(plane_x & ~(plane_0 | ... | plane_{x-1})).
// OR is idempotent and commutative, so the result does not depend on the order in which
// lanes, probes or colour expansions arrive, and is bit-identical to the serial rule.
//
// Built with -ffp-contract=off: the reference is compiled for baseline x86-64 (no FMA,
// makefile:7), and the likelihood follows its operation order.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "kr_common.h"
#include "kr_devutil.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// Compile-time shape of the probe kernel
// ---------------------------------------------------------------------------
constexpr int kWave = 64;
constexpr int kSegPos = 128;      // k-mer positions per segment = 4 plane words
constexpr int kPlaneWords = kSegPos / 32;
constexpr int kMaxLibs = 16;
constexpr int kStackCap = 256;    // colour work stack (items of 8 B)
constexpr int kLeanStack = 192;   // ... of the single-segment accumulate instantiation (128 pushes per step + DAG depth)
// its stack region also hosts the key bitmap (4 B per word) and the ordinal prefix (1 B per word)
__host__ __device__ inline uint32_t lean_stack_bytes(uint32_t bm_words) { uint32_t b = 5u * bm_words; b = (b + 15u) & ~15u; return b > kLeanStack * 8u ? b : kLeanStack * 8u; }
constexpr int kLdsSlots = 64;     // level-1 (LDS) accumulator slots per wave: lane t owns slot t in the epilogue
constexpr int kLdsProbeMax = 8;   // bounded probe sequence of the level-1 table
constexpr int kMaxPlanes = KR_MAX_HDIST_TH + 1;
constexpr int kHistWords = (kMaxPlanes + 3) / 4; // packed 8-bit histogram counters

struct DevLib {
  const uint64_t* bkt;  // [nrows]  (start << 24) | len
  const uint32_t* enc;  // [nkmers + pad] residual codes, bucket-contiguous
  const uint32_t* se;   // [nkmers] colour ids
  const uint2* pse;     // [nsubsets] colour DAG: colour = union of .x and .y
  const double* rho;    // [nnodes] subsampling rates, already scaled
  const uint32_t* slots; // [nrows << slot_log2w] slotted copy of the head of every bucket (dense tables), or null
  uint64_t nkmers;
  uint32_t nrows, nsubsets, nnodes, numer;
  uint32_t slot_log2w;  // words per slot = 1 << slot_log2w: {len, start, first (words - 2) residual codes}
};

struct DevIndex {
  uint32_t k, h, m, nlibs, tree_nnodes;
  uint32_t m_shift;      // log2(m) if m is a power of two, else 0xFFFFFFFF
  PextMask pmask, nmask; // bit masks of the LSH / non-LSH positions within the k-bit half-codes
  uint32_t nleaves;
  const uint32_t* node_info; // [tree_nnodes+1] kind (0 null, 1 leaf, 2 internal) | leaf_rank << 2
  const uint32_t* leaf_se;   // [nleaves] colour id of the leaf with a given rank (ranks follow se order)
  const int32_t* res_lib; // [m] library serving each residue, or -1
  const DevLib* libs;    // [nlibs] in device memory
  uint64_t res_mask;     // nlibs == 1 && m <= 64: bit r set iff residue r is served
  DevLib lib0;           // copy of libs[0]: single-library indexes never touch `libs`
};

struct LlhConst {
  uint32_t k, h, th, dbg; // dbg: timing experiments only (KR_DEBUG_LLH): 1 no pow, 2 no log(d)/log(1-d), 4 no final log, 8 short loop
  double binom_k[32];
  double binom_hnk[kMaxPlanes];
};

struct DevParams {
  uint32_t th, np;       // np = th + 1 planes
  uint32_t multi, no_filter, dmax_set;
  uint32_t dbg; // KR_DEBUG_SKIP (timing experiments / tests only): 1 drop hits, 2 drop expansion, 4 skip scan, 8 no event mode,
                // 16 drop events, 64 no batches, 128 no plane pass, 256 no record output, 512 statistics, 1024 never / 2048
                // eagerly use the global single batch, 4096 planes for single-event keys too, 8192 plane tables for reads of
                // several segments
  double chisq, dist_max;
};

// Everything the kernels write for one batch.
struct BatchOut {
  uint32_t* counters;    // [0] record slots handed out  [1] error flags  [2] reads that used level 2  [3] nhits(tap)  [4] records
                         // [5] LLH chunk cursor  [6] item slots handed out  [22] distinct likelihood problems
  uint32_t* cursors;     // read cursors of the scan and the accumulate kernel, [2][kCursors * kCursorStride]
  uint32_t* rd_off;
  uint32_t* rd_cnt;
  uint32_t* rd_onmers;
  uint32_t* rd_filt;     // [2*nreads] raw per-strand hdist_filt (tap)
  uint8_t* rd_na;
  uint32_t* rec_read;
  uint32_t* rec_key;
  uint32_t* rec_hist;    // [np][rec_cap]: hist[x] of record i at x * rec_cap + i (coalesced across records)
  double* rec_d;
  double* rec_v;
  double* rec_chisq;
  uint8_t* rec_sel;
  // likelihood de-duplication (kr_dedup_kernel): records with the same (leaf, histogram, #k-mers) are one problem
  uint64_t* rec_w0;      // [rec_cap] the record's likelihood problem packed in one word by kr_acc_kernel (0: not packable)
  uint32_t* rec_rep;     // [rec_cap] position of the record's representative in rep_list (0xFFFFFFFF: hole)
  uint32_t* rep_list;    // [rec_cap] record index of every distinct problem
  double2* rep_dv;       // [rec_cap] (d_llh, v_llh) per distinct problem
  ulonglong2* dd_table;  // [dd_slots] open-addressing table: x = histogram word (0 = empty), y = leaf | (list position + 1) << 32
  uint32_t dd_slots;     // power of two
  uint32_t dd_shift;     // slots used per batch = records >> dd_shift (rounded up to a power of two)
  uint32_t rec_cap;
  kr_hit* hits;
  uint32_t hit_cap;
  // scan kernel -> accumulate kernel: resolved hits (colour, tag) of every read, contiguous per read
  uint2* items;          // x = entry index (low 32), y = pos(7) | strand<<7 | lib(4)<<8 | hd(5)<<12 | index high(8)<<17;
                         // y bit 31: segment marker, x = segment number
  uint32_t item_cap;
  uint32_t* rd_it_off;
  uint32_t* rd_it_cnt;
  uint32_t* long_list;   // [max_reads] reads of several segments, set aside by the first accumulate launch (counters[25])
  // level-2 accumulator scratch, one region per resident wave
  uint32_t* g_planes; // [nwaves][nslots2][np][4]
  uint32_t* g_counts; // [nwaves][nslots2][np]
  uint32_t* g_list;   // [nwaves][nslots2]
  uint64_t* stk_spill; // [nwaves][kStackSpill] lower part of a work stack that outgrew the LDS (large clades)
  uint32_t nslots2;   // 2 * nleaves
  uint32_t g_list_words; // per wave: max(nslots2, ev_spill + tab_spill * (hist words + 1))
  uint32_t ev_spill, tab_spill, kt_spill; // event-mode spill capacities per wave (events, table entries, keys)
  uint32_t bm_words;  // ceil(nslots2 / 32)
};

enum : uint32_t { kErrRecCap = 1u, kErrStack = 2u, kErrTable = 4u, kErrHitCap = 8u, kErrItemCap = 16u };

struct BatchIn {
  const uint8_t* bases;
  const uint64_t* offsets;
  uint32_t nreads;
};

// ---------------------------------------------------------------------------
// Device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// LDS pointers carry their address space in the type so that every access is a ds_ instruction
// (a generic pointer that the compiler cannot trace back to LDS becomes a flat_ access, which
// waits on both the LDS and the vector-memory counters).
#define KR_LDS __attribute__((address_space(3)))
typedef KR_LDS uint32_t lds_u32;
typedef KR_LDS uint64_t lds_u64;
typedef KR_LDS uint16_t lds_u16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_cas(lds_u32* p, uint32_t expected, uint32_t desired)
{
  __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return expected;
}
__device__ __forceinline__ void lds_or(lds_u32* p, uint32_t v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_ld(lds_u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The probe kernel runs one wave64 per workgroup, so LDS hand-offs are between lanes of ONE wave:
// LDS operations of a wave execute in issue order, and all that is needed is that the compiler
// keeps that order.  __syncthreads() would also drain every outstanding global load
// (s_waitcnt vmcnt(0)) and so serialise the prefetched bucket loads behind each LDS exchange.
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// hd = number of non-LSH positions that differ (popcount_lr32, src/common.hpp:175)
__device__ __forceinline__ uint32_t hd_lr32(uint32_t a, uint32_t b)
{
  uint32_t z = a ^ b, f;
#ifdef KR_NO_SDWA
  return __popc((z | (z >> 16)) & 0xFFFFu);
#endif
  // (z | z >> 16) & 0xffff in one instruction: OR of the two 16-bit halves, zero-extended (SDWA operand selects)
  asm("v_or_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(f) : "v"(z));
  return __popc(f);
}

__device__ __forceinline__ uint32_t hash_key(uint32_t key) { return key * 0x9E3779B1u; }

// Per-(k-mer, strand) front end shared by the probe kernel and the debug tap.
struct FrontEnd {
  uint32_t rix[2], enc32[2]; // [strand]
  bool valid;
};

// Three 64-bit ballots per bit array cover 192 bases of the segment; position j's window
// is bits [j, j+k) in read order.
struct SegBits {
  uint64_t L[3], H[3], N[3];
};

__device__ __forceinline__ void load_segment(const uint8_t* seq, uint64_t len, uint64_t base0, SegBits& sb)
{
  uint32_t lane = lane_id();
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    uint64_t bi = base0 + 64u * p + lane;
    uint32_t code = 4;
    if (bi < len) code = base_code(seq[bi]);
    sb.L[p] = __ballot(code & 1u && code < 4);
    sb.H[p] = __ballot((code >> 1) & 1u && code < 4);
    sb.N[p] = __ballot(code >= 4);
  }
}

__device__ __forceinline__ uint32_t window32(uint64_t w0, uint64_t w1, uint32_t s)
{
  uint64_t v = s ? ((w0 >> s) | (w1 << (64 - s))) : w0;
  return (uint32_t)v;
}

// pp = 0/1: positions j = 64*pp + lane of the segment.
__device__ __forceinline__ FrontEnd front_end(const DevIndex& ix, const SegBits& sb, int pp, uint32_t npos_seg)
{
  FrontEnd fe;
  uint32_t lane = lane_id();
  uint32_t j = 64u * pp + lane;
  uint32_t mk = (1u << ix.k) - 1u;
  uint32_t wl = window32(sb.L[pp], sb.L[pp + 1], lane) & mk;
  uint32_t wh = window32(sb.H[pp], sb.H[pp + 1], lane) & mk;
  uint32_t wn = window32(sb.N[pp], sb.N[pp + 1], lane) & mk;
  fe.valid = (wn == 0) && (j < npos_seg);
  // position p of the k-mer counts from its LAST base
  uint32_t lo_f = __brev(wl) >> (32 - ix.k), hi_f = __brev(wh) >> (32 - ix.k);
  uint32_t lo_r = ~wl & mk, hi_r = ~wh & mk; // reverse complement: complement, order already reversed
  fe.rix[0] = spread16(pext32(lo_f, ix.pmask)) | (spread16(pext32(hi_f, ix.pmask)) << 1);
  fe.rix[1] = spread16(pext32(lo_r, ix.pmask)) | (spread16(pext32(hi_r, ix.pmask)) << 1);
  fe.enc32[0] = pext32(lo_f, ix.nmask) | (pext32(hi_f, ix.nmask) << 16);
  fe.enc32[1] = pext32(lo_r, ix.nmask) | (pext32(hi_r, ix.nmask) << 16);
  return fe;
}

// Index::check_partial + Index::bucket_indices (src/index.hpp:27, src/index.cpp:160-168)
template <bool SL>
__device__ __forceinline__ bool locate_row(const DevIndex& ix, uint32_t rix, int& lib, uint32_t& row)
{
  uint32_t res, q;
  if (ix.m_shift != 0xFFFFFFFFu) {
    res = rix & (ix.m - 1u);
    q = rix >> ix.m_shift;
  } else {
    q = rix / ix.m;
    res = rix - q * ix.m;
  }
  uint32_t numer, nrows;
  if (SL) { // everything from kernel arguments
    if (!((ix.res_mask >> res) & 1ull)) return false;
    lib = 0;
    numer = ix.lib0.numer;
    nrows = ix.lib0.nrows;
  } else {
    lib = ix.res_lib[res];
    if (lib < 0) return false;
    numer = ix.libs[lib].numer;
    nrows = ix.libs[lib].nrows;
  }
  row = numer > 1 ? q * numer + res : q;
  return row < nrows;
}

// ---------------------------------------------------------------------------
// Accumulators.  Level 1: 64-slot open-addressing hash table in LDS, keyed
// (se << 1) | strand, probe sequences bounded to 8 slots.  Level 2 (only for reads whose
// keys do not fit level 1): direct-indexed by slot2 = 2 * leaf_rank + strand in a per-wave
// global scratch region, with an LDS bitmap of the touched slots.  A key lives in exactly one
// level: slots never empty during a read, so a bounded probe that failed once fails again.
// ---------------------------------------------------------------------------
struct Acc {
  // level 1 (LDS)
  lds_u32* keys;    // [kLdsSlots]
  lds_u32* planes;  // [kLdsSlots * np * 4]
  lds_u32* counts;  // [kLdsSlots * np]
  // level 2
  lds_u32* bitmap;   // LDS [bm_words] (bm_words even, 8-byte aligned)
  lds_u32* rbitmap;  // LDS [bm_words]: keys of the READ so far (event mode over several segments)
  lds_u16* pre;      // LDS [bm_words / 2]: key ordinal prefix per 64-bit bitmap block (event mode)
  uint32_t* g_planes; // global [nslots2 * np * 4]
  uint32_t* g_counts; // global [nslots2 * np]
  uint32_t* g_list;   // global [nslots2]
  uint32_t nslots2, bm_words, np;
};

__device__ __forceinline__ uint32_t gload(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Work-stack item (8 B): lo = tagged colour id, hi = pos(7) | strand(1)<<7 | lib(4)<<8 | hd(5)<<12.
// The items the scan kernel hands over name the table ENTRY instead (index low in lo, high 8 bits in hi
// bits 17..24); the accumulate kernel fetches the colour.
__device__ __forceinline__ uint32_t tag_pos(uint32_t t) { return t & 127u; }
__device__ __forceinline__ uint32_t tag_strand(uint32_t t) { return (t >> 7) & 1u; }
__device__ __forceinline__ uint32_t tag_lib(uint32_t t) { return (t >> 8) & 15u; }
__device__ __forceinline__ uint32_t tag_hd(uint32_t t) { return (t >> 12) & 31u; }

struct WaveState {
  lds_u64* stack;   // LDS [stack_cap]: lo | hi << 32
  uint32_t stack_cap; // entries (wave-uniform)
  uint32_t top;   // wave-uniform
  uint64_t* gstk; // global spill of the stack's oldest entries (wave-private)
  uint32_t gs_top; // entries spilled (wave-uniform)
  bool l2;        // this lane sent something to level 2 during this read
  uint32_t err;
  uint32_t rec_next, rec_end; // wave-private range of record slots (wave-uniform)
  uint32_t n_l2;              // reads of this wave that used level 2
  uint32_t n_rec;             // records this wave emitted
  uint32_t n_spill;           // times the work stack moved its older half to global memory
  uint32_t ll_next, ll_end;   // wave-private chunk of out.long_list (wave-uniform)
  // event mode (reads of a single segment): leaf updates are appended as 32-bit events
  bool evmode;      // wave-uniform
  bool ev_full;     // wave-uniform: the event buffer overflowed
  uint32_t nev;     // events buffered (wave-uniform)
  uint32_t ev_cap;  // power of two
  uint32_t ev_words; // words of the region the events and the epilogue's batch arrays share (level-1 planes + counts)
  bool dirty;       // wave-uniform: event mode left data in the level-1 table regions (zeroed on demand)
  bool lean;        // single-segment instantiation: the key bitmap aliases the (idle) work stack, zeroed per read
  uint32_t* gev;    // global spill: events beyond ev_cap, then table entries beyond the LDS table
  uint32_t gev_cap, gtab_cap, gkt_cap; // spill capacities (events / table entries / keys)
  lds_u32* ev;      // aliases the level-1 planes + counts region
};

// SL (single library, m <= 64) is a compile-time property of the launched kernel: the library
// descriptor and the residue mask then come from kernel arguments (SGPRs) with no load at all.
template <bool SL>
__device__ __forceinline__ DevLib get_lib(const DevIndex& ix, uint32_t lib)
{
  if (SL) return ix.lib0;
  return ix.libs[lib];
}

// A leaf update as one 32-bit event, ordered so that an ascending sort groups by (leaf, strand),
// then position, then Hamming distance:  rank << 13 | strand << 12 | pos << 5 | hd.
__device__ __forceinline__ uint32_t make_event(uint32_t rank, uint32_t tag)
{
  return (rank << 13) | (tag_strand(tag) << 12) | (tag_pos(tag) << 5) | tag_hd(tag);
}

// Minfo::update_match (src/query.hpp:153-176) as an idempotent OR: bit `pos` of plane `hd`.
// Keys are (leaf rank + 1) << 1 | strand; level 2 is indexed by slot2 = 2 * rank + strand.
__device__ __forceinline__ void accumulate_planes(const Acc& A, WaveState& ws, uint32_t ev)
{
  const uint32_t rs = ev >> 12; // rank << 1 | strand
  const uint32_t key = rs + 2u;
  const uint32_t hd = ev & 31u, pos = (ev >> 5) & 127u;
  uint32_t s = (hash_key(key) >> 8) & (kLdsSlots - 1);
#pragma unroll 1
  for (int i = 0; i < kLdsProbeMax; ++i) {
    uint32_t cur = lds_ld(&A.keys[s]);
    if (cur == 0) {
      uint32_t old = lds_cas(&A.keys[s], 0u, key);
      cur = old == 0 ? key : old;
    }
    if (cur == key) {
      lds_or(&A.planes[(s * A.np + hd) * kPlaneWords + (pos >> 5)], 1u << (pos & 31));
      return;
    }
    s = (s + 1) & (kLdsSlots - 1);
  }
  // level 2
  lds_or(&A.bitmap[rs >> 5], 1u << (rs & 31));
  __hip_atomic_fetch_or(&A.g_planes[((uint64_t)rs * A.np + hd) * kPlaneWords + (pos >> 5)], 1u << (pos & 31),
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ws.l2 = true;
}

// One leaf update per lane (f = this lane has one): event append, or plane OR after fallback.
__device__ __forceinline__ void add_leaf_events(const Acc& A, WaveState& ws, bool f, uint32_t ev)
{
  const uint64_t m = __ballot(f);
  if (m == 0) return;
  if (ws.evmode) {
    const uint32_t c = __popcll(m);
    if (ws.nev + c <= ws.ev_cap + ws.gev_cap) {
      if (f) {
        const uint32_t i = ws.nev + __popcll(m & ((1ull << lane_id()) - 1ull));
        if (i < ws.ev_cap)
          ws.ev[i] = ev;
        else
          gstore(&ws.gev[i - ws.ev_cap], ev); // long event lists continue in (L2-resident) global scratch
      }
      ws.nev += c;
    } else {
      ws.ev_full = true; // the read is redone with the plane tables (rare)
    }
    return;
  }
  if (f) accumulate_planes(A, ws, ev);
}

// Colour ids in HBM carry their class in the top two bits (set once at upload, see
// kr_tag_colours): 0 = drop (empty set / null tree node), 1 = tree leaf (low bits = leaf RANK, the
// index among leaves in colour-id order), 2 = expand through se_to_pse (low bits = colour id).
// This replaces Tree::check_node + get_node + check_leaf (src/query.cpp:371-381) and a dependent
// load per colour.
constexpr uint32_t kColMask = 0x3FFFFFFFu;

// One step of the colour expansion (the BFS of src/query.cpp:369-387, order-free here) for one item
// per lane: a leaf becomes an event, a colour that expands is looked up in se_to_pse and its two
// parts become events or new work items.  The caller guarantees room for 128 pushes.
template <bool SL>
__device__ __forceinline__ void expand_step(const DevIndex& ix, const Acc& A, WaveState& ws, bool have, uint32_t se, uint32_t tag)
{
  const uint64_t lt = (1ull << lane_id()) - 1ull;
  uint32_t c0 = 0, c1 = 0;
  bool p0 = false, p1 = false;
  bool f0 = false, f1 = false, f2 = false; // leaf updates found by this lane: the item, child 0, child 1
  if (have) {
    f0 = (se >> 30) == 1u;
    if ((se >> 30) == 2u) {
      const uint2 pr = get_lib<SL>(ix, tag_lib(tag)).pse[se & kColMask];
      c0 = pr.x;
      c1 = pr.y;
      f1 = (c0 >> 30) == 1u, p0 = (c0 >> 30) == 2u;
      f2 = (c1 >> 30) == 1u, p1 = (c1 >> 30) == 2u;
    }
  }
  add_leaf_events(A, ws, f0, make_event(se & kColMask, tag));
  add_leaf_events(A, ws, f1, make_event(c0 & kColMask, tag));
  add_leaf_events(A, ws, f2, make_event(c1 & kColMask, tag));
  const uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
  if (p0) ws.stack[ws.top + __popcll(m0 & lt)] = (uint64_t)c0 | ((uint64_t)tag << 32);
  ws.top += __popcll(m0);
  if (p1) ws.stack[ws.top + __popcll(m1 & lt)] = (uint64_t)c1 | ((uint64_t)tag << 32);
  ws.top += __popcll(m1);
  WAVE_SYNC();
}

// drain the work stack
// The LDS stack is the top of a two-level stack.  A colour that covers a large clade fans out faster than it is
// consumed (64 items popped, up to 128 pushed): when no room is left the OLDER half moves to the wave's global spill
// (LIFO never needs it before the newer half is gone), and it comes back when the LDS part has run empty.
constexpr uint32_t kStackSpill = 8192; // entries per wave (64 KiB)
// (by value: a WaveState passed by reference to a real call would have to live in scratch memory)
__device__ __noinline__ uint32_t stack_spill(lds_u64* stack, uint64_t* gstk, uint32_t top, uint32_t gs_top)
{ // returns the number of entries moved out (0: the spill is full)
  const uint32_t lane = lane_id(), half = top / 2u;
  if (half == 0 || gs_top + half > kStackSpill) return 0;
  for (uint32_t i = lane; i < half; i += 64) gstk[gs_top + i] = stack[i];
  WAVE_SYNC();
  for (uint32_t i0 = 0; i0 < top - half; i0 += 64) { // move the newer part down, tile by tile (ascending: no overlap hazard)
    const uint32_t i = i0 + lane;
    uint64_t v = 0;
    if (i < top - half) v = stack[half + i];
    WAVE_SYNC();
    if (i < top - half) stack[i] = v;
    WAVE_SYNC();
  }
  return half;
}
__device__ __noinline__ uint32_t stack_refill(lds_u64* stack, const uint64_t* gstk, uint32_t cap, uint32_t gs_top)
{ // returns the number of entries brought back
  const uint32_t lane = lane_id(), m = min(gs_top, cap / 2u);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // this wave's earlier spill stores have landed
  for (uint32_t i = lane; i < m; i += 64) stack[i] = gstk[gs_top - m + i];
  WAVE_SYNC();
  return m;
}

template <bool SL>
__device__ __forceinline__ void expand_all(const DevIndex& ix, const Acc& A, WaveState& ws)
{
  const uint32_t lane = lane_id();
  for (;;) {
    if (ws.top == 0) {
      if (ws.gs_top == 0) break;
      const uint32_t m = stack_refill(ws.stack, ws.gstk, ws.stack_cap, ws.gs_top);
      ws.gs_top -= m, ws.top = m;
    }
    const uint32_t room = ws.stack_cap - ws.top;
    const uint32_t n = min(min(64u, ws.top), room);
    if (n == 0) { // no room for the children of even one item
      const uint32_t half = stack_spill(ws.stack, ws.gstk, ws.top, ws.gs_top);
      if (half) {
        ws.gs_top += half, ws.top -= half;
        ws.n_spill++;
        continue;
      }
      ws.err |= kErrStack; // the spill is full too: report, drop the rest
      ws.top = 0, ws.gs_top = 0;
      break;
    }
    const uint32_t base = ws.top - n;
    const bool have = lane < n;
    uint64_t raw = 0;
    if (have) raw = ws.stack[base + lane];
    ws.top = base;
    expand_step<SL>(ix, A, ws, have, (uint32_t)raw, (uint32_t)(raw >> 32));
  }
}

// ---------------------------------------------------------------------------
// Probe list of one group (64 positions x 2 strands) and the bucket scan.
// ---------------------------------------------------------------------------
constexpr int kListCap = 128;
struct ProbeList {
  lds_u64* bkt; // [128] (start << 24) | len
  lds_u32* q;   // [128] residual code of the query k-mer
  lds_u32* tag; // [128] pos | strand<<7 | lib<<8
};

struct Cand { // one lane's two candidate probes (forward, reverse) of a position
  uint64_t b0, b1;
  uint32_t q0, q1;
  uint32_t lib0, lib1;
};

// front end + descriptor loads for positions 64*pp + lane.  SLOT: the table has a slotted copy, the bucket is
// found by its row alone (b = 1 << 32 | row, no load here).
template <bool SL, bool SLOT>
__device__ __forceinline__ Cand fetch_group(const DevIndex& ix, const SegBits& sb, int pp, uint32_t npos_seg,
                                            uint32_t& nvalid)
{
  Cand c;
  FrontEnd fe = front_end(ix, sb, pp, npos_seg);
  nvalid = __popcll(__ballot(fe.valid));
  int lib = -1;
  uint32_t row = 0;
  c.b0 = 0, c.b1 = 0, c.lib0 = 0, c.lib1 = 0;
  c.q0 = fe.enc32[0];
  c.q1 = fe.enc32[1];
  if (fe.valid && locate_row<SL>(ix, fe.rix[0], lib, row)) {
    c.lib0 = (uint32_t)lib;
    c.b0 = SLOT ? (1ull << 32) | row : get_lib<SL>(ix, (uint32_t)lib).bkt[row];
  }
  if (fe.valid && locate_row<SL>(ix, fe.rix[1], lib, row)) {
    c.lib1 = (uint32_t)lib;
    c.b1 = SLOT ? (1ull << 32) | row : get_lib<SL>(ix, (uint32_t)lib).bkt[row];
  }
  return c;
}

// 4 entries of one 16-byte chunk against the query code.  `lo`/`hi` bound the entries of the chunk
// that belong to the bucket (0..4).  Returns the 4-bit hit mask; hds = 4 x 8-bit Hamming distances.
__device__ __forceinline__ uint32_t chunk_hits(uint4 v, int lo, int hi, uint32_t q, uint32_t th, uint32_t& hds)
{
  uint32_t h0 = hd_lr32(v.x, q), h1 = hd_lr32(v.y, q), h2 = hd_lr32(v.z, q), h3 = hd_lr32(v.w, q);
  hds = h0 | (h1 << 8) | (h2 << 16) | (h3 << 24);
  uint32_t m = (h0 <= th ? 1u : 0u) | (h1 <= th ? 2u : 0u) | (h2 <= th ? 4u : 0u) | (h3 <= th ? 8u : 0u);
  lo = max(lo, 0), hi = min(hi, 4);
  uint32_t in = hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
  return m & in;
}

// ---------------------------------------------------------------------------
// Kernel 1: the table scan.  One wave per read; per group of 64 positions the wave builds the probe
// list (front end + bucket descriptors) and scans the listed buckets: G = 2^LOG_G consecutive lanes
// share one probe and read consecutive aligned 16-byte chunks of its bucket (G*16 contiguous bytes per
// step), 64/G probes per pass, CPL chunks per lane per pass.  No search, no prefix sums: the probe of a
// lane is fixed by its lane id.  A pass covers G*CPL*4 entries of each bucket; longer buckets take
// extra rounds.  The kernel keeps no accumulator state, so it runs at 8 waves per SIMD and hides the
// HBM latency of its dependent steps (descriptor -> bucket -> colour) by occupancy alone.
//
// The scan proper only asks, per chunk, "is the smallest of the four Hamming distances within th?"
// (hits are ~1 % of the entries) and keeps the answers as one bit per (pass, chunk) in a per-lane
// register.  After the group's last pass the hit chunks of all lanes are dealt out again, one per
// lane: the lane re-reads its chunk (L2), applies the bucket bounds, fetches the colour of every
// matching entry and appends (colour, tag) items to the read's item list in HBM, which the
// accumulate kernel consumes.
// ---------------------------------------------------------------------------
struct ScanWave {
  uint32_t it_next, it_end; // wave-private range of item slots (wave-uniform)
  uint32_t rd_start;        // first item of the current read
  uint32_t err;
  uint32_t filt0, filt1;    // per-lane running min hd per strand (hdist_filt, src/query.cpp:366-368)
  uint32_t cur_seg;         // segment of the read's most recent item (wave-uniform)
};
constexpr uint32_t kItemChunk = 2048;
constexpr uint32_t kReadChunk = 8; // reads a wave takes per visit to a read cursor
// Reads are handed out in small chunks (heavy reads do not pile up on one wave, no tail).  One shared
// cursor would serve only ~90 M atomics/s -- 1.4 ms for a million reads -- so the batch is cut into
// kCursors ranges with a cursor each (on cache lines of their own); a wave starts on its "home" range
// and moves on to the next when that is used up.
constexpr uint32_t kCursors = 32, kCursorStride = 32; // words
struct ReadCursor {
  uint32_t* cur;   // [kCursors * kCursorStride]
  uint32_t nreads, range, c, tried;
  __device__ __forceinline__ void init(uint32_t* base, uint32_t n, uint32_t home)
  {
    cur = base, nreads = n, range = (n + kCursors - 1) / kCursors, c = home % kCursors, tried = n ? 0 : kCursors;
  }
  // next chunk [r0, r1); false when every range is used up (wave-uniform)
  __device__ __forceinline__ bool next(uint32_t& r0, uint32_t& r1)
  {
    while (tried < kCursors) {
      const uint32_t lo = c * range, hi = min(lo + range, nreads);
      uint32_t o = 0;
      if (lane_id() == 0) o = atomicAdd(&cur[c * kCursorStride], kReadChunk);
      o = __shfl(o, 0);
      if (lo + o < hi) {
        r0 = lo + o, r1 = min(r0 + kReadChunk, hi);
        return true;
      }
      c = (c + 1) % kCursors, ++tried;
    }
    return false;
  }
};

// Make room for n more items directly behind the current read's items (wave-uniform).
__device__ __forceinline__ bool item_reserve(const BatchOut& out, ScanWave& sw, uint32_t n)
{
  if (sw.it_next + n <= sw.it_end) return true;
  const uint32_t have = sw.it_next - sw.rd_start;
  const uint32_t size = have + n + kItemChunk;
  uint32_t base = 0;
  if (lane_id() == 0) base = atomicAdd(&out.counters[6], size);
  base = __shfl(base, 0);
  if ((uint64_t)base + size > out.item_cap) {
    sw.err |= kErrItemCap;
    return false;
  }
  if (have) { // the read's items so far move to the new chunk (they were written by this wave: read them from L2)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (uint32_t i = lane_id(); i < have; i += 64) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(out.items + sw.rd_start + i);
      out.items[base + i] = make_uint2(gload(src), gload(src + 1));
    }
  }
  sw.rd_start = base;
  sw.it_next = base + have;
  sw.it_end = base + size;
  return true;
}

// Entries `pend` (4-bit mask) of chunk c of the bucket at e_al matched: append one item per entry.  The item
// names the entry (its colour id is fetched by the accumulate kernel: this kernel runs at the chip's
// random-request rate, that one has memory bandwidth to spare).  Reads of more than one segment get a
// marker item in front of the first item of every later segment.
constexpr uint32_t kItemMarker = 0x80000000u;
template <bool SL, bool TAP>
__device__ __forceinline__ void emit_hits(const DevIndex& ix, const BatchOut& out, ScanWave& sw, uint32_t read, uint32_t base0,
                                          uint32_t tg, uint64_t e_al, uint32_t c, uint32_t pend, uint32_t hds)
{
  const uint64_t lt = (1ull << lane_id()) - 1ull;
  const uint32_t seg = base0 >> 7;
  while (__ballot(pend != 0) != 0) {
    const bool has = pend != 0;
    const uint32_t e = has ? (uint32_t)__ffs((int)pend) - 1u : 0u;
    const uint32_t hd = (hds >> (8u * e)) & 31u;
    const uint64_t idx = e_al + 4ull * c + e;
    if (has) {
      if (tag_strand(tg))
        sw.filt1 = min(sw.filt1, hd);
      else
        sw.filt0 = min(sw.filt0, hd);
      if (TAP) {
        const uint32_t se = get_lib<SL>(ix, tag_lib(tg)).se[idx];
        const uint32_t hix = atomicAdd(&out.counters[3], 1u);
        if (hix < out.hit_cap) {
          kr_hit h;
          h.read = read;
          h.kpos = base0 + tag_pos(tg);
          h.strand = tag_strand(tg);
          h.lib = tag_lib(tg);
          h.cmer_index = idx;
          h.hd = hd;
          h.se = (se >> 30) == 1u ? ix.leaf_se[se & kColMask] : (se & kColMask);
          out.hits[hix] = h;
        } else {
          atomicOr(&out.counters[1], kErrHitCap);
        }
      }
    }
    const uint64_t hm = __ballot(has);
    const uint32_t mark = seg != sw.cur_seg ? 1u : 0u; // wave-uniform
    if (item_reserve(out, sw, (uint32_t)__popcll(hm) + mark)) {
      if (mark && lane_id() == 0) out.items[sw.it_next] = make_uint2(seg, kItemMarker);
      sw.it_next += mark;
      sw.cur_seg = seg;
      if (has)
        out.items[sw.it_next + __popcll(hm & lt)] =
          make_uint2((uint32_t)idx, (tg & 0xFFFu) | (hd << 12) | ((uint32_t)(idx >> 32) << 17));
      sw.it_next += (uint32_t)__popcll(hm);
    }
    pend &= pend - 1u;
  }
}

// Scan of one group's probe list.  Step s of the scan = (pass p0, chunk block cb): lane l looks at chunk
// (l mod G) + cb*G of probe p0 + l/G and records in bit s of `hitbits` whether it holds an entry
// within th; stepinfo[s] (LDS, wave-uniform) remembers (p0, cb*G).  When the group is done -- or the
// bits are used up, for very long buckets -- the hit chunks are dealt out and resolved.
template <int LOG_G, int CPL, bool SL, bool TAP>
__device__ __forceinline__ void scan_group(const DevIndex& ix, const DevParams& P, const BatchOut& out, ScanWave& sw,
                                           const ProbeList& pl, lds_u32* queue, lds_u32* stepinfo, uint32_t nact, uint32_t read,
                                           uint32_t base0)
{
  constexpr uint32_t G = 1u << LOG_G, PPS = 64u >> LOG_G;
  constexpr uint32_t kBits = 64u / CPL * CPL; // steps per flush
  const uint32_t lane = lane_id(), sub = lane & (G - 1u);
  uint32_t p0 = 0, cb = 0; // scan position: pass start, chunk block (units of G chunks)
  while (p0 < nact) {
    // ---- scan until the group is done or the step bits are used up
    uint64_t hitbits = 0;
    uint32_t step = 0;
    while (p0 < nact && step < kBits) {
      const uint32_t pi = p0 + (lane >> LOG_G);
      const bool on = pi < nact;
      const uint64_t b = on ? pl.bkt[pi] : 0ull;
      const uint32_t q = on ? pl.q[pi] : 0u;
      const uint64_t st = b >> 24;
      const uint32_t tot = (uint32_t)(st & 3u) + (uint32_t)(b & 0xFFFFFFu);
      const uint32_t nch = on ? (tot + 3u) >> 2 : 0u;
      const uint32_t* enc = get_lib<SL>(ix, SL ? 0u : tag_lib(pl.tag[on ? pi : 0u])).enc + (st & ~3ull);
      for (; step < kBits && __ballot(sub + cb * G < nch) != 0; cb += CPL, step += CPL) {
        if (lane < (uint32_t)CPL) stepinfo[step + lane] = p0 | (((cb + lane) * G) << 8);
        // (loads stay masked for chunks past the end of the bucket: clamping them onto the last chunk instead
        //  saves instructions but costs 20 % in run time -- the extra lanes load the texture path)
        uint4 v[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const uint32_t c = sub + (cb + (uint32_t)j) * G;
          v[j] = make_uint4(0, 0, 0, 0);
          if (c < nch) v[j] = *reinterpret_cast<const uint4*>(enc + 4u * c);
        }
        // entries of the neighbouring buckets in the first / last chunk may give false positives, rejected below
        uint32_t r = 0;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const uint32_t m = min(min(hd_lr32(v[j].x, q), hd_lr32(v[j].y, q)), min(hd_lr32(v[j].z, q), hd_lr32(v[j].w, q)));
          r |= (m <= P.th && sub + (cb + (uint32_t)j) * G < nch) ? (1u << j) : 0u;
        }
        hitbits |= (uint64_t)r << step;
      }
      if (__ballot(sub + cb * G < nch) == 0) p0 += PPS, cb = 0; // pass finished
    }
    if (P.dbg & 1u) hitbits = 0;
    // ---- deal the hit chunks out, 64 at a time: re-read the chunk, apply the bucket bounds, emit
    const uint32_t cnt = (uint32_t)__popcll(hitbits);
    uint32_t inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d);
      if (lane >= (uint32_t)d) inc += o;
    }
    const uint32_t H = __shfl(inc, 63);
    uint64_t hb = hitbits;
    uint32_t slot = inc - cnt;
    WAVE_SYNC();
    for (uint32_t w0 = 0; w0 < H; w0 += 64) {
      while (__ballot(hb != 0 && slot < w0 + 64u) != 0) {
        if (hb != 0 && slot < w0 + 64u) {
          const uint32_t si = stepinfo[(uint32_t)__ffsll((long long)hb) - 1u];
          queue[slot - w0] = ((si & 255u) + (lane >> LOG_G)) | ((sub + (si >> 8)) << 8);
          hb &= hb - 1;
          ++slot;
        }
      }
      WAVE_SYNC();
      uint32_t pend = 0, hds = 0, tg = 0, c = 0;
      uint64_t e_al = 0;
      if (lane < min(64u, H - w0)) {
        const uint32_t d = queue[lane], probe = d & 255u;
        c = d >> 8;
        const uint64_t b = pl.bkt[probe];
        const uint32_t q = pl.q[probe];
        tg = pl.tag[probe];
        const uint64_t st = b >> 24;
        e_al = st & ~3ull;
        const int rel0 = (int)(st & 3u), tot = rel0 + (int)(uint32_t)(b & 0xFFFFFFu);
        const uint4 v = *reinterpret_cast<const uint4*>(get_lib<SL>(ix, tag_lib(tg)).enc + e_al + 4u * c);
        pend = chunk_hits(v, rel0 - 4 * (int)c, tot - 4 * (int)c, q, P.th, hds);
      }
      WAVE_SYNC();
      emit_hits<SL, TAP>(ix, out, sw, read, base0, tg, e_al, c, pend, hds);
    }
  }
}

// Scan of one group's probe list through the SLOTTED table (dense tables): row r owns an aligned slot of
// W = G*CPL*4 words = {bucket length, packed start index, the first W-2 residual codes}.  One pass reads the
// whole slot of 64/G probes -- no descriptor gather in front (one dependent HBM round trip and one
// L2->fabric request per probe less: the kernel runs at the chip's random-request rate), always the same
// two lines per probe.  The two header words take part in the hit test like entries and are rejected, like
// every entry beyond the bucket's length, when the hit chunks are resolved.  The rest of a bucket longer than
// W-2 entries is read from the packed array and resolved on the spot.
template <int LOG_G, int CPL, bool SL, bool TAP>
__device__ __forceinline__ void scan_group_slots(const DevIndex& ix, const DevParams& P, const BatchOut& out, ScanWave& sw,
                                                 const ProbeList& pl, lds_u32* queue, uint32_t nact, uint32_t read, uint32_t base0)
{
  constexpr uint32_t G = 1u << LOG_G, PPS = 64u >> LOG_G, W = G * CPL * 4u, CAP = W - 2u;
  static_assert((kListCap / PPS) * CPL <= 64, "one hit bit per (pass, chunk)");
  const uint32_t lane = lane_id(), sub = lane & (G - 1u);
  uint64_t hitbits = 0;
  uint32_t step = 0;
  for (uint32_t p0 = 0; p0 < nact; p0 += PPS, step += CPL) {
    const uint32_t pi = p0 + (lane >> LOG_G);
    const bool on = pi < nact;
    const uint32_t row = on ? (uint32_t)pl.bkt[pi] : 0u;
    const uint32_t q = on ? pl.q[pi] : 0u;
    const DevLib L = get_lib<SL>(ix, SL ? 0u : tag_lib(pl.tag[on ? pi : 0u]));
    const uint32_t* slot = L.slots + (uint64_t)row * W;
    uint4 v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      v[j] = make_uint4(0, 0, 0, 0);
      if (on) v[j] = *reinterpret_cast<const uint4*>(slot + 4u * (sub + (uint32_t)j * G));
    }
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const uint32_t m = min(min(hd_lr32(v[j].x, q), hd_lr32(v[j].y, q)), min(hd_lr32(v[j].z, q), hd_lr32(v[j].w, q)));
      r |= (m <= P.th && on) ? (1u << j) : 0u;
    }
    hitbits |= (uint64_t)r << step;
    // ---- buckets longer than the slot: the rest from the packed array, resolved on the spot
    if (__ballot(on && sub == 0 && v[0].x > CAP) != 0) {
      const uint32_t blen = __shfl(v[0].x, lane & ~(G - 1u)), bstart = __shfl(v[0].y, lane & ~(G - 1u));
      const bool lg = on && blen > CAP;
      const uint64_t st = (uint64_t)bstart + CAP;
      const uint64_t e_al = st & ~3ull;
      const int rel0 = (int)(st & 3u), tot = rel0 + (lg ? (int)(blen - CAP) : 0);
      const uint32_t nch = lg ? (uint32_t)(tot + 3) >> 2 : 0u;
      const uint32_t tg = pl.tag[on ? pi : 0u];
      for (uint32_t c = sub; __ballot(c < nch) != 0; c += G) {
        uint32_t pend = 0, hds = 0;
        if (c < nch) {
          const uint4 w = *reinterpret_cast<const uint4*>(L.enc + e_al + 4u * c);
          pend = chunk_hits(w, rel0 - 4 * (int)c, tot - 4 * (int)c, q, P.th, hds);
        }
        if (P.dbg & 1u) pend = 0;
        emit_hits<SL, TAP>(ix, out, sw, read, base0, tg, e_al, c, pend, hds);
      }
    }
  }
  if (P.dbg & 1u) hitbits = 0;
  // ---- deal the hit chunks out, 64 at a time: re-read header + chunk, apply the bucket bounds, emit
  const uint32_t cnt = (uint32_t)__popcll(hitbits);
  uint32_t inc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (lane >= (uint32_t)d) inc += o;
  }
  const uint32_t H = __shfl(inc, 63);
  uint64_t hb = hitbits;
  uint32_t slot_i = inc - cnt;
  for (uint32_t w0 = 0; w0 < H; w0 += 64) {
    while (__ballot(hb != 0 && slot_i < w0 + 64u) != 0) {
      if (hb != 0 && slot_i < w0 + 64u) {
        const uint32_t bit = (uint32_t)__ffsll((long long)hb) - 1u;
        const uint32_t pass = bit / (uint32_t)CPL, j = bit - pass * (uint32_t)CPL;
        queue[slot_i - w0] = (pass * PPS + (lane >> LOG_G)) | ((sub + j * G) << 8);
        hb &= hb - 1;
        ++slot_i;
      }
    }
    WAVE_SYNC();
    uint32_t pend = 0, hds = 0, tg = 0, c = 0;
    uint64_t e_al = 0;
    if (lane < min(64u, H - w0)) {
      const uint32_t d = queue[lane], probe = d & 255u;
      c = d >> 8;
      const uint32_t row = (uint32_t)pl.bkt[probe];
      const uint32_t q = pl.q[probe];
      tg = pl.tag[probe];
      const uint32_t* slot = get_lib<SL>(ix, tag_lib(tg)).slots + (uint64_t)row * W;
      const uint2 hdr = *reinterpret_cast<const uint2*>(slot);
      const uint4 v = *reinterpret_cast<const uint4*>(slot + 4u * c);
      // word 4c+e of the slot is entry 4c+e-2 of the bucket = packed index start + 4c+e-2
      const int nin = (int)min(hdr.x, CAP);
      pend = chunk_hits(v, 2 - 4 * (int)c, nin + 2 - 4 * (int)c, q, P.th, hds);
      e_al = (uint64_t)hdr.y - 2ull;
    }
    WAVE_SYNC();
    emit_hits<SL, TAP>(ix, out, sw, read, base0, tg, e_al, c, pend, hds);
  }
}

template <int LOG_G, int CPL, bool SL, bool TAP, bool SLOT>
__device__ __forceinline__ void scan_read(const DevIndex& ix, const DevParams& P, const BatchIn& in, const BatchOut& out,
                                          uint32_t read, ScanWave& sw, const ProbeList& pl, lds_u32* queue, lds_u32* stepinfo)
{
  const uint32_t lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint64_t off0 = in.offsets[read], off1 = in.offsets[read + 1];
  const uint8_t* seq = in.bases + off0;
  const uint64_t len = off1 - off0;
  const uint32_t k = ix.k;
  const uint64_t nkm = len >= k ? len - k + 1 : 0; // enmers (src/query.cpp:42)
  uint32_t onmers = 0;
  sw.rd_start = sw.it_next;
  sw.cur_seg = 0;
  sw.filt0 = 0xFFFFFFFFu, sw.filt1 = 0xFFFFFFFFu;
  for (uint64_t base0 = 0; base0 < nkm; base0 += kSegPos) {
    const uint32_t npos_seg = (uint32_t)min((uint64_t)kSegPos, nkm - base0);
    SegBits sb;
    load_segment(seq, len, base0, sb);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) { // unrolled: SegBits stays in scalar registers
      if (pp == 1 && npos_seg <= 64) break;
      uint32_t nv = 0;
      const Cand cur = fetch_group<SL, SLOT>(ix, sb, pp, npos_seg, nv);
      onmers += nv;
      // ---- compact the non-empty probes of this group into the LDS list (slotted: every located probe)
      const bool a0 = SLOT ? cur.b0 != 0 : (cur.b0 & 0xFFFFFFu) != 0, a1 = SLOT ? cur.b1 != 0 : (cur.b1 & 0xFFFFFFu) != 0;
      const uint64_t m0 = __ballot(a0), m1 = __ballot(a1);
      const uint32_t n0 = __popcll(m0), nact = n0 + __popcll(m1);
      if (a0) {
        const uint32_t i = __popcll(m0 & lt);
        pl.bkt[i] = cur.b0;
        pl.q[i] = cur.q0;
        pl.tag[i] = (64u * pp + lane) | (cur.lib0 << 8);
      }
      if (a1) {
        const uint32_t i = n0 + __popcll(m1 & lt);
        pl.bkt[i] = cur.b1;
        pl.q[i] = cur.q1;
        pl.tag[i] = (64u * pp + lane) | (1u << 7) | (cur.lib1 << 8);
      }
      WAVE_SYNC();
      if (!(P.dbg & 4u)) {
        if (SLOT)
          scan_group_slots<LOG_G, CPL, SL, TAP>(ix, P, out, sw, pl, queue, nact, read, (uint32_t)base0);
        else
          scan_group<LOG_G, CPL, SL, TAP>(ix, P, out, sw, pl, queue, stepinfo, nact, read, (uint32_t)base0);
      }
      WAVE_SYNC();
    }
  }
  // ---- per-strand hdist_filt = min hd over kept table entries (src/query.cpp:366-368)
  uint32_t f0 = sw.filt0, f1 = sw.filt1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    f0 = min(f0, (uint32_t)__shfl_xor(f0, d));
    f1 = min(f1, (uint32_t)__shfl_xor(f1, d));
  }
  if (lane == 0) {
    out.rd_it_off[read] = sw.rd_start;
    out.rd_it_cnt[read] = sw.it_next - sw.rd_start;
    out.rd_onmers[read] = onmers;
    out.rd_filt[2 * read] = f0;
    out.rd_filt[2 * read + 1] = f1;
  }
}

#ifndef KR_SCAN_WPE
#define KR_SCAN_WPE 6 // resident scan waves per SIMD the register allocation is sized for (8 spills, 5 hides less latency)
#endif
#ifndef KR_SCAN_WPE_SLOT
#define KR_SCAN_WPE_SLOT 4 // ... of the slotted variant (one more chunk per lane; 5: 6.0 ms, 6 spills: 7.3 ms, 4: 5.6 ms)
#endif
constexpr int kScanWaves = 4; // waves per workgroup of the scan kernel (they share nothing)
template <int LOG_G, int CPL, bool SL, bool TAP, bool SLOT>
__global__ __launch_bounds__(kScanWaves* kWave, (SLOT ? KR_SCAN_WPE_SLOT : KR_SCAN_WPE)) void kr_scan_kernel_t(DevIndex ix, DevParams P, BatchIn in, BatchOut out)
{
  __shared__ __attribute__((aligned(16))) uint64_t s_bkt[kScanWaves][kListCap];
  __shared__ uint32_t s_q[kScanWaves][kListCap], s_tag[kScanWaves][kListCap], s_queue[kScanWaves][64], s_step[kScanWaves][64];
  const uint32_t w = threadIdx.x / kWave;
  ProbeList pl{(lds_u64*)s_bkt[w], (lds_u32*)s_q[w], (lds_u32*)s_tag[w]};
  lds_u32* queue = (lds_u32*)s_queue[w];
  lds_u32* stepinfo = (lds_u32*)s_step[w];
  ScanWave sw;
  sw.it_next = sw.it_end = sw.rd_start = 0;
  sw.err = 0;
  sw.cur_seg = 0;
  sw.filt0 = sw.filt1 = 0xFFFFFFFFu;
  ReadCursor rc;
  rc.init(out.cursors, in.nreads, blockIdx.x * kScanWaves + w);
  uint32_t r0, r1;
  while (rc.next(r0, r1))
    for (uint32_t r = r0; r < r1; ++r) scan_read<LOG_G, CPL, SL, TAP, SLOT>(ix, P, in, out, r, sw, pl, queue, stepinfo);
  if (sw.err && lane_id() == 0) atomicOr(&out.counters[1], sw.err);
}

// fold the planes of one level-1 slot / level-2 slot into its running counts and zero them
__device__ __forceinline__ void fold_l1(const Acc& A, uint32_t s)
{
  uint32_t cum0 = 0, cum1 = 0, cum2 = 0, cum3 = 0;
  for (uint32_t x = 0; x < A.np; ++x) {
    lds_u32* p = &A.planes[(s * A.np + x) * kPlaneWords];
    uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
    uint32_t add = __popc(w0 & ~cum0) + __popc(w1 & ~cum1) + __popc(w2 & ~cum2) + __popc(w3 & ~cum3);
    cum0 |= w0, cum1 |= w1, cum2 |= w2, cum3 |= w3;
    if (w0 | w1 | w2 | w3) p[0] = 0, p[1] = 0, p[2] = 0, p[3] = 0;
    if (add) A.counts[s * A.np + x] += add;
  }
}
__device__ __forceinline__ void fold_l2(const Acc& A, uint32_t slot2)
{
  uint32_t cum0 = 0, cum1 = 0, cum2 = 0, cum3 = 0;
  for (uint32_t x = 0; x < A.np; ++x) {
    uint32_t* p = &A.g_planes[((uint64_t)slot2 * A.np + x) * kPlaneWords];
    uint32_t w0 = gload(p), w1 = gload(p + 1), w2 = gload(p + 2), w3 = gload(p + 3);
    uint32_t add = __popc(w0 & ~cum0) + __popc(w1 & ~cum1) + __popc(w2 & ~cum2) + __popc(w3 & ~cum3);
    cum0 |= w0, cum1 |= w1, cum2 |= w2, cum3 |= w3;
    if (w0) gstore(p, 0);
    if (w1) gstore(p + 1, 0);
    if (w2) gstore(p + 2, 0);
    if (w3) gstore(p + 3, 0);
    if (add) {
      uint32_t* cp = &A.g_counts[(uint64_t)slot2 * A.np + x];
      gstore(cp, gload(cp) + add);
    }
  }
}

// list of touched level-2 slots, ascending (= ascending key); returns its length
__device__ __forceinline__ uint32_t l2_build_list(const Acc& A, lds_u32* bitmap)
{
  const uint32_t lane = lane_id();
  uint32_t n = 0;
  for (uint32_t w0 = 0; w0 < A.bm_words; w0 += 64) {
    uint32_t wi = w0 + lane;
    uint32_t word = wi < A.bm_words ? bitmap[wi] : 0u;
    uint32_t c = __popc(word), inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(inc, d);
      if (lane >= (uint32_t)d) inc += t;
    }
    uint32_t o = n + inc - c;
    while (word) {
      uint32_t bit = (uint32_t)__ffs((int)word) - 1u;
      A.g_list[o++] = wi * 32u + bit;
      word &= word - 1u;
    }
    n += __shfl(inc, 63);
  }
  __syncthreads();
  return n;
}

__device__ __forceinline__ uint32_t hmin_l1(const Acc& A, uint32_t s)
{
  for (uint32_t x = 0; x < A.np; ++x)
    if (A.counts[s * A.np + x]) return x;
  return 0xFFFFFFFFu;
}
__device__ __forceinline__ uint32_t hmin_l2(const Acc& A, uint32_t slot2)
{
  for (uint32_t x = 0; x < A.np; ++x)
    if (gload(&A.g_counts[(uint64_t)slot2 * A.np + x])) return x;
  return 0xFFFFFFFFu;
}

// Record slots are handed out in wave-private chunks: one returning atomic on the shared counter
// per ~1000 records instead of one per read (a single word sustains only ~90 M atomics/s chip-wide,
// which is the read rate of the small-index configuration).  Unused slots at the end of a chunk stay
// zero (key 0 = hole; the record arrays are zeroed per batch).
constexpr uint32_t kRecChunk = 256;
constexpr uint32_t kEvSpill = 15360; // events of one read beyond the LDS buffer (60 KiB of global scratch per wave)
__device__ __forceinline__ uint32_t alloc_records(const BatchOut& out, WaveState& ws, uint32_t n)
{
  if (ws.rec_next + n > ws.rec_end) {
    uint32_t chunk = max(n, kRecChunk), base = 0;
    if (lane_id() == 0) base = atomicAdd(&out.counters[0], chunk);
    base = __shfl(base, 0);
    if ((uint64_t)base + chunk > out.rec_cap) {
      if (lane_id() == 0) atomicOr(&out.counters[1], kErrRecCap);
      return 0xFFFFFFFFu;
    }
    ws.rec_next = base;
    ws.rec_end = base + chunk;
  }
  uint32_t r = ws.rec_next;
  ws.rec_next += n;
  ws.n_rec += n;
  return r;
}

// ---------------------------------------------------------------------------
// Event mode epilogue.  No sort:
//  1. the keys (2 * leaf rank + strand) of the read's events are marked in the LDS bitmap; a popcount
//     prefix over the bitmap gives every key its ORDINAL among the read's keys (ascending key =
//     ascending colour id, strands adjacent);
//  2. every event is rewritten in place as  ordinal << 12 | pos << 5 | hd  and its key stored in
//     keytab[ordinal];
//  3. the per-position minimum of Minfo::update_match (src/query.hpp:153-176) is taken by visiting the
//     events in ascending hd LEVELS: an event sets bit `pos` of its key's 128-bit position map with a
//     returning LDS atomic OR, and only the event that finds the bit clear increments the key's
//     histogram counter of that level (8-bit counters, four per word: a segment has <= 128 positions);
//  4. keys that pass `hdist_min <= 2*hdist_filt+1` (src/query.cpp:101-106,119) are appended to a
//     compact table (counters + key) and written out at the end.
// Position maps and counters hold KB keys at a time (4 + hw words per key, all of the idle LDS behind
// the events: a single batch for all but the largest reads).  LDS use: [tab ... keytab] in the idle
// stack / probe-list / level-1 key region, [events | batch arrays] in the level-1 plane region.
// Returns false if the read does not fit (more keys than keytab holds, more passing keys than the
// table and its global spill hold): the caller falls back to the plane tables.
// ---------------------------------------------------------------------------
// (Combining lanes that OR into the same LDS word in registers first -- a segmented OR-scan on DPP -- was
//  measured: the serialisation it removes costs less than its ~45 instructions, +1..5 % run time.)
// Batch arrays of the event epilogue live in LDS, or -- for the rare read with more keys than the LDS
// holds -- in the wave's global scratch (L2): same code, memory operations by pointer type.
__device__ __forceinline__ void mem_or(lds_u32* p, uint32_t v) { lds_or(p, v); }
__device__ __forceinline__ void mem_or(uint32_t* p, uint32_t v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mem_st(lds_u32* p, uint32_t v) { *p = v; }
__device__ __forceinline__ void mem_st(uint32_t* p, uint32_t v) { gstore(p, v); }
__device__ __forceinline__ void mem_sync(lds_u32*) { WAVE_SYNC(); }
__device__ __forceinline__ void mem_sync(uint32_t*)
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // this wave's global atomics / stores have landed in L2
  WAVE_SYNC();
}

// One pass over the events whose key ordinal lies in [k0, k0 + kn): an event ORs its position bit into
// plane `hd` of its key (np planes of 128 bits per key; bit `pos >> 2` of word `pos & 3`, so that a run
// of neighbouring positions spreads over the four words).  The per-position minimum is taken afterwards,
// by the lane that owns the key: count[hd] = popc(plane[hd] & ~(plane[0] | ... | plane[hd-1])).
template <typename PT, typename EV>
__device__ __forceinline__ void plane_pass(PT bt, uint32_t np, uint32_t k0, uint32_t kn, uint32_t nev, EV ev_at)
{
  const uint32_t lane = lane_id();
  for (uint32_t i = lane; i < kn * np * kPlaneWords; i += 64) mem_st(bt + i, 0u);
  mem_sync(bt);
  for (uint32_t t0 = 0; t0 < nev; t0 += 64) {
    const uint32_t i = t0 + lane;
    if (i < nev) {
      const uint32_t v = ev_at(t0, i);
      const uint32_t o = (v >> 12) - k0;
      if (o < kn) {
        const uint32_t pos = (v >> 5) & 127u;
        mem_or(bt + (o * np + (v & 31u)) * kPlaneWords + (pos & 3u), 1u << (pos >> 2));
      }
    }
  }
  mem_sync(bt);
}
// packed 8-bit counters (four hd values per word) of key j of the batch
__device__ __forceinline__ void plane_counts(lds_u32* bt, uint32_t np, uint32_t j, uint32_t* c)
{
  u32x4 seen = {0, 0, 0, 0};
#pragma unroll
  for (int x = 0; x < kMaxPlanes; ++x)
    if ((uint32_t)x < np) {
      const u32x4 pw = *(KR_LDS u32x4*)(bt + (j * np + x) * kPlaneWords); // ds_read_b128
      c[x >> 2] += (uint32_t)(__popc(pw.x & ~seen.x) + __popc(pw.y & ~seen.y) + __popc(pw.z & ~seen.z) + __popc(pw.w & ~seen.w)) << (8 * (x & 3));
      seen |= pw;
    }
}
__device__ __forceinline__ void plane_counts(uint32_t* bt, uint32_t np, uint32_t j, uint32_t* c)
{
  uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
  for (int x = 0; x < kMaxPlanes; ++x)
    if ((uint32_t)x < np) {
      uint32_t* p = bt + (uint64_t)(j * np + x) * kPlaneWords;
      const uint32_t w0 = gload(p), w1 = gload(p + 1), w2 = gload(p + 2), w3 = gload(p + 3);
      c[x >> 2] += (uint32_t)(__popc(w0 & ~s0) + __popc(w1 & ~s1) + __popc(w2 & ~s2) + __popc(w3 & ~s3)) << (8 * (x & 3));
      s0 |= w0, s1 |= w1, s2 |= w2, s3 |= w3;
    }
}

__device__ __forceinline__ uint32_t key_ordinal(const Acc& A, uint32_t rs)
{
  const uint32_t blk = rs >> 6;
  const uint64_t bits = *(lds_u64*)(A.bitmap + 2u * blk);
  return (uint32_t)A.pre[blk] + __popcll(bits & ((1ull << (rs & 63u)) - 1ull));
}

// MERGE: one segment of a longer read -- every key is kept (the hdist_filt test needs the whole read), the
// segment's counts are added to the wave's global count table (disjoint positions: histograms add) and the key
// is noted in the read's bitmap; records are written by the caller when the read is complete.
template <bool MERGE>
__device__ __forceinline__ bool finalize_events(const DevIndex& ix, const BatchOut& out, const Acc& A, WaveState& ws,
                                                lds_u32* lo, uint32_t lo_words, uint32_t read, uint32_t onmers,
                                                uint32_t filt0, uint32_t filt1, uint32_t dbg)
{
  const uint32_t lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t nev = ws.nev;
  const uint32_t lim0 = 2u * filt0 + 1u, lim1 = 2u * filt1 + 1u; // u32 wrap keeps "none" = max
  const uint32_t hw = (A.np + 3u) >> 2; // histogram words per key
  const uint32_t ew = hw + 1u;          // table entry: packed counters, key
  const uint32_t kw = A.np * kPlaneWords; // batch words per key: np planes of 128 position bits
  lds_u32* e = ws.ev;
  uint32_t nrec = 0;
  bool fits = true;
  // event i of the read (a 64-aligned tile is entirely in LDS or entirely spilled)
  auto ev_at = [&](uint32_t t0, uint32_t i) -> uint32_t { return t0 < ws.ev_cap ? e[i] : gload(&ws.gev[i - ws.ev_cap]); };
  uint32_t* gtab = ws.gev + ws.gev_cap;
  if (nev) {
    if (nev > ws.ev_cap) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // spilled events are complete
    if (ws.lean) { // the bitmap aliases the work stack (empty now)
      for (uint32_t q = lane; q < A.bm_words; q += 64) A.bitmap[q] = 0;
      WAVE_SYNC();
    }
    // ---- 1. mark keys, ordinal prefix per 64-bit block
    for (uint32_t t0 = 0; t0 < nev; t0 += 64) {
      const uint32_t i = t0 + lane;
      // only keys that can pass `hdist_min <= 2*hdist_filt+1` at all: an event within the limit marks its key; the
      // events of unmarked keys are dropped in step 2 (a read with an exact match keeps a third of its keys)
      const uint32_t v = i < nev ? ev_at(t0, i) : 0xFFFFFFFFu;
      const uint32_t rs = v >> 12;
      if (i < nev && (MERGE || (v & 31u) <= ((rs & 1u) ? lim1 : lim0))) lds_or(&A.bitmap[rs >> 5], 1u << (rs & 31u));
    }
    WAVE_SYNC();
    uint32_t nkeys = 0;
    const uint32_t nblk = A.bm_words >> 1;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) {
      const uint32_t b = b0 + lane;
      uint32_t c = b < nblk ? __popcll(*(lds_u64*)(A.bitmap + 2u * b)) : 0u;
      uint32_t inc = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += o;
      }
      if (b < nblk) A.pre[b] = (uint16_t)(nkeys + inc - c);
      nkeys += __shfl(inc, 63);
    }
    WAVE_SYNC();
    // keytab[nkeys] behind the events if that leaves room for a batch of 8 keys, else in global scratch
    const uint32_t nev_lds = (min(nev, ws.ev_cap) + 1u) & ~1u;
    const bool kt_lds = nev_lds + nkeys + 4u + 8u * kw <= ws.ev_words;
    fits = kt_lds || nkeys <= ws.gkt_cap;
    lds_u32* keytab = e + nev_lds;
    uint32_t* gkt = gtab + (uint64_t)ws.gtab_cap * ew;
    const uint32_t tab_cap = lo_words / ew;
    // ---- 2. ordinals into the events, keys into keytab, events per key into kinfo
    //         (most keys of a read have ONE event -- relatives reached by a single k-mer -- and need no planes)
    lds_u32* kinfo = keytab + nkeys;
    // only where the plain planes would need more than one batch
    const bool sparse_try = kt_lds && nev_lds + 2u * nkeys + 4u <= ws.ev_words && !(dbg & 4096u) &&
                            nkeys > (ws.ev_words - ((nev_lds + nkeys + 3u) & ~3u)) / kw;
    if (sparse_try) {
      for (uint32_t o = lane; o < nkeys; o += 64) kinfo[o] = 0;
      WAVE_SYNC();
    }
    uint32_t lv = 0;
    if (fits)
      for (uint32_t t0 = 0; t0 < nev; t0 += 64) {
        const uint32_t i = t0 + lane;
        if (i < nev) {
          const uint32_t v = ev_at(t0, i), rs = v >> 12;
          const bool live = (A.bitmap[rs >> 5] >> (rs & 31u)) & 1u; // its key was marked
          const uint32_t o = key_ordinal(A, rs);
          const uint32_t nv = live ? (o << 12) | (v & 0xFFFu) : 0xFFFFFFFFu; // dead events match no batch
          if (t0 < ws.ev_cap)
            e[i] = nv;
          else
            gstore(&ws.gev[i - ws.ev_cap], nv);
          if (live) {
            if (kt_lds)
              keytab[o] = rs;
            else
              gstore(&gkt[o], rs);
            if (sparse_try) __hip_atomic_fetch_add(&kinfo[o], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          lv |= 1u << (v & 31u);
        }
      }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lv |= __shfl_xor(lv, d);
    if (nev > ws.ev_cap || !kt_lds) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    WAVE_SYNC();
    // the bitmap goes back to all-zero
    if (fits) {
      for (uint32_t o = lane; o < nkeys; o += 64) A.bitmap[(kt_lds ? keytab[o] : gload(&gkt[o])) >> 5] = 0;
    } else {
      for (uint32_t t0 = 0; t0 < nev; t0 += 64) {
        const uint32_t i = t0 + lane;
        if (i < nev) A.bitmap[ev_at(t0, i) >> 17] = 0;
      }
      nkeys = 0;
    }
    if (dbg & 64u) nkeys = 0;
    // ---- 3. batches of KB ordinals; a read with more than 16 LDS batches of keys runs as ONE batch in the
    //         wave's global scratch (A.g_planes, all-zero between reads like the plane path needs it):
    //         L2 atomics are ~10x slower than LDS ones, the batches re-read the events
    // kinfo[o] becomes: keys with several events -> their plane slot (consecutive); keys with one event ->
    // 0x80000000 | hd of that event (filled in by the plane pass)
    constexpr uint32_t kSingle = 0x80000000u;
    uint32_t nmulti = 0;
    if (sparse_try) {
      for (uint32_t j0 = 0; j0 < nkeys; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool multi = j < nkeys && kinfo[j] > 1u;
        const uint64_t mm = __ballot(multi);
        if (j < nkeys) kinfo[j] = multi ? nmulti + (uint32_t)__popcll(mm & lt) : (kSingle | 0x40u);
        nmulti += (uint32_t)__popcll(mm);
      }
      WAVE_SYNC();
    }
    const bool sparse = sparse_try && nkeys != 0 && ((nev_lds + 2u * nkeys + 3u) & ~3u) + nmulti * kw <= ws.ev_words;
    const uint32_t bt_off = (nev_lds + (kt_lds ? nkeys : 0u) + (sparse ? nkeys : 0u) + 3u) & ~3u; // 16-byte aligned
    lds_u32* bt = e + bt_off; // [KB][kw]
    const uint32_t KB_lds = sparse ? nkeys : (ws.ev_words - bt_off) / kw;
    const bool big = !sparse && nkeys > ((dbg & 2048u) ? 1u : 16u) * KB_lds && (uint64_t)nkeys * kw <= (uint64_t)A.nslots2 * A.np * kPlaneWords && !(dbg & 1024u);
    const uint32_t KB = big ? nkeys : KB_lds;
    for (uint32_t k0 = 0; k0 < nkeys; k0 += KB) {
      const uint32_t kn = min(KB, nkeys - k0);
      const uint64_t tl0 = (dbg & 512u) ? __builtin_readcyclecounter() : 0;
      if (!(dbg & 128u)) {
        if (sparse) { // one pass, planes for the keys with several events only
          for (uint32_t i = lane; i < nmulti * kw; i += 64) bt[i] = 0;
          WAVE_SYNC();
          for (uint32_t t0 = 0; t0 < nev; t0 += 64) {
            const uint32_t i = t0 + lane;
            if (i < nev) {
              const uint32_t v = ev_at(t0, i), o = v >> 12, ki = o < nkeys ? kinfo[o] : 0u;
              if (o >= nkeys) {
                // dead event
              } else if (ki & kSingle) {
                kinfo[o] = kSingle | (v & 31u);
              } else {
                const uint32_t pos = (v >> 5) & 127u;
                lds_or(bt + (ki * A.np + (v & 31u)) * kPlaneWords + (pos & 3u), 1u << (pos >> 2));
              }
            }
          }
          WAVE_SYNC();
        } else if (big)
          plane_pass(A.g_planes, A.np, k0, kn, nev, ev_at);
        else
          plane_pass(bt, A.np, k0, kn, nev, ev_at);
      }
      if ((dbg & 512u) && lane == 0) atomicAdd(&out.counters[16], (uint32_t)((__builtin_readcyclecounter() - tl0) >> 6));
      // ---- 4. one lane per key of the batch
      for (uint32_t j0 = 0; j0 < kn; j0 += 64) {
        const uint32_t j = j0 + lane;
        uint32_t c[kHistWords]; // packed 8-bit counters, four hd values per word
        bool ok = false;
        uint32_t rs = 0;
#pragma unroll
        for (int q = 0; q < kHistWords; ++q) c[q] = 0;
        if (j < kn) {
          if (sparse) {
            const uint32_t ki = kinfo[j];
            if (ki & kSingle) {
              const uint32_t hd1 = ki & 31u;
#pragma unroll
              for (int q = 0; q < kHistWords; ++q)
                if ((hd1 >> 2) == (uint32_t)q && !(ki & 0x40u)) c[q] = 1u << (8u * (hd1 & 3u));
            } else {
              plane_counts(bt, A.np, ki, c);
            }
          } else if (big)
            plane_counts(A.g_planes, A.np, j, c);
          else
            plane_counts(bt, A.np, j, c);
          rs = kt_lds ? keytab[k0 + j] : gload(&gkt[k0 + j]);
          uint32_t hmin = 0xFFFFFFFFu; // hdist_min = lowest hd with a non-zero counter
#pragma unroll
          for (int q = kHistWords - 1; q >= 0; --q)
            if (c[q]) hmin = 4u * q + ((uint32_t)(__ffs((int)c[q]) - 1) >> 3);
          ok = !MERGE && hmin <= ((rs & 1u) ? lim1 : lim0);
          if (MERGE && hmin != 0xFFFFFFFFu) { // this lane owns the key: plain read-modify-write
            for (uint32_t x = 0; x < A.np; ++x) {
              const uint32_t cnt = (c[x >> 2] >> (8u * (x & 3u))) & 255u;
              if (cnt) {
                uint32_t* cp = &A.g_counts[(uint64_t)rs * A.np + x];
                gstore(cp, gload(cp) + cnt);
              }
            }
            lds_or(&A.rbitmap[rs >> 5], 1u << (rs & 31u));
          }
        }
        const uint64_t okm = __ballot(ok);
        if (ok) {
          const uint32_t t = nrec + __popcll(okm & lt);
          if (t < tab_cap) {
#pragma unroll
            for (int q = 0; q < kHistWords; ++q)
              if ((uint32_t)q < hw) lo[t * ew + q] = c[q];
            lo[t * ew + hw] = rs;
          } else if (t - tab_cap < ws.gtab_cap) {
            uint32_t* g = gtab + (uint64_t)(t - tab_cap) * ew;
#pragma unroll
            for (int q = 0; q < kHistWords; ++q)
              if ((uint32_t)q < hw) gstore(&g[q], c[q]);
            gstore(&g[hw], rs);
          }
        }
        nrec += __popcll(okm);
      }
      WAVE_SYNC();
    }
    if ((dbg & 512u) && lane == 0) { // statistics for tuning
      atomicAdd(&out.counters[9], nev);
      atomicAdd(&out.counters[10], nkeys);
      atomicAdd(&out.counters[11], KB ? (nkeys + KB - 1) / KB : 0u);
      atomicAdd(&out.counters[12], big ? 1u : 0u);
      atomicAdd(&out.counters[13], (uint32_t)__popc(lv) * ((nev + 63) / 64));
      atomicMax(&out.counters[14], nkeys);
      atomicMax(&out.counters[15], nev);
    }
    if (big && nkeys) { // the plane path expects its global tables all-zero
      for (uint32_t i = lane; i < nkeys * kw; i += 64) gstore(&A.g_planes[i], 0u);
    }
    if (fits) {
      fits = nrec <= tab_cap + ws.gtab_cap;
      if (nrec > tab_cap) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    if (fits && !MERGE) {
      const uint32_t rbase = nrec ? alloc_records(out, ws, nrec) : 0u;
      if (lane == 0) {
        out.rd_off[read] = rbase == 0xFFFFFFFFu ? 0 : rbase;
        out.rd_cnt[read] = rbase == 0xFFFFFFFFu ? 0 : nrec;
      }
      if (nrec && rbase != 0xFFFFFFFFu && !(dbg & 256u)) {
        for (uint32_t t0 = 0; t0 < nrec; t0 += 64) {
          const uint32_t t = t0 + lane;
          if (t >= nrec) break;
          const uint32_t ri = rbase + t;
          const uint32_t* g = gtab + (uint64_t)(t - tab_cap) * ew;
          const bool in_lds = t < tab_cap;
          const uint32_t rs = in_lds ? lo[t * ew + hw] : gload(&g[hw]);
          out.rec_read[ri] = read;
          out.rec_key[ri] = (ix.leaf_se[rs >> 1] << 1) | (rs & 1u);
          uint64_t w0 = 0; // the likelihood problem in one word (kr_dedup_kernel): five 8-bit counts, #k-mers, bit 63
          for (uint32_t x = 0; x < A.np; ++x) {
            const uint32_t w = in_lds ? lo[t * ew + (x >> 2)] : gload(&g[x >> 2]);
            const uint32_t hv = (w >> (8u * (x & 3u))) & 255u;
            out.rec_hist[(uint64_t)x * out.rec_cap + ri] = hv;
            if (x < 5) w0 |= (uint64_t)hv << (8u * x);
          }
          out.rec_w0[ri] = (A.np == 5u && onmers < 65536u) ? (w0 | ((uint64_t)onmers << 40) | (1ull << 63)) : 0ull;
        }
      }
    }
  } else if (!MERGE && lane == 0) {
    out.rd_off[read] = 0;
    out.rd_cnt[read] = 0;
  }
  WAVE_SYNC();
  return fits;
}

// ---------------------------------------------------------------------------
// Kernel 2: accumulate.  One wave per read: the read's items (colour, tag) are expanded through the
// colour DAG to leaf events, the events are reduced to per-(leaf, strand) histograms and the
// records that pass the hdist_filt test are written out.
// ---------------------------------------------------------------------------
// fold this segment's planes into running counts (positions of different segments are distinct,
// so histograms add)
__device__ __forceinline__ void segment_fold(const Acc& A, WaveState& ws, bool& l2_any)
{
  WAVE_SYNC();
  if (A.keys[lane_id()]) fold_l1(A, lane_id()); // kLdsSlots == 64: lane t owns slot t
  l2_any = __ballot(ws.l2) != 0;
  if (l2_any) {
    const uint32_t n2 = l2_build_list(A, A.bitmap);
    for (uint32_t t = lane_id(); t < n2; t += 64) fold_l2(A, A.g_list[t]);
  }
  __syncthreads();
}

// MULTI: the instantiation for reads of several segments.  The single-segment instantiation only notes such
// reads in out.long_list (the kernel is launched a second time for them): with both in one kernel the merge code's
// registers cost the common case 5 %.
template <bool SL, bool MULTI>
__device__ __forceinline__ void process_read(const DevIndex& ix, const DevParams& P, const BatchIn& in,
                                             const BatchOut& out, uint32_t read, const Acc& A, WaveState& ws,
                                             lds_u32* hist_tbl, uint32_t hist_words)
{
  const uint32_t lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint64_t len = in.offsets[read + 1] - in.offsets[read];
  const uint32_t k = ix.k;
  const uint64_t nkm = len >= k ? len - k + 1 : 0; // enmers (src/query.cpp:42)
  const uint32_t nit = out.rd_it_cnt[read];
  const uint2* items = out.items + out.rd_it_off[read];
  const uint32_t onmers = out.rd_onmers[read];
  const uint32_t filt0 = out.rd_filt[2 * read], filt1 = out.rd_filt[2 * read + 1];
  bool l2_any = false; // wave-uniform: some key of this read lives in level 2
  const uint64_t tr0 = (P.dbg & 512u) ? __builtin_readcyclecounter() : 0;
  // Event mode: always for reads of one segment; reads of several segments run it per segment and merge the
  // segments' counts in the wave's global count table (debug bit 8192 sends them to the plane tables instead).
  // set aside for the second launch: list slots in wave-private chunks of 16 (one shared word serves ~90 M atomics/s)
  auto set_aside = [&]() {
    if (ws.ll_next == ws.ll_end) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&out.counters[25], 16u);
      ws.ll_next = __shfl(base, 0);
      ws.ll_end = ws.ll_next + 16u;
    }
    if (lane == 0) out.long_list[ws.ll_next] = read;
    ++ws.ll_next;
  };
  if (!MULTI && (nkm > (uint64_t)kSegPos || (P.dbg & 8u))) {
    set_aside();
    return;
  }
  const bool multi = MULTI;
  ws.evmode = !MULTI || (!(P.dbg & 8u) && !(P.dbg & 8192u));
  bool merged = false; // wave-uniform: the read's segments were merged through the count table
  // A read is processed once; only if its events do not fit (buffer or tables) is it processed a
  // second time with the plane tables.
  for (;;) {
  if (ws.evmode) {
    ws.dirty = true;
  } else if (MULTI && ws.dirty) { // the plane tables must start empty
    WAVE_SYNC();
    A.keys[lane] = 0;
    for (uint32_t i = lane; i < ws.ev_words; i += 64) A.planes[i] = 0;
    WAVE_SYNC();
    ws.dirty = false;
  }
  ws.top = 0;
  ws.l2 = false;
  ws.nev = 0;
  ws.ev_full = false;
  bool seg_ok = true; // every segment so far went through the event epilogue
  // the segment before a marker (or the last one) is complete
  auto segment_done = [&]() {
    if (MULTI && !ws.evmode) {
      segment_fold(A, ws, l2_any);
    } else if (multi) {
      seg_ok = seg_ok && !ws.ev_full &&
               finalize_events<true>(ix, out, A, ws, hist_tbl, hist_words, read, onmers, filt0, filt1, P.dbg);
      ws.nev = 0;
      ws.ev_full = false;
    }
  };
  for (uint32_t t0 = 0; t0 < ((P.dbg & 2u) ? 0u : nit); t0 += 64) {
    const uint32_t i = t0 + lane;
    const bool valid = i < nit;
    uint2 it = make_uint2(0, kItemMarker);
    if (valid) it = items[i];
    const bool marker = (it.y & kItemMarker) != 0;
    uint32_t se = 0;
    if (valid && !marker) // the colour of the table entry the scan kernel matched
      se = get_lib<SL>(ix, tag_lib(it.y)).se[(uint64_t)it.x | ((uint64_t)((it.y >> 17) & 0xFFu) << 32)];
    uint64_t todo = __ballot(valid);
    for (;;) { // markers (reads of more than one segment only) split the tile
      const uint64_t mk = __ballot(valid && marker) & todo;
      const uint64_t upto = mk ? (1ull << (__ffsll((long long)mk) - 1)) - 1ull : ~0ull;
      if (ws.top + 128u > ws.stack_cap) expand_all<SL>(ix, A, ws);
      expand_step<SL>(ix, A, ws, valid && !marker && ((todo & upto) >> lane & 1ull), se, it.y & 0x1FFFFu);
      if (mk == 0) break;
      expand_all<SL>(ix, A, ws);
      segment_done();
      todo &= ~(upto | (upto + 1ull));
    }
  }
  expand_all<SL>(ix, A, ws);
  if (MULTI) segment_done();
  if (ws.err && lane == 0) atomicOr(&out.counters[1], ws.err);
  ws.err = 0;
  if (MULTI && !ws.evmode) break;
  if (multi) {
    if (seg_ok) { // records from the merged counts, below
      merged = true;
      break;
    }
    // a segment did not fit: take back what the others added, then the plane tables
    const uint32_t n2 = l2_build_list(A, A.rbitmap);
    for (uint32_t t = lane; t < n2; t += 64)
      for (uint32_t x = 0; x < A.np; ++x) gstore(&A.g_counts[(uint64_t)A.g_list[t] * A.np + x], 0);
    for (uint32_t w = lane; w < A.bm_words; w += 64) A.rbitmap[w] = 0;
    __syncthreads();
    ws.evmode = false;
    continue;
  }
  if (P.dbg & 16u) ws.nev = 0, ws.ev_full = false;
  if ((P.dbg & 32u) && ws.ev_full) ws.nev = 0, ws.ev_full = false;
  const uint64_t tf0 = (P.dbg & 512u) ? __builtin_readcyclecounter() : 0;
  const bool fin_ok = !ws.ev_full && finalize_events<false>(ix, out, A, ws, hist_tbl, hist_words, read, onmers, filt0, filt1, P.dbg);
  if ((P.dbg & 512u) && lane == 0) {
    const uint64_t tf1 = __builtin_readcyclecounter();
    atomicAdd(&out.counters[17], (uint32_t)((tf1 - tf0) >> 6));
    atomicAdd(&out.counters[18], (uint32_t)((tf1 - tr0) >> 6));
  }
  if (fin_ok) return;
  if (!MULTI) { // does not fit: the second launch has the plane tables
    set_aside();
    return;
  }
  ws.evmode = false; // does not fit: redo the read with the plane tables
  } // redo loop
  if (!MULTI) return; // (unreachable: keeps the plane code out of this instantiation)
  // records that pass `hdist_min <= 2*hdist_filt+1` (src/query.cpp:101-106,119), ordered by key so
  // that the two strands of a leaf are adjacent
  const uint32_t lim0 = 2u * filt0 + 1u, lim1 = 2u * filt1 + 1u; // u32 wrap keeps "none" = max

  if (!l2_any && !merged) {
    // ---- level 1 only: lane t owns slot t
    const uint32_t key = A.keys[lane]; // (rank + 1) << 1 | strand: ascending key == ascending colour id
    const bool ok = key && hmin_l1(A, lane) <= ((key & 1u) ? lim1 : lim0);
    uint64_t okm = __ballot(ok);
    const uint32_t nrec = __popcll(okm);
    const uint32_t rbase = nrec ? alloc_records(out, ws, nrec) : 0u;
    if (lane == 0) {
      out.rd_off[read] = rbase == 0xFFFFFFFFu ? 0 : rbase;
      out.rd_cnt[read] = rbase == 0xFFFFFFFFu ? 0 : nrec;
    }
    uint32_t rank = 0; // number of passing keys smaller than mine
    while (okm) {
      int u = __ffsll((long long)okm) - 1;
      uint32_t kb = __shfl(key, u);
      rank += (kb < key) ? 1u : 0u;
      okm &= okm - 1;
    }
    if (ok && rbase != 0xFFFFFFFFu) {
      uint32_t ri = rbase + rank;
      out.rec_read[ri] = read;
      out.rec_key[ri] = (ix.leaf_se[(key >> 1) - 1u] << 1) | (key & 1u);
      uint64_t w0 = 0; // the likelihood problem in one word (kr_dedup_kernel), if the counts fit 8 bits
      bool fit = A.np == 5u && onmers < 65536u;
      for (uint32_t x = 0; x < A.np; ++x) {
        const uint32_t hv = A.counts[lane * A.np + x];
        out.rec_hist[(uint64_t)x * out.rec_cap + ri] = hv;
        fit = fit && hv < 256u;
        if (x < 5) w0 |= (uint64_t)(hv & 255u) << (8u * x);
      }
      out.rec_w0[ri] = fit ? (w0 | ((uint64_t)onmers << 40) | (1ull << 63)) : 0ull;
    }
    if (key) { // leave the slot empty for the next read
      A.keys[lane] = 0;
      for (uint32_t x = 0; x < A.np; ++x) A.counts[lane * A.np + x] = 0;
    }
    WAVE_SYNC();
    return;
  }

  // ---- level 2 in use (or segments merged through the count table): move the level-1 entries over, then emit
  //      from the sorted slot list
  lds_u32* const bitmap = merged ? A.rbitmap : A.bitmap;
  if (!merged) {
    const uint32_t key = A.keys[lane];
    if (key) {
      uint32_t slot2 = key - 2u;
      for (uint32_t x = 0; x < A.np; ++x) {
        uint32_t c = A.counts[lane * A.np + x];
        if (c) {
          uint32_t* cp = &A.g_counts[(uint64_t)slot2 * A.np + x];
          gstore(cp, gload(cp) + c);
          A.counts[lane * A.np + x] = 0;
        }
      }
      lds_or(&A.bitmap[slot2 >> 5], 1u << (slot2 & 31));
      A.keys[lane] = 0;
    }
    __syncthreads();
  }
  const uint32_t n2 = l2_build_list(A, bitmap);
  uint32_t nrec = 0;
  for (uint32_t t0 = 0; t0 < n2; t0 += 64) {
    uint32_t t = t0 + lane;
    bool ok = false;
    if (t < n2) {
      uint32_t slot2 = A.g_list[t];
      ok = hmin_l2(A, slot2) <= ((slot2 & 1u) ? lim1 : lim0);
    }
    nrec += __popcll(__ballot(ok));
  }
  if (!merged) ws.n_l2++;
  const uint32_t rbase = nrec ? alloc_records(out, ws, nrec) : 0u;
  if (lane == 0) {
    out.rd_off[read] = rbase == 0xFFFFFFFFu ? 0 : rbase;
    out.rd_cnt[read] = rbase == 0xFFFFFFFFu ? 0 : nrec;
  }
  uint32_t run = 0;
  for (uint32_t t0 = 0; t0 < n2; t0 += 64) {
    uint32_t t = t0 + lane;
    bool ok = false;
    uint32_t slot2 = 0;
    if (t < n2) {
      slot2 = A.g_list[t];
      ok = hmin_l2(A, slot2) <= ((slot2 & 1u) ? lim1 : lim0);
    }
    uint64_t okm = __ballot(ok);
    if (ok && rbase != 0xFFFFFFFFu) {
      uint32_t ri = rbase + run + __popcll(okm & lt);
      out.rec_read[ri] = read;
      out.rec_key[ri] = (ix.leaf_se[slot2 >> 1] << 1) | (slot2 & 1u);
      uint64_t w0 = 0;
      bool fit = A.np == 5u && onmers < 65536u;
      for (uint32_t x = 0; x < A.np; ++x) {
        const uint32_t hv = gload(&A.g_counts[(uint64_t)slot2 * A.np + x]);
        out.rec_hist[(uint64_t)x * out.rec_cap + ri] = hv;
        fit = fit && hv < 256u;
        if (x < 5) w0 |= (uint64_t)(hv & 255u) << (8u * x);
      }
      out.rec_w0[ri] = fit ? (w0 | ((uint64_t)onmers << 40) | (1ull << 63)) : 0ull;
    }
    if (t < n2)
      for (uint32_t x = 0; x < A.np; ++x) gstore(&A.g_counts[(uint64_t)slot2 * A.np + x], 0);
    run += __popcll(okm);
  }
  for (uint32_t w = lane; w < A.bm_words; w += 64) bitmap[w] = 0;
  __syncthreads();
}

// NP = th + 1 planes when known at compile time (5: --hdist-th default; the plane loops unroll and the address
// arithmetic folds), 0 = any threshold.
template <bool SL, int NP, bool MULTI>
__global__ __launch_bounds__(kWave, (MULTI ? 4 : 5)) void kr_acc_kernel_t(DevIndex ix, DevParams P, BatchIn in, BatchOut out)
{
  if (NP) P.np = NP, P.th = NP - 1;
  // One carve of dynamic LDS (base is 16-byte aligned: no static __shared__ in front):
  //   MULTI : stack | level-1 table (keys, planes, counts) | level-2 bitmap | read bitmap | ordinal prefix
  //   !MULTI: stack (the key bitmap and its prefix alias its start: the stack is empty in the epilogue)
  //           | table region for the passing keys (the level-1 key slots) | event / batch region
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  KR_LDS uint8_t* s_base = (KR_LDS uint8_t*)s_dyn;
  lds_u64* s_stack = (lds_u64*)s_base;
  const uint32_t stack_bytes = MULTI ? kStackCap * 8u : lean_stack_bytes(out.bm_words);
  lds_u32* s_tbl = (lds_u32*)(s_base + stack_bytes);

  Acc A;
  A.np = P.np;
  A.keys = s_tbl;
  A.planes = A.keys + kLdsSlots;
  A.counts = A.planes + kLdsSlots * P.np * kPlaneWords;
  A.bitmap = MULTI ? A.counts + kLdsSlots * P.np : (lds_u32*)s_base;
  A.rbitmap = A.bitmap + out.bm_words; // MULTI only
  A.pre = (lds_u16*)((MULTI ? A.rbitmap : A.bitmap) + out.bm_words);
  A.nslots2 = out.nslots2;
  A.bm_words = out.bm_words;
  const uint64_t w = blockIdx.x;
  A.g_planes = out.g_planes + w * (uint64_t)out.nslots2 * P.np * kPlaneWords;
  A.g_counts = out.g_counts + w * (uint64_t)out.nslots2 * P.np;
  A.g_list = out.g_list + w * (uint64_t)out.g_list_words;
  if (MULTI)
  { // tables start empty; every read leaves them empty again
    const uint32_t lane = lane_id();
    A.keys[lane] = 0;
    for (uint32_t x = 0; x < P.np; ++x) {
      A.counts[lane * P.np + x] = 0;
#pragma unroll
      for (int q = 0; q < kPlaneWords; ++q) A.planes[(lane * P.np + x) * kPlaneWords + q] = 0;
    }
    for (uint32_t q = lane; q < A.bm_words; q += 64) A.bitmap[q] = 0, A.rbitmap[q] = 0;
  }
  __syncthreads();
  WaveState ws;
  ws.stack = s_stack;
  ws.stack_cap = stack_bytes / 8u;
  ws.lean = !MULTI;
  ws.top = 0;
  ws.gstk = out.stk_spill + w * (uint64_t)kStackSpill;
  ws.gs_top = 0;
  ws.l2 = false;
  ws.err = 0;
  ws.rec_next = 0;
  ws.rec_end = 0;
  ws.n_l2 = 0;
  ws.n_rec = 0;
  ws.n_spill = 0;
  ws.ll_next = ws.ll_end = 0;
  ws.evmode = false;
  ws.nev = 0;
  ws.ev = A.planes; // planes + counts are contiguous: kLdsSlots * np * 5 words
  ws.ev_cap = 64;
  while (ws.ev_cap * 2 <= (uint32_t)kLdsSlots * P.np * (kPlaneWords + 1)) ws.ev_cap <<= 1;
  if (ws.ev_cap == (uint32_t)kLdsSlots * P.np * (kPlaneWords + 1)) ws.ev_cap >>= 1;
  ws.ev_words = (uint32_t)kLdsSlots * P.np * (kPlaneWords + 1);
  ws.dirty = false;
  ws.gev = A.g_list;
  ws.gev_cap = out.ev_spill;
  ws.gtab_cap = out.tab_spill;
  ws.gkt_cap = out.kt_spill;
  ReadCursor rc;
  // second launch: the reads the first one set aside (their number is final: kernel boundary)
  rc.init(out.cursors + (MULTI ? 2u : 1u) * kCursors * kCursorStride, MULTI ? out.counters[25] : in.nreads, blockIdx.x);
  uint32_t r0, r1;
  while (rc.next(r0, r1))
    for (uint32_t r = r0; r < r1; ++r)
    {
      const uint32_t rd = MULTI ? out.long_list[r] : r;
      if (rd != 0xFFFFFFFFu) process_read<SL, MULTI>(ix, P, in, out, rd, A, ws, (lds_u32*)s_base, stack_bytes / 4u + kLdsSlots); // else: unused list slot
    }
  for (uint32_t q = ws.ll_next + lane_id(); q < ws.ll_end; q += 64) out.long_list[q] = 0xFFFFFFFFu;
  if (ws.n_l2 && lane_id() == 0) atomicAdd(&out.counters[2], ws.n_l2);
  if (ws.n_rec && lane_id() == 0) atomicAdd(&out.counters[4], ws.n_rec);
  if (ws.n_spill && lane_id() == 0) atomicAdd(&out.counters[26], ws.n_spill);
}

// ---------------------------------------------------------------------------
// Likelihood (HDistHistLLH::operator(), src/hdhistllh.hpp:71-89) and Brent
// ---------------------------------------------------------------------------
struct LlhProblem {
  double mc[kMaxPlanes];
  double uc, rho;
};

// x^n for a small positive integer n in double-double arithmetic (error-free products through fma).
// The chain carries ~100 bits, so the rounded result is the correctly rounded power except in
// near-tie cases: the contract of glibc's pow, which the reference calls at src/hdhistllh.hpp:74,
// and tighter (and ~3x cheaper) than the general-purpose device pow.
struct DD {
  double hi, lo;
};
__device__ __forceinline__ DD dd_mul(DD a, DD b)
{
  double p = a.hi * b.hi;
  double e = fma(a.hi, b.hi, -p);
  e = fma(a.hi, b.lo, e);
  e = fma(a.lo, b.hi, e);
  double s = p + e;
  return DD{s, e - (s - p)};
}
__device__ __forceinline__ double pown_dd(double x, uint32_t n)
{
  DD r{1.0, 0.0}, b{x, 0.0};
  while (n) {
    if (n & 1u) r = dd_mul(r, b);
    n >>= 1;
    if (n) b = dd_mul(b, b);
  }
  return r.hi + r.lo;
}

// Natural logarithm of a positive normal double, error < 1 ulp: the argument reduction and the degree-14
// minimax polynomial of the classic freely distributable libm `log` (x = 2^k (1+f), s = f/(2+f),
// log(1+f) = f - s (f - R(s^2)), k ln2 added in two pieces), restated without its special cases -- the
// likelihood only takes logs of numbers in (1e-300, 1].  40 instructions against the 75 of the general
// device `log`, and the objective takes three per evaluation (55 % of its instructions before).
__device__ __attribute__((noinline)) double kr_log_general(double x) { return log(x); }
__device__ __forceinline__ double kr_log(double x)
{
#ifdef KR_OCML_LOG
  return log(x);
#endif
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return kr_log_general(x); // zero, subnormal, negative, inf, nan
  int32_t hx = __double2hiint(x);
  const uint32_t lx = (uint32_t)__double2loint(x);
  int32_t k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  x = __hiloint2double(hx | (i ^ 0x3ff00000), (int32_t)lx); // normalise x or x/2
  k += i >> 20;
  const double f = x - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s;
  i = hx - 0x6147a;
  const double w = z * z;
  const int32_t j = 0x6b851 - hx;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  const double R = t2 + t1;
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

typedef KR_LDS double lds_f64;
struct LlhTables { // per-workgroup copies of the binomial tables (uniform LDS reads, no scalar-load stalls)
  lds_f64* bk;  // [k+1]
  lds_f64* hnk; // [th+1]
};
__device__ __forceinline__ void llh_tables_init(const LlhConst& C, lds_f64* bk, lds_f64* hnk)
{
  for (uint32_t i = threadIdx.x; i <= C.k; i += blockDim.x) bk[i] = C.binom_k[i];
  for (uint32_t i = threadIdx.x; i <= C.th; i += blockDim.x) hnk[i] = C.binom_hnk[i];
  __syncthreads();
}

// HDistHistLLH::operator() (src/hdhistllh.hpp:71-89), same operation order.  NPT = th+1 when known
// at compile time (histogram in registers), 0 = any th.
// The objective splits into a part that depends on d alone -- (1-d)^k, two logs, the (k+1)-term sum lv_m:
// 85 % of the work -- and a cheap combination with the record's histogram.  Same operations, same order,
// as the single loop of the reference (the two accumulators never meet before the last line).
struct LlhShared {
  double logdn, logdp, lv_m; // k log(1-d), log(d) - log(1-d), sum of the likelihood weights
};
template <int NPT>
__device__ __forceinline__ LlhShared llh_dpart(const LlhConst& C, const LlhTables& T, double d)
{
  LlhShared g;
  double lv_m = 0.0;
  double powdc = (C.dbg & 1u) ? pow(1.0 - d, (double)C.k) : pown_dd(1.0 - d, C.k);
  g.logdn = kr_log(1.0 - d);
  g.logdp = kr_log(d) - g.logdn;
  g.logdn *= (double)C.k;
  const double dratio = d / (1.0 - d);
  if (NPT > 0) {
#pragma unroll
    for (int x = 0; x < NPT; ++x) {
      lv_m += T.hnk[x] * powdc;
      powdc *= dratio;
    }
  } else {
    for (uint32_t x = 0; x <= C.th; ++x) {
      lv_m += T.hnk[x] * powdc;
      powdc *= dratio;
    }
  }
  // (a fully unrolled k = 29 tail with the binomials in SGPRs was tried: 37 SGPR spills, 2 % slower)
#pragma unroll 4
  for (uint32_t x = C.th + 1; x <= C.k; ++x) {
    lv_m += powdc * T.bk[x];
    powdc *= dratio;
  }
  g.lv_m = lv_m;
  return g;
}
template <int NPT>
__device__ __forceinline__ double llh_combine(const LlhConst& C, const LlhShared& g, const LlhProblem& p)
{
  double sum = 0.0;
  if (NPT > 0) {
#pragma unroll
    for (int x = 0; x < NPT; ++x) sum -= (g.logdn + (double)x * g.logdp) * p.mc[x];
  } else {
    for (uint32_t x = 0; x <= C.th; ++x) sum -= (g.logdn + (double)x * g.logdp) * p.mc[x];
  }
  return sum - kr_log(p.rho * g.lv_m + 1.0 - p.rho) * p.uc;
}
template <int NPT>
__device__ __forceinline__ double llh_eval(const LlhConst& C, const LlhTables& T, const LlhProblem& p, double d)
{
  return llh_combine<NPT>(C, llh_dpart<NPT>(C, T, d), p);
}

// boost::math::tools::brent_find_minima(f, 1e-10, 0.5, 16) (src/query.cpp:430);
// published algorithm, see SURVEY.md Appendix B.  The minimisation is written as a resumable state
// machine -- one objective evaluation per step() -- so that the record kernel can hand a lane a new
// record the moment its minimisation converges (the number of evaluations varies 10..23 between
// records; a wave that waits for its slowest lane idles 40 % of its lanes).
struct BrentState {
  double x, w, v, fx, fw, fv, mn, mx, delta, delta2;
  int it;
};
constexpr double kBrentTol = 0x1p-15; // ldexp(1, 1 - min(53/2, 16))

// Before an evaluation: false if converged (result is x, fx), else the next abscissa in u.
__device__ __forceinline__ bool brent_next(BrentState& s, double& u)
{
  const double golden = 0.3819660f;
  const double mid = (s.mn + s.mx) / 2;
  const double fract1 = kBrentTol * fabs(s.x) + kBrentTol / 4;
  const double fract2 = 2 * fract1;
  if (s.it >= 1000 || fabs(s.x - mid) <= (fract2 - (s.mx - s.mn) / 2)) return false;
  ++s.it;
  if (fabs(s.delta2) > fract1) {
    double r = (s.x - s.w) * (s.fx - s.fv);
    double q = (s.x - s.v) * (s.fx - s.fw);
    double pp = (s.x - s.v) * q - (s.x - s.w) * r;
    q = 2 * (q - r);
    if (q > 0) pp = -pp;
    q = fabs(q);
    double td = s.delta2;
    s.delta2 = s.delta;
    if ((fabs(pp) >= fabs(q * td / 2)) || (pp <= q * (s.mn - s.x)) || (pp >= q * (s.mx - s.x))) {
      s.delta2 = (s.x >= mid) ? s.mn - s.x : s.mx - s.x;
      s.delta = golden * s.delta2;
    } else {
      s.delta = pp / q;
      u = s.x + s.delta;
      if (((u - s.mn) < fract2) || ((s.mx - u) < fract2)) s.delta = (mid - s.x) < 0 ? -fabs(fract1) : fabs(fract1);
    }
  } else {
    s.delta2 = (s.x >= mid) ? s.mn - s.x : s.mx - s.x;
    s.delta = golden * s.delta2;
  }
  u = (fabs(s.delta) >= fract1) ? (s.x + s.delta) : (s.delta > 0 ? (s.x + fabs(fract1)) : (s.x - fabs(fract1)));
  return true;
}
// First evaluation (at the upper bracket end).
__device__ __forceinline__ void brent_start(BrentState& s, double u, double fu)
{
  s.mn = 1e-10, s.mx = 0.5;
  s.x = s.w = s.v = u;
  s.fw = s.fv = s.fx = fu;
  s.delta2 = s.delta = 0;
  s.it = 0;
}
// After an evaluation at u.
__device__ __forceinline__ void brent_update(BrentState& s, double u, double fu)
{
  if (fu <= s.fx) {
    if (u >= s.x)
      s.mn = s.x;
    else
      s.mx = s.x;
    s.v = s.w, s.w = s.x, s.x = u;
    s.fv = s.fw, s.fw = s.fx, s.fx = fu;
  } else {
    if (u < s.x)
      s.mn = u;
    else
      s.mx = u;
    if ((fu <= s.fw) || (s.w == s.x)) {
      s.v = s.w, s.w = u;
      s.fv = s.fw, s.fw = fu;
    } else if ((fu <= s.fv) || (s.v == s.x) || (s.v == s.w)) {
      s.v = u;
      s.fv = fu;
    }
  }
}

template <int NPT>
__device__ __forceinline__ void brent_min(const LlhConst& C, const LlhTables& T, const LlhProblem& p, double& d_out, double& v_out)
{
  BrentState s;
  double u = 0.5;
  brent_start(s, u, llh_eval<NPT>(C, T, p, u));
  while (brent_next(s, u)) brent_update(s, u, llh_eval<NPT>(C, T, p, u));
  d_out = s.x;
  v_out = s.fx;
}

template <int NPT>
__device__ __forceinline__ void load_problem(const LlhConst& C, const uint32_t* hist, uint64_t stride, uint32_t onmers,
                                             double rho, LlhProblem& p)
{ // hist[x] of this record at hist[x * stride]
  uint32_t mc = 0;
  if (NPT > 0) {
#pragma unroll
    for (int x = 0; x < NPT; ++x) {
      uint32_t hv = hist[(uint64_t)x * stride];
      p.mc[x] = (double)hv;
      mc += hv;
    }
  } else {
    for (uint32_t x = 0; x <= C.th; ++x) {
      uint32_t hv = hist[(uint64_t)x * stride];
      p.mc[x] = (double)hv;
      mc += hv;
    }
  }
  p.uc = (double)onmers - (double)mc; // mismatch_count = onmers - match_count (src/query.cpp:104)
  p.rho = rho;
}

// ---------------------------------------------------------------------------
// Likelihood de-duplication.  The ML distance of a record is a pure function of (leaf -> rho, histogram,
// number of k-mers of the read), and a batch repeats the same few problems over and over -- a reference
// reached by one k-mer at Hamming distance 1 from a 150-bp read is THE most common record.  kr_dedup_kernel
// finds the distinct problems of the batch with an open-addressing table in HBM (slot = 64-bit histogram
// word claimed by CAS + 64-bit {leaf, list position}), the likelihood kernels run on the distinct ones, and
// kr_llh_copy_kernel hands the result to the duplicates.  Nothing is kept between batches.  Records the
// 64-bit word cannot describe (th != 4, a count above 255, more than 65535 k-mers) are their own problem.  The word is packed by kr_acc_kernel as it writes the record.
// ---------------------------------------------------------------------------
constexpr uint32_t kRepChunk = 16;
__device__ __forceinline__ uint32_t dd_mask(const BatchOut& out)
{ // table slots used for this batch: a power of two >= (number of record slots) >> dd_shift, at most dd_slots
  const uint32_t n = min(out.counters[0], out.rec_cap);
  uint32_t m = 1024;
  while (m < out.dd_slots && m < (n >> out.dd_shift)) m <<= 1;
  return m - 1u;
}
__global__ __launch_bounds__(256) void kr_dedup_clear_kernel(BatchOut out)
{
  const uint32_t n = dd_mask(out) + 1u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out.dd_table[i] = make_ulonglong2(0, 0);
}
__global__ __launch_bounds__(256) void kr_dedup_kernel(BatchOut out)
{
  const uint32_t nrec = min(out.counters[0], out.rec_cap);
  const uint32_t mask = dd_mask(out);
  unsigned long long* tab = reinterpret_cast<unsigned long long*>(out.dd_table);
  // Once the table is half full a wave stops inserting (look-ups only; new problems stand alone).  The wave learns
  // it from the list positions its own insertions get: a shared flag polled by every wave was measured at 3x the
  // kernel's run time -- one word read by the whole chip serialises on its L2 channel.
  bool crowded = false;
  // list positions come from wave-private chunks of kRepChunk (one shared-counter atomic per chunk: distinct
  // problems are ~5 % of the records, one winner per wave and round, and a single word serves ~90 M atomics/s);
  // the unused tail of a wave's last chunk is marked as holes
  uint32_t lp_next = 0, lp_end = 0;
  auto take_positions = [&](uint32_t n) -> uint32_t { // wave-uniform
    if (lp_next + n > lp_end) {
      for (uint32_t q = lp_next + lane_id(); q < lp_end; q += 64) out.rep_list[q] = 0xFFFFFFFFu;
      const uint32_t size = max(n, kRepChunk);
      uint32_t base = 0;
      if (lane_id() == 0) base = atomicAdd(&out.counters[22], size);
      base = __shfl(base, 0);
      lp_next = base, lp_end = base + size;
    }
    const uint32_t r = lp_next;
    lp_next += n;
    return r;
  };
  for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < nrec; i0 += gridDim.x * blockDim.x) { // wave-uniform trip count
    const uint32_t i = i0 + threadIdx.x;
    const uint32_t key = i < nrec ? out.rec_key[i] : 0u;
    bool todo = key != 0; // 0 = hole at the end of a wave's record chunk
    bool own = false;     // becomes its own problem without the table
    uint64_t w0 = 0;
    uint32_t pos = 0xFFFFFFFFu;
    const uint32_t se = key >> 1;
    if (todo) {
      w0 = out.rec_w0[i];
      own = w0 == 0;
    }
    uint32_t slot = (uint32_t)((w0 * 0x9E3779B97F4A7C15ull) >> 32) ^ (se * 0x85EBCA6Bu);
    // Retry loop without an inner spin: a lane that finds its histogram word in a slot whose second word is not
    // published yet simply comes round again (the publishing lane may be in this very wave).
    const uint64_t lt = (1ull << lane_id()) - 1ull;
    for (int it = 0; it < 48 && __ballot(todo && !own) != 0; ++it) {
      bool won = false;
      unsigned long long old = 1ull;
      if (todo && !own) {
        slot &= mask;
        // most records find their problem already there: look before the (much slower) atomic
        old = __hip_atomic_load(&tab[2ull * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0ull) {
          if (crowded) {
            own = true; // not in the table and no room for it
            old = 1ull;
          } else {
            old = atomicCAS(&tab[2ull * slot], 0ull, (unsigned long long)w0);
          }
        }
        won = old == 0ull; // claimed: this record represents the problem
      }
      // list positions for the winners of this round: ONE atomic per wave (a single word serves ~90 M atomics/s)
      const uint64_t wm = __ballot(won);
      if (wm != 0) {
        const uint32_t base = take_positions((uint32_t)__popcll(wm));
        crowded = crowded || base > (mask >> 1);
        if (won) {
          pos = base + (uint32_t)__popcll(wm & lt);
          out.rep_list[pos] = i;
          __hip_atomic_store(&tab[2ull * slot + 1], (unsigned long long)se | ((unsigned long long)(pos + 1u) << 32), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT); // nothing else is published through it in this kernel
          todo = false;
        }
      }
      if (todo && !own && !won) {
        if (old == (unsigned long long)w0) {
          const unsigned long long w1 = __hip_atomic_load(&tab[2ull * slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(w1 >> 32) != 0u) {
            if ((uint32_t)w1 == se) { // the same problem
              pos = (uint32_t)(w1 >> 32) - 1u;
              todo = false;
            } else {
              ++slot; // same histogram, another leaf
            }
          } // else: not published yet, same slot again
        } else {
          ++slot;
        }
      }
    }
    { // not describable, or the table is crowded: its own problem
      const uint64_t om = __ballot(todo);
      if (om != 0) {
        const uint32_t base = take_positions((uint32_t)__popcll(om));
        if (todo) {
          pos = base + (uint32_t)__popcll(om & lt);
          out.rep_list[pos] = i;
        }
      }
    }
    if (i < nrec) out.rec_rep[i] = pos;
  }
  for (uint32_t q = lp_next + lane_id(); q < lp_end; q += 64) out.rep_list[q] = 0xFFFFFFFFu;
}
__global__ __launch_bounds__(256) void kr_llh_copy_kernel(BatchOut out)
{
  const uint32_t nrec = min(out.counters[0], out.rec_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += gridDim.x * blockDim.x) {
    const uint32_t pos = out.rec_rep[i];
    if (pos == 0xFFFFFFFFu) continue;
    const double2 dv = out.rep_dv[pos]; // every record takes its value from the table of distinct problems
    out.rec_d[i] = dv.x;
    out.rec_v[i] = dv.y;
  }
}

// Every minimisation starts with the same abscissas: the upper bracket end 0.5, the golden-section point below
// it, and -- two coincident points make the parabolic step degenerate (p = q = 0 exactly) -- a second
// golden-section point that depends only on WHICH of the first two values is smaller.  kr_llh_pre_kernel
// evaluates the d-part of the objective at those four points ONCE per workgroup and combines it with every
// record's histogram (one log per evaluation instead of a full evaluation): 3 of the ~11.6 evaluations of a
// record leave the divergent main loop.  The three values travel in rec_d / rec_v / rec_chisq, which the later
// kernels overwrite with their results.
__device__ __forceinline__ void brent_shared_points(double& u1, double& u2a, double& u2b)
{
  BrentState s;
  double u = 0.0;
  brent_start(s, 0.5, 1.0);
  brent_next(s, u1); // golden section from the upper end: no objective value enters
  BrentState sa = s, sb = s;
  brent_update(sa, u1, 0.0); // f(u1) <= f(0.5)
  brent_next(sa, u2a);
  brent_update(sb, u1, 2.0); // f(u1) > f(0.5)
  brent_next(sb, u2b);
  (void)u;
}
template <int NPT>
__global__ __launch_bounds__(256) void kr_llh_pre_kernel(LlhConst C, DevIndex ix, BatchOut out)
{
  __shared__ double s_bk[32], s_hnk[kMaxPlanes], s_g[4][3];
  LlhTables T{(lds_f64*)s_bk, (lds_f64*)s_hnk};
  llh_tables_init(C, T.bk, T.hnk);
  double u1, u2a, u2b;
  brent_shared_points(u1, u2a, u2b);
  if (threadIdx.x < 4) {
    const double d = threadIdx.x == 0 ? 0.5 : (threadIdx.x == 1 ? u1 : (threadIdx.x == 2 ? u2a : u2b));
    const LlhShared g = llh_dpart<NPT>(C, T, d);
    s_g[threadIdx.x][0] = g.logdn, s_g[threadIdx.x][1] = g.logdp, s_g[threadIdx.x][2] = g.lv_m;
  }
  __syncthreads();
  const LlhShared g0{s_g[0][0], s_g[0][1], s_g[0][2]}, g1{s_g[1][0], s_g[1][1], s_g[1][2]};
  const uint32_t nrep = out.counters[22];
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nrep; j += gridDim.x * blockDim.x) {
    const uint32_t i = out.rep_list[j];
    if (i == 0xFFFFFFFFu) continue; // hole at the end of a wave's chunk of list positions
    const uint32_t key = out.rec_key[i];
    LlhProblem p;
    load_problem<NPT>(C, out.rec_hist + i, out.rec_cap, out.rd_onmers[out.rec_read[i]], ix.libs[0].rho[key >> 1], p);
    const double f0 = llh_combine<NPT>(C, g0, p), f1 = llh_combine<NPT>(C, g1, p);
    const int w = f1 <= f0 ? 2 : 3; // the branch brent_update takes
    const LlhShared g2{s_g[w][0], s_g[w][1], s_g[w][2]};
    out.rec_d[i] = f0;
    out.rec_v[i] = f1;
    out.rec_chisq[i] = llh_combine<NPT>(C, g2, p);
  }
}

// One record per lane, refilled: a lane whose minimisation has converged stores its result and, once
// kLlhRefill lanes are idle, the idle lanes take the next records of the wave's current chunk (chunks of
// kLlhChunk records are handed out through counters[5]) together with their first two objective values.
// Every step evaluates the objective once for all busy lanes.
#ifndef KR_LLH_REFILL
#define KR_LLH_REFILL 8
#endif
#ifndef KR_LLH_WPE
#define KR_LLH_WPE 4
#endif
constexpr uint32_t kLlhChunkMax = 2048, kLlhRefill = KR_LLH_REFILL;
template <int NPT>
__device__ __forceinline__ void llh_records(const LlhConst& C, const LlhTables& T, const DevIndex& ix, const BatchOut& out)
{
  const uint32_t nrec = out.counters[22]; // distinct problems (rep_list)
  const uint32_t lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  // chunk size: about four chunks per wave, 64 .. kLlhChunkMax records
  const uint32_t kLlhChunk = min(kLlhChunkMax, max(64u, (nrec / (gridDim.x * (blockDim.x / kWave) * 4u)) & ~63u));
  uint32_t next = 0, end = 0; // wave-uniform cursor into the current chunk
  bool more = true;           // wave-uniform: chunks may remain
  bool busy = false;
  uint32_t rec = 0, pos = 0;
  LlhProblem p;
  BrentState s;
  for (;;) {
    double u = 0.5;
    bool has_u = false;
    if (busy) {
      has_u = brent_next(s, u);
      if (!has_u) {
        out.rep_dv[pos] = make_double2(s.x, s.fx);
        busy = false;
      }
    }
    // ---- refill
    const uint64_t idle = __ballot(!busy);
    if (idle != 0 && (more || next < end) && ((uint32_t)__popcll(idle) >= kLlhRefill || idle == __ballot(true))) {
      if (next == end) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&out.counters[5], kLlhChunk);
        base = __shfl(base, 0);
        next = min(base, nrec);
        end = min(base + kLlhChunk, nrec);
        more = base + kLlhChunk < nrec;
      }
      const uint32_t mine = next + __popcll(idle & lt);
      if (!busy && mine < end) {
        pos = mine;
        rec = out.rep_list[mine];
        if (rec != 0xFFFFFFFFu) { // else: hole at the end of a wave's chunk of list positions
          const uint32_t key = out.rec_key[rec];
          load_problem<NPT>(C, out.rec_hist + rec, out.rec_cap, out.rd_onmers[out.rec_read[rec]], ix.libs[0].rho[key >> 1], p);
          // the first three steps of the minimisation, with the objective values of kr_llh_pre_kernel
          const double f0 = out.rec_d[rec], f1 = out.rec_v[rec], f2 = out.rec_chisq[rec];
          brent_start(s, 0.5, f0);
          double u1 = 0.0, u2 = 0.0;
          brent_next(s, u1);
          brent_update(s, u1, f1);
          brent_next(s, u2);
          brent_update(s, u2, f2);
          has_u = brent_next(s, u);
          busy = has_u;
          if (!has_u) out.rep_dv[pos] = make_double2(s.x, s.fx); // converged at once (not with these brackets, but cheap to honour)
        }
      }
      next = min(end, next + (uint32_t)__popcll(idle));
    }
    if (__ballot(busy) == 0) {
      if (!more && next == end) break;
      continue;
    }
    // ---- one objective evaluation for every busy lane
    if (has_u) brent_update(s, u, llh_eval<NPT>(C, T, p, u));
  }
}

// NPT = 5: --hdist-th default, histogram in registers; NPT = 0: any threshold
template <int NPT>
__global__ __launch_bounds__(256, KR_LLH_WPE) void kr_llh_kernel(LlhConst C, DevIndex ix, BatchOut out)
{
  __shared__ double s_bk[32], s_hnk[kMaxPlanes];
  LlhTables T{(lds_f64*)s_bk, (lds_f64*)s_hnk};
  llh_tables_init(C, T.bk, T.hnk);
  llh_records<NPT>(C, T, ix, out);
}

// summarize_matches' strand merge and closest (src/query.cpp:96-139) + the row selection
// of report_distances (src/query.cpp:158-196).  Records of a read are sorted by
// key = (se << 1) | strand.  Iteration order of the reference's maps is arbitrary; the
// order used here (all forward leaves by ascending se, then all reverse ones) is the
// oracle's, so `<=` ties resolve identically.
// 32 lanes per read, one record per lane (reads have tens of records; all record arrays are read
// coalesced).
// FILT: --filter (the chi-square needs an objective evaluation per record); without it the kernel carries no
// likelihood code and runs at full occupancy.
template <int NPT, bool FILT>
__global__ __launch_bounds__(256) void kr_select_kernel(LlhConst C, DevIndex ix, DevParams P, BatchOut out,
                                                        uint32_t nreads)
{
  if (!FILT) P.no_filter = 1;
  __shared__ double s_bk[32], s_hnk[kMaxPlanes];
  LlhTables T{(lds_f64*)s_bk, (lds_f64*)s_hnk};
  llh_tables_init(C, T.bk, T.hnk);
  constexpr uint32_t GL = 32, GPB = 256 / GL;
  const uint32_t gl = threadIdx.x & (GL - 1u);
  const double kMax = 1.7976931348623157e308;
  for (uint32_t r = blockIdx.x * GPB + threadIdx.x / GL; r < nreads; r += gridDim.x * GPB) {
    const uint32_t o = out.rd_off[r], n = out.rd_cnt[r];
    // ---- closest: last record in (strand, se) order with d <= best, i.e. the smallest d and among
    //      equal d the largest (strand, index)
    double bd = kMax;
    uint32_t bo = 0;
    bool any = false;
    for (uint32_t i = gl; i < n; i += GL) {
      const double d = out.rec_d[o + i];
      const uint32_t ord = ((out.rec_key[o + i] & 1u) << 31) | i;
      if (d <= kMax && (!any || d < bd || (d == bd && ord > bo))) bd = d, bo = ord, any = true;
    }
#pragma unroll
    for (int sft = GL / 2; sft >= 1; sft >>= 1) {
      const double od = __shfl_xor(bd, sft, GL);
      const uint32_t oo = __shfl_xor(bo, sft, GL);
      const bool oa = __shfl_xor(any ? 1 : 0, sft, GL) != 0;
      if (oa && (!any || od < bd || (od == bd && oo > bo))) bd = od, bo = oo, any = true;
    }
    const int cl = any ? (int)(o + (bo & 0x7FFFFFFFu)) : -1;
    const double best = bd;
    const bool na = (n == 0) || (P.dmax_set && best > P.dist_max);
    if (gl == 0) out.rd_na[r] = na ? 1 : 0;
    LlhProblem pc;
    double vcl = 0;
    uint32_t kcl = 0;
    if (cl >= 0) kcl = out.rec_key[cl];
    if (FILT && cl >= 0 && !P.no_filter) {
      load_problem<NPT>(C, out.rec_hist + cl, out.rec_cap, out.rd_onmers[r], ix.libs[0].rho[kcl >> 1], pc);
      vcl = out.rec_v[cl];
    }
    for (uint32_t t = gl; t < n; t += GL) {
      const uint32_t i = o + t;
      const uint32_t key = out.rec_key[i];
      // which record represents this leaf in node_to_minfo?
      bool chosen;
      const bool has_other = (key & 1u) ? (t > 0 && out.rec_key[i - 1] == (key ^ 1u)) : (t + 1 < n && out.rec_key[i + 1] == (key ^ 1u));
      if (!has_other) {
        chosen = true;
      } else {
        const uint32_t io = (key & 1u) ? i - 1 : i, ir = (key & 1u) ? i : i + 1;
        const double d_or = out.rec_d[io], d_rc = out.rec_d[ir];
        uint32_t m_or = 0, m_rc = 0;
        for (uint32_t x = 0; x <= C.th; ++x) {
          m_or += out.rec_hist[(uint64_t)x * out.rec_cap + io];
          m_rc += out.rec_hist[(uint64_t)x * out.rec_cap + ir];
        }
        bool take_or = (d_rc > d_or) || ((d_rc == d_or) && (m_rc < m_or)); // src/query.cpp:129-133
        // the closest overrides (src/query.cpp:136-138)
        if (cl >= 0 && (kcl >> 1) == (key >> 1)) take_or = ((uint32_t)cl == io);
        chosen = (key & 1u) ? !take_or : take_or;
      }
      const double d = out.rec_d[i];
      double chi = nan("");
      bool sel = false;
      if (chosen && !na) {
        const bool dm = !P.dmax_set || d < P.dist_max;
        if (!P.multi) {
          sel = (int)i == cl;
        } else if (P.no_filter) {
          sel = dm;
        } else if (FILT) {
          chi = 2 * (llh_eval<NPT>(C, T, pc, d) - vcl); // Minfo::likelihood_ratio (src/query.cpp:420-424)
          sel = (chi < P.chisq) && dm;
        }
      }
      out.rec_sel[i] = sel ? 1 : 0;
      out.rec_chisq[i] = chi;
    }
  }
}

// Slotted copy of the head of every bucket: slot r = {len, start, enc[start .. start + W - 2)}, unused words 0xFFFFFFFF.
// One thread per slot word: reads are contiguous within a bucket, writes fully coalesced.
__global__ void kr_build_slots(const uint64_t* bkt, const uint32_t* enc, uint32_t nrows, uint32_t log2w, uint32_t* slots)
{
  const uint64_t n = (uint64_t)nrows << log2w;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t row = (uint32_t)(i >> log2w), wd = (uint32_t)i & ((1u << log2w) - 1u);
    const uint64_t b = bkt[row];
    const uint64_t st = b >> 24;
    const uint32_t ln = (uint32_t)(b & 0xFFFFFFu);
    uint32_t x = 0xFFFFFFFFu;
    if (wd == 0)
      x = ln;
    else if (wd == 1)
      x = (uint32_t)st;
    else if (wd - 2u < ln)
      x = enc[st + wd - 2u];
    slots[i] = x;
  }
}

// ---------------------------------------------------------------------------
// Debug kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void kr_front_end_kernel(DevIndex ix, BatchIn in, uint32_t stride, uint32_t* rix,
                                                            uint32_t* enc32, uint8_t* valid, uint8_t* pass)
{
  const uint32_t lane = lane_id();
  for (uint32_t r = blockIdx.x; r < in.nreads; r += gridDim.x) {
    const uint64_t off0 = in.offsets[r], len = in.offsets[r + 1] - off0;
    const uint8_t* seq = in.bases + off0;
    const uint64_t nkm = len >= ix.k ? len - ix.k + 1 : 0;
    for (uint64_t base0 = 0; base0 < nkm; base0 += kSegPos) {
      const uint32_t npos_seg = (uint32_t)min((uint64_t)kSegPos, nkm - base0);
      SegBits sb;
      load_segment(seq, len, base0, sb);
      for (int pp = 0; pp < 2; ++pp) {
        FrontEnd fe = front_end(ix, sb, pp, npos_seg);
        uint64_t j = base0 + 64u * pp + lane;
        if (64u * pp + lane < npos_seg && j < stride) {
          for (int s = 0; s < 2; ++s) {
            uint64_t o = ((uint64_t)r * stride + j) * 2 + s;
            int lib;
            uint32_t row;
            rix[o] = fe.rix[s];
            enc32[o] = fe.enc32[s];
            valid[o] = fe.valid;
            pass[o] = fe.valid && locate_row<false>(ix, fe.rix[s], lib, row);
          }
        }
      }
    }
  }
}

template <int NPT>
__global__ void kr_brent_kernel(LlhConst C, uint32_t n, const uint32_t* hist, const uint32_t* onmers, const double* rho,
                                double* d_out, double* v_out)
{
  __shared__ double s_bk[32], s_hnk[kMaxPlanes];
  LlhTables T{(lds_f64*)s_bk, (lds_f64*)s_hnk};
  llh_tables_init(C, T.bk, T.hnk);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  LlhProblem p;
  load_problem<NPT>(C, hist + (uint64_t)i * (C.th + 1), 1, onmers[i], rho[i], p);
  brent_min<NPT>(C, T, p, d_out[i], v_out[i]);
}

// pidx != nullptr: evaluation i uses problem pidx[i] (several evaluations of one histogram at different d)
__global__ __launch_bounds__(256) void kr_llh_batch_kernel(LlhConst C, uint32_t mode, uint64_t n, const double* hist,
                                                           const double* uc, const double* rho, const double* d_in,
                                                           double* d_out, double* v_out, const uint32_t* pidx)
{
  __shared__ double s_bk[32], s_hnk[kMaxPlanes];
  LlhTables T{(lds_f64*)s_bk, (lds_f64*)s_hnk};
  llh_tables_init(C, T.bk, T.hnk);
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    LlhProblem p;
    const uint64_t q = pidx ? pidx[i] : i;
    for (uint32_t x = 0; x <= C.th; ++x) p.mc[x] = hist[q * (C.th + 1) + x];
    p.uc = uc[q];
    p.rho = rho[q];
    if (mode == 0) {
      double d, v;
      brent_min<0>(C, T, p, d, v);
      d_out[i] = d;
      v_out[i] = v;
    } else {
      v_out[i] = llh_eval<0>(C, T, p, d_in[i]);
    }
  }
}

// Re-layout kernels used by kr_index_upload.
// class of a colour id: 0 drop, 1 leaf, 2 expand (see colour_needs_expansion)
__device__ __forceinline__ uint32_t tag_colour(uint32_t se, const uint32_t* node_info, uint32_t tree_nnodes, uint32_t nsubsets)
{
  if (se == 0 || se >= nsubsets || se > kColMask) return 0; // empty set, or an id the crecord does not define
  if (se <= tree_nnodes) {
    uint32_t info = node_info[se], kd = info & 3u;
    if (kd == 1u) return (info >> 2) | (1u << 30); // leaf: its rank
    return kd ? (se | (kd << 30)) : 0u;
  }
  return se | (2u << 30);
}
__global__ void kr_relayout_cmer(const uint32_t* cmer, uint64_t n, uint32_t* enc, uint32_t* se, const uint32_t* node_info,
                                 uint32_t tree_nnodes, uint32_t nsubsets)
{
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint2 v = reinterpret_cast<const uint2*>(cmer)[i];
    enc[i] = v.x;
    se[i] = tag_colour(v.y, node_info, tree_nnodes, nsubsets);
  }
}
__global__ void kr_tag_colours(uint2* pse, uint32_t nsubsets, const uint32_t* node_info, uint32_t tree_nnodes)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nsubsets; i += gridDim.x * blockDim.x) {
    uint2 v = pse[i];
    pse[i] = make_uint2(tag_colour(v.x, node_info, tree_nnodes, nsubsets), tag_colour(v.y, node_info, tree_nnodes, nsubsets));
  }
}
__global__ void kr_relayout_inc(const uint64_t* inc, uint32_t nrows, uint64_t* bkt, uint32_t* bad)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += gridDim.x * blockDim.x) {
    uint64_t e = inc[i], s = i ? inc[i - 1] : 0; // FlatHT::bucket_start/next (src/table.hpp:121-136)
    uint64_t l = e >= s ? e - s : 0;
    if (e < s || l > 0xFFFFFFull || s >= (1ull << 40)) atomicOr(bad, 1u);
    bkt[i] = (s << 24) | (l & 0xFFFFFFull);
  }
}

// ---------------------------------------------------------------------------
// Host side of the device ABI
// ---------------------------------------------------------------------------
#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t e__ = (expr);                                                                                \
    if (e__ != hipSuccess)                                                                                  \
      return kr::fail(e__ == hipErrorOutOfMemory ? KR_ERR_NOMEM : KR_ERR_NO_DEVICE,                          \
                      std::string(#expr) + ": " + hipGetErrorString(e__));                                  \
  } while (0)

// dynamic LDS bytes of the probe kernel: stack + probe list + ntouched (+ table)
uint32_t probe_lds_bytes(uint32_t np, uint32_t bm_words, bool lean = false)
{
  if (lean) // single-segment instantiation: short stack (bitmap aliases it) | key slots | event region
    return (lean_stack_bytes(bm_words) + kLdsSlots * 4 + kLdsSlots * np * (kPlaneWords + 1) * 4 + 15u) & ~15u;
  uint32_t b = kStackCap * 8;
  b += kLdsSlots * 4 + kLdsSlots * np * kPlaneWords * 4 + kLdsSlots * np * 4 + 2 * bm_words * 4 + bm_words; // two bitmaps + u16 prefix per 2 words
  if (getenv("KR_DEBUG_LDS_PAD")) b += (uint32_t)atoi(getenv("KR_DEBUG_LDS_PAD")); // occupancy experiments
  return (b + 15u) & ~15u;
}

LlhConst make_llh_const(uint32_t k, uint32_t h, uint32_t th)
{ // HDistHistLLH ctor (src/hdhistllh.hpp:51-69): uint64 arithmetic, then exact conversion
  LlhConst C;
  memset(&C, 0, sizeof(C));
  C.k = k, C.h = h, C.th = th;
  C.dbg = getenv("KR_DEBUG_LLH") ? (uint32_t)atoi(getenv("KR_DEBUG_LLH")) : 0u;
  uint64_t bk[32] = {0};
  bk[0] = 1;
  for (uint32_t i = 0; i < k; ++i) bk[i + 1] = (bk[i] * (k - i)) / (i + 1);
  for (uint32_t i = 0; i <= k; ++i) C.binom_k[i] = (double)bk[i];
  uint64_t vc = 1, nh = k - h;
  C.binom_hnk[0] = 0.0;
  for (uint32_t i = 1; i <= th; ++i) {
    vc = (vc * (nh - i + 1)) / i;
    C.binom_hnk[i] = (double)(bk[i] - vc);
  }
  return C;
}

} // namespace

struct kr_index {
  int device = 0;
  uint32_t log_g = 0;           // lanes per probe in the bucket scan = 2^log_g (from the mean bucket length)
  uint32_t slot_log2w = 0;      // slotted table copy: words per slot = 1 << slot_log2w (0: none)
  DevIndex dix;
  std::vector<DevLib> hlibs;    // host copy of the device DevLib array
  std::vector<void*> allocs;    // everything to hipFree
  std::vector<kr_index_buffer> bufs; // export order
  std::vector<uint8_t> desc;    // export descriptor
  uint64_t bytes = 0;
  // workspace of kr_llh_batch (grown on demand, reused across calls): one device buffer and one pinned host
  // buffer laid out [hist n*np | uc n | rho n | d_in n | d_out n | v n]
  mutable std::mutex llh_mu;
  mutable double* llh_dev = nullptr;
  mutable double* llh_pin = nullptr;
  mutable uint64_t llh_cap = 0; // doubles
};

namespace {

struct DescHeader {
  uint32_t magic, k, h, m, nlibs, tree_nnodes, nleaves, log_g;
  uint32_t slot_log2w, pad_; // words per slot of the slotted table copy = 1 << slot_log2w (0: none)
  uint64_t res_mask;
  uint8_t ppos[32], npos[32];
};
struct DescLib {
  uint64_t nkmers;
  uint32_t nrows, nsubsets, nnodes, numer;
};

int dev_alloc(kr_index* ix, void** p, uint64_t bytes)
{
  HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
  ix->allocs.push_back(*p);
  ix->bytes += bytes;
  return KR_OK;
}

// Allocate every device buffer of an index from its descriptor; fills ix->dix/hlibs/bufs.
int alloc_from_desc(kr_index* ix, const DescHeader& H, const std::vector<DescLib>& L)
{
  ix->hlibs.resize(H.nlibs);
  ix->bufs.clear();
  for (uint32_t i = 0; i < H.nlibs; ++i) {
    DevLib& d = ix->hlibs[i];
    memset(&d, 0, sizeof(d));
    d.nkmers = L[i].nkmers, d.nrows = L[i].nrows, d.nsubsets = L[i].nsubsets, d.nnodes = L[i].nnodes,
    d.numer = L[i].numer;
    void* p;
    int rc;
    uint64_t b;
    b = (uint64_t)d.nrows * 8;
    if ((rc = dev_alloc(ix, &p, b))) return rc;
    d.bkt = (const uint64_t*)p;
    ix->bufs.push_back({p, b});
    b = (d.nkmers + 16) * 4; // 16 entries of slack: 16-byte chunk loads may run past the end
    if ((rc = dev_alloc(ix, &p, b))) return rc;
    d.enc = (const uint32_t*)p;
    ix->bufs.push_back({p, b});
    b = (d.nkmers + 16) * 4;
    if ((rc = dev_alloc(ix, &p, b))) return rc;
    d.se = (const uint32_t*)p;
    ix->bufs.push_back({p, b});
    b = (uint64_t)d.nsubsets * 8;
    if ((rc = dev_alloc(ix, &p, b))) return rc;
    d.pse = (const uint2*)p;
    ix->bufs.push_back({p, b});
    b = (uint64_t)d.nnodes * 8;
    if ((rc = dev_alloc(ix, &p, b))) return rc;
    d.rho = (const double*)p;
    ix->bufs.push_back({p, b});
    d.slot_log2w = H.slot_log2w;
    if (H.slot_log2w) {
      b = ((uint64_t)d.nrows << H.slot_log2w) * 4;
      if ((rc = dev_alloc(ix, &p, b))) return rc;
      d.slots = (const uint32_t*)p;
      ix->bufs.push_back({p, b});
    }
  }
  void* p;
  int rc;
  uint64_t b = ((uint64_t)H.tree_nnodes + 1) * 4;
  if ((rc = dev_alloc(ix, &p, b))) return rc;
  ix->dix.node_info = (const uint32_t*)p;
  ix->bufs.push_back({p, b});
  b = (uint64_t)std::max<uint32_t>(1u, H.nleaves) * 4;
  if ((rc = dev_alloc(ix, &p, b))) return rc;
  ix->dix.leaf_se = (const uint32_t*)p;
  ix->bufs.push_back({p, b});
  ix->dix.nleaves = H.nleaves;
  ix->log_g = H.log_g;
  ix->slot_log2w = H.slot_log2w;
  b = (uint64_t)H.m * 4;
  if ((rc = dev_alloc(ix, &p, b))) return rc;
  ix->dix.res_lib = (const int32_t*)p;
  ix->bufs.push_back({p, b});
  b = (uint64_t)H.nlibs * sizeof(DevLib);
  if ((rc = dev_alloc(ix, &p, b))) return rc;
  ix->dix.libs = (const DevLib*)p;
  HIP_TRY(hipMemcpy(p, ix->hlibs.data(), b, hipMemcpyHostToDevice)); // pointers are per-device: never exported
  ix->dix.lib0 = ix->hlibs[0];
  ix->dix.res_mask = H.res_mask;

  ix->dix.k = H.k, ix->dix.h = H.h, ix->dix.m = H.m, ix->dix.nlibs = H.nlibs, ix->dix.tree_nnodes = H.tree_nnodes;
  ix->dix.m_shift = 0xFFFFFFFFu;
  if ((H.m & (H.m - 1)) == 0) {
    uint32_t s = 0;
    while ((1u << s) < H.m) ++s;
    ix->dix.m_shift = s;
  }
  std::vector<uint8_t> pasc(H.ppos, H.ppos + H.h), nasc(H.npos, H.npos + (H.k - H.h));
  ix->dix.pmask = make_pext(pasc);
  ix->dix.nmask = make_pext(nasc);
  ix->desc.resize(sizeof(DescHeader) + L.size() * sizeof(DescLib));
  memcpy(ix->desc.data(), &H, sizeof(H));
  memcpy(ix->desc.data() + sizeof(H), L.data(), L.size() * sizeof(DescLib));
  return KR_OK;
}

constexpr uint32_t kDescMagic = 0x4b524958u; // "KRIX"

} // namespace

extern "C" {

int kr_index_upload(const kr_index_view* v, int device, uint32_t flags, kr_index** out)
{
  kr::clear_error();
  if (!v || !out || !v->libs || !v->ppos || !v->npos || !v->node_kind) return kr::fail(KR_ERR_ARG, "kr_index_upload: null argument");
  *out = nullptr;
  if (v->k < 3 || v->k > 31 || v->h == 0 || v->h >= v->k || v->k - v->h > 16 || v->h > 15)
    return kr::fail(KR_ERR_ARG, "kr_index_upload: unsupported k/h (need k<=31, k-h<=16, h<=15)");
  if (v->nlibs == 0 || v->nlibs > (uint32_t)kMaxLibs) return kr::fail(KR_ERR_ARG, "kr_index_upload: 1..16 partial libraries supported");
  if (v->m == 0 || v->m > 65536) return kr::fail(KR_ERR_ARG, "kr_index_upload: m out of range");
  if (v->tree_nnodes >= (1u << 30)) return kr::fail(KR_ERR_ARG, "kr_index_upload: tree too large");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return kr::fail(KR_ERR_NO_DEVICE, "no HIP device available");
  if (device < 0 || device >= ndev) return kr::fail(KR_ERR_ARG, "kr_index_upload: bad device ordinal");
  HIP_TRY(hipSetDevice(device));

  DescHeader H;
  memset(&H, 0, sizeof(H));
  H.magic = kDescMagic, H.k = v->k, H.h = v->h, H.m = v->m, H.nlibs = v->nlibs, H.tree_nnodes = v->tree_nnodes;
  memcpy(H.ppos, v->ppos, v->h);
  memcpy(H.npos, v->npos, v->k - v->h);
  std::vector<DescLib> L(v->nlibs);
  std::vector<int32_t> res_lib(v->m, -1);
  for (uint32_t i = 0; i < v->nlibs; ++i) {
    const kr_lib_view& lv = v->libs[i];
    if (lv.r >= v->m) return kr::fail(KR_ERR_FORMAT, "library residue r >= m");
    if (lv.nnodes != v->tree_nnodes + 1) return kr::fail(KR_ERR_FORMAT, "crecord nnodes does not match the tree");
    L[i] = DescLib{lv.nkmers, lv.nrows, lv.nsubsets, lv.nnodes, lv.frac ? lv.r + 1 : 1u};
    // src/index.cpp:144-157
    if (lv.frac)
      for (uint32_t q = 0; q <= lv.r; ++q) res_lib[q] = (int32_t)i;
    else
      res_lib[lv.r] = (int32_t)i;
  }
  for (uint32_t q = 0; q < v->m && q < 64; ++q)
    if (res_lib[q] >= 0) H.res_mask |= 1ull << q;
  std::vector<uint32_t> node_info(v->tree_nnodes + 1, 0), leaf_se;
  for (uint32_t se = 1; se <= v->tree_nnodes; ++se) {
    uint32_t kd = v->node_kind[se] & 3u;
    node_info[se] = kd;
    if (kd == 1) {
      node_info[se] |= (uint32_t)leaf_se.size() << 2;
      leaf_se.push_back(se);
    }
  }
  H.nleaves = (uint32_t)leaf_se.size();
  // 2 key bits per leaf in the accumulate kernel's LDS bitmaps: 65,536 leaves = 45 KB of the 64 KB a workgroup may use
  if (2ull * H.nleaves > 131072) return kr::fail(KR_ERR_ARG, "kr_index_upload: more than 65536 reference leaves is not supported");
  { // lanes per probe: a bucket of L entries spans about (L + 4.5) / 4 aligned 16-byte chunks
    double nk = 0, nr = 0;
    for (uint32_t i = 0; i < v->nlibs; ++i) nk += (double)v->libs[i].nkmers, nr += (double)v->libs[i].nrows;
    double mean_len = nr > 0 ? nk / nr : 0; // empty buckets never reach the scan, so this underestimates slightly
    H.log_g = mean_len <= 3.0 ? 0u : (mean_len <= 44.0 ? 2u : 3u);
    if (const char* e = getenv("KR_LOG_G")) H.log_g = (uint32_t)atoi(e) > 3 ? 3u : (uint32_t)atoi(e);
    if (H.log_g == 1) H.log_g = 2;
    // dense tables get a slotted copy of the head of every bucket (see scan_group_slots): the smallest slot
    // of 32 / 64 / 128 words whose W-2 entries cover the mean bucket length + 3 sigma (Poisson)
    bool fits32 = true;
    for (uint32_t i = 0; i < v->nlibs; ++i) fits32 = fits32 && v->libs[i].nkmers < (1ull << 32);
    const double need = mean_len + 3.0 * std::sqrt(mean_len);
    if (mean_len >= 12.0 && fits32) H.slot_log2w = need <= 30.0 ? 5u : (need <= 62.0 ? 6u : (need <= 126.0 ? 7u : 0u));
    if (const char* e = getenv("KR_SLOT_LOG2W")) { // tests / experiments: 0 (packed only), 5, 6, 7
      const uint32_t w = (uint32_t)atoi(e);
      H.slot_log2w = (fits32 && w >= 5 && w <= 7) ? w : 0u;
    }
  }
  std::unique_ptr<kr_index> ix(new kr_index());
  ix->device = device;
  int rc = alloc_from_desc(ix.get(), H, L);
  if (rc) {
    kr_index_free(ix.release());
    return rc;
  }
  const hipMemcpyKind kind = (flags & KR_VIEW_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIP_TRY(hipMemcpy((void*)ix->dix.node_info, node_info.data(), node_info.size() * 4, hipMemcpyHostToDevice));
  if (!leaf_se.empty()) HIP_TRY(hipMemcpy((void*)ix->dix.leaf_se, leaf_se.data(), leaf_se.size() * 4, hipMemcpyHostToDevice));
  uint32_t* d_bad = nullptr;
  HIP_TRY(hipMalloc(&d_bad, 4));
  HIP_TRY(hipMemset(d_bad, 0, 4));
  for (uint32_t i = 0; i < v->nlibs; ++i) {
    const kr_lib_view& lv = v->libs[i];
    const DevLib& d = ix->hlibs[i];
    // stage the on-disk arrays, then re-lay them out on the device
    const uint32_t* src_cmer = lv.cmer;
    const uint64_t* src_inc = lv.inc;
    void *t_cmer = nullptr, *t_inc = nullptr;
    if (!(flags & KR_VIEW_DEVICE)) {
      HIP_TRY(hipMalloc(&t_cmer, lv.nkmers * 8 + 16));
      HIP_TRY(hipMemcpy(t_cmer, lv.cmer, lv.nkmers * 8, hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&t_inc, (uint64_t)lv.nrows * 8 + 16));
      HIP_TRY(hipMemcpy(t_inc, lv.inc, (uint64_t)lv.nrows * 8, hipMemcpyHostToDevice));
      src_cmer = (const uint32_t*)t_cmer;
      src_inc = (const uint64_t*)t_inc;
    }
    HIP_TRY(hipMemset((void*)d.enc, 0xFF, (d.nkmers + 16) * 4));
    HIP_TRY(hipMemset((void*)d.se, 0, (d.nkmers + 16) * 4));
    if (lv.nsubsets > kColMask) return kr::fail(KR_ERR_ARG, "kr_index_upload: more than 2^30 colours is not supported");
    if (lv.nkmers)
      hipLaunchKernelGGL(kr_relayout_cmer, dim3(2048), dim3(256), 0, 0, src_cmer, lv.nkmers, (uint32_t*)d.enc, (uint32_t*)d.se,
                         ix->dix.node_info, v->tree_nnodes, lv.nsubsets);
    if (lv.nrows) hipLaunchKernelGGL(kr_relayout_inc, dim3(1024), dim3(256), 0, 0, src_inc, lv.nrows, (uint64_t*)d.bkt, d_bad);
    if (lv.nrows && H.slot_log2w)
      hipLaunchKernelGGL(kr_build_slots, dim3(8192), dim3(256), 0, 0, d.bkt, d.enc, lv.nrows, H.slot_log2w, (uint32_t*)d.slots);
    HIP_TRY(hipDeviceSynchronize());
    if (t_cmer) hipFree(t_cmer);
    if (t_inc) hipFree(t_inc);
    HIP_TRY(hipMemcpy((void*)d.pse, lv.pse, (uint64_t)lv.nsubsets * 8, kind));
    if (lv.nsubsets) hipLaunchKernelGGL(kr_tag_colours, dim3(1024), dim3(256), 0, 0, (uint2*)d.pse, lv.nsubsets, ix->dix.node_info, v->tree_nnodes);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy((void*)d.rho, lv.rho, (uint64_t)lv.nnodes * 8, kind));
  }
  uint32_t bad = 0;
  HIP_TRY(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  hipFree(d_bad);
  if (bad) {
    kr_index_free(ix.release());
    return kr::fail(KR_ERR_FORMAT, "inc-* is not monotone or a bucket exceeds 2^24 entries / 2^40 offset");
  }
  HIP_TRY(hipMemcpy((void*)ix->dix.res_lib, res_lib.data(), (uint64_t)v->m * 4, hipMemcpyHostToDevice));
  *out = ix.release();
  return KR_OK;
}

void kr_index_free(kr_index* ix)
{
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  for (void* p : ix->allocs) (void)hipFree(p);
  if (ix->llh_dev) (void)hipFree(ix->llh_dev);
  if (ix->llh_pin) (void)hipHostFree(ix->llh_pin);
  delete ix;
}

uint64_t kr_index_device_bytes(const kr_index* ix) { return ix ? ix->bytes : 0; }

int kr_index_export(const kr_index* ix, void* desc, uint64_t* desc_bytes, kr_index_buffer* bufs, uint32_t* nbufs)
{
  if (!ix || !desc_bytes || !nbufs) return kr::fail(KR_ERR_ARG, "kr_index_export: null argument");
  uint64_t need = ix->desc.size();
  uint32_t nb = (uint32_t)ix->bufs.size();
  bool fits = desc && *desc_bytes >= need && bufs && *nbufs >= nb;
  *desc_bytes = need;
  *nbufs = nb;
  if (!fits) return (desc || bufs) ? kr::fail(KR_ERR_ARG, "kr_index_export: buffers too small") : KR_OK;
  memcpy(desc, ix->desc.data(), need);
  memcpy(bufs, ix->bufs.data(), nb * sizeof(kr_index_buffer));
  return KR_OK;
}

int kr_index_import(const void* desc, uint64_t desc_bytes, int device, kr_index** out, kr_index_buffer* bufs, uint32_t* nbufs)
{
  kr::clear_error();
  if (!desc || !out || !nbufs || desc_bytes < sizeof(DescHeader)) return kr::fail(KR_ERR_ARG, "kr_index_import: bad argument");
  DescHeader H;
  memcpy(&H, desc, sizeof(H));
  if (H.magic != kDescMagic || desc_bytes != sizeof(H) + (uint64_t)H.nlibs * sizeof(DescLib) || H.nlibs == 0 || H.nlibs > (uint32_t)kMaxLibs)
    return kr::fail(KR_ERR_FORMAT, "kr_index_import: bad descriptor");
  std::vector<DescLib> L(H.nlibs);
  memcpy(L.data(), (const uint8_t*)desc + sizeof(H), H.nlibs * sizeof(DescLib));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return kr::fail(KR_ERR_NO_DEVICE, "no HIP device available");
  if (device < 0 || device >= ndev) return kr::fail(KR_ERR_ARG, "kr_index_import: bad device ordinal");
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<kr_index> ix(new kr_index());
  ix->device = device;
  int rc = alloc_from_desc(ix.get(), H, L);
  if (rc) {
    kr_index_free(ix.release());
    return rc;
  }
  uint32_t nb = (uint32_t)ix->bufs.size();
  if (!bufs || *nbufs < nb) {
    *nbufs = nb;
    kr_index_free(ix.release());
    return kr::fail(KR_ERR_ARG, "kr_index_import: bufs too small");
  }
  memcpy(bufs, ix->bufs.data(), nb * sizeof(kr_index_buffer));
  *nbufs = nb;
  *out = ix.release();
  return KR_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------
// kr_stream
// ---------------------------------------------------------------------------
struct kr_stream {
  const kr_index* ix = nullptr;
  kr_params params;
  DevParams dp;
  LlhConst llh;
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t max_reads = 0;
  uint64_t max_bases = 0;
  uint32_t rec_cap = 0, hit_cap = 0;
  uint64_t rec_user_cap = 0; // the caller's max_records (rec_cap adds per-wave chunk slack)
  uint32_t nwaves = 0, nwaves_full = 0, nwaves_lean = 0; // per-wave scratch slots; grids of the two accumulate launches
  // device
  uint8_t* d_bases = nullptr;
  uint64_t* d_offsets = nullptr;
  BatchOut out;
  std::vector<void*> dallocs;
  // pinned host
  uint8_t* h_bases = nullptr;
  uint64_t* h_offsets = nullptr;
  uint32_t* h_counters = nullptr;
  uint32_t *h_rd_off = nullptr, *h_rd_cnt = nullptr, *h_rd_onmers = nullptr, *h_rd_filt = nullptr;
  uint8_t* h_rd_na = nullptr;
  uint32_t *h_rec_key = nullptr, *h_rec_hist = nullptr;
  uint8_t* h_rec_sel = nullptr;
  double *h_rec_d = nullptr, *h_rec_v = nullptr, *h_rec_chisq = nullptr;
  kr_hit* h_hits = nullptr;
  std::vector<void*> hallocs;
  // state
  uint64_t h_rec_cap = 0; // pinned record buffers grow on demand in kr_batch_collect
  bool submitted = false, waited = false;
  uint32_t nreads = 0, flags = 0, nrecs = 0;
  uint32_t scan_blocks = 0;
  uint64_t nhits = 0;
  BatchIn in;
};

namespace {

template <typename T>
int salloc(kr_stream* s, T** p, uint64_t n)
{
  HIP_TRY(hipMalloc((void**)p, std::max<uint64_t>(16, n * sizeof(T))));
  s->dallocs.push_back(*p);
  return KR_OK;
}
template <typename T>
int halloc(kr_stream* s, T** p, uint64_t n)
{
  HIP_TRY(hipHostMalloc((void**)p, std::max<uint64_t>(16, n * sizeof(T)), hipHostMallocDefault));
  s->hallocs.push_back(*p);
  return KR_OK;
}

uint32_t next_pow2(uint32_t v)
{
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

int check_errflags(uint32_t e)
{
  if (e & kErrRecCap) return kr::fail(KR_ERR_CAPACITY, "record buffer overflow: submit fewer reads per batch");
  if (e & kErrStack) return kr::fail(KR_ERR_CAPACITY, "colour work stack overflow (a colour expands into more pending work than the LDS stack and its spill hold)");
  if (e & kErrTable) return kr::fail(KR_ERR_CAPACITY, "global accumulator table overflow");
  if (e & kErrHitCap) return kr::fail(KR_ERR_CAPACITY, "hit tap buffer overflow");
  if (e & kErrItemCap) return kr::fail(KR_ERR_CAPACITY, "hit list overflow (more than 256 table hits per read on average): submit fewer reads per batch");
  return KR_OK;
}

} // namespace

extern "C" {

int kr_stream_create(const kr_index* ix, const kr_params* p, uint32_t max_reads, uint64_t max_bases, uint64_t max_records,
                     kr_stream** out)
{
  kr::clear_error();
  if (!ix || !p || !out || max_reads == 0) return kr::fail(KR_ERR_ARG, "kr_stream_create: bad argument");
  if (p->hdist_th > KR_MAX_HDIST_TH) return kr::fail(KR_ERR_ARG, "--hdist-th above 16 is not supported (k-h <= 16 bounds hd)");
  HIP_TRY(hipSetDevice(ix->device));
  std::unique_ptr<kr_stream> s(new kr_stream());
  s->ix = ix;
  s->params = *p;
  s->dp.th = p->hdist_th, s->dp.np = p->hdist_th + 1;
  s->dp.multi = p->multi, s->dp.no_filter = p->no_filter;
  s->dp.dmax_set = std::isnan(p->dist_max) ? 0 : 1;
  s->dp.dbg = getenv("KR_DEBUG_SKIP") ? (uint32_t)atoi(getenv("KR_DEBUG_SKIP")) : 0u;
  s->dp.chisq = p->chisq, s->dp.dist_max = p->dist_max;
  s->llh = make_llh_const(ix->dix.k, ix->dix.h, p->hdist_th);
  s->max_reads = max_reads, s->max_bases = max_bases;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
  // resident waves per CU: LDS-limited (160 KiB per CU), VGPR-limited to 4 waves per SIMD
  const uint32_t nslots2 = std::max<uint32_t>(2u, 2u * ix->dix.nleaves), bm_words = ((nslots2 + 63) / 64) * 2;
  // resident accumulate waves per CU: LDS-limited, at most 16 by registers (4 per SIMD); reads are handed out
  // dynamically, so a grid that is not fully resident costs nothing
  if (probe_lds_bytes(p->hdist_th + 1, bm_words) > 65536u)
    return kr::fail(KR_ERR_ARG, "kr_stream_create: this many reference leaves with this --hdist-th needs more LDS than a workgroup has");
  uint32_t per_cu = std::min<uint32_t>(16u, 163840u / probe_lds_bytes(p->hdist_th + 1, bm_words));
  // the single-segment instantiation has a lean LDS layout and 96 registers: 5 waves per SIMD
  uint32_t per_cu_lean = std::min<uint32_t>(20u, 163840u / probe_lds_bytes(p->hdist_th + 1, bm_words, true));
  if (getenv("KR_DEBUG_ACC_WAVES")) {
    per_cu = std::min<uint32_t>(per_cu, (uint32_t)atoi(getenv("KR_DEBUG_ACC_WAVES")));
    per_cu_lean = std::min<uint32_t>(per_cu_lean, (uint32_t)atoi(getenv("KR_DEBUG_ACC_WAVES")));
  }
  s->nwaves_full = (uint32_t)prop.multiProcessorCount * std::max<uint32_t>(1u, per_cu);
  s->nwaves_lean = (uint32_t)prop.multiProcessorCount * std::max<uint32_t>(1u, per_cu_lean);
  { // per-wave global scratch grows with the tree (100 B per leaf for the level-2 tables): bound the total by running
    // fewer waves on very large trees (reads are handed out dynamically, so any grid size is correct)
    const uint64_t np_ = p->hdist_th + 1;
    const uint64_t per_wave = (uint64_t)nslots2 * np_ * (kPlaneWords + 1) * 4 + (uint64_t)std::max<uint32_t>(nslots2, kEvSpill + 8192u) * 4;
    const uint64_t budget = (getenv("KR_ACC_SCRATCH_GB") ? (uint64_t)atoi(getenv("KR_ACC_SCRATCH_GB")) : 16ull) << 30;
    const uint32_t max_waves = (uint32_t)std::max<uint64_t>((uint64_t)prop.multiProcessorCount, budget / per_wave);
    s->nwaves_full = std::min(s->nwaves_full, max_waves);
    s->nwaves_lean = std::min(s->nwaves_lean, max_waves);
  }
  s->nwaves = std::max(s->nwaves_full, s->nwaves_lean);
  // default record capacity: up to 2 * leaves per read, at most 16 per read on average
  uint64_t per_read = std::min<uint64_t>(16, std::max<uint64_t>(8, (uint64_t)ix->dix.tree_nnodes + 1));
  uint64_t rc64 = max_records ? max_records : std::max<uint64_t>(1u << 16, (uint64_t)max_reads * per_read);
  s->rec_user_cap = rc64;
  // every resident wave may leave one partly used chunk behind: add that slack to the caller's bound
  s->rec_cap = (uint32_t)std::min<uint64_t>(rc64 + (uint64_t)s->nwaves * kRecChunk, 1ull << 30);
  s->hit_cap = 1u << 22;
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  for (auto& e : s->ev) HIP_TRY(hipEventCreate(&e));
  int rc = 0;
  BatchOut& o = s->out;
  memset(&o, 0, sizeof(o));
  uint32_t np = s->dp.np;
#define SA(ptr, n) \
  if ((rc = salloc(s.get(), &ptr, (n)))) { kr_stream_destroy(s.release()); return rc; }
#define HA(ptr, n) \
  if ((rc = halloc(s.get(), &ptr, (n)))) { kr_stream_destroy(s.release()); return rc; }
  SA(s->d_bases, max_bases + 256);
  SA(s->d_offsets, (uint64_t)max_reads + 1);
  SA(o.counters, 32);
  SA(o.cursors, 3 * kCursors * kCursorStride);
  SA(o.rd_off, max_reads);
  SA(o.rd_cnt, max_reads);
  SA(o.rd_onmers, max_reads);
  SA(o.rd_filt, 2ull * max_reads);
  SA(o.rd_na, max_reads);
  SA(o.rec_read, s->rec_cap);
  SA(o.rec_key, s->rec_cap);
  SA(o.rec_hist, (uint64_t)s->rec_cap * np);
  SA(o.rec_d, s->rec_cap);
  SA(o.rec_v, s->rec_cap);
  SA(o.rec_chisq, s->rec_cap);
  SA(o.rec_sel, s->rec_cap);
  SA(o.rec_w0, s->rec_cap);
  SA(o.rec_rep, s->rec_cap);
  SA(o.rep_list, s->rec_cap);
  SA(o.rep_dv, s->rec_cap);
  o.dd_shift = getenv("KR_DD_SHIFT") ? (uint32_t)atoi(getenv("KR_DD_SHIFT")) : 1u; // measured: 1: 5.1 ms, 3: 5.6, 5: 6.9 (llh + select, syn1000)
  o.dd_slots = std::min<uint32_t>(next_pow2(std::max<uint32_t>(2048u, s->rec_cap >> o.dd_shift)), 1u << 26);
  SA(o.dd_table, o.dd_slots);
  o.rec_cap = s->rec_cap;
  o.hit_cap = s->hit_cap;
  // item list between the two kernels: 256 hits per read on average, plus one partly used chunk per scan wave
  s->scan_blocks = (uint32_t)prop.multiProcessorCount * (4u * (ix->slot_log2w ? KR_SCAN_WPE_SLOT : KR_SCAN_WPE) / kScanWaves); // resident by construction (launch bounds)
  o.item_cap = (uint32_t)std::min<uint64_t>((uint64_t)max_reads * 256u + (uint64_t)s->scan_blocks * kScanWaves * 2u * kItemChunk, 1ull << 31);
  SA(o.items, o.item_cap);
  SA(o.rd_it_off, max_reads);
  SA(o.rd_it_cnt, max_reads);
  SA(o.long_list, (uint64_t)max_reads + 16ull * s->nwaves);
  SA(o.stk_spill, (uint64_t)s->nwaves * kStackSpill);
  o.nslots2 = nslots2;
  o.ev_spill = kEvSpill;
  o.tab_spill = std::min<uint32_t>(nslots2, 4096u);
  o.kt_spill = std::min<uint32_t>(nslots2, 16384u);
  const uint32_t g_list_words = std::max<uint32_t>(nslots2, o.ev_spill + o.tab_spill * ((np + 3) / 4 + 1) + o.kt_spill);
  o.g_list_words = g_list_words;
  o.bm_words = bm_words;
  SA(o.g_planes, (uint64_t)s->nwaves * nslots2 * np * kPlaneWords);
  SA(o.g_counts, (uint64_t)s->nwaves * nslots2 * np);
  SA(o.g_list, (uint64_t)s->nwaves * g_list_words);
  // on the stream's own stream and waited for: the stream does not synchronise with the null stream, and a
  // multi-GB clear (large trees) would otherwise still be running when the first batch arrives
  HIP_TRY(hipMemsetAsync(o.g_planes, 0, (uint64_t)s->nwaves * nslots2 * np * kPlaneWords * 4, s->stream));
  HIP_TRY(hipMemsetAsync(o.g_counts, 0, (uint64_t)s->nwaves * nslots2 * np * 4, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HA(s->h_bases, max_bases + 256);
  HA(s->h_offsets, (uint64_t)max_reads + 1);
  HA(s->h_counters, 32);
  HA(s->h_rd_off, max_reads);
  HA(s->h_rd_cnt, max_reads);
  HA(s->h_rd_onmers, max_reads);
  HA(s->h_rd_filt, 2ull * max_reads);
  HA(s->h_rd_na, max_reads);
#undef SA
#undef HA
  *out = s.release();
  return KR_OK;
}

void kr_stream_destroy(kr_stream* s)
{
  if (!s) return;
  if (s->ix) (void)hipSetDevice(s->ix->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (void* p : s->dallocs) (void)hipFree(p);
  for (void* p : s->hallocs) (void)hipHostFree(p);
  for (auto& e : s->ev)
    if (e) (void)hipEventDestroy(e);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

int kr_batch_submit(kr_stream* s, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads, uint32_t flags)
{
  kr::clear_error();
  if (!s || !bases || !offsets) return kr::fail(KR_ERR_ARG, "kr_batch_submit: null argument");
  if (nreads == 0 || nreads > s->max_reads) return kr::fail(KR_ERR_ARG, "kr_batch_submit: nreads out of range for this stream");
  HIP_TRY(hipSetDevice(s->ix->device));
  if (s->submitted && !s->waited) HIP_TRY(hipStreamSynchronize(s->stream));
  s->nreads = nreads;
  s->flags = flags;
  s->submitted = true;
  s->waited = false;
  hipStream_t st = s->stream;
  HIP_TRY(hipEventRecord(s->ev[0], st));
  if (flags & KR_BASES_DEVICE) {
    s->in.bases = bases;
    s->in.offsets = offsets;
  } else {
    uint64_t nb = offsets[nreads];
    if (nb > s->max_bases) return kr::fail(KR_ERR_ARG, "kr_batch_submit: more bases than the stream was created for");
    memcpy(s->h_bases, bases, nb);
    memcpy(s->h_offsets, offsets, ((uint64_t)nreads + 1) * 8);
    HIP_TRY(hipMemcpyAsync(s->d_bases, s->h_bases, nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->d_offsets, s->h_offsets, ((uint64_t)nreads + 1) * 8, hipMemcpyHostToDevice, st));
    s->in.bases = s->d_bases;
    s->in.offsets = s->d_offsets;
  }
  s->in.nreads = nreads;
  HIP_TRY(hipMemsetAsync(s->out.counters, 0, 128, st));
  HIP_TRY(hipMemsetAsync(s->out.cursors, 0, 3 * kCursors * kCursorStride * 4, st));
  HIP_TRY(hipMemsetAsync(s->out.rec_key, 0, (uint64_t)s->rec_cap * 4, st));
  HIP_TRY(hipMemsetAsync(s->out.rec_sel, 0, (uint64_t)s->rec_cap, st));
  HIP_TRY(hipEventRecord(s->ev[1], st));
  const DevIndex& dix = s->ix->dix;
  if (flags & KR_TAP_HITS) {
    if (!s->h_hits) {
      HIP_TRY(hipMalloc((void**)&s->out.hits, (uint64_t)s->hit_cap * sizeof(kr_hit)));
      s->dallocs.push_back(s->out.hits);
      HIP_TRY(hipHostMalloc((void**)&s->h_hits, (uint64_t)s->hit_cap * sizeof(kr_hit), hipHostMallocDefault));
      s->hallocs.push_back(s->h_hits);
    }
  }
  {
    const bool tap = (flags & KR_TAP_HITS) != 0;
    const uint32_t sgrid = std::min<uint32_t>((nreads + kScanWaves - 1) / kScanWaves, s->scan_blocks);
#define KR_LAUNCH2(LG, CP, SLV, SLT)                                                                                          \
  do {                                                                                                                   \
    if (tap)                                                                                                             \
      hipLaunchKernelGGL((kr_scan_kernel_t<LG, CP, SLV, true, SLT>), dim3(sgrid), dim3(kScanWaves* kWave), 0, st, dix, s->dp, s->in, s->out); \
    else                                                                                                                 \
      hipLaunchKernelGGL((kr_scan_kernel_t<LG, CP, SLV, false, SLT>), dim3(sgrid), dim3(kScanWaves* kWave), 0, st, dix, s->dp, s->in, s->out); \
  } while (0)
#define KR_LAUNCH(LG, CP, SLT)          \
  do {                                  \
    if (single)                         \
      KR_LAUNCH2(LG, CP, true, SLT);    \
    else                                \
      KR_LAUNCH2(LG, CP, false, SLT);   \
  } while (0)
    const bool single = dix.nlibs == 1 && dix.m <= 64;
    switch (s->ix->slot_log2w) {
      case 5: KR_LAUNCH(2, 2, true); break; // slotted table, 128-byte slots: 4 lanes x 2 chunks
      case 6: KR_LAUNCH(2, 4, true); break; // 256-byte slots: 4 lanes x 4 chunks
      case 7: KR_LAUNCH(3, 4, true); break; // 512-byte slots: 8 lanes x 4 chunks
      default:
        switch (s->ix->log_g) {
          case 0: KR_LAUNCH(0, 2, false); break; // sparse tables: a lane per probe
          case 2: KR_LAUNCH(2, 3, false); break; // 4 lanes x 3 chunks = 48 entries per pass
          default: KR_LAUNCH(3, 3, false); break; // 8 lanes x 3 chunks = 96 entries per pass
        }
    }
#undef KR_LAUNCH2
#undef KR_LAUNCH
    HIP_TRY(hipEventRecord(s->ev[2], st));
    const uint32_t lds = probe_lds_bytes(s->dp.np, s->out.bm_words), lds_lean = probe_lds_bytes(s->dp.np, s->out.bm_words, true);
    const uint32_t grid_lean = std::min(nreads, s->nwaves_lean), grid_full = std::min(nreads, s->nwaves_full);
    const bool np5 = s->dp.np == 5 && !getenv("KR_DEBUG_NP0");
#define KR_ACC(SLV, NPV)                                                                                                       \
  do {                                                                                                                       \
    hipLaunchKernelGGL((kr_acc_kernel_t<SLV, NPV, false>), dim3(grid_lean), dim3(kWave), lds_lean, st, dix, s->dp, s->in, s->out); \
    hipLaunchKernelGGL((kr_acc_kernel_t<SLV, NPV, true>), dim3(grid_full), dim3(kWave), lds, st, dix, s->dp, s->in, s->out);  \
  } while (0)
    if (single && np5)
      KR_ACC(true, 5);
    else if (single)
      KR_ACC(true, 0);
    else if (np5)
      KR_ACC(false, 5);
    else
      KR_ACC(false, 0);
#undef KR_ACC
  }
  HIP_TRY(hipEventRecord(s->ev[3], st));
  hipLaunchKernelGGL(kr_dedup_clear_kernel, dim3(4096), dim3(256), 0, st, s->out);
  if (s->llh.th == 4) {
    hipLaunchKernelGGL(kr_dedup_kernel, dim3(4096), dim3(256), 0, st, s->out);
    hipLaunchKernelGGL(kr_llh_pre_kernel<5>, dim3(4096), dim3(256), 0, st, s->llh, dix, s->out);
    hipLaunchKernelGGL(kr_llh_kernel<5>, dim3(2048), dim3(256), 0, st, s->llh, dix, s->out);
  } else {
    hipLaunchKernelGGL(kr_dedup_kernel, dim3(4096), dim3(256), 0, st, s->out);
    hipLaunchKernelGGL(kr_llh_pre_kernel<0>, dim3(4096), dim3(256), 0, st, s->llh, dix, s->out);
    hipLaunchKernelGGL(kr_llh_kernel<0>, dim3(2048), dim3(256), 0, st, s->llh, dix, s->out);
  }
  hipLaunchKernelGGL(kr_llh_copy_kernel, dim3(4096), dim3(256), 0, st, s->out);
  {
    const uint32_t sgrid = std::min<uint32_t>((nreads + 7) / 8, 16384u);
    const bool filt = !s->dp.no_filter && s->dp.multi;
    if (s->llh.th == 4) {
      if (filt)
        hipLaunchKernelGGL((kr_select_kernel<5, true>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, s->out, nreads);
      else
        hipLaunchKernelGGL((kr_select_kernel<5, false>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, s->out, nreads);
    } else {
      if (filt)
        hipLaunchKernelGGL((kr_select_kernel<0, true>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, s->out, nreads);
      else
        hipLaunchKernelGGL((kr_select_kernel<0, false>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, s->out, nreads);
    }
  }
  HIP_TRY(hipEventRecord(s->ev[4], st));
  HIP_TRY(hipGetLastError());
  return KR_OK;
}

int kr_batch_wait(kr_stream* s)
{
  if (!s || !s->submitted) return kr::fail(KR_ERR_STATE, "kr_batch_wait: nothing submitted");
  if (s->waited) return KR_OK;
  HIP_TRY(hipSetDevice(s->ix->device));
  HIP_TRY(hipMemcpyAsync(s->h_counters, s->out.counters, 128, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->waited = true;
  if (s->dp.dbg & 512u)
    fprintf(stderr, "[kr stats] reads %u events %u keys %u batches %u big %u level-tiles %u max keys %u max events %u records %u\n", s->nreads,
            s->h_counters[9], s->h_counters[10], s->h_counters[11], s->h_counters[12], s->h_counters[13], s->h_counters[14],
            s->h_counters[15], s->h_counters[4]);
  if (s->dp.dbg & 512u)
    fprintf(stderr, "[kr stats] wave cycles/64 per launch: level passes %u (zero %u, event passes %u, key passes %u), finalize %u, whole read %u\n",
            s->h_counters[16], s->h_counters[19], s->h_counters[20], s->h_counters[21], s->h_counters[17], s->h_counters[18]);
  s->nrecs = std::min(s->h_counters[0], s->rec_cap);
  s->nhits = std::min<uint64_t>(s->h_counters[3], s->hit_cap);
  if (s->h_counters[4] > s->rec_user_cap)
    return kr::fail(KR_ERR_CAPACITY, "the batch produced more records than max_records: submit fewer reads per batch");
  return check_errflags(s->h_counters[1]);
}

static void fill_view(kr_stream* s, kr_result_view* v, bool device)
{
  memset(v, 0, sizeof(*v));
  v->nreads = s->nreads;
  v->nrecs = s->nrecs;
  if (device) {
    v->read_off = s->out.rd_off, v->read_cnt = s->out.rd_cnt, v->read_onmers = s->out.rd_onmers, v->read_na = s->out.rd_na;
    v->rec_key = s->out.rec_key, v->rec_sel = s->out.rec_sel, v->rec_d = s->out.rec_d, v->rec_v = s->out.rec_v;
    v->rec_chisq = s->out.rec_chisq, v->rec_hist = s->out.rec_hist;
    v->rec_hist_stride = s->rec_cap;
  } else {
    v->read_off = s->h_rd_off, v->read_cnt = s->h_rd_cnt, v->read_onmers = s->h_rd_onmers, v->read_na = s->h_rd_na;
    v->rec_key = s->h_rec_key, v->rec_sel = s->h_rec_sel, v->rec_d = s->h_rec_d, v->rec_v = s->h_rec_v;
    v->rec_chisq = s->h_rec_chisq, v->rec_hist = (s->flags & KR_TAP_ACCS) ? s->h_rec_hist : nullptr;
    v->rec_hist_stride = s->nrecs;
  }
}

int kr_batch_collect(kr_stream* s, kr_result_view* v)
{
  kr::clear_error();
  if (!s || !v) return kr::fail(KR_ERR_ARG, "kr_batch_collect: null argument");
  int rc = kr_batch_wait(s);
  if (rc) return rc;
  hipStream_t st = s->stream;
  uint64_t nr = s->nreads, nc = s->nrecs;
  if (nc > s->h_rec_cap) { // (re)allocate pinned record buffers
    uint64_t cap = std::max<uint64_t>(nc + nc / 4, 1u << 16);
    void** olds[] = {(void**)&s->h_rec_key, (void**)&s->h_rec_hist, (void**)&s->h_rec_sel, (void**)&s->h_rec_d, (void**)&s->h_rec_v, (void**)&s->h_rec_chisq};
    for (void** o : olds)
      if (*o) {
        s->hallocs.erase(std::remove(s->hallocs.begin(), s->hallocs.end(), *o), s->hallocs.end());
        (void)hipHostFree(*o);
        *o = nullptr;
      }
    int rc2 = 0;
    if ((rc2 = halloc(s, &s->h_rec_key, cap)) || (rc2 = halloc(s, &s->h_rec_hist, cap * s->dp.np)) || (rc2 = halloc(s, &s->h_rec_sel, cap)) ||
        (rc2 = halloc(s, &s->h_rec_d, cap)) || (rc2 = halloc(s, &s->h_rec_v, cap)) || (rc2 = halloc(s, &s->h_rec_chisq, cap)))
      return rc2;
    s->h_rec_cap = cap;
  }
  HIP_TRY(hipMemcpyAsync(s->h_rd_off, s->out.rd_off, nr * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(s->h_rd_cnt, s->out.rd_cnt, nr * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(s->h_rd_onmers, s->out.rd_onmers, nr * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(s->h_rd_filt, s->out.rd_filt, nr * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(s->h_rd_na, s->out.rd_na, nr, hipMemcpyDeviceToHost, st));
  if (nc) {
    HIP_TRY(hipMemcpyAsync(s->h_rec_key, s->out.rec_key, nc * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s->h_rec_sel, s->out.rec_sel, nc, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s->h_rec_d, s->out.rec_d, nc * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s->h_rec_v, s->out.rec_v, nc * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s->h_rec_chisq, s->out.rec_chisq, nc * 8, hipMemcpyDeviceToHost, st));
    if (s->flags & KR_TAP_ACCS)
      for (uint32_t x = 0; x < s->dp.np; ++x)
        HIP_TRY(hipMemcpyAsync(s->h_rec_hist + (uint64_t)x * nc, s->out.rec_hist + (uint64_t)x * s->rec_cap, nc * 4, hipMemcpyDeviceToHost, st));
  }
  if ((s->flags & KR_TAP_HITS) && s->nhits)
    HIP_TRY(hipMemcpyAsync(s->h_hits, s->out.hits, s->nhits * sizeof(kr_hit), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  fill_view(s, v, false);
  uint64_t nrows = 0;
  for (uint64_t i = 0; i < nc; ++i) nrows += s->h_rec_sel[i];
  v->nrows = nrows;
  return KR_OK;
}

int kr_batch_collect_device(kr_stream* s, kr_result_view* v)
{
  kr::clear_error();
  if (!s || !v) return kr::fail(KR_ERR_ARG, "kr_batch_collect_device: null argument");
  int rc = kr_batch_wait(s);
  if (rc) return rc;
  fill_view(s, v, true);
  return KR_OK;
}

int kr_batch_hits(kr_stream* s, const kr_hit** hits, uint64_t* nhits)
{
  if (!s || !hits || !nhits || !s->waited) return kr::fail(KR_ERR_STATE, "kr_batch_hits: collect a KR_TAP_HITS batch first");
  *hits = s->h_hits;
  *nhits = s->nhits;
  return KR_OK;
}

int kr_batch_readtaps(kr_stream* s, const kr_readtap** taps)
{
  if (!s || !taps || !s->waited) return kr::fail(KR_ERR_STATE, "kr_batch_readtaps: collect a batch first");
  *taps = reinterpret_cast<const kr_readtap*>(s->h_rd_filt);
  return KR_OK;
}

int kr_batch_timing(kr_stream* s, kr_timing* t)
{
  if (!s || !t || !s->waited) return kr::fail(KR_ERR_STATE, "kr_batch_timing: wait for a batch first");
  memset(t, 0, sizeof(*t));
  HIP_TRY(hipEventElapsedTime(&t->ms_h2d, s->ev[0], s->ev[1]));
  HIP_TRY(hipEventElapsedTime(&t->ms_scan, s->ev[1], s->ev[2]));
  HIP_TRY(hipEventElapsedTime(&t->ms_acc, s->ev[2], s->ev[3]));
  HIP_TRY(hipEventElapsedTime(&t->ms_llh, s->ev[3], s->ev[4]));
  HIP_TRY(hipEventElapsedTime(&t->ms_total, s->ev[1], s->ev[4]));
  t->overflow_reads = s->h_counters[2];
  t->stack_spills = s->h_counters[26];
  return KR_OK;
}

int kr_debug_front_end(const kr_index* ix, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads, uint32_t stride,
                       uint32_t* rix, uint32_t* enc32, uint8_t* valid, uint8_t* pass)
{
  kr::clear_error();
  if (!ix || !bases || !offsets || !rix || !enc32 || !valid || !pass || !nreads) return kr::fail(KR_ERR_ARG, "kr_debug_front_end: bad argument");
  HIP_TRY(hipSetDevice(ix->device));
  uint64_t nb = offsets[nreads], n = (uint64_t)nreads * stride * 2;
  uint8_t *d_b = nullptr, *d_valid = nullptr, *d_pass = nullptr;
  uint64_t* d_o = nullptr;
  uint32_t *d_rix = nullptr, *d_enc = nullptr;
  HIP_TRY(hipMalloc((void**)&d_b, nb + 256));
  HIP_TRY(hipMalloc((void**)&d_o, ((uint64_t)nreads + 1) * 8));
  HIP_TRY(hipMalloc((void**)&d_rix, n * 4 + 16));
  HIP_TRY(hipMalloc((void**)&d_enc, n * 4 + 16));
  HIP_TRY(hipMalloc((void**)&d_valid, n + 16));
  HIP_TRY(hipMalloc((void**)&d_pass, n + 16));
  HIP_TRY(hipMemcpy(d_b, bases, nb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_o, offsets, ((uint64_t)nreads + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_rix, 0, n * 4));
  HIP_TRY(hipMemset(d_enc, 0, n * 4));
  HIP_TRY(hipMemset(d_valid, 0, n));
  HIP_TRY(hipMemset(d_pass, 0, n));
  BatchIn in{d_b, d_o, nreads};
  hipLaunchKernelGGL(kr_front_end_kernel, dim3(std::min(nreads, 4096u)), dim3(kWave), 0, 0, ix->dix, in, stride, d_rix, d_enc, d_valid, d_pass);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(rix, d_rix, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(enc32, d_enc, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(valid, d_valid, n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(pass, d_pass, n, hipMemcpyDeviceToHost));
  hipFree(d_b), hipFree(d_o), hipFree(d_rix), hipFree(d_enc), hipFree(d_valid), hipFree(d_pass);
  return KR_OK;
}

int kr_llh_batch(const kr_index* ix, uint32_t th, uint32_t mode, uint64_t n, const double* hist, const double* uc, const double* rho,
                 const double* d_in, double* d_out, double* v_out)
{
  kr::clear_error();
  if (!ix || th > KR_MAX_HDIST_TH || mode > 1 || (n && (!hist || !uc || !rho || !v_out)) || (mode == 0 && n && !d_out) ||
      (mode == 1 && n && !d_in))
    return kr::fail(KR_ERR_ARG, "kr_llh_batch: bad argument");
  if (n == 0) return KR_OK;
  HIP_TRY(hipSetDevice(ix->device));
  LlhConst C = make_llh_const(ix->dix.k, ix->dix.h, th);
  const uint64_t np = th + 1, need = n * (np + 5);
  std::lock_guard<std::mutex> lk(ix->llh_mu);
  if (need > ix->llh_cap) {
    if (ix->llh_dev) (void)hipFree(ix->llh_dev);
    if (ix->llh_pin) (void)hipHostFree(ix->llh_pin);
    ix->llh_dev = ix->llh_pin = nullptr, ix->llh_cap = 0;
    const uint64_t cap = need + need / 4;
    HIP_TRY(hipMalloc((void**)&ix->llh_dev, cap * 8));
    HIP_TRY(hipHostMalloc((void**)&ix->llh_pin, cap * 8, hipHostMallocDefault));
    ix->llh_cap = cap;
  }
  double *pin = ix->llh_pin, *dev = ix->llh_dev;
  const uint64_t o_uc = n * np, o_rho = o_uc + n, o_di = o_rho + n, o_do = o_di + n, o_v = o_do + n;
  { // stage the inputs in pinned memory (a few threads: tens of MB for a large batch), one H2D copy
    const int nt = n >= (1u << 16) ? 4 : 1;
    auto piece = [&](int t) {
      const uint64_t a = n * (uint64_t)t / nt, b = n * (uint64_t)(t + 1) / nt;
      memcpy(pin + a * np, hist + a * np, (b - a) * np * 8);
      memcpy(pin + o_uc + a, uc + a, (b - a) * 8);
      memcpy(pin + o_rho + a, rho + a, (b - a) * 8);
      if (mode == 1) memcpy(pin + o_di + a, d_in + a, (b - a) * 8);
    };
    std::vector<std::thread> th_;
    for (int t = 1; t < nt; ++t) th_.emplace_back(piece, t);
    piece(0);
    for (auto& t : th_) t.join();
  }
  HIP_TRY(hipMemcpy(dev, pin, (mode == 1 ? o_do : o_di) * 8, hipMemcpyHostToDevice));
  uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(kr_llh_batch_kernel, dim3(grid), dim3(256), 0, 0, C, mode, n, dev, dev + o_uc, dev + o_rho, dev + o_di, dev + o_do,
                     dev + o_v, (const uint32_t*)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(pin + o_do, dev + o_do, 2 * n * 8, hipMemcpyDeviceToHost)); // synchronises with the kernel (null stream)
  if (mode == 0) memcpy(d_out, pin + o_do, n * 8);
  memcpy(v_out, pin + o_v, n * 8);
  return KR_OK;
}

// f_{problem pidx[i]}(d_in[i]) for i < n: many evaluations of few problems (the chi-square tests of `place`: every
// candidate of a read against the read's closest leaf).  Workspace layout [hist nprob*np | uc | rho | d_in n | v n | pidx n].
int kr_llh_eval_indexed(const kr_index* ix, uint32_t th, uint64_t nprob, const double* hist, const double* uc, const double* rho,
                        uint64_t n, const uint32_t* pidx, const double* d_in, double* v_out)
{
  kr::clear_error();
  if (!ix || th > KR_MAX_HDIST_TH || (n && (!hist || !uc || !rho || !pidx || !d_in || !v_out || !nprob)))
    return kr::fail(KR_ERR_ARG, "kr_llh_eval_indexed: bad argument");
  if (n == 0) return KR_OK;
  HIP_TRY(hipSetDevice(ix->device));
  LlhConst C = make_llh_const(ix->dix.k, ix->dix.h, th);
  const uint64_t np = th + 1, o_uc = nprob * np, o_rho = o_uc + nprob, o_di = o_rho + nprob, o_v = o_di + n, o_ix = o_v + n,
                 need = o_ix + (n + 1) / 2;
  std::lock_guard<std::mutex> lk(ix->llh_mu);
  if (need > ix->llh_cap) {
    if (ix->llh_dev) (void)hipFree(ix->llh_dev);
    if (ix->llh_pin) (void)hipHostFree(ix->llh_pin);
    ix->llh_dev = ix->llh_pin = nullptr, ix->llh_cap = 0;
    const uint64_t cap = need + need / 4;
    HIP_TRY(hipMalloc((void**)&ix->llh_dev, cap * 8));
    HIP_TRY(hipHostMalloc((void**)&ix->llh_pin, cap * 8, hipHostMallocDefault));
    ix->llh_cap = cap;
  }
  double *pin = ix->llh_pin, *dev = ix->llh_dev;
  memcpy(pin, hist, nprob * np * 8);
  memcpy(pin + o_uc, uc, nprob * 8);
  memcpy(pin + o_rho, rho, nprob * 8);
  memcpy(pin + o_di, d_in, n * 8);
  memcpy(pin + o_ix, pidx, n * 4);
  HIP_TRY(hipMemcpy(dev, pin, o_v * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dev + o_ix, pin + o_ix, n * 4, hipMemcpyHostToDevice));
  uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(kr_llh_batch_kernel, dim3(grid), dim3(256), 0, 0, C, 1u, n, dev, dev + o_uc, dev + o_rho, dev + o_di, (double*)nullptr,
                     dev + o_v, (const uint32_t*)(dev + o_ix));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(pin + o_v, dev + o_v, n * 8, hipMemcpyDeviceToHost));
  memcpy(v_out, pin + o_v, n * 8);
  return KR_OK;
}

int kr_debug_brent(const kr_index* ix, uint32_t th, uint32_t n, const uint32_t* hist, const uint32_t* onmers, const double* rho,
                   double* d_out, double* v_out)
{
  kr::clear_error();
  if (!ix || !hist || !onmers || !rho || !d_out || !v_out || !n || th > KR_MAX_HDIST_TH) return kr::fail(KR_ERR_ARG, "kr_debug_brent: bad argument");
  HIP_TRY(hipSetDevice(ix->device));
  LlhConst C = make_llh_const(ix->dix.k, ix->dix.h, th);
  uint32_t *d_h = nullptr, *d_on = nullptr;
  double *d_rho = nullptr, *d_d = nullptr, *d_v = nullptr;
  HIP_TRY(hipMalloc((void**)&d_h, (uint64_t)n * (th + 1) * 4));
  HIP_TRY(hipMalloc((void**)&d_on, (uint64_t)n * 4));
  HIP_TRY(hipMalloc((void**)&d_rho, (uint64_t)n * 8));
  HIP_TRY(hipMalloc((void**)&d_d, (uint64_t)n * 8));
  HIP_TRY(hipMalloc((void**)&d_v, (uint64_t)n * 8));
  HIP_TRY(hipMemcpy(d_h, hist, (uint64_t)n * (th + 1) * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_on, onmers, (uint64_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_rho, rho, (uint64_t)n * 8, hipMemcpyHostToDevice));
  if (th == 4)
    hipLaunchKernelGGL(kr_brent_kernel<5>, dim3((n + 127) / 128), dim3(128), 0, 0, C, n, d_h, d_on, d_rho, d_d, d_v);
  else
    hipLaunchKernelGGL(kr_brent_kernel<0>, dim3((n + 127) / 128), dim3(128), 0, 0, C, n, d_h, d_on, d_rho, d_d, d_v);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(d_out, d_d, (uint64_t)n * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(v_out, d_v, (uint64_t)n * 8, hipMemcpyDeviceToHost));
  hipFree(d_h), hipFree(d_on), hipFree(d_rho), hipFree(d_d), hipFree(d_v);
  return KR_OK;
}

} // extern "C"
