// kr_device.hip — the per-read `krepp dist` hot path as hand-written HIP for gfx950
// (MI355X), and the device half of the C ABI (include/krepp_amd.h).
//
// Pipeline per submitted batch (one HIP stream per kr_stream):
//
//   kr_probe_kernel      one wave64 per read.  Front end from wave ballots (no LDS, no
//                        rolling state): every k-mer x strand -> LSH row (rix) and
//                        residual code (enc32)          [src/query.cpp:40-94,
//                        src/common.hpp:177-243, src/lshf.cpp:39-69]; bucket lookup
//                        [src/index.cpp:160-168]; bucket scan with lanes flattened over
//                        16-byte chunks of the bucket, Hamming filter
//                        [src/query.cpp:361-368]; colour-DAG expansion from an LDS work
//                        stack [src/query.cpp:369-387]; per-(strand, leaf) accumulation
//                        as position bit-planes in an LDS hash table
//                        [Minfo::update_match, src/query.hpp:153-176]; hdist_filt test
//                        [src/query.cpp:101-106,119] and record emission.
//   kr_probe_overflow_kernel  same code, accumulator table in global memory, for reads
//                        whose leaf set does not fit the LDS table.
//   kr_llh_kernel        one lane per (read, strand, leaf) record: Brent minimisation of
//                        HDistHistLLH in fp64 [src/hdhistllh.hpp:51-96,
//                        src/query.cpp:426-433, boost::math::tools::brent_find_minima].
//   kr_select_kernel     one lane per read: strand merge, closest reference, --filter /
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

#include "kr_common.h"
#include "kr_devutil.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
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
constexpr int kLdsSlots = 64;     // level-1 (LDS) accumulator slots per wave: lane t owns slot t in the epilogue
constexpr int kLdsProbeMax = 8;   // bounded probe sequence of the level-1 table
constexpr int kMaxPlanes = KR_MAX_HDIST_TH + 1;

struct DevLib {
  const uint64_t* bkt;  // [nrows]  (start << 24) | len
  const uint32_t* enc;  // [nkmers + pad] residual codes, bucket-contiguous
  const uint32_t* se;   // [nkmers] colour ids
  const uint2* pse;     // [nsubsets] colour DAG: colour = union of .x and .y
  const double* rho;    // [nnodes] subsampling rates, already scaled
  uint64_t nkmers;
  uint32_t nrows, nsubsets, nnodes, numer;
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
  uint32_t dbg; // KR_DEBUG_SKIP (timing experiments only): 1 drop hits, 2 drop expansion, 4 skip scan
  double chisq, dist_max;
};

// Everything the kernels write for one batch.
struct BatchOut {
  uint32_t* counters;    // [0] record slots handed out  [1] error flags  [2] reads that used level 2  [3] nhits(tap)  [4] records
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
  uint32_t rec_cap;
  kr_hit* hits;
  uint32_t hit_cap;
  // level-2 accumulator scratch, one region per resident wave
  uint32_t* g_planes; // [nwaves][nslots2][np][4]
  uint32_t* g_counts; // [nwaves][nslots2][np]
  uint32_t* g_list;   // [nwaves][nslots2]
  uint32_t nslots2;   // 2 * nleaves
  uint32_t g_list_words; // per wave: max(nslots2, event capacity)
  uint32_t bm_words;  // ceil(nslots2 / 32)
};

enum : uint32_t { kErrRecCap = 1u, kErrStack = 2u, kErrTable = 4u, kErrHitCap = 8u };

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
  uint32_t z = a ^ b;
  return __popc((z | (z >> 16)) & 0xFFFFu);
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
  lds_u32* bitmap;   // LDS [bm_words]
  uint32_t* g_planes; // global [nslots2 * np * 4]
  uint32_t* g_counts; // global [nslots2 * np]
  uint32_t* g_list;   // global [nslots2]
  uint32_t nslots2, bm_words, np;
};

__device__ __forceinline__ uint32_t gload(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Work-stack item (8 B).  hi = pos(7) | strand(1)<<7 | lib(4)<<8 | hd(5)<<12 | idx_hi(8)<<17 |
// unresolved<<31.  Resolved: lo = colour id.  Unresolved (a fresh table hit whose colour has not
// been fetched yet): lo | idx_hi<<32 = entry index into the library's se[] array.
constexpr uint32_t kItemUnresolved = 0x80000000u;
__device__ __forceinline__ uint32_t tag_pos(uint32_t t) { return t & 127u; }
__device__ __forceinline__ uint32_t tag_strand(uint32_t t) { return (t >> 7) & 1u; }
__device__ __forceinline__ uint32_t tag_lib(uint32_t t) { return (t >> 8) & 15u; }
__device__ __forceinline__ uint32_t tag_hd(uint32_t t) { return (t >> 12) & 31u; }

struct WaveState {
  lds_u64* stack;   // LDS [kStackCap]: lo | hi << 32
  uint32_t top;   // wave-uniform
  bool l2;        // this lane sent something to level 2 during this read
  uint32_t err;
  uint32_t read;  // for the hit tap
  uint32_t base0; // first k-mer position of the current segment
  uint32_t rec_next, rec_end; // wave-private range of record slots (wave-uniform)
  uint32_t n_l2;              // reads of this wave that used level 2
  uint32_t n_rec;             // records this wave emitted
  // event mode (reads of a single segment): leaf updates are appended as 32-bit events
  bool evmode;      // wave-uniform
  bool ev_full;     // wave-uniform: the event buffer overflowed
  uint32_t nev;     // events buffered (wave-uniform)
  uint32_t ev_cap;  // power of two
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
    if (ws.nev + c <= ws.ev_cap) {
      if (f) ws.ev[ws.nev + __popcll(m & ((1ull << lane_id()) - 1ull))] = ev;
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

// Colour expansion (the BFS of src/query.cpp:369-387, order-free here): drain the work stack.
template <bool SL, bool TAP>
__device__ __forceinline__ void expand_all(const DevIndex& ix, const BatchOut& out, const Acc& A, WaveState& ws)
{
  const uint32_t lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  while (ws.top > 0) {
    uint32_t room = kStackCap - ws.top;
    uint32_t n = min(min(64u, ws.top), room);
    if (n == 0) { // cannot make progress: report, drop the rest
      ws.err |= kErrStack;
      ws.top = 0;
      break;
    }
    uint32_t base = ws.top - n;
    bool have = lane < n;
    uint2 item = make_uint2(0, 0);
    if (have) {
      uint64_t raw = ws.stack[base + lane];
      item = make_uint2((uint32_t)raw, (uint32_t)(raw >> 32));
    }
    ws.top = base;
    uint32_t se = item.x, tag = item.y;
    uint32_t c0 = 0, c1 = 0;
    bool p0 = false, p1 = false;
    bool f0 = false, f1 = false, f2 = false; // leaf updates found by this lane: the item, child 0, child 1
    if (have) {
      DevLib L = get_lib<SL>(ix, tag_lib(tag));
      if (tag & kItemUnresolved) { // fetch the colour of a fresh hit
        uint64_t idx = (uint64_t)item.x | ((uint64_t)((tag >> 17) & 0xFFu) << 32);
        se = L.se[idx];
        if (TAP) {
          uint32_t hix = atomicAdd(&out.counters[3], 1u);
          if (hix < out.hit_cap) {
            kr_hit h;
            h.read = ws.read;
            h.kpos = ws.base0 + tag_pos(tag);
            h.strand = tag_strand(tag);
            h.lib = tag_lib(tag);
            h.cmer_index = idx;
            h.hd = tag_hd(tag);
            h.se = (se >> 30) == 1u ? ix.leaf_se[se & kColMask] : (se & kColMask);
            out.hits[hix] = h;
          } else {
            atomicOr(&out.counters[1], kErrHitCap);
          }
        }
        tag &= 0x1FFFFu;
      }
      f0 = (se >> 30) == 1u;
      if ((se >> 30) == 2u) {
        uint2 pr = L.pse[se & kColMask];
        c0 = pr.x;
        c1 = pr.y;
        f1 = (c0 >> 30) == 1u, p0 = (c0 >> 30) == 2u;
        f2 = (c1 >> 30) == 1u, p1 = (c1 >> 30) == 2u;
      }
    }
    add_leaf_events(A, ws, f0, make_event(se & kColMask, tag));
    add_leaf_events(A, ws, f1, make_event(c0 & kColMask, tag));
    add_leaf_events(A, ws, f2, make_event(c1 & kColMask, tag));
    uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
    if (p0) ws.stack[ws.top + __popcll(m0 & lt)] = (uint64_t)c0 | ((uint64_t)tag << 32);
    ws.top += __popcll(m0);
    if (p1) ws.stack[ws.top + __popcll(m1 & lt)] = (uint64_t)c1 | ((uint64_t)tag << 32);
    ws.top += __popcll(m1);
    WAVE_SYNC();
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

// front end + descriptor loads for positions 64*pp + lane
template <bool SL>
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
    c.b0 = get_lib<SL>(ix, (uint32_t)lib).bkt[row];
  }
  if (fe.valid && locate_row<SL>(ix, fe.rix[1], lib, row)) {
    c.lib1 = (uint32_t)lib;
    c.b1 = get_lib<SL>(ix, (uint32_t)lib).bkt[row];
  }
  return c;
}

// 4 entries of one 16-byte chunk against the query code: 4-bit hit mask, 4 x 8-bit hd
__device__ __forceinline__ void chunk_hits(uint4 v, uint64_t e0, uint64_t st, uint32_t ln, uint32_t q, uint32_t th,
                                           uint32_t& mask, uint32_t& hds)
{
  uint32_t h0 = hd_lr32(v.x, q), h1 = hd_lr32(v.y, q), h2 = hd_lr32(v.z, q), h3 = hd_lr32(v.w, q);
  uint64_t en = st + ln;
  bool b0 = e0 >= st && e0 < en && h0 <= th;
  bool b1 = e0 + 1 >= st && e0 + 1 < en && h1 <= th;
  bool b2 = e0 + 2 >= st && e0 + 2 < en && h2 <= th;
  bool b3 = e0 + 3 >= st && e0 + 3 < en && h3 <= th;
  mask = (b0 ? 1u : 0u) | (b1 ? 2u : 0u) | (b2 ? 4u : 0u) | (b3 ? 8u : 0u);
  hds = h0 | (h1 << 8) | (h2 << 16) | (h3 << 24);
}

// Scan the listed buckets: G = 2^LOG_G consecutive lanes share one probe and read consecutive
// aligned 16-byte chunks of its bucket (G*16 contiguous bytes per step); 64/G probes per pass;
// two chunks per lane per pass, and the loads of the NEXT pass are issued before the current
// pass is examined (software pipeline: four 16-byte loads in flight per lane).  No search, no
// prefix sums: the probe of a lane is fixed by its lane id.
struct ScanStep {
  uint64_t st, eA;
  uint32_t ln, q, tg, nch;
  uint4 vA, vB;
};

template <int LOG_G, bool SL>
__device__ __forceinline__ void scan_issue(const DevIndex& ix, const ProbeList& pl, uint32_t nact, uint32_t p0, ScanStep& S)
{
  constexpr uint32_t G = 1u << LOG_G;
  const uint32_t lane = lane_id();
  const uint32_t sub = lane & (G - 1u), pi = p0 + (lane >> LOG_G);
  const bool on = pi < nact;
  const uint64_t b = on ? pl.bkt[pi] : 0ull;
  S.q = on ? pl.q[pi] : 0u;
  S.tg = on ? pl.tag[pi] : 0u;
  S.st = b >> 24;
  S.ln = (uint32_t)(b & 0xFFFFFFu);
  S.nch = on ? (uint32_t)(((S.st & 3u) + S.ln + 3u) >> 2) : 0u;
  S.eA = (S.st & ~3ull) + 4ull * sub;
  S.vA = make_uint4(0, 0, 0, 0);
  S.vB = S.vA;
  const uint32_t* enc = get_lib<SL>(ix, tag_lib(S.tg)).enc;
  if (sub < S.nch) S.vA = *reinterpret_cast<const uint4*>(enc + S.eA);
  if (sub + G < S.nch) S.vB = *reinterpret_cast<const uint4*>(enc + S.eA + 4ull * G);
}

// push the hits of two chunks (A at entry eA, B at entry eB) as unresolved work items
template <bool SL, bool TAP>
__device__ __forceinline__ void push_hits(const DevIndex& ix, const BatchOut& out, const Acc& A, WaveState& ws, uint32_t tg,
                                          uint64_t eA, uint64_t eB, uint32_t mA, uint32_t hA, uint32_t mB, uint32_t hB,
                                          uint32_t& filt0, uint32_t& filt1, uint32_t P_dbg)
{
  const uint32_t lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t pend = mA | (mB << 4);
  if (P_dbg & 1u) pend = 0;
  while (__ballot(pend != 0) != 0) {
    if (ws.top > (uint32_t)(kStackCap - 64)) expand_all<SL, TAP>(ix, out, A, ws);
    const bool has = pend != 0;
    const uint32_t bit = has ? (uint32_t)__ffs((int)pend) - 1u : 0u;
    const uint32_t e = bit & 3u;
    const uint64_t hm = __ballot(has);
    if (has) {
      const uint32_t hd = (((bit >> 2) ? hB : hA) >> (8 * e)) & 31u;
      const uint64_t idx = ((bit >> 2) ? eB : eA) + e;
      if (tag_strand(tg))
        filt1 = min(filt1, hd);
      else
        filt0 = min(filt0, hd);
      ws.stack[ws.top + __popcll(hm & lt)] =
        (uint64_t)(uint32_t)idx | ((uint64_t)(tg | (hd << 12) | ((uint32_t)(idx >> 32) << 17) | kItemUnresolved) << 32);
      pend &= pend - 1u;
    }
    ws.top += __popcll(hm);
    WAVE_SYNC();
  }
}

template <int LOG_G, bool SL, bool TAP>
__device__ __forceinline__ void scan_list(const DevIndex& ix, const DevParams& P, const BatchOut& out, const Acc& A,
                                          WaveState& ws, const ProbeList& pl, uint32_t nact, uint32_t& filt0,
                                          uint32_t& filt1)
{
  constexpr uint32_t G = 1u << LOG_G, PPS = 64u >> LOG_G;
  const uint32_t sub = lane_id() & (G - 1u);
  if (nact == 0) return;
  ScanStep cur, nxt;
  scan_issue<LOG_G, SL>(ix, pl, nact, 0, cur);
  for (uint32_t p0 = 0; p0 < nact; p0 += PPS) {
    nxt = cur;
    if (p0 + PPS < nact) scan_issue<LOG_G, SL>(ix, pl, nact, p0 + PPS, nxt);
    // ---- first two chunks of every probe of this pass (already loaded)
    {
      uint32_t mA = 0, hA = 0, mB = 0, hB = 0;
      const uint64_t eB = cur.eA + 4ull * G;
      if (sub < cur.nch) chunk_hits(cur.vA, cur.eA, cur.st, cur.ln, cur.q, P.th, mA, hA);
      if (sub + G < cur.nch) chunk_hits(cur.vB, eB, cur.st, cur.ln, cur.q, P.th, mB, hB);
      push_hits<SL, TAP>(ix, out, A, ws, cur.tg, cur.eA, eB, mA, hA, mB, hB, filt0, filt1, P.dbg);
    }
    // ---- buckets longer than 2*G chunks: the rest, two chunks per lane at a time
    if (__ballot(cur.nch > 2u * G) != 0) {
      const uint32_t* enc = get_lib<SL>(ix, tag_lib(cur.tg)).enc;
      for (uint32_t c = sub + 2u * G; __ballot(c < cur.nch) != 0; c += 2u * G) {
        const bool onA = c < cur.nch, onB = c + G < cur.nch;
        const uint64_t eA = (cur.st & ~3ull) + 4ull * c, eB = eA + 4ull * G;
        uint4 vA = make_uint4(0, 0, 0, 0), vB = vA;
        if (onA) vA = *reinterpret_cast<const uint4*>(enc + eA);
        if (onB) vB = *reinterpret_cast<const uint4*>(enc + eB);
        uint32_t mA = 0, hA = 0, mB = 0, hB = 0;
        if (onA) chunk_hits(vA, eA, cur.st, cur.ln, cur.q, P.th, mA, hA);
        if (onB) chunk_hits(vB, eB, cur.st, cur.ln, cur.q, P.th, mB, hB);
        push_hits<SL, TAP>(ix, out, A, ws, cur.tg, eA, eB, mA, hA, mB, hB, filt0, filt1, P.dbg);
      }
    }
    cur = nxt;
  }
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
__device__ __forceinline__ uint32_t l2_build_list(const Acc& A)
{
  const uint32_t lane = lane_id();
  uint32_t n = 0;
  for (uint32_t w0 = 0; w0 < A.bm_words; w0 += 64) {
    uint32_t wi = w0 + lane;
    uint32_t word = wi < A.bm_words ? A.bitmap[wi] : 0u;
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
// Event mode epilogue: sort the events, keep the first event of every (leaf, strand, pos) group
// (= the smallest hd at that position: the rule of Minfo::update_match, src/query.hpp:153-176),
// histogram per (leaf, strand), apply the hdist_filt test, emit records in key order.
// All in LDS; the histogram table lives in the (now idle) stack + probe-list regions.
// Returns false if the read has more distinct keys than the table holds (caller falls back).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_lds(lds_u32* e, uint32_t n) // n: power of two >= 64
{
  const uint32_t lane = lane_id();
  if (n == 64) { // one element per lane: sort in registers
    uint32_t v = e[lane];
#pragma unroll
    for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        uint32_t o = __shfl_xor(v, j);
        bool up = (lane & k) == 0, lower = (lane & j) == 0;
        v = (lower == up) ? min(v, o) : max(v, o);
      }
    }
    e[lane] = v;
    WAVE_SYNC();
    return;
  }
  for (uint32_t k = 2; k <= n; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = lane; t < (n >> 1); t += 64) {
        uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), q = i + j;
        uint32_t a = e[i], b = e[q];
        bool up = (i & k) == 0;
        if ((a > b) == up) {
          e[i] = b;
          e[q] = a;
        }
      }
      WAVE_SYNC();
    }
  }
}

__device__ __forceinline__ bool finalize_events(const DevIndex& ix, const BatchOut& out, const Acc& A, WaveState& ws,
                                                lds_u32* hist, uint32_t hist_words, uint32_t read, uint32_t onmers,
                                                uint32_t filt0, uint32_t filt1)
{
  const uint32_t lane = lane_id();
  const uint64_t le = (2ull << lane) - 1ull, lt = (1ull << lane) - 1ull;
  const uint32_t nev = ws.nev;
  const uint32_t lim0 = 2u * filt0 + 1u, lim1 = 2u * filt1 + 1u; // u32 wrap keeps "none" = max
  lds_u32* e = ws.ev;
  uint32_t nrec = 0, nkeys = 0, npad = 0;
  if (nev) {
    npad = 64;
    while (npad < nev) npad <<= 1;
    for (uint32_t i = nev + lane; i < npad; i += 64) e[i] = 0xFFFFFFFFu;
    WAVE_SYNC();
    bitonic_sort_lds(e, npad);
    // ---- distinct (leaf, strand) keys
    for (uint32_t t0 = 0; t0 < nev; t0 += 64) {
      uint32_t i = t0 + lane;
      bool kl = i < nev && (i == 0 || (e[i] >> 12) != (e[i - 1] >> 12));
      nkeys += __popcll(__ballot(kl));
    }
    const uint32_t hw = (A.np + 3u) >> 2; // four 8-bit counters per word: a segment has <= 128 positions
    if (nkeys * hw > hist_words) return false;
    for (uint32_t i = lane; i < nkeys * hw; i += 64) hist[i] = 0;
    WAVE_SYNC();
    // ---- histogram of the position leaders; key table written in place over the sorted events
    uint32_t run = 0, carry = 0xFFFFFFFFu; // carry = last event of the previous tile (kept in a register:
                                           // the key table is written over the events as we go)
    for (uint32_t t0 = 0; t0 < nev; t0 += 64) {
      uint32_t i = t0 + lane;
      bool valid = i < nev;
      uint32_t v = valid ? e[i] : 0xFFFFFFFFu;
      uint32_t pv = __shfl_up(v, 1);
      if (lane == 0) pv = carry;
      carry = __shfl(v, 63);
      bool kl = valid && (i == 0 || (v >> 12) != (pv >> 12));
      bool plead = valid && (i == 0 || (v >> 5) != (pv >> 5));
      uint64_t km = __ballot(kl);
      uint32_t ord = run + __popcll(km & le) - 1u; // ordinal of my key (ord <= i)
      WAVE_SYNC();                                  // every lane holds its event before the table is written
      if (kl) e[ord] = v >> 12;
      if (plead)
        __hip_atomic_fetch_add(&hist[ord * hw + ((v & 31u) >> 2)], 1u << (8u * (v & 3u)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      run += __popcll(km);
      WAVE_SYNC();
    }
    // ---- records that pass `hdist_min <= 2*hdist_filt+1` (src/query.cpp:101-106,119)
    for (uint32_t t0 = 0; t0 < nkeys; t0 += 64) {
      uint32_t o = t0 + lane;
      bool ok = false;
      if (o < nkeys) {
        uint32_t hmin = 0xFFFFFFFFu;
        for (uint32_t x = 0; x < A.np; ++x)
          if ((hist[o * hw + (x >> 2)] >> (8u * (x & 3u))) & 255u) {
            hmin = x;
            break;
          }
        ok = hmin <= ((e[o] & 1u) ? lim1 : lim0);
      }
      nrec += __popcll(__ballot(ok));
    }
  }
  const uint32_t rbase = nrec ? alloc_records(out, ws, nrec) : 0u;
  if (lane == 0) {
    out.rd_off[read] = rbase == 0xFFFFFFFFu ? 0 : rbase;
    out.rd_cnt[read] = rbase == 0xFFFFFFFFu ? 0 : nrec;
    out.rd_onmers[read] = onmers;
    out.rd_filt[2 * read] = filt0;
    out.rd_filt[2 * read + 1] = filt1;
  }
  if (nrec && rbase != 0xFFFFFFFFu) {
    const uint32_t hw = (A.np + 3u) >> 2;
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < nkeys; t0 += 64) {
      uint32_t o = t0 + lane;
      bool ok = false;
      uint32_t rs = 0;
      if (o < nkeys) {
        rs = e[o];
        uint32_t hmin = 0xFFFFFFFFu;
        for (uint32_t x = 0; x < A.np; ++x)
          if ((hist[o * hw + (x >> 2)] >> (8u * (x & 3u))) & 255u) {
            hmin = x;
            break;
          }
        ok = hmin <= ((rs & 1u) ? lim1 : lim0);
      }
      uint64_t okm = __ballot(ok);
      if (ok) {
        uint32_t ri = rbase + run + __popcll(okm & lt);
        out.rec_read[ri] = read;
        out.rec_key[ri] = (ix.leaf_se[rs >> 1] << 1) | (rs & 1u);
        for (uint32_t x = 0; x < A.np; ++x)
          out.rec_hist[(uint64_t)x * out.rec_cap + ri] = (hist[o * hw + (x >> 2)] >> (8u * (x & 3u))) & 255u;
      }
      run += __popcll(okm);
    }
  }
  // the event buffer aliases the plane tables: leave it zeroed
  WAVE_SYNC();
  for (uint32_t i = lane; i < npad; i += 64) e[i] = 0;
  WAVE_SYNC();
  return true;
}

// ---------------------------------------------------------------------------
// One read.
// ---------------------------------------------------------------------------
template <int LOG_G, bool SL, bool TAP>
__device__ __forceinline__ void process_read(const DevIndex& ix, const DevParams& P, const BatchIn& in,
                                             const BatchOut& out, uint32_t read, const Acc& A, WaveState& ws,
                                             const ProbeList& pl, lds_u32* hist_tbl, uint32_t hist_words)
{
  const uint32_t lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint64_t off0 = in.offsets[read], off1 = in.offsets[read + 1];
  const uint8_t* seq = in.bases + off0;
  const uint64_t len = off1 - off0;
  const uint32_t k = ix.k;
  const uint64_t nkm = len >= k ? len - k + 1 : 0; // enmers (src/query.cpp:42)
  uint32_t onmers = 0;
  uint32_t filt0 = 0xFFFFFFFFu, filt1 = 0xFFFFFFFFu;
  bool l2_any = false; // wave-uniform: some key of this read lives in level 2
  ws.read = read;
  ws.evmode = nkm <= (uint64_t)kSegPos && !(P.dbg & 8u); // single segment: event mode
  // A read is processed once; only if its events do not fit (buffer or histogram table) is it
  // processed a second time with the plane tables.
  for (;;) {
  onmers = 0;
  filt0 = 0xFFFFFFFFu, filt1 = 0xFFFFFFFFu;
  ws.top = 0;
  ws.l2 = false;
  ws.nev = 0;
  ws.ev_full = false;
  for (uint64_t base0 = 0; base0 < nkm; base0 += kSegPos) {
    const uint32_t npos_seg = (uint32_t)min((uint64_t)kSegPos, nkm - base0);
    ws.base0 = (uint32_t)base0;
    SegBits sb;
    load_segment(seq, len, base0, sb);
    uint32_t nv = 0;
    Cand cur = fetch_group<SL>(ix, sb, 0, npos_seg, nv);
    onmers += nv;
    const int ngroups = npos_seg > 64 ? 2 : 1;
    for (int pp = 0; pp < ngroups; ++pp) {
      // descriptors of the NEXT group are requested before this group is scanned
      Cand nxt = cur;
      if (pp + 1 < ngroups) {
        nxt = fetch_group<SL>(ix, sb, pp + 1, npos_seg, nv);
        onmers += nv;
      }
      // ---- compact the non-empty probes of this group into the LDS list
      const bool a0 = (cur.b0 & 0xFFFFFFu) != 0, a1 = (cur.b1 & 0xFFFFFFu) != 0;
      const uint64_t m0 = __ballot(a0), m1 = __ballot(a1);
      const uint32_t n0 = __popcll(m0), nact = n0 + __popcll(m1);
      if (a0) {
        uint32_t i = __popcll(m0 & lt);
        pl.bkt[i] = cur.b0;
        pl.q[i] = cur.q0;
        pl.tag[i] = (64u * pp + lane) | (cur.lib0 << 8);
      }
      if (a1) {
        uint32_t i = n0 + __popcll(m1 & lt);
        pl.bkt[i] = cur.b1;
        pl.q[i] = cur.q1;
        pl.tag[i] = (64u * pp + lane) | (1u << 7) | (cur.lib1 << 8);
      }
      WAVE_SYNC();
      if (!(P.dbg & 4u)) scan_list<LOG_G, SL, TAP>(ix, P, out, A, ws, pl, nact, filt0, filt1);
      WAVE_SYNC();
      cur = nxt;
    }
    if (P.dbg & 2u) ws.top = 0;
    expand_all<SL, TAP>(ix, out, A, ws);
    if (ws.evmode) break; // single segment, nothing to fold: finalize_events does the rest
    // ---- fold this segment's planes into running counts (positions of different
    //      segments are distinct, so histograms add)
    WAVE_SYNC();
    if (A.keys[lane]) fold_l1(A, lane); // kLdsSlots == 64: lane t owns slot t
    l2_any = __ballot(ws.l2) != 0;
    if (l2_any) {
      uint32_t n2 = l2_build_list(A);
      for (uint32_t t = lane; t < n2; t += 64) fold_l2(A, A.g_list[t]);
    }
    __syncthreads();
  }

  // ---- per-strand hdist_filt = min hd over kept table entries (src/query.cpp:366-368)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    filt0 = min(filt0, (uint32_t)__shfl_xor(filt0, d));
    filt1 = min(filt1, (uint32_t)__shfl_xor(filt1, d));
  }
  if (ws.err && lane == 0) atomicOr(&out.counters[1], ws.err);
  ws.err = 0;
  if (!ws.evmode) break;
  if (!ws.ev_full && finalize_events(ix, out, A, ws, hist_tbl, hist_words, read, onmers, filt0, filt1)) return;
  // does not fit: zero the aliased region and redo the read with the plane tables
  WAVE_SYNC();
  for (uint32_t i = lane; i < (uint32_t)kLdsSlots * A.np * (kPlaneWords + 1); i += 64) A.planes[i] = 0;
  WAVE_SYNC();
  ws.evmode = false;
  } // redo loop
  // records that pass `hdist_min <= 2*hdist_filt+1` (src/query.cpp:101-106,119), ordered by key so
  // that the two strands of a leaf are adjacent
  const uint32_t lim0 = 2u * filt0 + 1u, lim1 = 2u * filt1 + 1u; // u32 wrap keeps "none" = max

  if (!l2_any) {
    // ---- level 1 only: lane t owns slot t
    const uint32_t key = A.keys[lane]; // (rank + 1) << 1 | strand: ascending key == ascending colour id
    const bool ok = key && hmin_l1(A, lane) <= ((key & 1u) ? lim1 : lim0);
    uint64_t okm = __ballot(ok);
    const uint32_t nrec = __popcll(okm);
    const uint32_t rbase = nrec ? alloc_records(out, ws, nrec) : 0u;
    if (lane == 0) {
      out.rd_off[read] = rbase == 0xFFFFFFFFu ? 0 : rbase;
      out.rd_cnt[read] = rbase == 0xFFFFFFFFu ? 0 : nrec;
      out.rd_onmers[read] = onmers;
      out.rd_filt[2 * read] = filt0;
      out.rd_filt[2 * read + 1] = filt1;
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
      for (uint32_t x = 0; x < A.np; ++x) out.rec_hist[(uint64_t)x * out.rec_cap + ri] = A.counts[lane * A.np + x];
    }
    if (key) { // leave the slot empty for the next read
      A.keys[lane] = 0;
      for (uint32_t x = 0; x < A.np; ++x) A.counts[lane * A.np + x] = 0;
    }
    WAVE_SYNC();
    return;
  }

  // ---- level 2 in use: move the level-1 entries over, then emit from the sorted slot list
  {
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
  const uint32_t n2 = l2_build_list(A);
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
  ws.n_l2++;
  const uint32_t rbase = nrec ? alloc_records(out, ws, nrec) : 0u;
  if (lane == 0) {
    out.rd_off[read] = rbase == 0xFFFFFFFFu ? 0 : rbase;
    out.rd_cnt[read] = rbase == 0xFFFFFFFFu ? 0 : nrec;
    out.rd_onmers[read] = onmers;
    out.rd_filt[2 * read] = filt0;
    out.rd_filt[2 * read + 1] = filt1;
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
      for (uint32_t x = 0; x < A.np; ++x)
        out.rec_hist[(uint64_t)x * out.rec_cap + ri] = gload(&A.g_counts[(uint64_t)slot2 * A.np + x]);
    }
    if (t < n2)
      for (uint32_t x = 0; x < A.np; ++x) gstore(&A.g_counts[(uint64_t)slot2 * A.np + x], 0);
    run += __popcll(okm);
  }
  for (uint32_t w = lane; w < A.bm_words; w += 64) A.bitmap[w] = 0;
  __syncthreads();
}

template <int LOG_G, bool SL, bool TAP>
__global__ __launch_bounds__(kWave, 4) void kr_probe_kernel_t(DevIndex ix, DevParams P, BatchIn in, BatchOut out)
{
  // One carve of dynamic LDS (base is 16-byte aligned: no static __shared__ in front):
  //   stack | probe list | level-1 table (keys, planes, counts) | level-2 bitmap
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  KR_LDS uint8_t* s_base = (KR_LDS uint8_t*)s_dyn;
  lds_u64* s_stack = (lds_u64*)s_base;
  lds_u64* s_bkt = (lds_u64*)(s_base + kStackCap * 8);
  lds_u32* s_q = (lds_u32*)(s_base + kStackCap * 8 + kListCap * 8);
  lds_u32* s_tag = s_q + kListCap;
  lds_u32* s_tbl = s_tag + kListCap;

  Acc A;
  A.np = P.np;
  A.keys = s_tbl;
  A.planes = A.keys + kLdsSlots;
  A.counts = A.planes + kLdsSlots * P.np * kPlaneWords;
  A.bitmap = A.counts + kLdsSlots * P.np;
  A.nslots2 = out.nslots2;
  A.bm_words = out.bm_words;
  const uint64_t w = blockIdx.x;
  A.g_planes = out.g_planes + w * (uint64_t)out.nslots2 * P.np * kPlaneWords;
  A.g_counts = out.g_counts + w * (uint64_t)out.nslots2 * P.np;
  A.g_list = out.g_list + w * (uint64_t)out.g_list_words;
  { // tables start empty; every read leaves them empty again
    const uint32_t lane = lane_id();
    A.keys[lane] = 0;
    for (uint32_t x = 0; x < P.np; ++x) {
      A.counts[lane * P.np + x] = 0;
#pragma unroll
      for (int q = 0; q < kPlaneWords; ++q) A.planes[(lane * P.np + x) * kPlaneWords + q] = 0;
    }
    for (uint32_t q = lane; q < A.bm_words; q += 64) A.bitmap[q] = 0;
  }
  __syncthreads();
  WaveState ws;
  ws.stack = s_stack;
  ws.top = 0;
  ws.l2 = false;
  ws.err = 0;
  ws.read = 0;
  ws.base0 = 0;
  ws.rec_next = 0;
  ws.rec_end = 0;
  ws.n_l2 = 0;
  ws.n_rec = 0;
  ws.evmode = false;
  ws.nev = 0;
  ws.ev = A.planes; // planes + counts are contiguous: kLdsSlots * np * 5 words
  ws.ev_cap = 64;
  while (ws.ev_cap * 2 <= (uint32_t)kLdsSlots * P.np * (kPlaneWords + 1)) ws.ev_cap <<= 1;
  ProbeList pl{s_bkt, s_q, s_tag};
  for (uint32_t r = blockIdx.x; r < in.nreads; r += gridDim.x) process_read<LOG_G, SL, TAP>(ix, P, in, out, r, A, ws, pl, (lds_u32*)s_base, (kStackCap * 8 + kListCap * 16) / 4);
  if (ws.n_l2 && lane_id() == 0) atomicAdd(&out.counters[2], ws.n_l2);
  if (ws.n_rec && lane_id() == 0) atomicAdd(&out.counters[4], ws.n_rec);
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
template <int NPT>
__device__ __forceinline__ double llh_eval(const LlhConst& C, const LlhTables& T, const LlhProblem& p, double d)
{
  double sum = 0.0, lv_m = 0.0;
  double powdc = (C.dbg & 1u) ? pow(1.0 - d, (double)C.k) : pown_dd(1.0 - d, C.k);
  double logdn = log(1.0 - d);
  double logdp = log(d) - logdn;
  logdn *= (double)C.k;
  const double dratio = d / (1.0 - d);
  if (NPT > 0) {
#pragma unroll
    for (int x = 0; x < NPT; ++x) {
      sum -= (logdn + (double)x * logdp) * p.mc[x];
      lv_m += T.hnk[x] * powdc;
      powdc *= dratio;
    }
  } else {
    for (uint32_t x = 0; x <= C.th; ++x) {
      sum -= (logdn + (double)x * logdp) * p.mc[x];
      lv_m += T.hnk[x] * powdc;
      powdc *= dratio;
    }
  }
#pragma unroll 4
  for (uint32_t x = C.th + 1; x <= C.k; ++x) {
    lv_m += powdc * T.bk[x];
    powdc *= dratio;
  }
  return sum - log(p.rho * lv_m + 1.0 - p.rho) * p.uc;
}

// boost::math::tools::brent_find_minima(f, 1e-10, 0.5, 16) (src/query.cpp:430);
// published algorithm, see SURVEY.md Appendix B.
template <int NPT>
__device__ __forceinline__ void brent_min(const LlhConst& C, const LlhTables& T, const LlhProblem& p, double& d_out, double& v_out)
{
  double mn = 1e-10, mx = 0.5;
  const double tolerance = 0x1p-15; // ldexp(1, 1 - min(53/2, 16))
  const double golden = 0.3819660f;
  double x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
  x = w = v = mx;
  fw = fv = fx = llh_eval<NPT>(C, T, p, x);
  delta2 = delta = 0;
  for (int it = 0; it < 1000; ++it) {
    mid = (mn + mx) / 2;
    fract1 = tolerance * fabs(x) + tolerance / 4;
    fract2 = 2 * fract1;
    if (fabs(x - mid) <= (fract2 - (mx - mn) / 2)) break;
    if (fabs(delta2) > fract1) {
      double r = (x - w) * (fx - fv);
      double q = (x - v) * (fx - fw);
      double pp = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) pp = -pp;
      q = fabs(q);
      double td = delta2;
      delta2 = delta;
      if ((fabs(pp) >= fabs(q * td / 2)) || (pp <= q * (mn - x)) || (pp >= q * (mx - x))) {
        delta2 = (x >= mid) ? mn - x : mx - x;
        delta = golden * delta2;
      } else {
        delta = pp / q;
        u = x + delta;
        if (((u - mn) < fract2) || ((mx - u) < fract2)) delta = (mid - x) < 0 ? -fabs(fract1) : fabs(fract1);
      }
    } else {
      delta2 = (x >= mid) ? mn - x : mx - x;
      delta = golden * delta2;
    }
    u = (fabs(delta) >= fract1) ? (x + delta) : (delta > 0 ? (x + fabs(fract1)) : (x - fabs(fract1)));
    fu = llh_eval<NPT>(C, T, p, u);
    if (fu <= fx) {
      if (u >= x)
        mn = x;
      else
        mx = x;
      v = w, w = x, x = u;
      fv = fw, fw = fx, fx = fu;
    } else {
      if (u < x)
        mn = u;
      else
        mx = u;
      if ((fu <= fw) || (w == x)) {
        v = w, w = u;
        fv = fw, fw = fu;
      } else if ((fu <= fv) || (v == x) || (v == w)) {
        v = u;
        fv = fu;
      }
    }
  }
  d_out = x;
  v_out = fx;
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

template <int NPT>
__device__ __forceinline__ void llh_records(const LlhConst& C, const LlhTables& T, const DevIndex& ix, const BatchOut& out)
{
  uint32_t nrec = min(out.counters[0], out.rec_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += gridDim.x * blockDim.x) {
    uint32_t key = out.rec_key[i];
    if (key == 0) continue; // hole at the end of a wave's record chunk
    uint32_t read = out.rec_read[i];
    LlhProblem p;
    load_problem<NPT>(C, out.rec_hist + i, out.rec_cap, out.rd_onmers[read], ix.libs[0].rho[key >> 1], p);
    double d, v;
    brent_min<NPT>(C, T, p, d, v);
    out.rec_d[i] = d;
    out.rec_v[i] = v;
  }
}

// NPT = 5: --hdist-th default, histogram in registers; NPT = 0: any threshold
template <int NPT>
__global__ __launch_bounds__(256) void kr_llh_kernel(LlhConst C, DevIndex ix, BatchOut out)
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
__global__ __launch_bounds__(256) void kr_select_kernel(LlhConst C, DevIndex ix, DevParams P, BatchOut out,
                                                        uint32_t nreads)
{
  __shared__ double s_bk[32], s_hnk[kMaxPlanes];
  LlhTables T{(lds_f64*)s_bk, (lds_f64*)s_hnk};
  llh_tables_init(C, T.bk, T.hnk);
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += gridDim.x * blockDim.x) {
    uint32_t o = out.rd_off[r], n = out.rd_cnt[r];
    // closest: last record in (strand, se) order with d <= best
    double best = 1.7976931348623157e308;
    int cl = -1;
    for (int strand = 0; strand < 2; ++strand)
      for (uint32_t i = o; i < o + n; ++i)
        if ((out.rec_key[i] & 1u) == (uint32_t)strand && out.rec_d[i] <= best) {
          best = out.rec_d[i];
          cl = (int)i;
        }
    bool na = (n == 0) || (P.dmax_set && best > P.dist_max);
    out.rd_na[r] = na ? 1 : 0;
    LlhProblem pc;
    double vcl = 0;
    if (cl >= 0 && !P.no_filter) {
      load_problem<0>(C, out.rec_hist + cl, out.rec_cap, out.rd_onmers[r], ix.libs[0].rho[out.rec_key[cl] >> 1], pc);
      vcl = out.rec_v[cl];
    }
    for (uint32_t i = o; i < o + n; ++i) {
      uint32_t key = out.rec_key[i];
      // which record represents this leaf in node_to_minfo?
      bool chosen;
      bool has_other = (key & 1u) ? (i > o && out.rec_key[i - 1] == (key ^ 1u)) : (i + 1 < o + n && out.rec_key[i + 1] == (key ^ 1u));
      if (!has_other) {
        chosen = true;
      } else {
        uint32_t io = (key & 1u) ? i - 1 : i, ir = (key & 1u) ? i : i + 1;
        double d_or = out.rec_d[io], d_rc = out.rec_d[ir];
        uint32_t m_or = 0, m_rc = 0;
        for (uint32_t x = 0; x <= C.th; ++x) {
          m_or += out.rec_hist[(uint64_t)x * out.rec_cap + io];
          m_rc += out.rec_hist[(uint64_t)x * out.rec_cap + ir];
        }
        bool take_or = (d_rc > d_or) || ((d_rc == d_or) && (m_rc < m_or)); // src/query.cpp:129-133
        // the closest overrides (src/query.cpp:136-138)
        if (cl >= 0 && (out.rec_key[cl] >> 1) == (key >> 1)) take_or = ((uint32_t)cl == io);
        chosen = (key & 1u) ? !take_or : take_or;
      }
      double d = out.rec_d[i];
      double chi = nan("");
      bool sel = false;
      if (chosen && !na) {
        bool dm = !P.dmax_set || d < P.dist_max;
        if (!P.multi) {
          sel = (int)i == cl;
        } else if (P.no_filter) {
          sel = dm;
        } else {
          chi = 2 * (llh_eval<0>(C, T, pc, d) - vcl); // Minfo::likelihood_ratio (src/query.cpp:420-424)
          sel = (chi < P.chisq) && dm;
        }
      }
      out.rec_sel[i] = sel ? 1 : 0;
      out.rec_chisq[i] = chi;
    }
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

__global__ __launch_bounds__(256) void kr_llh_batch_kernel(LlhConst C, uint32_t mode, uint64_t n, const double* hist,
                                                           const double* uc, const double* rho, const double* d_in,
                                                           double* d_out, double* v_out)
{
  __shared__ double s_bk[32], s_hnk[kMaxPlanes];
  LlhTables T{(lds_f64*)s_bk, (lds_f64*)s_hnk};
  llh_tables_init(C, T.bk, T.hnk);
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    LlhProblem p;
    for (uint32_t x = 0; x <= C.th; ++x) p.mc[x] = hist[i * (C.th + 1) + x];
    p.uc = uc[i];
    p.rho = rho[i];
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
uint32_t probe_lds_bytes(uint32_t np, uint32_t bm_words)
{
  uint32_t b = kStackCap * 8 + kListCap * 8 + 2 * kListCap * 4;
  b += kLdsSlots * 4 + kLdsSlots * np * kPlaneWords * 4 + kLdsSlots * np * 4 + bm_words * 4;
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
  DevIndex dix;
  std::vector<DevLib> hlibs;    // host copy of the device DevLib array
  std::vector<void*> allocs;    // everything to hipFree
  std::vector<kr_index_buffer> bufs; // export order
  std::vector<uint8_t> desc;    // export descriptor
  uint64_t bytes = 0;
};

namespace {

struct DescHeader {
  uint32_t magic, k, h, m, nlibs, tree_nnodes, nleaves, log_g;
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
  if (2ull * H.nleaves > 65536) return kr::fail(KR_ERR_ARG, "kr_index_upload: more than 32768 reference leaves is not supported");
  { // lanes per probe: a bucket of L entries spans about (L + 4.5) / 4 aligned 16-byte chunks
    double nk = 0, nr = 0;
    for (uint32_t i = 0; i < v->nlibs; ++i) nk += (double)v->libs[i].nkmers, nr += (double)v->libs[i].nrows;
    double mean_len = nr > 0 ? nk / nr : 0; // empty buckets never reach the scan, so this underestimates slightly
    H.log_g = mean_len <= 3.0 ? 0u : (mean_len <= 20.0 ? 2u : 3u);
    if (const char* e = getenv("KR_LOG_G")) H.log_g = (uint32_t)atoi(e) > 3 ? 3u : (uint32_t)atoi(e);
    if (H.log_g == 1) H.log_g = 2;
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
  uint32_t nwaves = 0;
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
  if (e & kErrStack) return kr::fail(KR_ERR_CAPACITY, "colour work stack overflow (colour DAG deeper than supported)");
  if (e & kErrTable) return kr::fail(KR_ERR_CAPACITY, "global accumulator table overflow");
  if (e & kErrHitCap) return kr::fail(KR_ERR_CAPACITY, "hit tap buffer overflow");
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
  const uint32_t nslots2 = std::max<uint32_t>(2u, 2u * ix->dix.nleaves), bm_words = (nslots2 + 31) / 32;
  uint32_t per_cu = std::min<uint32_t>(16u, 163840u / probe_lds_bytes(p->hdist_th + 1, bm_words));
  s->nwaves = (uint32_t)prop.multiProcessorCount * std::max<uint32_t>(1u, per_cu);
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
  SA(o.counters, 8);
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
  o.rec_cap = s->rec_cap;
  o.hit_cap = s->hit_cap;
  o.nslots2 = nslots2;
  const uint32_t g_list_words = std::max<uint32_t>(nslots2, (uint32_t)kLdsSlots * np * (kPlaneWords + 1));
  o.g_list_words = g_list_words;
  o.bm_words = bm_words;
  SA(o.g_planes, (uint64_t)s->nwaves * nslots2 * np * kPlaneWords);
  SA(o.g_counts, (uint64_t)s->nwaves * nslots2 * np);
  SA(o.g_list, (uint64_t)s->nwaves * g_list_words);
  HIP_TRY(hipMemset(o.g_planes, 0, (uint64_t)s->nwaves * nslots2 * np * kPlaneWords * 4));
  HIP_TRY(hipMemset(o.g_counts, 0, (uint64_t)s->nwaves * nslots2 * np * 4));
  HA(s->h_bases, max_bases + 256);
  HA(s->h_offsets, (uint64_t)max_reads + 1);
  HA(s->h_counters, 8);
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
  HIP_TRY(hipMemsetAsync(s->out.counters, 0, 32, st));
  HIP_TRY(hipMemsetAsync(s->out.rec_key, 0, (uint64_t)s->rec_cap * 4, st));
  HIP_TRY(hipMemsetAsync(s->out.rec_sel, 0, (uint64_t)s->rec_cap, st));
  HIP_TRY(hipEventRecord(s->ev[1], st));
  const DevIndex& dix = s->ix->dix;
  uint32_t grid = std::min(nreads, s->nwaves);
  if (flags & KR_TAP_HITS) {
    if (!s->h_hits) {
      HIP_TRY(hipMalloc((void**)&s->out.hits, (uint64_t)s->hit_cap * sizeof(kr_hit)));
      s->dallocs.push_back(s->out.hits);
      HIP_TRY(hipHostMalloc((void**)&s->h_hits, (uint64_t)s->hit_cap * sizeof(kr_hit), hipHostMallocDefault));
      s->hallocs.push_back(s->h_hits);
    }
  }
  {
    const uint32_t lds = probe_lds_bytes(s->dp.np, s->out.bm_words);
    const bool tap = (flags & KR_TAP_HITS) != 0;
#define KR_LAUNCH2(LG, SLV)                                                                                             \
  do {                                                                                                                 \
    if (tap)                                                                                                           \
      hipLaunchKernelGGL((kr_probe_kernel_t<LG, SLV, true>), dim3(grid), dim3(kWave), lds, st, dix, s->dp, s->in, s->out); \
    else                                                                                                               \
      hipLaunchKernelGGL((kr_probe_kernel_t<LG, SLV, false>), dim3(grid), dim3(kWave), lds, st, dix, s->dp, s->in, s->out); \
  } while (0)
#define KR_LAUNCH(LG)          \
  do {                         \
    if (single)                \
      KR_LAUNCH2(LG, true);    \
    else                       \
      KR_LAUNCH2(LG, false);   \
  } while (0)
    const bool single = dix.nlibs == 1 && dix.m <= 64;
    switch (s->ix->log_g) {
      case 0: KR_LAUNCH(0); break;
      case 2: KR_LAUNCH(2); break;
      default: KR_LAUNCH(3); break;
    }
#undef KR_LAUNCH2
#undef KR_LAUNCH
    HIP_TRY(hipEventRecord(s->ev[2], st));
  }
  HIP_TRY(hipEventRecord(s->ev[3], st));
  if (s->llh.th == 4)
    hipLaunchKernelGGL(kr_llh_kernel<5>, dim3(2048), dim3(256), 0, st, s->llh, dix, s->out);
  else
    hipLaunchKernelGGL(kr_llh_kernel<0>, dim3(2048), dim3(256), 0, st, s->llh, dix, s->out);
  hipLaunchKernelGGL(kr_select_kernel, dim3((nreads + 255) / 256), dim3(256), 0, st, s->llh, dix, s->dp, s->out, nreads);
  HIP_TRY(hipEventRecord(s->ev[4], st));
  HIP_TRY(hipGetLastError());
  return KR_OK;
}

int kr_batch_wait(kr_stream* s)
{
  if (!s || !s->submitted) return kr::fail(KR_ERR_STATE, "kr_batch_wait: nothing submitted");
  if (s->waited) return KR_OK;
  HIP_TRY(hipSetDevice(s->ix->device));
  HIP_TRY(hipMemcpyAsync(s->h_counters, s->out.counters, 32, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->waited = true;
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
  HIP_TRY(hipEventElapsedTime(&t->ms_probe, s->ev[1], s->ev[2]));
  HIP_TRY(hipEventElapsedTime(&t->ms_overflow, s->ev[2], s->ev[3]));
  HIP_TRY(hipEventElapsedTime(&t->ms_llh, s->ev[3], s->ev[4]));
  HIP_TRY(hipEventElapsedTime(&t->ms_total, s->ev[1], s->ev[4]));
  t->overflow_reads = s->h_counters[2];
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
  double *d_h = nullptr, *d_uc = nullptr, *d_rho = nullptr, *d_di = nullptr, *d_do = nullptr, *d_v = nullptr;
  HIP_TRY(hipMalloc((void**)&d_h, n * (th + 1) * 8));
  HIP_TRY(hipMalloc((void**)&d_uc, n * 8));
  HIP_TRY(hipMalloc((void**)&d_rho, n * 8));
  HIP_TRY(hipMalloc((void**)&d_di, n * 8));
  HIP_TRY(hipMalloc((void**)&d_do, n * 8));
  HIP_TRY(hipMalloc((void**)&d_v, n * 8));
  HIP_TRY(hipMemcpy(d_h, hist, n * (th + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_uc, uc, n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_rho, rho, n * 8, hipMemcpyHostToDevice));
  if (mode == 1) HIP_TRY(hipMemcpy(d_di, d_in, n * 8, hipMemcpyHostToDevice));
  uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(kr_llh_batch_kernel, dim3(grid), dim3(256), 0, 0, C, mode, n, d_h, d_uc, d_rho, d_di, d_do, d_v);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  if (mode == 0) HIP_TRY(hipMemcpy(d_out, d_do, n * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(v_out, d_v, n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d_h), (void)hipFree(d_uc), (void)hipFree(d_rho), (void)hipFree(d_di), (void)hipFree(d_do), (void)hipFree(d_v);
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
