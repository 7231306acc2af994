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
//
// One translation unit in several files, included below in this order (device code first):
//   kr_dev_common.inc      constants, device structs and helpers, accumulator tables, colour expansion
//   kr_dev_scan.inc        kernel 1: probe list, bucket scan, hit items
//   kr_dev_accumulate.inc  kernel 2: event epilogue, plane tables, records
//   kr_dev_likelihood.inc  likelihood, Brent, de-duplication, selection kernels
//   kr_dev_place.inc       back end of `place`: ancestor accumulation, candidates, their likelihoods (kr_place_kernel)
//   kr_dev_debug.inc       debug / tap kernels and the re-layout kernels of kr_index_upload
// and, in this file, the host side: kr_index_upload / export / import, kr_stream_*, kr_batch_*, kr_llh_batch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <type_traits>

#include "kr_common.h"
#include "kr_devutil.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace {

#include "kr_dev_common.inc"
#include "kr_dev_scan.inc"
#include "kr_dev_accumulate.inc"
#include "kr_dev_likelihood.inc"
#include "kr_dev_place.inc"
#include "kr_dev_debug.inc"

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
    return (lean_stack_bytes(bm_words) + kLdsSlots * 4 + lean_ev_words(np) * 4 + 15u) & ~15u;
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
  // kernel chain: the kernels of every batch (every lane of every kr_stream) on this index run one batch after the
  // other, in submission order -- each batch's first kernel waits for the event recorded behind the previous batch's
  // last -- while the H2D copy in front of them and the D2H copy behind them overlap other batches' kernels.
  // Concurrent batches only slow each other down (measured), and a pipeline whose stages take turns
  // keeps the chip busy with two streams.
  mutable std::mutex chain_mu;
  mutable hipEvent_t chain_ev[16] = {nullptr};
  mutable uint32_t chain_n = 0;
  mutable bool chain_off = false;
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

// The HIP runtime folds every stream after the third onto its last hardware queue (GPU_MAX_HW_QUEUES, default 4), where
// the copies of one kr_stream would queue behind the kernels of another as blit kernels instead of running beside them
// on an SDMA engine.  Raised here, when the library is loaded, unless the host application has set it; it only takes
// effect if the runtime has not initialised yet.
__attribute__((constructor)) void kr_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

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
  ix->chain_off = getenv("KR_NO_KERNEL_CHAIN") != nullptr;
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
  for (auto& e : ix->chain_ev)
    if (e) (void)hipEventDestroy(e);
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
  ix->chain_off = getenv("KR_NO_KERNEL_CHAIN") != nullptr;
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
// kr_index_broadcast: replicas over RCCL (xGMI), load time only
// ---------------------------------------------------------------------------
namespace {

// RCCL is loaded on first use: a process that never replicates an index (every single-GPU run, every rank of a
// one-process-per-GPU job that replicates through its own transport) neither maps nor initialises it.
struct Rccl {
  void* h = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
  bool load()
  {
    if (h) return true;
    const char* names[] = {getenv("KR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
      err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "librccl.so.1 not found");
      return false;
    }
#define KR_SYM(field, name)                                                            \
  if (!(field = reinterpret_cast<decltype(field)>(dlsym(h, name)))) {                  \
    err = std::string("RCCL lacks ") + name;                                           \
    dlclose(h), h = nullptr;                                                           \
    return false;                                                                      \
  }
    KR_SYM(CommInitAll, "ncclCommInitAll")
    KR_SYM(CommDestroy, "ncclCommDestroy")
    KR_SYM(Broadcast, "ncclBroadcast")
    KR_SYM(GroupStart, "ncclGroupStart")
    KR_SYM(GroupEnd, "ncclGroupEnd")
    KR_SYM(GetErrorString, "ncclGetErrorString")
#undef KR_SYM
    return true;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

} // namespace

extern "C" int kr_index_broadcast(const kr_index* root, int ndev, const int* devices, kr_index** replicas)
{
  kr::clear_error();
  if (!root || ndev < 0 || (ndev && (!devices || !replicas))) return kr::fail(KR_ERR_ARG, "kr_index_broadcast: bad argument");
  if (ndev == 0) return KR_OK;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return kr::fail(KR_ERR_NO_DEVICE, "no HIP device available");
  for (int i = 0; i < ndev; ++i) {
    replicas[i] = nullptr;
    if (devices[i] < 0 || devices[i] >= have) return kr::fail(KR_ERR_ARG, "kr_index_broadcast: bad device ordinal");
    for (int j = 0; j < i; ++j)
      if (devices[j] == devices[i]) return kr::fail(KR_ERR_ARG, "kr_index_broadcast: a device is listed twice");
  }
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.load()) return kr::fail(KR_ERR_NO_DEVICE, g_rccl.err);
  Rccl& R = g_rccl;
  auto cleanup = [&](int rc) {
    for (int i = 0; i < ndev; ++i)
      if (replicas[i]) kr_index_free(replicas[i]), replicas[i] = nullptr;
    return rc;
  };
  // same buffers, same order, on every target device (kr_index_import's allocation)
  std::vector<std::vector<kr_index_buffer>> rb(ndev);
  for (int i = 0; i < ndev; ++i) {
    uint32_t nb = 0;
    rb[i].resize(root->bufs.size());
    nb = (uint32_t)rb[i].size();
    int rc = kr_index_import(root->desc.data(), root->desc.size(), devices[i], &replicas[i], rb[i].data(), &nb);
    if (rc) return cleanup(rc);
    if (nb != root->bufs.size()) return cleanup(kr::fail(KR_ERR_FORMAT, "kr_index_broadcast: replica layout differs from the root's"));
  }
  // one communicator over {root device} + the target devices that are not the root's; a target ON the root's device
  // (a second replica in the same HBM: tests, A/B runs) is served by a one-rank communicator, whose out-of-place
  // broadcast is a device copy
  std::vector<int> devlist{root->device};
  std::vector<int> rank_of(ndev, -1);
  for (int i = 0; i < ndev; ++i)
    if (devices[i] != root->device) rank_of[i] = (int)devlist.size(), devlist.push_back(devices[i]);
  const int nranks = (int)devlist.size();
  std::vector<ncclComm_t> comm(nranks, nullptr);
  std::vector<hipStream_t> strm(nranks, nullptr);
  auto nccl_fail = [&](ncclResult_t r, const char* what) {
    return kr::fail(KR_ERR_NO_DEVICE, std::string(what) + ": " + R.GetErrorString(r));
  };
  int rc = KR_OK;
  ncclResult_t nr = R.CommInitAll(comm.data(), nranks, devlist.data());
  if (nr != ncclSuccess) return cleanup(nccl_fail(nr, "ncclCommInitAll"));
  for (int r = 0; r < nranks && rc == KR_OK; ++r) {
    if (hipSetDevice(devlist[r]) != hipSuccess || hipStreamCreateWithFlags(&strm[r], hipStreamNonBlocking) != hipSuccess)
      rc = kr::fail(KR_ERR_NO_DEVICE, "kr_index_broadcast: cannot create a stream");
  }
  // few, large buffers: one ring broadcast each (per-link bound over xGMI), all ranks of one buffer in one group
  for (size_t b = 0; b < root->bufs.size() && rc == KR_OK; ++b) {
    const uint64_t bytes = root->bufs[b].bytes;
    if (!bytes) continue;
    if ((nr = R.GroupStart()) != ncclSuccess) { rc = nccl_fail(nr, "ncclGroupStart"); break; }
    // rank 0 = the root: in place unless a same-device replica wants the bytes too
    void* root_recv = root->bufs[b].dptr;
    for (int i = 0; i < ndev; ++i)
      if (rank_of[i] < 0) root_recv = rb[i][b].dptr; // (at most one: devices are distinct)
    nr = R.Broadcast(root->bufs[b].dptr, root_recv, bytes, ncclUint8, 0, comm[0], strm[0]);
    for (int i = 0; i < ndev && nr == ncclSuccess; ++i)
      if (rank_of[i] > 0) nr = R.Broadcast(rb[i][b].dptr, rb[i][b].dptr, bytes, ncclUint8, 0, comm[rank_of[i]], strm[rank_of[i]]);
    ncclResult_t ge = R.GroupEnd();
    if (nr != ncclSuccess || ge != ncclSuccess) rc = nccl_fail(nr != ncclSuccess ? nr : ge, "ncclBroadcast");
  }
  for (int r = 0; r < nranks; ++r) {
    if (!strm[r]) continue;
    (void)hipSetDevice(devlist[r]);
    if (hipStreamSynchronize(strm[r]) != hipSuccess && rc == KR_OK) rc = kr::fail(KR_ERR_NO_DEVICE, "kr_index_broadcast: stream failed");
    (void)hipStreamDestroy(strm[r]);
  }
  for (auto c : comm)
    if (c) (void)R.CommDestroy(c);
  (void)hipSetDevice(root->device);
  return rc == KR_OK ? KR_OK : cleanup(rc);
}

// ---------------------------------------------------------------------------
// kr_stream
// ---------------------------------------------------------------------------
// One batch is cut into up to kMaxLanes contiguous read ranges ("lanes").  A lane is a complete pipeline of its own --
// HIP stream, H2D copy of its reads, the kernels, its counters -- working on its slice of the stream's result arrays,
// so that the copies of one lane and the kernels of another overlap, and the three kernel families (scan: HBM request
// rate; accumulate: latency; likelihood: fp64 ALU) of different lanes can share the chip.  Lanes change nothing in any
// result: reads are independent, and every lane runs the same kernels on its reads.
constexpr uint32_t kMaxLanes = 8;
struct Lane {
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  BatchIn in;
  BatchOut out;                // pointers: the lane's private scratch + its slices of the stream's arrays
  uint32_t* d_counters = nullptr;
  uint32_t* d_cursors = nullptr;
  ulonglong2* d_dd = nullptr;  // private de-duplication table
  uint32_t dd_slots = 0;
  uint32_t *d_g_planes = nullptr, *d_g_counts = nullptr, *d_g_list = nullptr;
  uint64_t* d_stk = nullptr;
  uint32_t* h_counters = nullptr; // pinned [32]
  // this batch
  uint32_t read0 = 0, nreads = 0, rec_base = 0, rec_cap = 0, nrecs = 0;
  uint64_t host_off = 0;       // where the lane's records start in the compacted host arrays
};

struct kr_stream {
  const kr_index* ix = nullptr;
  int device = 0; // copy of ix->device: destroying a stream must not read an index that may already be gone
  kr_params params;
  DevParams dp;
  LlhConst llh;
  uint32_t max_reads = 0;
  uint64_t max_bases = 0;
  uint32_t rec_cap = 0, hit_cap = 0, item_cap = 0;
  uint64_t rec_user_cap = 0; // the caller's max_records (rec_cap adds per-wave chunk slack per lane)
  uint32_t nwaves = 0, nwaves_full = 0, nwaves_lean = 0; // per-wave scratch slots; grids of the two accumulate launches
  uint32_t max_lanes = 1, nlanes = 1;  // lanes created / lanes of the current batch
  uint32_t lane_min_reads = 1u << 16;
  bool lanes_for_device_input = false;
  Lane lanes[kMaxLanes];
  // device: input staging and the result arrays every lane writes its slice of
  uint8_t* d_bases = nullptr;
  uint64_t* d_offsets = nullptr; // [max_reads + max_lanes]: every lane has its own nreads + 1 entries
  BatchOut out;                  // the stream-wide arrays (and the constants every lane copies)
  std::vector<void*> dallocs;
  // pinned host
  uint8_t* h_bases = nullptr;
  uint64_t* h_offsets = nullptr;
  uint32_t h_counters[32] = {0}; // aggregate of the lanes' counters
  uint32_t *h_rd_off = nullptr, *h_rd_cnt = nullptr, *h_rd_onmers = nullptr, *h_rd_filt = nullptr;
  uint8_t* h_rd_na = nullptr;
  uint32_t *h_rec_key = nullptr, *h_rec_hist = nullptr;
  uint8_t* h_rec_sel = nullptr;
  double *h_rec_d = nullptr, *h_rec_v = nullptr, *h_rec_chisq = nullptr;
  kr_hit* h_hits = nullptr;
  std::vector<void*> hallocs;
  // `place` back end on the device (kr::place_on_device): the placement tree as device arrays (for one tree at a
  // time), per-read / per-candidate outputs and their page-locked mirrors; grown on demand
  struct PlaceWs {
    const void* tree_tag = nullptr;
    uint32_t pn = 0, nidx = 0;
    uint32_t *d_parent = nullptr, *d_eff = nullptr, *d_lo = nullptr, *d_idx_to_pt = nullptr;
    uint8_t* d_elig = nullptr;
    uint32_t *d_len = nullptr, *d_c0 = nullptr, *d_info = nullptr, *d_cse = nullptr, *d_cread = nullptr, *d_cnt = nullptr;
    double *d_cd = nullptr, *d_cv = nullptr, *d_cchi = nullptr, *d_cprob = nullptr, *d_rprob = nullptr;
    uint64_t cprob_cap = 0, rprob_cap = 0; // doubles
    uint32_t *h_len = nullptr, *h_c0 = nullptr, *h_info = nullptr, *h_cse = nullptr, *h_cnt = nullptr;
    double *h_cd = nullptr, *h_cv = nullptr, *h_cchi = nullptr;
    uint64_t reads_cap = 0, cand_cap = 0, h_cand_cap = 0;
  } pw;
  // state
  uint64_t h_rec_cap = 0; // pinned record buffers grow on demand in kr_batch_collect
  bool h_rec_full = false; // ... and hold v / chisq / hist only once a batch asked for them
  bool submitted = false, waited = false, collected = false;
  int batch_rc = 0; // result of the batch, returned by every wait / collect until the next submit (errors are sticky)
  std::string batch_msg;
  uint32_t nreads = 0, flags = 0, nrecs = 0;
  uint32_t scan_blocks = 0;
  uint64_t nhits = 0;
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

// Queue the kernels of one lane on its stream (everything the lane needs is in L.in / L.out).
int launch_lane(kr_stream* s, Lane& L, uint32_t flags)
{
  hipStream_t st = L.stream;
  const uint32_t nreads = L.nreads;
  BatchOut& o = L.out;
  { // about four chunks' worth of reads per wave, between 32 and kRecChunk slots (one shared counter serves ~90 M atomics/s:
    // a million-read batch must not take its slots 32 at a time)
    uint32_t per_wave = (uint32_t)std::min<uint64_t>(4ull * nreads / std::max<uint32_t>(1u, s->nwaves_lean), kRecChunk), c = 32;
    while (c < per_wave) c <<= 1;
    o.rec_chunk = std::min<uint32_t>(c, kRecChunk);
  }
  HIP_TRY(hipMemsetAsync(o.counters, 0, 128, st));
  HIP_TRY(hipMemsetAsync(o.cursors, 0, 3 * kCursors * kCursorStride * 4, st));
  HIP_TRY(hipMemsetAsync(o.rec_key, 0, (uint64_t)o.rec_cap * 4, st));
  HIP_TRY(hipMemsetAsync(o.rec_sel, 0, (uint64_t)o.rec_cap, st));
  // the kernel chain of the index (see kr_index): wait for the previous batch's kernels, record behind ours
  const kr_index* ixp = s->ix;
  std::unique_lock<std::mutex> chain(ixp->chain_mu);
  if (!ixp->chain_off && ixp->chain_n) HIP_TRY(hipStreamWaitEvent(st, ixp->chain_ev[(ixp->chain_n - 1) % 16], 0));
  HIP_TRY(hipEventRecord(L.ev[1], st));
  const DevIndex& dix = s->ix->dix;
  {
    const bool tap = (flags & KR_TAP_HITS) != 0;
    const uint32_t sgrid = std::min<uint32_t>((nreads + kScanWaves - 1) / kScanWaves, s->scan_blocks);
#define KR_LAUNCH2(LG, CP, SLV, SLT)                                                                                          \
  do {                                                                                                                   \
    if (tap)                                                                                                             \
      hipLaunchKernelGGL((kr_scan_kernel_t<LG, CP, SLV, true, SLT>), dim3(sgrid), dim3(kScanWaves* kWave), 0, st, dix, s->dp, L.in, o); \
    else                                                                                                                 \
      hipLaunchKernelGGL((kr_scan_kernel_t<LG, CP, SLV, false, SLT>), dim3(sgrid), dim3(kScanWaves* kWave), 0, st, dix, s->dp, L.in, o); \
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
    HIP_TRY(hipEventRecord(L.ev[2], st));
    const uint32_t lds = probe_lds_bytes(s->dp.np, o.bm_words), lds_lean = probe_lds_bytes(s->dp.np, o.bm_words, true);
    const uint32_t grid_lean = std::min(nreads, s->nwaves_lean), grid_full = std::min(nreads, s->nwaves_full);
    const bool np5 = s->dp.np == 5 && !getenv("KR_DEBUG_NP0");
#define KR_ACC(SLV, NPV)                                                                                                       \
  do {                                                                                                                       \
    hipLaunchKernelGGL((kr_acc_kernel_t<SLV, NPV, false>), dim3(grid_lean), dim3(kWave), lds_lean, st, dix, s->dp, L.in, o); \
    hipLaunchKernelGGL((kr_acc_kernel_t<SLV, NPV, true>), dim3(grid_full), dim3(kWave), lds, st, dix, s->dp, L.in, o);  \
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
  HIP_TRY(hipEventRecord(L.ev[3], st));
  hipLaunchKernelGGL(kr_dedup_clear_kernel, dim3(4096), dim3(256), 0, st, o);
  if (s->llh.th == 4) {
    hipLaunchKernelGGL(kr_dedup_kernel, dim3(4096), dim3(256), 0, st, o);
    hipLaunchKernelGGL(kr_llh_pre_kernel<5>, dim3(4096), dim3(256), 0, st, s->llh, dix, o);
    hipLaunchKernelGGL(kr_llh_kernel<5>, dim3(2048), dim3(256), 0, st, s->llh, dix, o);
  } else {
    hipLaunchKernelGGL(kr_dedup_kernel, dim3(4096), dim3(256), 0, st, o);
    hipLaunchKernelGGL(kr_llh_pre_kernel<0>, dim3(4096), dim3(256), 0, st, s->llh, dix, o);
    hipLaunchKernelGGL(kr_llh_kernel<0>, dim3(2048), dim3(256), 0, st, s->llh, dix, o);
  }
  hipLaunchKernelGGL(kr_llh_copy_kernel, dim3(4096), dim3(256), 0, st, o);
  {
    const uint32_t sgrid = std::min<uint32_t>((nreads + 7) / 8, 16384u);
    const bool filt = !s->dp.no_filter && s->dp.multi;
    if (s->llh.th == 4) {
      if (filt)
        hipLaunchKernelGGL((kr_select_kernel<5, true>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, o, nreads);
      else
        hipLaunchKernelGGL((kr_select_kernel<5, false>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, o, nreads);
    } else {
      if (filt)
        hipLaunchKernelGGL((kr_select_kernel<0, true>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, o, nreads);
      else
        hipLaunchKernelGGL((kr_select_kernel<0, false>), dim3(sgrid), dim3(256), 0, st, s->llh, dix, s->dp, o, nreads);
    }
  }
  if (L.rec_base) // the result view indexes the stream's arrays, the lane's kernels its slice
    hipLaunchKernelGGL(kr_rebase_kernel, dim3(std::min<uint32_t>((nreads + 255) / 256, 1024u)), dim3(256), 0, st, o.rd_off, o.rd_cnt, nreads, L.rec_base);
  HIP_TRY(hipEventRecord(L.ev[4], st));
  if (!ixp->chain_off) {
    hipEvent_t& ce = ixp->chain_ev[ixp->chain_n % 16];
    if (!ce) HIP_TRY(hipEventCreateWithFlags(&ce, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ce, st));
    ++ixp->chain_n;
  }
  chain.unlock();
  HIP_TRY(hipMemcpyAsync(L.h_counters, o.counters, 128, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipGetLastError());
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
  s->device = ix->device;
  s->params = *p;
  s->dp.th = p->hdist_th, s->dp.np = p->hdist_th + 1;
  s->dp.multi = p->multi, s->dp.no_filter = p->no_filter;
  s->dp.dmax_set = std::isnan(p->dist_max) ? 0 : 1;
  s->dp.dbg = getenv("KR_DEBUG_SKIP") ? (uint32_t)atoi(getenv("KR_DEBUG_SKIP")) : 0u;
  s->dp.chisq = p->chisq, s->dp.dist_max = p->dist_max;
  s->llh = make_llh_const(ix->dix.k, ix->dix.h, p->hdist_th);
  s->max_reads = max_reads, s->max_bases = max_bases;
  // lanes: host batches of at least 2 * lane_min_reads reads are cut into up to KR_LANES ranges (default 2: the copies of
  // one range overlap the kernels of the other; more ranges only add launches -- measured on the 10 GB index, reads
  // resident in HBM: 1 lane 13.3 ms per million reads, 2: 13.9, 4: 14.9, 8: 16.2, whatever share of the chip the
  // persistent scan / accumulate grids are given: the kernels of different lanes do not speed each other up)
  if (const char* e = getenv("KR_LANE_MIN_READS")) s->lane_min_reads = (uint32_t)std::max(1, atoi(e));
  s->lanes_for_device_input = getenv("KR_LANES_DEVICE") != nullptr; // experiments: lanes for batches already in HBM too
  {
    uint32_t want = getenv("KR_LANES") ? (uint32_t)std::max(1, atoi(getenv("KR_LANES"))) : 2u;
    s->max_lanes = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(want, kMaxLanes), max_reads / s->lane_min_reads));
  }
  const uint32_t ML = s->max_lanes;
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
  uint32_t per_cu_lean = std::min<uint32_t>(4u * KR_ACC_LEAN_WPE, 163840u / probe_lds_bytes(p->hdist_th + 1, bm_words, true));
  if (getenv("KR_DEBUG_ACC_WAVES")) {
    per_cu = std::min<uint32_t>(per_cu, (uint32_t)atoi(getenv("KR_DEBUG_ACC_WAVES")));
    per_cu_lean = std::min<uint32_t>(per_cu_lean, (uint32_t)atoi(getenv("KR_DEBUG_ACC_WAVES")));
  }
  s->nwaves_full = (uint32_t)prop.multiProcessorCount * std::max<uint32_t>(1u, per_cu);
  s->nwaves_lean = (uint32_t)prop.multiProcessorCount * std::max<uint32_t>(1u, per_cu_lean);
  const uint32_t np = s->dp.np;
  { // per-wave global scratch grows with the tree (100 B per leaf for the level-2 tables): bound the total by running
    // fewer waves on very large trees (reads are handed out dynamically, so any grid size is correct)
    const uint64_t np_ = np;
    const uint64_t tab_spill = std::min<uint32_t>(nslots2, 16384u), kt_spill = std::min<uint32_t>(nslots2, 65536u); // as below
    const uint64_t list_words = std::max<uint64_t>(nslots2, kEvSpill + tab_spill * ((np_ + 3) / 4 + 1) + kt_spill);
    const uint64_t per_wave = (uint64_t)nslots2 * np_ * (kPlaneWords + 1) * 4 + list_words * 4 + (uint64_t)kStackSpill * 8;
    const uint64_t budget = ((getenv("KR_ACC_SCRATCH_GB") ? (uint64_t)atoi(getenv("KR_ACC_SCRATCH_GB")) : 16ull) << 30) / ML;
    const uint32_t max_waves = (uint32_t)std::max<uint64_t>((uint64_t)prop.multiProcessorCount, budget / per_wave);
    s->nwaves_full = std::min(s->nwaves_full, max_waves);
    s->nwaves_lean = std::min(s->nwaves_lean, max_waves);
  }
  s->nwaves = std::max(s->nwaves_full, s->nwaves_lean);
  // default record capacity: up to 2 * leaves per read, at most 16 per read on average
  uint64_t per_read = std::min<uint64_t>(16, std::max<uint64_t>(8, (uint64_t)ix->dix.tree_nnodes + 1));
  uint64_t rc64 = max_records ? max_records : std::max<uint64_t>(1u << 16, (uint64_t)max_reads * per_read);
  s->rec_user_cap = rc64;
  // every resident wave of every lane may leave one partly used chunk behind: add that slack to the caller's bound
  s->rec_cap = (uint32_t)std::min<uint64_t>(rc64 + (uint64_t)ML * s->nwaves * kRecChunk, 1ull << 30);
  s->hit_cap = 1u << 22;
  int rc = 0;
  BatchOut& o = s->out;
  memset(&o, 0, sizeof(o));
#define SA(ptr, n) \
  if ((rc = salloc(s.get(), &ptr, (n)))) { kr_stream_destroy(s.release()); return rc; }
#define HA(ptr, n) \
  if ((rc = halloc(s.get(), &ptr, (n)))) { kr_stream_destroy(s.release()); return rc; }
  SA(s->d_bases, max_bases + 256);
  SA(s->d_offsets, (uint64_t)max_reads + ML);
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
  o.rec_cap = s->rec_cap;
  o.rec_stride = s->rec_cap;
  o.hit_cap = s->hit_cap;
  // item list between the two kernels: 256 hits per read on average, plus one partly used chunk per scan wave of every lane
  s->scan_blocks = (uint32_t)prop.multiProcessorCount * (4u * (ix->slot_log2w ? KR_SCAN_WPE_SLOT : KR_SCAN_WPE) / kScanWaves); // resident by construction (launch bounds)
  if (const char* e = getenv("KR_DEBUG_SCAN_BLOCKS_PER_CU")) s->scan_blocks = (uint32_t)prop.multiProcessorCount * (uint32_t)std::max(1, atoi(e));
  s->item_cap = (uint32_t)std::min<uint64_t>((uint64_t)max_reads * 256u + (uint64_t)ML * s->scan_blocks * kScanWaves * 2u * kItemChunk, 1ull << 31);
  SA(o.items, s->item_cap);
  SA(o.rd_it_off, max_reads);
  SA(o.rd_it_cnt, max_reads);
  SA(o.long_list, (uint64_t)max_reads + 16ull * s->nwaves * ML);
  o.nslots2 = nslots2;
  o.ev_spill = kEvSpill;
  o.tab_spill = std::min<uint32_t>(nslots2, 16384u);
  o.kt_spill = std::min<uint32_t>(nslots2, 65536u);
  const uint32_t g_list_words = std::max<uint32_t>(nslots2, o.ev_spill + o.tab_spill * ((np + 3) / 4 + 1) + o.kt_spill);
  o.g_list_words = g_list_words;
  o.bm_words = bm_words;
  for (uint32_t l = 0; l < ML; ++l) {
    Lane& L = s->lanes[l];
    // HIP streams are a scarce resource (the runtime folds every stream after the third onto one hardware queue unless
    // GPU_MAX_HW_QUEUES says otherwise): lane 0 has one from the start, the others get theirs when a batch first uses them
    if (l == 0) HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    for (auto& e : L.ev) HIP_TRY(hipEventCreate(&e));
    SA(L.d_counters, 32);
    SA(L.d_cursors, 3 * kCursors * kCursorStride);
    // lane 0 may be the only lane of a batch; a later lane never holds more than half of one
    const uint32_t lane_recs = l == 0 ? s->rec_cap : s->rec_cap / 2;
    L.dd_slots = std::min<uint32_t>(next_pow2(std::max<uint32_t>(2048u, lane_recs >> o.dd_shift)), 1u << 26);
    SA(L.d_dd, L.dd_slots);
    SA(L.d_stk, (uint64_t)s->nwaves * kStackSpill);
    SA(L.d_g_planes, (uint64_t)s->nwaves * nslots2 * np * kPlaneWords);
    SA(L.d_g_counts, (uint64_t)s->nwaves * nslots2 * np);
    SA(L.d_g_list, (uint64_t)s->nwaves * g_list_words);
    // on the lane's own stream and waited for: it does not synchronise with the null stream, and a
    // multi-GB clear (large trees) would otherwise still be running when the first batch arrives
    HIP_TRY(hipMemsetAsync(L.d_g_planes, 0, (uint64_t)s->nwaves * nslots2 * np * kPlaneWords * 4, s->lanes[0].stream));
    HIP_TRY(hipMemsetAsync(L.d_g_counts, 0, (uint64_t)s->nwaves * nslots2 * np * 4, s->lanes[0].stream));
    HA(L.h_counters, 32);
  }
  HIP_TRY(hipStreamSynchronize(s->lanes[0].stream));
  HA(s->h_bases, max_bases + 256);
  HA(s->h_offsets, (uint64_t)max_reads + 1);
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
  (void)hipSetDevice(s->device);
  for (auto& L : s->lanes)
    if (L.stream) (void)hipStreamSynchronize(L.stream);
  {
    kr_stream::PlaceWs& w = s->pw;
    for (void* p : {(void*)w.d_parent, (void*)w.d_eff, (void*)w.d_lo, (void*)w.d_idx_to_pt, (void*)w.d_elig, (void*)w.d_len, (void*)w.d_c0,
                    (void*)w.d_info, (void*)w.d_cse, (void*)w.d_cread, (void*)w.d_cnt, (void*)w.d_cd, (void*)w.d_cv, (void*)w.d_cchi, (void*)w.d_cprob,
                    (void*)w.d_rprob})
      if (p) (void)hipFree(p);
    for (void* p : {(void*)w.h_len, (void*)w.h_c0, (void*)w.h_info, (void*)w.h_cse, (void*)w.h_cnt, (void*)w.h_cd, (void*)w.h_cv, (void*)w.h_cchi})
      if (p) (void)hipHostFree(p);
  }
  for (void* p : s->dallocs) (void)hipFree(p);
  for (void* p : s->hallocs) (void)hipHostFree(p);
  for (auto& L : s->lanes) {
    for (auto& e : L.ev)
      if (e) (void)hipEventDestroy(e);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  delete s;
}

int kr_batch_submit(kr_stream* s, const uint8_t* bases, const uint64_t* offsets, uint32_t nreads, uint32_t flags)
{
  kr::clear_error();
  if (!s || !bases || !offsets) return kr::fail(KR_ERR_ARG, "kr_batch_submit: null argument");
  if (nreads == 0 || nreads > s->max_reads) return kr::fail(KR_ERR_ARG, "kr_batch_submit: nreads out of range for this stream");
  // every argument is checked before the stream's state is touched: a rejected submit leaves the previous batch as it was
  if (!(flags & KR_BASES_DEVICE) && offsets[nreads] - offsets[0] > s->max_bases)
    return kr::fail(KR_ERR_ARG, "kr_batch_submit: more bases than the stream was created for");
  HIP_TRY(hipSetDevice(s->ix->device));
  (void)hipGetLastError(); // a stale error of an earlier, unrelated call on this thread is not this batch's
  if (s->submitted && !s->waited)
    for (uint32_t l = 0; l < s->nlanes; ++l) HIP_TRY(hipStreamSynchronize(s->lanes[l].stream));
  s->nreads = nreads;
  s->flags = flags;
  s->submitted = true;
  s->waited = false;
  s->collected = false;
  s->batch_rc = KR_OK;
  // lanes of this batch: ranges of at least lane_min_reads reads; the hit tap (tests) keeps one hit buffer, hence one lane
  uint32_t P = std::min<uint32_t>(s->max_lanes, std::max<uint32_t>(1u, nreads / s->lane_min_reads));
  if (flags & KR_TAP_HITS) P = 1;
  if ((flags & KR_BASES_DEVICE) && !s->lanes_for_device_input) P = 1; // nothing to copy, nothing to overlap
  s->nlanes = P;
  if (flags & KR_TAP_HITS) {
    if (!s->h_hits) {
      HIP_TRY(hipMalloc((void**)&s->out.hits, (uint64_t)s->hit_cap * sizeof(kr_hit)));
      s->dallocs.push_back(s->out.hits);
      HIP_TRY(hipHostMalloc((void**)&s->h_hits, (uint64_t)s->hit_cap * sizeof(kr_hit), hipHostMallocDefault));
      s->hallocs.push_back(s->h_hits);
    }
  }
  const uint32_t lane_rec_cap = P == 1 ? s->rec_cap : (s->rec_cap / P) & ~63u;
  const uint32_t lane_item_cap = s->item_cap / P;
  for (uint32_t l = 0; l < P; ++l) {
    Lane& L = s->lanes[l];
    const uint32_t r0 = (uint32_t)((uint64_t)nreads * l / P), r1 = (uint32_t)((uint64_t)nreads * (l + 1) / P);
    L.read0 = r0, L.nreads = r1 - r0, L.rec_base = l * lane_rec_cap, L.rec_cap = lane_rec_cap, L.nrecs = 0;
    if (!L.stream) HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    hipStream_t st = L.stream;
    HIP_TRY(hipEventRecord(L.ev[0], st));
    if (flags & KR_BASES_DEVICE) {
      L.in.bases = bases;
      L.in.offsets = offsets + r0;
    } else {
      // the lane's bases land at the same offsets in d_bases as in the caller's buffer (relative to offsets[0]); its
      // offsets are rebased to the staging buffer
      const uint64_t b0 = offsets[r0] - offsets[0], b1 = offsets[r1] - offsets[0];
      const uint64_t* ho = s->h_offsets; // [nreads + 1], shared boundary entries are written with the same value
      const uint8_t* src = bases + offsets[r0];
      if (!(flags & KR_BASES_PINNED)) {
        memcpy(s->h_bases + b0, src, b1 - b0);
        src = s->h_bases + b0;
      }
      if ((flags & KR_BASES_PINNED) && offsets[0] == 0)
        ho = offsets; // page-locked and already relative to the staging buffer: no host pass at all
      else
        for (uint32_t r = r0; r <= r1; ++r) s->h_offsets[r] = offsets[r] - offsets[0];
      HIP_TRY(hipMemcpyAsync(s->d_bases + b0, src, b1 - b0, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(s->d_offsets + r0 + l, ho + r0, ((uint64_t)(r1 - r0) + 1) * 8, hipMemcpyHostToDevice, st));
      L.in.bases = s->d_bases;
      L.in.offsets = s->d_offsets + r0 + l;
    }
    L.in.nreads = r1 - r0;
    // the lane's view of the output: private scratch + slices of the stream's arrays
    BatchOut& o = L.out;
    o = s->out;
    o.counters = L.d_counters, o.cursors = L.d_cursors;
    o.dd_table = L.d_dd, o.dd_slots = L.dd_slots;
    o.g_planes = L.d_g_planes, o.g_counts = L.d_g_counts, o.g_list = L.d_g_list, o.stk_spill = L.d_stk;
    o.rd_off += r0, o.rd_cnt += r0, o.rd_onmers += r0, o.rd_filt += 2ull * r0, o.rd_na += r0;
    o.rd_it_off += r0, o.rd_it_cnt += r0;
    o.long_list += r0 + 16ull * s->nwaves * l;
    o.items += (uint64_t)l * lane_item_cap, o.item_cap = lane_item_cap;
    const uint64_t rb = L.rec_base;
    o.rec_read += rb, o.rec_key += rb, o.rec_hist += rb, o.rec_d += rb, o.rec_v += rb, o.rec_chisq += rb, o.rec_sel += rb;
    o.rec_w0 += rb, o.rec_rep += rb, o.rep_list += rb, o.rep_dv += rb;
    o.rec_cap = lane_rec_cap;
    o.hist_always = (flags & KR_TAP_ACCS) ? 1u : 0u; // else a record's planes exist only where its packed word cannot describe it
    int rc = launch_lane(s, L, flags);
    if (rc) return rc;
  }
  return KR_OK;
}

int kr_batch_wait(kr_stream* s)
{
  if (!s || !s->submitted) return kr::fail(KR_ERR_STATE, "kr_batch_wait: nothing submitted");
  if (s->waited) return s->batch_rc ? kr::fail(s->batch_rc, s->batch_msg) : KR_OK;
  HIP_TRY(hipSetDevice(s->ix->device));
  memset(s->h_counters, 0, sizeof(s->h_counters));
  uint32_t recs = 0, extent = 0;
  bool lane_full = false;
  for (uint32_t l = 0; l < s->nlanes; ++l) {
    Lane& L = s->lanes[l];
    HIP_TRY(hipStreamSynchronize(L.stream)); // kernels done, counters in L.h_counters
    const uint32_t* c = L.h_counters;
    L.nrecs = std::min(c[0], L.rec_cap);
    extent = std::max(extent, L.nrecs ? L.rec_base + L.nrecs : 0u);
    recs += c[4];
    lane_full = lane_full || c[0] > L.rec_cap;
    s->h_counters[1] |= c[1];
    for (int i : {2, 3, 4, 9, 10, 11, 12, 13, 16, 17, 18, 19, 20, 21, 22, 26}) s->h_counters[i] += c[i];
    for (int i : {14, 15}) s->h_counters[i] = std::max(s->h_counters[i], c[i]);
  }
  s->waited = true;
  if (s->dp.dbg & 512u)
    fprintf(stderr, "[kr stats] reads %u events %u keys %u batches %u big %u level-tiles %u max keys %u max events %u records %u\n", s->nreads,
            s->h_counters[9], s->h_counters[10], s->h_counters[11], s->h_counters[12], s->h_counters[13], s->h_counters[14],
            s->h_counters[15], s->h_counters[4]);
  if (s->dp.dbg & 512u)
    fprintf(stderr, "[kr stats] wave cycles/64 per launch: level passes %u (zero %u, event passes %u, key passes %u), finalize %u, whole read %u\n",
            s->h_counters[16], s->h_counters[19], s->h_counters[20], s->h_counters[21], s->h_counters[17], s->h_counters[18]);
  s->nrecs = extent; // record slots of the device view, unused ones (rec_key == 0) included
  s->nhits = std::min<uint64_t>(s->h_counters[3], s->hit_cap);
  if (recs > s->rec_user_cap)
    s->batch_rc = kr::fail(KR_ERR_CAPACITY, "the batch produced more records than max_records: submit fewer reads per batch");
  else
    s->batch_rc = check_errflags(s->h_counters[1] | (lane_full ? kErrRecCap : 0u));
  if (s->batch_rc) s->batch_msg = kr_last_error();
  return s->batch_rc;
}

static void fill_view(kr_stream* s, kr_result_view* v, bool device)
{
  memset(v, 0, sizeof(*v));
  v->nreads = s->nreads;
  v->nrecs = s->nrecs;
  const bool rows_only = (s->flags & KR_ROWS_ONLY) != 0;
  if (device) {
    v->read_off = s->out.rd_off, v->read_cnt = s->out.rd_cnt, v->read_onmers = s->out.rd_onmers, v->read_na = s->out.rd_na;
    v->rec_key = s->out.rec_key, v->rec_sel = s->out.rec_sel, v->rec_d = s->out.rec_d, v->rec_v = s->out.rec_v;
    v->rec_chisq = s->out.rec_chisq, v->rec_hist = s->out.rec_hist;
    v->rec_hist_stride = s->rec_cap;
  } else {
    v->read_off = s->h_rd_off, v->read_cnt = s->h_rd_cnt, v->read_onmers = s->h_rd_onmers, v->read_na = s->h_rd_na;
    v->rec_key = s->h_rec_key, v->rec_sel = s->h_rec_sel, v->rec_d = s->h_rec_d;
    v->rec_v = rows_only ? nullptr : s->h_rec_v;
    v->rec_chisq = rows_only ? nullptr : s->h_rec_chisq;
    v->rec_hist = (s->flags & KR_TAP_ACCS) && !rows_only ? s->h_rec_hist : nullptr;
    v->rec_hist_stride = s->nrecs;
  }
}

int kr_batch_collect(kr_stream* s, kr_result_view* v)
{
  kr::clear_error();
  if (!s || !v) return kr::fail(KR_ERR_ARG, "kr_batch_collect: null argument");
  if (!s->submitted) return kr::fail(KR_ERR_STATE, "kr_batch_collect: nothing submitted");
  HIP_TRY(hipSetDevice(s->ix->device));
  const bool rows_only = (s->flags & KR_ROWS_ONLY) != 0, full = !rows_only;
  const bool pipelined = !s->waited && !s->collected;
  static const bool timing = getenv("KR_COLLECT_TIMING") != nullptr;
  auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = timing ? wall() : 0.0;
  auto lap = [&](const char* what) {
    if (!timing) return;
    const double now = wall();
    fprintf(stderr, "[collect] %s %.2f ms\n", what, (now - t_mark) * 1e3);
    t_mark = now;
  };
  // Lane by lane: as soon as a lane's kernels are done its results start their way to the host on the lane's own
  // stream, while later lanes still compute.  The host arrays are compact (no unused slots between lanes), so a lane's
  // records land behind those of the lanes before it.
  auto ensure_host = [&](uint64_t need) -> int {
    if (need <= s->h_rec_cap && (s->h_rec_full || !full)) return KR_OK;
    uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(need + need / 4, s->h_rec_cap), 1u << 16);
    void** olds[] = {(void**)&s->h_rec_key, (void**)&s->h_rec_hist, (void**)&s->h_rec_sel, (void**)&s->h_rec_d, (void**)&s->h_rec_v, (void**)&s->h_rec_chisq};
    for (void** o : olds)
      if (*o) {
        s->hallocs.erase(std::remove(s->hallocs.begin(), s->hallocs.end(), *o), s->hallocs.end());
        (void)hipHostFree(*o);
        *o = nullptr;
      }
    s->h_rec_cap = 0;
    int rc2 = 0;
    if ((rc2 = halloc(s, &s->h_rec_key, cap)) || (rc2 = halloc(s, &s->h_rec_sel, cap)) || (rc2 = halloc(s, &s->h_rec_d, cap))) return rc2;
    s->h_rec_full = s->h_rec_full || full;
    if (s->h_rec_full)
      if ((rc2 = halloc(s, &s->h_rec_hist, cap * s->dp.np)) || (rc2 = halloc(s, &s->h_rec_v, cap)) || (rc2 = halloc(s, &s->h_rec_chisq, cap))) return rc2;
    s->h_rec_cap = cap;
    return KR_OK;
  };
  if (!pipelined || s->nlanes == 1 || (s->flags & KR_TAP_ACCS)) {
    int rc = kr_batch_wait(s); // (the histogram planes are laid out by the total record count: it must be known first)
    if (rc) return rc;
  }
  lap("wait for the kernels");
  uint64_t hoff = 0;
  bool copies_started = false;
  if (s->waited) {
    uint64_t total = 0;
    for (uint32_t l = 0; l < s->nlanes; ++l) total += s->lanes[l].nrecs;
    int rc = ensure_host(total);
    if (rc) return rc;
  }
  const uint64_t hist_stride_host = [&] { uint64_t t = 0; for (uint32_t l = 0; l < s->nlanes; ++l) t += s->lanes[l].nrecs; return t; }();
  for (uint32_t l = 0; l < s->nlanes; ++l) {
    Lane& L = s->lanes[l];
    hipStream_t st = L.stream;
    if (!s->waited) {
      HIP_TRY(hipStreamSynchronize(st));
      L.nrecs = std::min(L.h_counters[0], L.rec_cap);
      // room for this lane and, by its measure, for the lanes still running; if not, start over once everything is known
      const uint64_t guess = hoff + (uint64_t)L.nrecs * (s->nlanes - l) + (uint64_t)L.nrecs / 8 * (s->nlanes - l - 1);
      if (guess > s->h_rec_cap || (full && !s->h_rec_full)) {
        for (uint32_t j = 0; j < s->nlanes; ++j) HIP_TRY(hipStreamSynchronize(s->lanes[j].stream));
        int rc = kr_batch_wait(s);
        if (rc) return rc;
        uint64_t total = 0;
        for (uint32_t j = 0; j < s->nlanes; ++j) total += s->lanes[j].nrecs;
        if ((rc = ensure_host(total))) return rc;
        if (copies_started) { // restart: the buffers moved
          l = (uint32_t)-1, hoff = 0, copies_started = false;
          continue;
        }
      }
    }
    L.host_off = hoff;
    const uint64_t nr = L.nreads, nc = L.nrecs, r0 = L.read0;
    const BatchOut& o = L.out;
    HIP_TRY(hipMemcpyAsync(s->h_rd_off + r0, o.rd_off, nr * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s->h_rd_cnt + r0, o.rd_cnt, nr * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s->h_rd_na + r0, o.rd_na, nr, hipMemcpyDeviceToHost, st));
    if (full) {
      HIP_TRY(hipMemcpyAsync(s->h_rd_onmers + r0, o.rd_onmers, nr * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(s->h_rd_filt + 2 * r0, o.rd_filt, nr * 8, hipMemcpyDeviceToHost, st));
    }
    if (nc) {
      HIP_TRY(hipMemcpyAsync(s->h_rec_key + hoff, o.rec_key, nc * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(s->h_rec_sel + hoff, o.rec_sel, nc, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(s->h_rec_d + hoff, o.rec_d, nc * 8, hipMemcpyDeviceToHost, st));
      if (full) {
        HIP_TRY(hipMemcpyAsync(s->h_rec_v + hoff, o.rec_v, nc * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(s->h_rec_chisq + hoff, o.rec_chisq, nc * 8, hipMemcpyDeviceToHost, st));
        if (s->flags & KR_TAP_ACCS)
          for (uint32_t x = 0; x < s->dp.np; ++x)
            HIP_TRY(hipMemcpyAsync(s->h_rec_hist + (uint64_t)x * hist_stride_host + hoff, o.rec_hist + (uint64_t)x * s->rec_cap, nc * 4, hipMemcpyDeviceToHost, st));
      }
    }
    copies_started = true;
    hoff += nc;
  }
  if ((s->flags & KR_TAP_HITS) && s->waited && s->nhits)
    HIP_TRY(hipMemcpyAsync(s->h_hits, s->out.hits, s->nhits * sizeof(kr_hit), hipMemcpyDeviceToHost, s->lanes[0].stream));
  int rc = kr_batch_wait(s); // (a no-op when it already ran; otherwise every lane is idle by now: aggregates the counters)
  for (uint32_t l = 0; l < s->nlanes; ++l) HIP_TRY(hipStreamSynchronize(s->lanes[l].stream));
  if (rc) return rc;
  lap("copies to the host");
  // device offsets index the stream's arrays (lane slices); the host arrays are compact
  for (uint32_t l = 0; l < s->nlanes; ++l) {
    const Lane& L = s->lanes[l];
    const uint32_t delta = (uint32_t)L.host_off - L.rec_base; // (mod 2^32)
    if (!delta) continue;
    uint32_t* off = s->h_rd_off + L.read0;
    const uint32_t* cnt = s->h_rd_cnt + L.read0;
    for (uint32_t r = 0; r < L.nreads; ++r) off[r] = cnt[r] ? off[r] + delta : 0u;
  }
  s->collected = true;
  fill_view(s, v, false);
  v->nrecs = (uint32_t)hoff;
  v->rec_hist_stride = hoff;
  { // number of output rows: tens of millions of flags for a large batch, summed by the host pool
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::min(kr::parallel_width(), 16), hoff >> 20));
    std::vector<uint64_t> part((size_t)nt, 0);
    const uint8_t* sel = s->h_rec_sel;
    kr::parallel_for(nt, [&](int t) {
      uint64_t a = hoff * (uint64_t)t / nt, b = hoff * (uint64_t)(t + 1) / nt, c = 0;
      for (uint64_t i = a; i < b; ++i) c += sel[i];
      part[(size_t)t] = c;
    });
    uint64_t nrows = 0;
    for (uint64_t c : part) nrows += c;
    v->nrows = nrows;
  }
  lap("offsets + row count");
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
  // per phase: the sum over the lanes of the time between the lane's events (with one lane: the kernels' own time;
  // with several, lanes share the chip and the sums exceed ms_total, the span from the first lane's first kernel to the
  // last lane's last)
  float first = 0, last = 0;
  for (uint32_t l = 0; l < s->nlanes; ++l) {
    Lane& L = s->lanes[l];
    float a = 0;
    HIP_TRY(hipEventElapsedTime(&a, L.ev[0], L.ev[1]));
    t->ms_h2d += a;
    HIP_TRY(hipEventElapsedTime(&a, L.ev[1], L.ev[2]));
    t->ms_scan += a;
    HIP_TRY(hipEventElapsedTime(&a, L.ev[2], L.ev[3]));
    t->ms_acc += a;
    HIP_TRY(hipEventElapsedTime(&a, L.ev[3], L.ev[4]));
    t->ms_llh += a;
    if (l) {
      HIP_TRY(hipEventElapsedTime(&a, s->lanes[0].ev[1], L.ev[1]));
      first = std::min(first, a);
    }
    HIP_TRY(hipEventElapsedTime(&a, s->lanes[0].ev[1], L.ev[4]));
    last = std::max(last, a);
  }
  t->ms_total = last - first;
  t->lanes = s->nlanes;
  t->overflow_reads = s->h_counters[2];
  t->stack_spills = s->h_counters[26];
  return KR_OK;
}

} // extern "C"

// `place` back end for the batch last submitted on the stream (see kr_common.h)
int kr::place_on_device(kr_stream* s, const void* tree_tag, const kr::PlaceTreeArrays& T, const uint32_t* read_len, uint32_t tau,
                        bool no_filter, kr::PlaceDeviceResult* out)
{
  if (!s || !out || !read_len || !T.parent || !T.eff || !T.elig || !T.lo || !T.idx_to_pt) return kr::fail(KR_ERR_ARG, "place_on_device: null argument");
  if (!s->submitted || !(s->flags & KR_TAP_ACCS))
    return kr::fail(KR_ERR_STATE, "place: the batch must be submitted with KR_TAP_ACCS (the back end reads every record's histogram)");
  int rc = kr_batch_wait(s);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(s->device));
  kr_stream::PlaceWs& w = s->pw;
  hipStream_t st = s->lanes[0].stream;
  auto dev_renew = [&](auto*& p, uint64_t n) -> int {
    if (p) (void)hipFree(p), p = nullptr;
    HIP_TRY(hipMalloc((void**)&p, std::max<uint64_t>(16, n * sizeof(*p))));
    return KR_OK;
  };
  auto pin_renew = [&](auto*& p, uint64_t n) -> int {
    if (p) (void)hipHostFree(p), p = nullptr;
    HIP_TRY(hipHostMalloc((void**)&p, std::max<uint64_t>(16, n * sizeof(*p)), hipHostMallocDefault));
    return KR_OK;
  };
  if (w.tree_tag != tree_tag || w.pn != T.pn || w.nidx != T.nidx) { // the tree as device arrays (once per tree)
    w.tree_tag = nullptr;
    const uint64_t n1 = (uint64_t)T.pn + 1, n2 = (uint64_t)T.nidx + 1;
    if ((rc = dev_renew(w.d_parent, n1)) || (rc = dev_renew(w.d_eff, n1)) || (rc = dev_renew(w.d_lo, n1)) || (rc = dev_renew(w.d_elig, n1)) ||
        (rc = dev_renew(w.d_idx_to_pt, n2)))
      return rc;
    HIP_TRY(hipMemcpy(w.d_parent, T.parent, n1 * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(w.d_eff, T.eff, n1 * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(w.d_lo, T.lo, n1 * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(w.d_elig, T.elig, n1, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(w.d_idx_to_pt, T.idx_to_pt, n2 * 4, hipMemcpyHostToDevice));
    w.tree_tag = tree_tag, w.pn = T.pn, w.nidx = T.nidx;
  }
  const uint32_t n = s->nreads;
  if (n > w.reads_cap) {
    const uint64_t cap = (uint64_t)n + n / 4;
    if ((rc = dev_renew(w.d_len, cap)) || (rc = dev_renew(w.d_c0, cap)) || (rc = dev_renew(w.d_info, cap)) || (rc = pin_renew(w.h_len, cap)) ||
        (rc = pin_renew(w.h_c0, cap)) || (rc = pin_renew(w.h_info, cap)))
      return rc;
    if (!w.d_cnt && ((rc = dev_renew(w.d_cnt, 4)) || (rc = pin_renew(w.h_cnt, 4)))) return rc;
    w.reads_cap = cap;
  }
  { // candidate slots: every leaf and every distinct ancestor of a read may be one; a batch that needs more reports it
    const uint64_t want = std::max<uint64_t>(1u << 20, (uint64_t)n * 24);
    if (want > w.cand_cap) {
      if ((rc = dev_renew(w.d_cse, want)) || (rc = dev_renew(w.d_cread, want)) || (rc = dev_renew(w.d_cd, want)) || (rc = dev_renew(w.d_cv, want)) ||
          (rc = dev_renew(w.d_cchi, want)))
        return rc;
      w.cand_cap = want;
    }
    const uint64_t np_ = s->dp.np;
    if (w.cand_cap * (np_ + 2) > w.cprob_cap) {
      if ((rc = dev_renew(w.d_cprob, w.cand_cap * (np_ + 2)))) return rc;
      w.cprob_cap = w.cand_cap * (np_ + 2);
    }
    if (w.reads_cap * (np_ + 3) > w.rprob_cap) {
      if ((rc = dev_renew(w.d_rprob, w.reads_cap * (np_ + 3)))) return rc;
      w.rprob_cap = w.reads_cap * (np_ + 3);
    }
  }
  memcpy(w.h_len, read_len, (uint64_t)n * 4);
  HIP_TRY(hipMemcpyAsync(w.d_len, w.h_len, (uint64_t)n * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(w.d_cnt, 0, 16, st));
  HIP_TRY(hipMemsetAsync(w.d_cse, 0, w.cand_cap * 4, st)); // 0 = unused slot: what the second kernel and the host skip
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, s->device));
  PlaceTree PT{w.d_parent, w.d_eff, w.d_elig, w.d_lo, w.d_idx_to_pt, T.pn, T.nidx};
  PlaceOut PO{w.d_c0, w.d_info, w.d_cse, w.d_cread, w.d_cd, w.d_cv, w.d_cchi, w.d_cprob, w.d_rprob, w.d_cnt, (uint32_t)std::min<uint64_t>(w.cand_cap, 0x3FFFFFFFu)};
  const uint32_t grid = std::min<uint32_t>(n, (uint32_t)prop.multiProcessorCount * 16u);
  hipLaunchKernelGGL(kr_place_kernel, dim3(grid), dim3(kWave), 0, st, s->llh, s->ix->dix, s->out, n, w.d_len, PT, PO, tau, no_filter ? 1u : 0u);
  hipLaunchKernelGGL(kr_place_llh_kernel, dim3((uint32_t)prop.multiProcessorCount * 8u), dim3(256), 0, st, s->llh, PO);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(w.h_cnt, w.d_cnt, 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(w.h_c0, w.d_c0, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(w.h_info, w.d_info, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  out->nreads = n;
  out->overflow = w.h_cnt[1] != 0;
  if (out->overflow) return KR_OK;
  const uint64_t used = std::min<uint64_t>(w.h_cnt[0], w.cand_cap);
  if (used > w.h_cand_cap) {
    const uint64_t cap = used + used / 4 + 1024;
    if ((rc = pin_renew(w.h_cse, cap)) || (rc = pin_renew(w.h_cd, cap)) || (rc = pin_renew(w.h_cv, cap)) || (rc = pin_renew(w.h_cchi, cap))) return rc;
    w.h_cand_cap = cap;
  }
  if (used) {
    HIP_TRY(hipMemcpyAsync(w.h_cse, w.d_cse, used * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(w.h_cd, w.d_cd, used * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(w.h_cv, w.d_cv, used * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(w.h_cchi, w.d_cchi, used * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  out->rd_c0 = w.h_c0, out->rd_info = w.h_info, out->c_se = w.h_cse, out->c_d = w.h_cd, out->c_v = w.h_cv, out->c_chisq = w.h_cchi;
  return KR_OK;
}

extern "C" {

void* kr_host_alloc(uint64_t bytes)
{
  void* p = nullptr;
  if (hipHostMalloc(&p, std::max<uint64_t>(bytes, 16), hipHostMallocDefault) != hipSuccess) {
    kr::fail(KR_ERR_NOMEM, "kr_host_alloc: cannot allocate page-locked memory");
    return nullptr;
  }
  return p;
}
void kr_host_free(void* p)
{
  if (p) (void)hipHostFree(p);
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
