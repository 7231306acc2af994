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
//   kr_select_kernel     a wave per read, one record per lane: (d, v) of its problem to every record, then strand
//                        merge, closest reference, --filter / --dist-max / --no-multi selection
//                        [src/query.cpp:96-139,158-196].
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
//   kr_dev_common.inc      constants, device structs and helpers shared by the kernels
//   kr_dev_scan.inc        kernel 1: probe list, bucket scan, hit items
//   kr_dev_scan_pipe.inc   kernel 1 for slotted tables as a software pipeline across probe groups
//   kr_dev_scan_filt.inc   kernel 1 for FILTER slots (format 9): one 128-byte line of 24-bit codes per probe, candidates verified in kernel 2
//   kr_dev_expand.inc      accumulator tables, leaf events, colour classes and colour expansion (kernel 2 only)
//   kr_dev_accumulate.inc  kernel 2: event epilogue, plane tables, records
//   kr_dev_tiles.inc       long sequences across waves: tiles of a host batch, their merge per key, the real reads' results
//   kr_dev_likelihood.inc  likelihood, Brent, de-duplication, selection kernels
//   kr_dev_text.inc        the report rows as text, written on the device (kr_batch_submit_text / kr_batch_collect_text)
//   kr_dev_place.inc       back end of `place`: ancestor accumulation, candidates, their likelihoods (kr_place_kernel)
//   kr_dev_debug.inc       debug / tap kernels and the re-layout kernels of kr_index_upload
// and the host side of the device ABI:
//   kr_host_index.inc      kr_index: kr_index_upload / export / import / free, kr_index_broadcast (RCCL)
//   kr_host_stream.inc     kr_stream: lanes, kr_stream_create, kr_batch_submit / wait / collect / timing
//   kr_host_place.inc      kr::place_on_device (launch of the place kernels for kr_place_stream)
//   (below, in this file)  kr_host_alloc, debug entry points, kr_llh_batch, kr_llh_eval_indexed
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>
#include <type_traits>

#include "kr_common.h"
#include "kr_devutil.h"

#include <algorithm>
#include <iterator>
#include <limits>
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
#include "kr_dev_scan_pipe.inc"
#include "kr_dev_scan_filt.inc"
#include "kr_dev_expand.inc"
#include "kr_dev_accumulate.inc"
#include "kr_dev_tiles.inc"
#include "kr_dev_likelihood.inc"
#include "kr_dev_text.inc"
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
uint32_t probe_lds_bytes(uint32_t np, uint32_t bm_words, bool lean = false, uint32_t segs = 1)
{
  if (lean) // single-segment layout (segs = 2: its two-segment instantiation): short stack (bitmap aliases it) | key slots | event region
    return (lean_stack_bytes(bm_words) + kLdsSlots * 4 + lean_ev_words(np, segs) * 4 + 15u) & ~15u;
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

#include "kr_host_index.inc"
#include "kr_host_stream.inc"
#include "kr_host_place.inc"

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
  const uint32_t planes = getenv("KR_DEBUG_FE_PLANES") ? (uint32_t)atoi(getenv("KR_DEBUG_FE_PLANES")) : 0u; // tests: 1 the bit-plane front end, 2 the byte-table one
  hipLaunchKernelGGL(kr_front_end_kernel, dim3(std::min(nreads, 4096u)), dim3(kWave), 0, 0, ix->dix, in, stride, d_rix, d_enc, d_valid, d_pass, planes);
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
