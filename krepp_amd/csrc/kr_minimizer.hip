// kr_minimizer.hip — windowed-minimizer extraction for the index build's leaf stage on gfx950.
//
// Reference: RSeq::extract_mers (src/rqseq.cpp:51-144) with xur64_hash = MurmurHash3's fmix64
// (src/common.hpp:147-155) as the ordering hash, LSHF::compute_hash / drop_ppos_lr
// (src/lshf.cpp:62-69) on the winning k-mer, the residue filter (src/rqseq.cpp:125-128) and the
// two HyperLogLog(12) sketches behind rho (src/rqseq.cpp:63-64,108-118,142-143).
//
// One 256-thread workgroup per tile of 2048 k-mer end positions of one contig:
//   1. every wave turns 64 bases at a time into three 64-bit ballots (low bit, high bit, invalid)
//      and parks them in LDS: the tile plus a 256-base halo is 36 words per bit string;
//   2. every thread extracts the k-bit windows of its positions with a funnel shift, interleaves
//      them into enc_bp, hashes (fmix64) and stores (hash, code, valid) in LDS; valid k-mers of
//      the tile go to the contig's first HyperLogLog (register max, pre-checked);
//   3. every thread takes the minimum hash of the w-k+1 k-mers ending at its positions (the
//      reference's ring buffer holds exactly those once w consecutive valid bases have been seen;
//      fmix64 is a bijection, so ties mean identical k-mers and the slot-order tie rule of
//      std::min_element is moot), feeds the second HyperLogLog, applies LSH + residue filter and
//      appends the (row << 32 | enc32) key by wave-aggregated atomics.
// The one case that depends on the ring buffer's history — a contig whose LAST run of valid bases
// is shorter than w (stale or never-written slots take part, src/rqseq.cpp:108-116) — is a single
// extra emission per contig and is computed on the host (kr::contig_end_special).
#include "kr_common.h"
#include "kr_devutil.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace kr {
int check_minimizer_params(const kr_build_params* bp, BuildCfg& c, LshPositions& lsh);
int finish_minimizers(std::vector<uint64_t>& keys, double n1, double n2, kr_minimizer_result* out);
} // namespace kr

namespace {

constexpr int kTile = 2048;  // k-mer end positions per workgroup
constexpr int kHalo = 256;   // bases in front of the tile (w <= 255)
constexpr int kWords = (kTile + kHalo) / 64;
constexpr int kMaxWin = 256; // w - k + 1 <= 237

struct MinParams {
  uint32_t k, w, ldiff, m, r, frac;
  PextMask pmask, nmask;
};

__device__ __forceinline__ uint64_t spread32(uint32_t v)
{ // bit j -> bit 2j
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}
__device__ __forceinline__ uint64_t dev_fmix64(uint64_t v)
{
  v ^= v >> 33;
  v *= 0xff51afd7ed558ccdull;
  v ^= v >> 33;
  v *= 0xc4ceb9fe1a85ec53ull;
  v ^= v >> 33;
  return v;
}
__device__ __forceinline__ uint32_t window_bits(const uint64_t* arr, uint32_t s)
{
  uint64_t w0 = arr[s >> 6], w1 = arr[(s >> 6) + 1];
  uint32_t sh = s & 63u;
  uint64_t v = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
  return (uint32_t)v;
}
// hll::HyperLogLog(12)::add (src/hyperloglog.hpp:98-105) on registers widened to u32
__device__ __forceinline__ void hll_add(uint32_t* reg, uint32_t hsh)
{
  uint32_t ix = hsh >> 20, rest = hsh << 12;
  uint32_t rank = (uint32_t)min(20, rest ? __clz((int)rest) : 32) + 1u;
  if (reg[ix] < rank) atomicMax(&reg[ix], rank); // registers saturate quickly: most adds stop at the read
}

__global__ __launch_bounds__(256) void kr_minimizer_kernel(MinParams P, const uint8_t* bases, const uint64_t* offsets,
                                                           const uint2* tiles, uint32_t* hll, uint64_t* keys,
                                                           unsigned long long* nkeys, unsigned long long key_cap)
{
  __shared__ uint64_t sL[kWords + 1], sH[kWords + 1], sN[kWords + 1];
  __shared__ uint64_t sZ[kTile + kMaxWin], sX[kTile + kMaxWin];
  __shared__ uint8_t sOk[kTile + kMaxWin];
  const uint2 tl = tiles[blockIdx.x];
  const uint32_t ci = tl.x;
  const int64_t p0 = tl.y;
  const uint8_t* seq = bases + offsets[ci];
  const int64_t len = (int64_t)(offsets[ci + 1] - offsets[ci]);
  uint32_t* reg1 = hll + (uint64_t)ci * 8192u;
  uint32_t* reg2 = reg1 + 4096;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const int64_t b0 = p0 - kHalo; // base index of bit 0 of the LDS strings
  // ---- 1. bases -> bit strings
  for (uint32_t c = wave; c < (uint32_t)kWords; c += 4) {
    int64_t b = b0 + 64 * (int64_t)c + lane;
    uint32_t code = 4;
    if (b >= 0 && b < len) code = base_code(seq[b]);
    uint64_t L = __ballot((code & 1u) && code < 4), H = __ballot(((code >> 1) & 1u) && code < 4), N = __ballot(code >= 4);
    if (lane == 0) sL[c] = L, sH[c] = H, sN[c] = N;
  }
  if (tid == 0) sL[kWords] = 0, sH[kWords] = 0, sN[kWords] = ~0ull;
  __syncthreads();
  // ---- 2. hash of the k-mer ENDING at base g, for g in [p0 - ldiff + 1, p0 + kTile)
  const int64_t g0 = p0 - (int64_t)P.ldiff + 1;
  const uint32_t count = kTile + P.ldiff - 1;
  const uint32_t mk = (P.k < 32) ? ((1u << P.k) - 1u) : 0xFFFFFFFFu;
  for (uint32_t t = tid; t < count; t += 256) {
    int64_t g = g0 + t;
    bool ok = false;
    uint64_t x = 0, z = ~0ull;
    if (g >= (int64_t)P.k - 1 && g < len) {
      uint32_t s = (uint32_t)(g - (int64_t)P.k + 1 - b0);
      uint32_t wn = window_bits(sN, s) & mk;
      if (wn == 0) {
        uint32_t wl = window_bits(sL, s) & mk, wh = window_bits(sH, s) & mk;
        uint32_t lo = __brev(wl) >> (32 - P.k), hi = __brev(wh) >> (32 - P.k); // position 0 = LAST base
        x = spread32(lo) | (spread32(hi) << 1);
        z = dev_fmix64(x);
        ok = true;
        if (g >= p0) hll_add(reg1, (uint32_t)z); // c1: every valid k-mer, once
      }
    }
    sZ[t] = z;
    sX[t] = x;
    sOk[t] = ok ? 1 : 0;
  }
  __syncthreads();
  // ---- 3. window minimum, second sketch, LSH + residue filter, key
  for (uint32_t t2 = tid; t2 < (uint32_t)kTile; t2 += 256) {
    int64_t g = p0 + t2;
    bool emit = false;
    uint64_t key = 0;
    if (g < len) {
      bool all = true;
      uint64_t bz = ~0ull, bx = 0;
      for (uint32_t q = 0; q < P.ldiff; ++q) {
        uint32_t t = t2 + q;
        all = all && sOk[t];
        uint64_t z = sZ[t];
        if (z < bz) bz = z, bx = sX[t];
      }
      if (all) { // w consecutive valid bases end here (src/rqseq.cpp:111: l >= w)
        hll_add(reg2, (uint32_t)bz);
        uint32_t lo = 0, hi = 0; // de-interleave the winning code
        {
          uint64_t e = bx & 0x5555555555555555ull, o = (bx >> 1) & 0x5555555555555555ull;
          e = (e | (e >> 1)) & 0x3333333333333333ull, o = (o | (o >> 1)) & 0x3333333333333333ull;
          e = (e | (e >> 2)) & 0x0F0F0F0F0F0F0F0Full, o = (o | (o >> 2)) & 0x0F0F0F0F0F0F0F0Full;
          e = (e | (e >> 4)) & 0x00FF00FF00FF00FFull, o = (o | (o >> 4)) & 0x00FF00FF00FF00FFull;
          e = (e | (e >> 8)) & 0x0000FFFF0000FFFFull, o = (o | (o >> 8)) & 0x0000FFFF0000FFFFull;
          e = (e | (e >> 16)) & 0x00000000FFFFFFFFull, o = (o | (o >> 16)) & 0x00000000FFFFFFFFull;
          lo = (uint32_t)e, hi = (uint32_t)o;
        }
        uint32_t rix = spread16(pext32(lo, P.pmask)) | (spread16(pext32(hi, P.pmask)) << 1);
        uint32_t res = rix % P.m;
        if (P.frac ? res <= P.r : res == P.r) {
          uint32_t row = P.frac ? rix / P.m * (P.r + 1) + res : rix / P.m;
          uint32_t enc = pext32(lo, P.nmask) | (pext32(hi, P.nmask) << 16);
          key = ((uint64_t)row << 32) | enc;
          emit = true;
        }
      }
    }
    uint64_t m = __ballot(emit);
    if (m) {
      unsigned long long base = 0;
      if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(nkeys, (unsigned long long)__popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1);
      unsigned long long at = base + __popcll(m & ((1ull << lane) - 1ull));
      if (emit && at < key_cap) keys[at] = key;
    }
  }
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e__ = (expr);                                                                            \
    if (e__ != hipSuccess)                                                                              \
      return kr::fail(e__ == hipErrorOutOfMemory ? KR_ERR_NOMEM : KR_ERR_NO_DEVICE,                      \
                      std::string(#expr) + ": " + hipGetErrorString(e__));                              \
  } while (0)

} // namespace

extern "C" int kr_minimizers_device(int device, const kr_build_params* bp, const uint8_t* bases, const uint64_t* offsets,
                                    uint32_t ncontigs, kr_minimizer_result* out)
{
  kr::clear_error();
  if (!bases || !offsets || !out) return kr::fail(KR_ERR_ARG, "kr_minimizers_device: null argument");
  kr::BuildCfg c;
  kr::LshPositions lsh;
  int rc = kr::check_minimizer_params(bp, c, lsh);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return kr::fail(KR_ERR_NO_DEVICE, "no HIP device available");
  if (device < 0 || device >= ndev) return kr::fail(KR_ERR_ARG, "kr_minimizers_device: bad device ordinal");
  HIP_TRY(hipSetDevice(device));
  const uint32_t w = std::max(c.w, c.k);
  MinParams P;
  P.k = c.k, P.w = w, P.ldiff = w - c.k + 1, P.m = c.m, P.r = c.r, P.frac = c.frac ? 1 : 0;
  P.pmask = make_pext(lsh.pasc);
  P.nmask = make_pext(lsh.npos);
  // contigs shorter than w are skipped altogether (RSeq::set_curr_seq, src/rqseq.hpp:81-87)
  std::vector<uint32_t> kept;
  std::vector<uint2> tiles;
  const uint64_t total = offsets[ncontigs];
  for (uint32_t q = 0; q < ncontigs; ++q) {
    uint64_t len = offsets[q + 1] - offsets[q];
    if (len < c.w) continue;
    if (len >= (1ull << 32)) return kr::fail(KR_ERR_ARG, "kr_minimizers_device: contig longer than 2^32 bases");
    kept.push_back(q);
    for (uint64_t p0 = 0; p0 < len; p0 += kTile) tiles.push_back(make_uint2(q, (uint32_t)p0));
  }
  std::vector<uint64_t> keys;
  double n1 = 0, n2 = 0;
  if (!tiles.empty()) {
    uint8_t* d_bases = nullptr;
    uint64_t *d_off = nullptr, *d_keys = nullptr;
    uint2* d_tiles = nullptr;
    uint32_t* d_hll = nullptr;
    unsigned long long* d_n = nullptr;
    const unsigned long long cap = total + 64;
    HIP_TRY(hipMalloc((void**)&d_bases, total + 64));
    HIP_TRY(hipMalloc((void**)&d_off, ((uint64_t)ncontigs + 1) * 8));
    HIP_TRY(hipMalloc((void**)&d_tiles, tiles.size() * sizeof(uint2)));
    HIP_TRY(hipMalloc((void**)&d_hll, (uint64_t)ncontigs * 8192 * 4));
    HIP_TRY(hipMalloc((void**)&d_keys, cap * 8));
    HIP_TRY(hipMalloc((void**)&d_n, 8));
    HIP_TRY(hipMemcpy(d_bases, bases, total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_off, offsets, ((uint64_t)ncontigs + 1) * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(uint2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_hll, 0, (uint64_t)ncontigs * 8192 * 4));
    HIP_TRY(hipMemset(d_n, 0, 8));
    hipLaunchKernelGGL(kr_minimizer_kernel, dim3((uint32_t)tiles.size()), dim3(256), 0, 0, P, d_bases, d_off, d_tiles, d_hll,
                       d_keys, d_n, cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long nk = 0;
    HIP_TRY(hipMemcpy(&nk, d_n, 8, hipMemcpyDeviceToHost));
    if (nk > cap) return kr::fail(KR_ERR_CAPACITY, "kr_minimizers_device: key buffer overflow");
    keys.resize(nk);
    if (nk) HIP_TRY(hipMemcpy(keys.data(), d_keys, nk * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> regs((size_t)ncontigs * 8192);
    HIP_TRY(hipMemcpy(regs.data(), d_hll, regs.size() * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_bases), (void)hipFree(d_off), (void)hipFree(d_tiles), (void)hipFree(d_hll), (void)hipFree(d_keys), (void)hipFree(d_n);
    // contig ends + HyperLogLog estimates, contig by contig as the reference adds them
    std::vector<uint8_t> r8(4096);
    for (uint32_t q : kept) {
      const uint8_t* seq = bases + offsets[q];
      uint64_t len = offsets[q + 1] - offsets[q];
      uint32_t* r1 = regs.data() + (size_t)q * 8192;
      uint32_t* r2 = r1 + 4096;
      uint64_t x = 0, z = 0;
      if (kr::contig_end_special(seq, len, c, x, z)) {
        uint32_t hsh = (uint32_t)z, ix = hsh >> 20, rest = hsh << 12;
        uint32_t rank = (uint32_t)std::min(20, rest ? __builtin_clz(rest) : 32) + 1u;
        r2[ix] = std::max(r2[ix], rank);
        int64_t row = kr::build_row(lsh.rix(x), c);
        if (row >= 0) keys.push_back(((uint64_t)row << 32) | lsh.enc32(x));
      }
      for (int i = 0; i < 4096; ++i) r8[i] = (uint8_t)r1[i];
      n1 += kr::hll12_estimate(r8.data());
      for (int i = 0; i < 4096; ++i) r8[i] = (uint8_t)r2[i];
      n2 += kr::hll12_estimate(r8.data());
    }
  }
  return kr::finish_minimizers(keys, n1, n2, out);
}
