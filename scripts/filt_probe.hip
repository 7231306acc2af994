// filt_probe.hip -- stand-alone experiment for the round-3 review's item 4 ("one line per probe in the scan").
// The product's slotted scan reads the whole 256-byte slot of a probe: two 128-byte lines, and the kernel pays per line
// (scripts/bin_probe.hip).  Here a slot is two FILTER lines: 8 chunks of 16 bytes = 5 x 24-bit codes (12 of the 16 non-LSH
// positions of a residual code: low 12 bits of each half) + 1 count byte (valid codes of the chunk; bit 7: the bucket goes
// on in the second line).  A probe reads line 0; only a bucket of more than 40 entries (Poisson(37.25): 28 %) reads line 1,
// a dependent load.  A code within th on its 12 positions is a CANDIDATE (necessary condition for hd <= th on all 16).
// Measures, on probes of the benchmark's shape (2^25 rows, ~124 probes per read, 500 M probes = one 4 M-read batch):
//   full256    : the product's access shape (4 lanes x 4 chunks, both lines, 32-bit codes)        -- the baseline
//   line128    : one line only, 32-bit codes (the floor of any one-line format)
//   filt_inline: filter lines, second line fetched on the spot by the probe's own lanes
//   filt_listed: filter lines, the probes that go on collected in an LDS list and scanned as passes of their own
// Build: hipcc -O3 --offload-arch=gfx950 scripts/filt_probe.hip -o /tmp/filt_probe ; run: /tmp/filt_probe [Mprobes]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kRowBits = 25;

__host__ __device__ __forceinline__ uint64_t mix(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// bucket length of a row: Poisson(37.25) by its normal approximation (sum of 12 uniforms)
__device__ __forceinline__ uint32_t row_len(uint32_t row)
{
  uint64_t h = mix(0xABCDEF12345ull + row);
  float s = 0.f;
  for (int i = 0; i < 12; ++i) {
    s += (float)(h & 0xFFFFu) * (1.0f / 65536.0f);
    h = mix(h);
  }
  const float z = s - 6.0f;
  const int l = (int)(37.25f + 6.103f * z + 0.5f);
  return (uint32_t)(l < 0 ? 0 : (l > 80 ? 80 : l));
}
__global__ void fill_plain(uint32_t* t, uint64_t nwords)
{
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) t[i] = (uint32_t)mix(i);
}
// one thread per 16-byte chunk
__global__ void fill_filter(uint4* t, uint64_t nrows)
{
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nrows * 16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t row = (uint32_t)(i >> 4), c = (uint32_t)(i & 15u), line = c >> 3, cc = c & 7u;
    const uint32_t len = row_len(row);
    const int left = (int)len - (int)(40u * line + 5u * cc);
    const uint32_t cnt = left <= 0 ? 0u : (left > 5 ? 5u : (uint32_t)left);
    const uint64_t a = mix(i * 2 + 1), b = mix(i * 2 + 2);
    uint4 v = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    v.w = (v.w & 0xFFFFFFu) | (cnt << 24) | ((line == 0 && len > 40) ? 0x80000000u : 0u);
    t[i] = v;
  }
}
struct Rec { uint32_t row, q, meta; };
__global__ void gen_probes(Rec* r, uint64_t n)
{
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix(i * 3 + 1);
    r[i] = Rec{(uint32_t)(h & ((1u << kRowBits) - 1u)), (uint32_t)(h >> 32), (uint32_t)i};
  }
}

__device__ __forceinline__ uint32_t hd_lr32(uint32_t e, uint32_t q)
{
  const uint32_t x = e ^ q;
  return __popc((x | (x >> 16)) & 0xFFFFu);
}
__device__ __forceinline__ uint32_t hd12(uint32_t c, uint32_t q)
{ // c, q: 24-bit codes (bits above 23 may hold anything)
  const uint32_t x = c ^ q;
  return __popc((x | (x >> 12)) & 0xFFFu);
}
// candidates among the first `cnt` codes of a chunk
__device__ __forceinline__ uint32_t chunk_cands(uint4 v, uint32_t q, uint32_t th)
{
  const uint32_t cnt = (v.w >> 24) & 7u;
  const uint32_t c0 = v.x, c1 = __builtin_amdgcn_alignbit(v.y, v.x, 24), c2 = __builtin_amdgcn_alignbit(v.z, v.y, 16), c3 = v.z >> 8, c4 = v.w;
  uint32_t m = (hd12(c0, q) <= th ? 1u : 0u) | (hd12(c1, q) <= th ? 2u : 0u) | (hd12(c2, q) <= th ? 4u : 0u) | (hd12(c3, q) <= th ? 8u : 0u) |
               (hd12(c4, q) <= th ? 16u : 0u);
  return m & ((1u << cnt) - 1u);
}

// ---- baseline shapes (bin_probe's scan_random): 4 lanes x CPL chunks of 16 bytes per probe, 32-bit codes
template <int CPL>
__global__ __launch_bounds__(256) void scan_plain(const uint32_t* slots, const Rec* recs, uint64_t n, uint32_t* nhits)
{
  constexpr uint32_t W = 16u * CPL;
  uint32_t hits = 0;
  const uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u, sub = lane & 3u;
  for (uint64_t i0 = w * 64; i0 < n; i0 += nw * 64) {
    const uint64_t i = i0 + lane;
    const bool on = i < n;
    const Rec r = on ? recs[i] : Rec{0, 0, 0};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int srcl = p * 16 + (lane >> 2);
      const uint32_t prow = __shfl(r.row, srcl), pq = __shfl(r.q, srcl);
      const bool pon = __shfl((int)on, srcl);
      const uint32_t* s = slots + (uint64_t)prow * W;
      uint4 v[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        v[j] = make_uint4(0, 0, 0, 0);
        if (pon) v[j] = *reinterpret_cast<const uint4*>(s + 4u * (sub + (uint32_t)j * 4u));
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const uint32_t m = min(min(hd_lr32(v[j].x, pq), hd_lr32(v[j].y, pq)), min(hd_lr32(v[j].z, pq), hd_lr32(v[j].w, pq)));
        hits += (m <= 4u && pon) ? 1u : 0u;
      }
    }
  }
  if (hits) atomicAdd(nhits, hits);
}

// ---- filter lines, LOG_G lanes-per-probe exponent, CPL = 8 >> LOG_G chunks per lane; second line on the spot
template <int LOG_G>
__global__ __launch_bounds__(256) void filt_inline(const uint4* slots, const Rec* recs, uint64_t n, uint32_t* nhits, uint32_t* nlong)
{
  constexpr uint32_t G = 1u << LOG_G, CPL = 8u >> LOG_G, PPS = 64u >> LOG_G;
  uint32_t hits = 0, longs = 0;
  const uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u, sub = lane & (G - 1u);
  for (uint64_t i0 = w * 64; i0 < n; i0 += nw * 64) {
    const uint64_t i = i0 + lane;
    const bool on = i < n;
    const Rec r = on ? recs[i] : Rec{0, 0, 0};
#pragma unroll
    for (uint32_t p = 0; p < G; ++p) {
      const int srcl = (int)(p * PPS + (lane >> LOG_G));
      const uint32_t prow = __shfl(r.row, srcl), pq = __shfl(r.q, srcl);
      const bool pon = __shfl((int)on, srcl);
      const uint4* s = slots + (uint64_t)prow * 16u;
      uint4 v[CPL];
#pragma unroll
      for (uint32_t j = 0; j < CPL; ++j) {
        v[j] = make_uint4(0, 0, 0, 0);
        if (pon) v[j] = s[sub + j * G];
      }
      uint32_t flag = 0;
#pragma unroll
      for (uint32_t j = 0; j < CPL; ++j) {
        hits += (uint32_t)__popc(chunk_cands(v[j], pq, 4u));
        flag |= v[j].w >> 31;
      }
      // every lane of the probe sees the flag in its own chunk (written to all chunks of line 0)
      if (flag) {
        longs += sub == 0 ? 1u : 0u;
#pragma unroll
        for (uint32_t j = 0; j < CPL; ++j) v[j] = s[8u + sub + j * G];
#pragma unroll
        for (uint32_t j = 0; j < CPL; ++j) hits += (uint32_t)__popc(chunk_cands(v[j], pq, 4u));
      }
    }
  }
  if (hits) atomicAdd(nhits, hits);
  if (longs) atomicAdd(nlong, longs);
}

// ---- filter lines, the probes that go on listed in LDS and scanned as passes of their own
template <int LOG_G>
__global__ __launch_bounds__(256) void filt_listed(const uint4* slots, const Rec* recs, uint64_t n, uint32_t* nhits, uint32_t* nlong)
{
  constexpr uint32_t G = 1u << LOG_G, CPL = 8u >> LOG_G, PPS = 64u >> LOG_G;
  __shared__ uint32_t s_row[4][64], s_q[4][64];
  uint32_t hits = 0, longs = 0;
  const uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u, sub = lane & (G - 1u), wv = threadIdx.x >> 6;
  for (uint64_t i0 = w * 64; i0 < n; i0 += nw * 64) {
    const uint64_t i = i0 + lane;
    const bool on = i < n;
    const Rec r = on ? recs[i] : Rec{0, 0, 0};
    uint32_t nl = 0; // wave-uniform: listed probes
#pragma unroll
    for (uint32_t p = 0; p < G; ++p) {
      const int srcl = (int)(p * PPS + (lane >> LOG_G));
      const uint32_t prow = __shfl(r.row, srcl), pq = __shfl(r.q, srcl);
      const bool pon = __shfl((int)on, srcl);
      const uint4* s = slots + (uint64_t)prow * 16u;
      uint4 v[CPL];
#pragma unroll
      for (uint32_t j = 0; j < CPL; ++j) {
        v[j] = make_uint4(0, 0, 0, 0);
        if (pon) v[j] = s[sub + j * G];
      }
      uint32_t flag = 0;
#pragma unroll
      for (uint32_t j = 0; j < CPL; ++j) {
        hits += (uint32_t)__popc(chunk_cands(v[j], pq, 4u));
        flag |= v[j].w >> 31;
      }
      const uint64_t fm = __ballot(flag && sub == 0);
      if (flag && sub == 0) {
        const uint32_t o = nl + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
        s_row[wv][o] = prow;
        s_q[wv][o] = pq;
      }
      nl += (uint32_t)__popcll(fm);
    }
    longs += lane == 0 ? nl : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t p0 = 0; p0 < nl; p0 += PPS) {
      const uint32_t pi = p0 + (lane >> LOG_G);
      const bool pon = pi < nl;
      const uint32_t prow = pon ? s_row[wv][pi] : 0u, pq = pon ? s_q[wv][pi] : 0u;
      const uint4* s = slots + (uint64_t)prow * 16u + 8u;
      uint4 v[CPL];
#pragma unroll
      for (uint32_t j = 0; j < CPL; ++j) {
        v[j] = make_uint4(0, 0, 0, 0);
        if (pon) v[j] = s[sub + j * G];
      }
#pragma unroll
      for (uint32_t j = 0; j < CPL; ++j) hits += (uint32_t)__popc(chunk_cands(v[j], pq, 4u));
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (hits) atomicAdd(nhits, hits);
  if (longs) atomicAdd(nlong, longs);
}

int main(int argc, char** argv)
{
  const uint64_t n = (uint64_t)(argc > 1 ? atof(argv[1]) : 500.0) * 1000000ull;
  const uint64_t nrows = 1ull << kRowBits;
  uint32_t *plain, *nhits, *nlong;
  uint4* filt;
  Rec* a;
  CK(hipMalloc(&plain, nrows * 256));
  CK(hipMalloc(&filt, nrows * 256));
  CK(hipMalloc(&a, n * sizeof(Rec)));
  CK(hipMalloc(&nhits, 4));
  CK(hipMalloc(&nlong, 4));
  fill_plain<<<8192, 256>>>(plain, nrows * 64);
  fill_filter<<<8192, 256>>>(filt, nrows);
  gen_probes<<<8192, 256>>>(a, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](const char* what, auto&& fn) {
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      CK(hipMemset(nhits, 0, 4));
      CK(hipMemset(nlong, 0, 4));
      CK(hipEventRecord(e0));
      fn();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    uint32_t h = 0, l = 0;
    CK(hipMemcpy(&h, nhits, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&l, nlong, 4, hipMemcpyDeviceToHost));
    printf("%-44s %8.3f ms  %7.2f ms per 500M probes   cands/probe %.3f  second-line probes %.3f\n", what, best, best * 500e6 / (double)n, (double)h / (double)n,
           (double)l / (double)n);
  };
  for (int blocks : {16, 24, 32}) {
    printf("-- %d workgroups of 256 per CU\n", blocks);
    const int g = 256 * blocks;
    timed("full256 (4 lanes x 4 chunks, 32-bit codes)", [&] { scan_plain<4><<<g, 256>>>(plain, a, n, nhits); });
    timed("line128 (4 lanes x 2 chunks, 32-bit codes)", [&] { scan_plain<2><<<g, 256>>>(plain, a, n, nhits); });
    timed("filt_inline G=8", [&] { filt_inline<3><<<g, 256>>>(filt, a, n, nhits, nlong); });
    timed("filt_inline G=4", [&] { filt_inline<2><<<g, 256>>>(filt, a, n, nhits, nlong); });
    timed("filt_inline G=2", [&] { filt_inline<1><<<g, 256>>>(filt, a, n, nhits, nlong); });
    timed("filt_listed G=8", [&] { filt_listed<3><<<g, 256>>>(filt, a, n, nhits, nlong); });
    timed("filt_listed G=4", [&] { filt_listed<2><<<g, 256>>>(filt, a, n, nhits, nlong); });
    timed("filt_listed G=2", [&] { filt_listed<1><<<g, 256>>>(filt, a, n, nhits, nlong); });
  }
  return 0;
}
