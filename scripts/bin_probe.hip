// bin_probe.hip -- stand-alone experiment for DESIGN 6-2 (probes grouped by table row before the scan).
// Measures, on synthetic probes of the benchmark's shape (2^25 rows x 256-byte slots, ~124 probes per read):
//   part<NB>    : tile-sort partition of 12-byte probe records into NB bins (LDS histogram, one global cursor add per bin and tile)
//   scan_random : every 4-lane group reads the 256-byte slot of its probe, probes in arrival order (today's access pattern)
//   scan_binned : the same scan over probes grouped into row bins small enough for an XCD's L2, bins dealt to XCDs
// Build: hipcc -O3 --offload-arch=gfx950 scripts/bin_probe.hip -o /tmp/bin_probe ; run: /tmp/bin_probe [Mprobes] [log2_fine_bins] [blocks_per_bin]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kRowBits = 25;
constexpr uint32_t kSlotWords = 64;

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void fill_table(uint32_t* t, uint64_t nwords)
{
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) t[i] = (uint32_t)mix(i);
}
struct Rec { uint32_t row, q, meta; };
__global__ void gen_probes(Rec* r, uint64_t n)
{
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix(i * 3 + 1);
    r[i] = Rec{(uint32_t)(h & ((1u << kRowBits) - 1u)), (uint32_t)(h >> 32), (uint32_t)i};
  }
}

// ---- tile-sort partition: block of 256 threads, U records per thread
template <int NB, int U>
__global__ __launch_bounds__(256) void part(const Rec* in, const uint32_t* in_count, uint32_t in_cap, uint32_t blocks_per_in, int shift, Rec* out,
                                            uint32_t out_cap, uint32_t* cursors, uint32_t* overflow)
{
  // input region blockIdx / blocks_per_in (one region = the whole input in pass 1), output bin = region * NB + sub
  __shared__ uint32_t hist[NB], start[NB], gbase[NB];
  __shared__ Rec tile[256 * U];
  const uint32_t region = blockIdx.x / blocks_per_in, part_i = blockIdx.x % blocks_per_in;
  const uint64_t n = in_count[region] < in_cap ? in_count[region] : in_cap;
  const Rec* src = in + (uint64_t)region * in_cap;
  const uint32_t tid = threadIdx.x;
  for (uint64_t t0 = (uint64_t)part_i * 256 * U; t0 < n; t0 += (uint64_t)blocks_per_in * 256 * U) {
    for (uint32_t b = tid; b < NB; b += 256) hist[b] = 0;
    __syncthreads();
    Rec r[U];
    uint32_t rank[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint64_t i = t0 + (uint64_t)j * 256 + tid;
      rank[j] = 0xFFFFFFFFu;
      if (i < n) {
        r[j] = src[i];
        rank[j] = atomicAdd(&hist[(r[j].row >> shift) & (NB - 1)], 1u);
      }
    }
    __syncthreads();
    if (tid < 64) { // exclusive prefix over NB bins by one wave
      uint32_t run = 0;
      for (uint32_t b0 = 0; b0 < NB; b0 += 64) {
        const uint32_t c = b0 + tid < NB ? hist[b0 + tid] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t o = __shfl_up(inc, d);
          if (tid >= (uint32_t)d) inc += o;
        }
        if (b0 + tid < NB) start[b0 + tid] = run + inc - c;
        run += __shfl(inc, 63);
      }
    }
    for (uint32_t b = tid; b < NB; b += 256) gbase[b] = hist[b] ? atomicAdd(&cursors[region * NB + b], hist[b]) : 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (rank[j] != 0xFFFFFFFFu) tile[start[(r[j].row >> shift) & (NB - 1)] + rank[j]] = r[j];
    __syncthreads();
    const uint32_t cnt = (uint32_t)((n - t0) < (uint64_t)(256 * U) ? (n - t0) : (uint64_t)(256 * U));
    for (uint32_t p = tid; p < cnt; p += 256) {
      const Rec x = tile[p];
      const uint32_t b = (x.row >> shift) & (NB - 1);
      const uint32_t o = gbase[b] + (p - start[b]);
      if (o < out_cap)
        out[(uint64_t)(region * NB + b) * out_cap + o] = x;
      else
        *overflow = 1;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ uint32_t hd_lr32(uint32_t e, uint32_t q)
{
  const uint32_t x = e ^ q;
  return __popc((x | (x >> 16)) & 0xFFFFu);
}

// one wave-iteration: 64 records in registers, 4 passes of 16 probes, G = 4 lanes x 4 chunks of 16 bytes per probe
template <int CPL = 4>
__device__ __forceinline__ uint32_t scan64(const uint32_t* slots, uint32_t row, uint32_t q, bool on)
{
  constexpr uint32_t kSlotWords = 16u * CPL; // (shadows the global: a slot of CPL chunks per lane of a 4-lane group)
  const uint32_t lane = threadIdx.x & 63u, sub = lane & 3u;
  uint32_t hits = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int srcl = p * 16 + (lane >> 2);
    const uint32_t prow = __shfl(row, srcl), pq = __shfl(q, srcl);
    const bool pon = __shfl((int)on, srcl);
    const uint32_t* s = slots + (uint64_t)prow * kSlotWords;
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
  return hits;
}

template <int CPL>
__global__ __launch_bounds__(256) void scan_random(const uint32_t* slots, const Rec* recs, uint64_t n, uint32_t* nhits)
{
  uint32_t hits = 0;
  const uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t i0 = w * 64; i0 < n; i0 += nw * 64) {
    const uint64_t i = i0 + (threadIdx.x & 63u);
    const bool on = i < n;
    const Rec r = on ? recs[i] : Rec{0, 0, 0};
    hits += scan64<CPL>(slots, r.row, r.q, on);
  }
  if (hits) atomicAdd(nhits, hits);
}

// bin f = 8 * (blockIdx / (8 * PB)) + blockIdx % 8 : consecutive bins go to different XCDs, the PB blocks of a bin to the same one
__global__ __launch_bounds__(256) void scan_binned(const uint32_t* slots, const Rec* recs, const uint32_t* counts, uint32_t cap, uint32_t PB, uint32_t* nhits)
{
  const uint32_t f = 8u * (blockIdx.x / (8u * PB)) + (blockIdx.x & 7u), part_i = (blockIdx.x >> 3) % PB;
  const uint32_t n = min(counts[f], cap);
  const Rec* src = recs + (uint64_t)f * cap;
  uint32_t hits = 0;
  const uint32_t w = part_i * 4u + (threadIdx.x >> 6), nw = PB * 4u;
  for (uint32_t i0 = w * 64u; i0 < n; i0 += nw * 64u) {
    const uint32_t i = i0 + (threadIdx.x & 63u);
    const bool on = i < n;
    const Rec r = on ? src[i] : Rec{0, 0, 0};
    hits += scan64(slots, r.row, r.q, on);
  }
  if (hits) atomicAdd(nhits, hits);
}

int main(int argc, char** argv)
{
  const uint64_t n = (uint64_t)(argc > 1 ? atof(argv[1]) : 250.0) * 1000000ull;
  const int fine_log2 = argc > 2 ? atoi(argv[2]) : 13; // fine bins = 2^fine_log2 = 128 coarse x 2^(fine_log2-7)
  const uint32_t PB = argc > 3 ? (uint32_t)atoi(argv[3]) : 32;
  constexpr int NB1 = 128;
  const int nb2 = 1 << (fine_log2 - 7);
  const uint64_t nrows = 1ull << kRowBits;
  uint32_t *slots, *cur1, *cur2, *ovf, *nhits;
  Rec *a, *b, *c;
  CK(hipMalloc(&slots, nrows * kSlotWords * 4));
  const uint32_t cap1 = (uint32_t)(n / NB1 * 1.1) + 4096, cap2 = (uint32_t)(n / (1u << fine_log2) * 1.25) + 1024;
  CK(hipMalloc(&a, n * sizeof(Rec)));
  CK(hipMalloc(&b, (uint64_t)NB1 * cap1 * sizeof(Rec)));
  CK(hipMalloc(&c, (uint64_t)(1u << fine_log2) * cap2 * sizeof(Rec)));
  CK(hipMalloc(&cur1, (NB1 + 1) * 4));
  CK(hipMalloc(&cur2, (1u << fine_log2) * 4));
  CK(hipMalloc(&ovf, 4));
  CK(hipMalloc(&nhits, 4));
  fill_table<<<8192, 256>>>(slots, nrows * kSlotWords);
  gen_probes<<<8192, 256>>>(a, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](const char* what, double gb, auto&& fn) {
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
      CK(hipMemset(ovf, 0, 4));
      CK(hipEventRecord(e0));
      fn();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    uint32_t o = 0, h = 0;
    CK(hipMemcpy(&o, ovf, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&h, nhits, 4, hipMemcpyDeviceToHost));
    printf("%-34s %8.3f ms  %7.2f ms per 500M probes  %6.2f TB/s of %5.1f GB  overflow=%u hits=%u\n", what, best, best * 500e6 / (double)n, gb / best, gb, o, h);
  };
  const uint32_t n32 = (uint32_t)n;
  uint32_t* d_n;
  CK(hipMalloc(&d_n, 4));
  CK(hipMemcpy(d_n, &n32, 4, hipMemcpyHostToDevice));
  const double recgb = (double)n * 12 / 1e9;
  timed("part 128 (pass 1)", 2 * recgb, [&] {
    CK(hipMemsetAsync(cur1, 0, NB1 * 4));
    part<NB1, 8><<<4096, 256>>>(a, d_n, n32, 4096, kRowBits - 7, b, cap1, cur1, ovf);
  });
  const uint32_t bpi = 64; // blocks per coarse bin in pass 2
  if (nb2 == 64)
    timed("part 64 per coarse bin (pass 2)", 2 * recgb, [&] {
      CK(hipMemsetAsync(cur2, 0, (1u << fine_log2) * 4));
      part<64, 8><<<NB1 * bpi, 256>>>(b, cur1, cap1, bpi, kRowBits - 13, c, cap2, cur2, ovf);
    });
  else if (nb2 == 256)
    timed("part 256 per coarse bin (pass 2)", 2 * recgb, [&] {
      CK(hipMemsetAsync(cur2, 0, (1u << fine_log2) * 4));
      part<256, 8><<<NB1 * bpi, 256>>>(b, cur1, cap1, bpi, kRowBits - 15, c, cap2, cur2, ovf);
    });
  else if (nb2 == 16)
    timed("part 16 per coarse bin (pass 2)", 2 * recgb, [&] {
      CK(hipMemsetAsync(cur2, 0, (1u << fine_log2) * 4));
      part<16, 8><<<NB1 * bpi, 256>>>(b, cur1, cap1, bpi, kRowBits - 11, c, cap2, cur2, ovf);
    });
  else {
    fprintf(stderr, "fine_log2 must be 11, 13 or 15\n");
    return 1;
  }
  const double slotgb = (double)n * 256 / 1e9;
  CK(hipMemset(nhits, 0, 4));
  timed("scan, arrival order", slotgb + recgb, [&] { scan_random<4><<<256 * 16, 256>>>(slots, a, n, nhits); });
  timed("scan, arrival order, 192-B slots", slotgb * 0.75 + recgb, [&] { scan_random<3><<<256 * 16, 256>>>(slots, a, n, nhits); });
  timed("scan, arrival order, 128-B slots", slotgb * 0.5 + recgb, [&] { scan_random<2><<<256 * 16, 256>>>(slots, a, n, nhits); });
  CK(hipMemset(nhits, 0, 4));
  for (uint32_t pb : {PB, PB * 4, PB / 4 ? PB / 4 : 1u}) {
    char nm[64];
    snprintf(nm, sizeof nm, "scan, 2^%d bins, %u blocks/bin", fine_log2, pb);
    timed(nm, slotgb + recgb, [&] { scan_binned<<<(1u << fine_log2) * pb, 256>>>(slots, c, cur2, cap2, pb, nhits); });
  }
  return 0;
}
