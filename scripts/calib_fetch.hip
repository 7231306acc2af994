// Calibration of rocprofv3's FETCH_SIZE on gfx950 for THIS kernel's access shape
// (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count in your own access pattern"):
// groups of 8 lanes read one random, 128-byte-aligned 128-byte block as 8 x dwordx4, from a
// buffer far larger than the Infinity Cache, each block exactly once.  Known bytes = nblocks * 128.
// Second kernel: one lane reads one random 8-byte word (the bucket-descriptor shape).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void gather128(const uint4* buf, const uint32_t* blk, uint64_t nblocks, uint32_t* sink)
{
  uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 3;
  uint32_t sub = threadIdx.x & 7;
  uint32_t acc = 0;
  for (; g < nblocks; g += ((uint64_t)gridDim.x * blockDim.x) >> 3) {
    uint4 v = buf[(uint64_t)blk[g] * 8 + sub];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// The scan kernel's shape: 4 lanes x dwordx4 = one 64-byte piece; a wave instruction reads 16 random
// pieces; `nj` such instructions read the following 64-byte pieces of the same 16 buckets.
__global__ void gather64(const uint4* buf, const uint32_t* blk, uint64_t nblocks, uint32_t nj, uint32_t* sink)
{
  uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 2;
  uint32_t sub = threadIdx.x & 3;
  uint32_t acc = 0;
  for (; g < nblocks; g += ((uint64_t)gridDim.x * blockDim.x) >> 2) {
    if (blk[g] & 1u) continue;                        // even blocks only: each owns 256 bytes
    const uint4* b = buf + (uint64_t)blk[g] * 8 + sub; // random 256-byte regions, each used once
    uint4 v0 = b[0], v1 = make_uint4(0, 0, 0, 0), v2 = make_uint4(0, 0, 0, 0);
    if (nj > 1) v1 = b[4];
    if (nj > 2) v2 = b[8];
    acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w ^ v1.x ^ v1.w ^ v2.x ^ v2.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void gather8(const uint64_t* buf, const uint32_t* idx, uint64_t n, uint32_t* sink)
{
  uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; g < n; g += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v = buf[(uint64_t)idx[g] * 16]; // one word per 128-byte line
    acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// 2^25 random 8-byte gathers confined to (mask+1) lines of 128 bytes
__global__ void gather8m(const uint64_t* buf, const uint32_t* idx, uint64_t n, uint32_t mask, uint32_t* sink)
{
  uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; g < n; g += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v = buf[(uint64_t)(idx[g] & mask) * 16];
    acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
  const uint64_t nblocks = 1ull << 25; // 4 GiB of 128-byte blocks
  uint4* buf;
  uint32_t *blk, *sink;
  hipMalloc(&buf, nblocks * 128);
  hipMemset(buf, 1, nblocks * 128);
  hipMalloc(&blk, nblocks * 4);
  hipMalloc(&sink, 4);
  std::vector<uint32_t> perm(nblocks);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  hipMemcpy(blk, perm.data(), nblocks * 4, hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gather128, dim3(8192), dim3(256), 0, 0, buf, blk, nblocks, sink);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gather8, dim3(8192), dim3(256), 0, 0, (const uint64_t*)buf, blk, nblocks, sink);
  // 2^24 random 256-byte slots of the same buffer: 64 / 128 / 192 bytes of each
  for (uint32_t nj = 1; nj <= 3; ++nj)
    hipLaunchKernelGGL(gather64, dim3(8192), dim3(256), 0, 0, buf, blk, nblocks, nj, sink);
  // descriptor-table shapes: 8-byte gathers from a 128 MiB and a 256 MiB region (Infinity Cache = 256 MiB)
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(gather8m, dim3(8192), dim3(256), 0, 0, (const uint64_t*)buf, blk, nblocks, (1u << 20) - 1u, sink);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(gather8m, dim3(8192), dim3(256), 0, 0, (const uint64_t*)buf, blk, nblocks, (1u << 21) - 1u, sink);
  hipDeviceSynchronize();
  printf("gather64 x nj: known bytes per launch: nj * 64 * %llu (+ index)\n", (unsigned long long)(nblocks / 2));
  printf("gather128 known bytes per launch: %llu (+ %llu index bytes)\n", (unsigned long long)(nblocks * 128), (unsigned long long)(nblocks * 4));
  printf("gather8   known useful bytes per launch: %llu, lines touched %llu x 128 B (+ index)\n", (unsigned long long)(nblocks * 8), (unsigned long long)nblocks);
  return 0;
}
