// kr_devutil.h — small device/host helpers shared by kr_device.hip and kr_minimizer.hip.
#ifndef KR_DEVUTIL_H
#define KR_DEVUTIL_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace {

// Software PEXT for a fixed 32-bit mask (Hacker's Delight 7-4 "compress"): five precomputed
// move masks; x86's _pext_u64 in LSHF::compute_hash / drop_ppos_lr (src/lshf.cpp:62-69) becomes
// 5 x (and, xor, shift, or) with wave-uniform constants held in SGPRs.
struct PextMask {
  uint32_t m;
  uint32_t mv[5];
};

__device__ __forceinline__ uint32_t pext32(uint32_t x, const PextMask& pm)
{
  x &= pm.m;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    uint32_t t = x & pm.mv[i];
    x = (x ^ t) | (t >> (1 << i));
  }
  return x;
}

// 16 -> 32 bit spread: bit j of v moves to bit 2j
__device__ __forceinline__ uint32_t spread16(uint32_t v)
{
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// seq_nt4_table (src/common.cpp:10-14): 0..3 for ACGT/acgt, 4 otherwise
__device__ __forceinline__ uint32_t base_code(uint32_t c)
{
  uint32_t u = c & 0xDFu; // fold case
  uint32_t code = 4;
  code = (u == 'A') ? 0u : code;
  code = (u == 'C') ? 1u : code;
  code = (u == 'G') ? 2u : code;
  code = (u == 'T') ? 3u : code;
  return (c & 0x80u) ? 4u : code;
}

inline PextMask make_pext(const std::vector<uint8_t>& positions)
{
  PextMask pm;
  uint32_t m = 0;
  for (uint8_t p : positions) m |= 1u << p;
  pm.m = m;
  uint32_t mk = ~m << 1; // count the 0s to the right of every bit
  for (int i = 0; i < 5; ++i) {
    uint32_t mp = mk ^ (mk << 1); // parallel suffix
    mp ^= mp << 2;
    mp ^= mp << 4;
    mp ^= mp << 8;
    mp ^= mp << 16;
    uint32_t mv = mp & m; // bits to move by 2^i
    pm.mv[i] = mv;
    m = (m ^ mv) | (mv >> (1 << i));
    mk &= ~mp;
  }
  return pm;
}


} // namespace

#endif
