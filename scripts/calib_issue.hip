// calib_issue.hip — how many wave instructions does a gfx950 SIMD retire per cycle?
//
// Round-5 review, item 2: docs/design/02 priced every kernel at "one wave instruction per SIMD per four
// cycles whatever its type"; MI355X_MICROARCH.md says a wave64 VALU takes 2 cycles on the SIMD-32 once more
// than one wave feeds it, and that scalar / LDS / memory instructions issue beside it.  This program measures
// it: straight-line instruction streams (inline asm, nothing for the compiler to fold) at 1, 2, 4, 6, 8
// resident waves per SIMD on every CU.  Residency is forced by LDS: a block is 256 threads (one wave per SIMD)
// and asks for floor(160 KiB / W) bytes, so exactly W blocks fit on a CU; the grid is 256 x W blocks and every
// wave reports the SIMD it ran on (HW_ID / XCC_ID) so that the host can check that placement.
//
// Streams (per loop iteration, unrolled):
//   valu      : 64 independent integer VALU (v_and / v_lshlrev / v_add_u32 / v_bcnt over 16 registers)
//   valu_dep  : 64 VALU in ONE dependent chain
//   salu      : 64 independent SALU (s_and / s_lshl / s_add / s_bcnt1 over 8 registers)
//   valu_salu : 64 VALU + 32 SALU interleaved 2:1
//   mix_lds   : 64 VALU + 32 SALU + 6 ds_read_b32 + 2 ds_write_b32 (kr_acc_kernel_t's proportions: 830 : 429 : 88)
//   lds       : 32 ds_read_b32 (addresses conflict-free), drained every 16
//   valu_f64  : 32 v_fma_f64 over 8 registers (the likelihood kernel's arithmetic)
//
// Output: one line per (stream, W): instructions per wave, median wave cycles (s_memtime), in-kernel clock
// (s_memtime / s_memrealtime x 100 MHz), wave instructions per SIMD per cycle = W x instructions / cycles,
// and the same split by type.   Build: hipcc -O2 --offload-arch=gfx950 -o calib_issue calib_issue.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct WaveStamp {
    uint64_t c0, c1;  // s_memtime
    uint64_t r0, r1;  // s_memrealtime (100 MHz)
    uint32_t hw_id, xcc_id;
    uint32_t sink, pad;
};

enum Stream { VALU = 0, VALU_DEP, SALU, VALU_SALU, MIX_LDS, LDS, VALU_F64, NSTREAM };
static const char* kName[NSTREAM] = {"valu", "valu_dep", "salu", "valu_salu", "mix_lds", "lds", "valu_f64"};
// wave instructions of one loop iteration: valu, salu, lds
static const int kCount[NSTREAM][3] = {{64, 0, 0}, {64, 0, 0}, {0, 64, 0}, {64, 32, 0}, {64, 32, 8}, {0, 0, 32}, {32, 0, 0}};

#define V4(a, b, c, d)                                                                              \
    asm volatile("v_and_b32 %0, %0, %4\n\tv_lshlrev_b32 %1, 1, %1\n\tv_add_u32 %2, %2, %4\n\t"       \
                 "v_bcnt_u32_b32 %3, %3, %4"                                                        \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                               \
                 : "v"(kk))
#define V16() V4(v[0], v[1], v[2], v[3]); V4(v[4], v[5], v[6], v[7]); V4(v[8], v[9], v[10], v[11]); V4(v[12], v[13], v[14], v[15])
#define VD4() asm volatile("v_and_b32 %0, %0, %1\n\tv_lshlrev_b32 %0, 1, %0\n\tv_add_u32 %0, %0, %1\n\tv_bcnt_u32_b32 %0, %0, %1" : "+v"(v[0]) : "v"(kk))
#define S4(a, b, c, d)                                                                                   \
    asm volatile("s_and_b32 %0, %0, %4\n\ts_lshl_b32 %1, %1, 1\n\ts_add_u32 %2, %2, %4\n\ts_bcnt1_i32_b32 %3, %3" \
                 : "+s"(a), "+s"(b), "+s"(c), "+s"(d)                                                    \
                 : "s"(sk)                                                                               \
                 : "scc")
// two VALU then one SALU, four times
#define VS(a, b, c, d, e, f, g, h, s0, s1, s2, s3)                                                                   \
    asm volatile("v_and_b32 %0, %0, %12\n\tv_lshlrev_b32 %1, 1, %1\n\ts_and_b32 %8, %8, %13\n\t"                      \
                 "v_add_u32 %2, %2, %12\n\tv_bcnt_u32_b32 %3, %3, %12\n\ts_lshl_b32 %9, %9, 1\n\t"                    \
                 "v_and_b32 %4, %4, %12\n\tv_lshlrev_b32 %5, 1, %5\n\ts_add_u32 %10, %10, %13\n\t"                    \
                 "v_add_u32 %6, %6, %12\n\tv_bcnt_u32_b32 %7, %7, %12\n\ts_bcnt1_i32_b32 %11, %11"                    \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(s0), "+s"(s1), "+s"(s2), \
                   "+s"(s3)                                                                                          \
                 : "v"(kk), "s"(sk)                                                                                  \
                 : "scc")

template <int STREAM>
__global__ __launch_bounds__(256) void calib_kernel(WaveStamp* out, int iters, uint32_t kk_in, uint32_t sk_in) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t v[16];
    uint32_t s[8];
    double f[8];
    uint32_t kk = kk_in | 1u;
    uint32_t sk = sk_in | 1u;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 2654435761u + i * 40503u + kk_in;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = __builtin_amdgcn_readfirstlane(kk_in * (i + 3) + blockIdx.x);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    const double fa = 1.0 - 1e-12 * kk_in, fb = 1e-13 * sk_in;
    // this wave's private, conflict-free LDS row: 64 dwords per wave of the block
    uint32_t laddr = ((threadIdx.x >> 6) * 64 + lane) * 4;
    lds[threadIdx.x] = kk_in + threadIdx.x;
    __syncthreads();

    uint64_t c0, c1, r0, r1;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
        if constexpr (STREAM == VALU) {
            V16(); V16(); V16(); V16();
        } else if constexpr (STREAM == VALU_DEP) {
#pragma unroll
            for (int j = 0; j < 16; ++j) VD4();
        } else if constexpr (STREAM == SALU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { S4(s[0], s[1], s[2], s[3]); S4(s[4], s[5], s[6], s[7]); }
        } else if constexpr (STREAM == VALU_SALU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                VS(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], s[0], s[1], s[2], s[3]);
                VS(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], s[4], s[5], s[6], s[7]);
            }
        } else if constexpr (STREAM == MIX_LDS) {
            uint32_t t0, t1, t2, t3, t4, t5;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:1024" : "=v"(t0), "=v"(t1) : "v"(laddr) : "memory");
            VS(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], s[0], s[1], s[2], s[3]);
            VS(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], s[4], s[5], s[6], s[7]);
            asm volatile("ds_read_b32 %0, %2 offset:2048\n\tds_read_b32 %1, %2 offset:3072" : "=v"(t2), "=v"(t3) : "v"(laddr) : "memory");
            VS(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], s[0], s[1], s[2], s[3]);
            VS(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], s[4], s[5], s[6], s[7]);
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:1024" : "=v"(t4), "=v"(t5) : "v"(laddr) : "memory");
            VS(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], s[0], s[1], s[2], s[3]);
            VS(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], s[4], s[5], s[6], s[7]);
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:1024" ::"v"(laddr), "v"(t0 ^ t2 ^ t4), "v"(t1 ^ t3 ^ t5) : "memory");
            VS(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], s[0], s[1], s[2], s[3]);
            VS(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], s[4], s[5], s[6], s[7]);
        } else if constexpr (STREAM == LDS) {
            uint32_t t[16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    asm volatile("ds_read_b32 %0, %1" : "=v"(t[j]) : "v"(laddr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("" ::"v"(t[j]));
            }
        } else if constexpr (STREAM == VALU_F64) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    __builtin_amdgcn_sched_barrier(0);

    uint32_t sink = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sink ^= v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sink ^= s[i] ^ (uint32_t)__double_as_longlong(f[i]);
    sink ^= lds[(threadIdx.x * 7) & 255];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // every lane folds into the wave's record so that nothing above is dead
    sink = __builtin_amdgcn_readfirstlane(sink) ^ (sink == 0x12345678u ? 1u : 0u);
    if (lane == 0) out[wave] = WaveStamp{c0, c1, r0, r1, hw, xcc, sink, 0};
}

static const char* kOpName[] = {"v_and_b32", "v_xor_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_mov_b32", "v_min_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_bcnt_u32_b32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_lshl_add_u32", "v_mad_u32_u24", "v_mul_lo_u32", "v_alignbit_b32", "v_perm_b32", "v_ffbl_b32", "v_cndmask_b32", "v_cmp_eq_u32(vcc)", "v_cmp_eq_u32(sgpr)", "v_mbcnt_lo", "v_readlane_b32", "v_readfirstlane", "v_mov_dpp_row_shr", "v_add_dpp_row_shr", "v_mov_dpp_bcast31", "ds_bpermute_b32", "ds_swizzle_b32", "v_lshlrev_b64", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f64", "v_log_f32", "s_and_b32", "s_bcnt1_i32_b64", "s_ff1_i32_b64", "s_lshl_b64", "s_mul_i32", "s_nop 0"};
static const int kNumOps = 46;
// 16 instructions of opcode OP in ONE asm statement (nothing of the compiler's in between), over 16 (8) registers
template <int OP>
__device__ __forceinline__ void op16(uint32_t (&v)[16], double (&f)[8], uint32_t (&s)[8], uint64_t (&q)[8], uint32_t kk, uint32_t sk, double fa) {
    if constexpr (OP == 0) asm volatile("v_and_b32 %0, %0, %16\n\tv_and_b32 %1, %1, %16\n\tv_and_b32 %2, %2, %16\n\tv_and_b32 %3, %3, %16\n\tv_and_b32 %4, %4, %16\n\tv_and_b32 %5, %5, %16\n\tv_and_b32 %6, %6, %16\n\tv_and_b32 %7, %7, %16\n\tv_and_b32 %8, %8, %16\n\tv_and_b32 %9, %9, %16\n\tv_and_b32 %10, %10, %16\n\tv_and_b32 %11, %11, %16\n\tv_and_b32 %12, %12, %16\n\tv_and_b32 %13, %13, %16\n\tv_and_b32 %14, %14, %16\n\tv_and_b32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 1) asm volatile("v_xor_b32 %0, %0, %16\n\tv_xor_b32 %1, %1, %16\n\tv_xor_b32 %2, %2, %16\n\tv_xor_b32 %3, %3, %16\n\tv_xor_b32 %4, %4, %16\n\tv_xor_b32 %5, %5, %16\n\tv_xor_b32 %6, %6, %16\n\tv_xor_b32 %7, %7, %16\n\tv_xor_b32 %8, %8, %16\n\tv_xor_b32 %9, %9, %16\n\tv_xor_b32 %10, %10, %16\n\tv_xor_b32 %11, %11, %16\n\tv_xor_b32 %12, %12, %16\n\tv_xor_b32 %13, %13, %16\n\tv_xor_b32 %14, %14, %16\n\tv_xor_b32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 2) asm volatile("v_or_b32 %0, %0, %16\n\tv_or_b32 %1, %1, %16\n\tv_or_b32 %2, %2, %16\n\tv_or_b32 %3, %3, %16\n\tv_or_b32 %4, %4, %16\n\tv_or_b32 %5, %5, %16\n\tv_or_b32 %6, %6, %16\n\tv_or_b32 %7, %7, %16\n\tv_or_b32 %8, %8, %16\n\tv_or_b32 %9, %9, %16\n\tv_or_b32 %10, %10, %16\n\tv_or_b32 %11, %11, %16\n\tv_or_b32 %12, %12, %16\n\tv_or_b32 %13, %13, %16\n\tv_or_b32 %14, %14, %16\n\tv_or_b32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 3) asm volatile("v_add_u32 %0, %0, %16\n\tv_add_u32 %1, %1, %16\n\tv_add_u32 %2, %2, %16\n\tv_add_u32 %3, %3, %16\n\tv_add_u32 %4, %4, %16\n\tv_add_u32 %5, %5, %16\n\tv_add_u32 %6, %6, %16\n\tv_add_u32 %7, %7, %16\n\tv_add_u32 %8, %8, %16\n\tv_add_u32 %9, %9, %16\n\tv_add_u32 %10, %10, %16\n\tv_add_u32 %11, %11, %16\n\tv_add_u32 %12, %12, %16\n\tv_add_u32 %13, %13, %16\n\tv_add_u32 %14, %14, %16\n\tv_add_u32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 4) asm volatile("v_sub_u32 %0, %0, %16\n\tv_sub_u32 %1, %1, %16\n\tv_sub_u32 %2, %2, %16\n\tv_sub_u32 %3, %3, %16\n\tv_sub_u32 %4, %4, %16\n\tv_sub_u32 %5, %5, %16\n\tv_sub_u32 %6, %6, %16\n\tv_sub_u32 %7, %7, %16\n\tv_sub_u32 %8, %8, %16\n\tv_sub_u32 %9, %9, %16\n\tv_sub_u32 %10, %10, %16\n\tv_sub_u32 %11, %11, %16\n\tv_sub_u32 %12, %12, %16\n\tv_sub_u32 %13, %13, %16\n\tv_sub_u32 %14, %14, %16\n\tv_sub_u32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 5) asm volatile("v_mov_b32 %0, %16\n\tv_mov_b32 %1, %16\n\tv_mov_b32 %2, %16\n\tv_mov_b32 %3, %16\n\tv_mov_b32 %4, %16\n\tv_mov_b32 %5, %16\n\tv_mov_b32 %6, %16\n\tv_mov_b32 %7, %16\n\tv_mov_b32 %8, %16\n\tv_mov_b32 %9, %16\n\tv_mov_b32 %10, %16\n\tv_mov_b32 %11, %16\n\tv_mov_b32 %12, %16\n\tv_mov_b32 %13, %16\n\tv_mov_b32 %14, %16\n\tv_mov_b32 %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 6) asm volatile("v_min_u32 %0, %0, %16\n\tv_min_u32 %1, %1, %16\n\tv_min_u32 %2, %2, %16\n\tv_min_u32 %3, %3, %16\n\tv_min_u32 %4, %4, %16\n\tv_min_u32 %5, %5, %16\n\tv_min_u32 %6, %6, %16\n\tv_min_u32 %7, %7, %16\n\tv_min_u32 %8, %8, %16\n\tv_min_u32 %9, %9, %16\n\tv_min_u32 %10, %10, %16\n\tv_min_u32 %11, %11, %16\n\tv_min_u32 %12, %12, %16\n\tv_min_u32 %13, %13, %16\n\tv_min_u32 %14, %14, %16\n\tv_min_u32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 7) asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\tv_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\tv_lshlrev_b32 %8, 1, %8\n\tv_lshlrev_b32 %9, 1, %9\n\tv_lshlrev_b32 %10, 1, %10\n\tv_lshlrev_b32 %11, 1, %11\n\tv_lshlrev_b32 %12, 1, %12\n\tv_lshlrev_b32 %13, 1, %13\n\tv_lshlrev_b32 %14, 1, %14\n\tv_lshlrev_b32 %15, 1, %15" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 8) asm volatile("v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %4, 1, %4\n\tv_lshrrev_b32 %5, 1, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_lshrrev_b32 %7, 1, %7\n\tv_lshrrev_b32 %8, 1, %8\n\tv_lshrrev_b32 %9, 1, %9\n\tv_lshrrev_b32 %10, 1, %10\n\tv_lshrrev_b32 %11, 1, %11\n\tv_lshrrev_b32 %12, 1, %12\n\tv_lshrrev_b32 %13, 1, %13\n\tv_lshrrev_b32 %14, 1, %14\n\tv_lshrrev_b32 %15, 1, %15" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 9) asm volatile("v_bcnt_u32_b32 %0, %0, %16\n\tv_bcnt_u32_b32 %1, %1, %16\n\tv_bcnt_u32_b32 %2, %2, %16\n\tv_bcnt_u32_b32 %3, %3, %16\n\tv_bcnt_u32_b32 %4, %4, %16\n\tv_bcnt_u32_b32 %5, %5, %16\n\tv_bcnt_u32_b32 %6, %6, %16\n\tv_bcnt_u32_b32 %7, %7, %16\n\tv_bcnt_u32_b32 %8, %8, %16\n\tv_bcnt_u32_b32 %9, %9, %16\n\tv_bcnt_u32_b32 %10, %10, %16\n\tv_bcnt_u32_b32 %11, %11, %16\n\tv_bcnt_u32_b32 %12, %12, %16\n\tv_bcnt_u32_b32 %13, %13, %16\n\tv_bcnt_u32_b32 %14, %14, %16\n\tv_bcnt_u32_b32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 10) asm volatile("v_add_f32 %0, %0, %16\n\tv_add_f32 %1, %1, %16\n\tv_add_f32 %2, %2, %16\n\tv_add_f32 %3, %3, %16\n\tv_add_f32 %4, %4, %16\n\tv_add_f32 %5, %5, %16\n\tv_add_f32 %6, %6, %16\n\tv_add_f32 %7, %7, %16\n\tv_add_f32 %8, %8, %16\n\tv_add_f32 %9, %9, %16\n\tv_add_f32 %10, %10, %16\n\tv_add_f32 %11, %11, %16\n\tv_add_f32 %12, %12, %16\n\tv_add_f32 %13, %13, %16\n\tv_add_f32 %14, %14, %16\n\tv_add_f32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 11) asm volatile("v_mul_f32 %0, %0, %16\n\tv_mul_f32 %1, %1, %16\n\tv_mul_f32 %2, %2, %16\n\tv_mul_f32 %3, %3, %16\n\tv_mul_f32 %4, %4, %16\n\tv_mul_f32 %5, %5, %16\n\tv_mul_f32 %6, %6, %16\n\tv_mul_f32 %7, %7, %16\n\tv_mul_f32 %8, %8, %16\n\tv_mul_f32 %9, %9, %16\n\tv_mul_f32 %10, %10, %16\n\tv_mul_f32 %11, %11, %16\n\tv_mul_f32 %12, %12, %16\n\tv_mul_f32 %13, %13, %16\n\tv_mul_f32 %14, %14, %16\n\tv_mul_f32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 12) asm volatile("v_fma_f32 %0, %0, %16, %16\n\tv_fma_f32 %1, %1, %16, %16\n\tv_fma_f32 %2, %2, %16, %16\n\tv_fma_f32 %3, %3, %16, %16\n\tv_fma_f32 %4, %4, %16, %16\n\tv_fma_f32 %5, %5, %16, %16\n\tv_fma_f32 %6, %6, %16, %16\n\tv_fma_f32 %7, %7, %16, %16\n\tv_fma_f32 %8, %8, %16, %16\n\tv_fma_f32 %9, %9, %16, %16\n\tv_fma_f32 %10, %10, %16, %16\n\tv_fma_f32 %11, %11, %16, %16\n\tv_fma_f32 %12, %12, %16, %16\n\tv_fma_f32 %13, %13, %16, %16\n\tv_fma_f32 %14, %14, %16, %16\n\tv_fma_f32 %15, %15, %16, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 13) asm volatile("v_add3_u32 %0, %0, %16, %16\n\tv_add3_u32 %1, %1, %16, %16\n\tv_add3_u32 %2, %2, %16, %16\n\tv_add3_u32 %3, %3, %16, %16\n\tv_add3_u32 %4, %4, %16, %16\n\tv_add3_u32 %5, %5, %16, %16\n\tv_add3_u32 %6, %6, %16, %16\n\tv_add3_u32 %7, %7, %16, %16\n\tv_add3_u32 %8, %8, %16, %16\n\tv_add3_u32 %9, %9, %16, %16\n\tv_add3_u32 %10, %10, %16, %16\n\tv_add3_u32 %11, %11, %16, %16\n\tv_add3_u32 %12, %12, %16, %16\n\tv_add3_u32 %13, %13, %16, %16\n\tv_add3_u32 %14, %14, %16, %16\n\tv_add3_u32 %15, %15, %16, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 1, %16\n\tv_lshl_or_b32 %1, %1, 1, %16\n\tv_lshl_or_b32 %2, %2, 1, %16\n\tv_lshl_or_b32 %3, %3, 1, %16\n\tv_lshl_or_b32 %4, %4, 1, %16\n\tv_lshl_or_b32 %5, %5, 1, %16\n\tv_lshl_or_b32 %6, %6, 1, %16\n\tv_lshl_or_b32 %7, %7, 1, %16\n\tv_lshl_or_b32 %8, %8, 1, %16\n\tv_lshl_or_b32 %9, %9, 1, %16\n\tv_lshl_or_b32 %10, %10, 1, %16\n\tv_lshl_or_b32 %11, %11, 1, %16\n\tv_lshl_or_b32 %12, %12, 1, %16\n\tv_lshl_or_b32 %13, %13, 1, %16\n\tv_lshl_or_b32 %14, %14, 1, %16\n\tv_lshl_or_b32 %15, %15, 1, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 15) asm volatile("v_and_or_b32 %0, %0, %16, %16\n\tv_and_or_b32 %1, %1, %16, %16\n\tv_and_or_b32 %2, %2, %16, %16\n\tv_and_or_b32 %3, %3, %16, %16\n\tv_and_or_b32 %4, %4, %16, %16\n\tv_and_or_b32 %5, %5, %16, %16\n\tv_and_or_b32 %6, %6, %16, %16\n\tv_and_or_b32 %7, %7, %16, %16\n\tv_and_or_b32 %8, %8, %16, %16\n\tv_and_or_b32 %9, %9, %16, %16\n\tv_and_or_b32 %10, %10, %16, %16\n\tv_and_or_b32 %11, %11, %16, %16\n\tv_and_or_b32 %12, %12, %16, %16\n\tv_and_or_b32 %13, %13, %16, %16\n\tv_and_or_b32 %14, %14, %16, %16\n\tv_and_or_b32 %15, %15, %16, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 16) asm volatile("v_bfe_u32 %0, %0, 1, 31\n\tv_bfe_u32 %1, %1, 1, 31\n\tv_bfe_u32 %2, %2, 1, 31\n\tv_bfe_u32 %3, %3, 1, 31\n\tv_bfe_u32 %4, %4, 1, 31\n\tv_bfe_u32 %5, %5, 1, 31\n\tv_bfe_u32 %6, %6, 1, 31\n\tv_bfe_u32 %7, %7, 1, 31\n\tv_bfe_u32 %8, %8, 1, 31\n\tv_bfe_u32 %9, %9, 1, 31\n\tv_bfe_u32 %10, %10, 1, 31\n\tv_bfe_u32 %11, %11, 1, 31\n\tv_bfe_u32 %12, %12, 1, 31\n\tv_bfe_u32 %13, %13, 1, 31\n\tv_bfe_u32 %14, %14, 1, 31\n\tv_bfe_u32 %15, %15, 1, 31" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 17) asm volatile("v_lshl_add_u32 %0, %0, 1, %16\n\tv_lshl_add_u32 %1, %1, 1, %16\n\tv_lshl_add_u32 %2, %2, 1, %16\n\tv_lshl_add_u32 %3, %3, 1, %16\n\tv_lshl_add_u32 %4, %4, 1, %16\n\tv_lshl_add_u32 %5, %5, 1, %16\n\tv_lshl_add_u32 %6, %6, 1, %16\n\tv_lshl_add_u32 %7, %7, 1, %16\n\tv_lshl_add_u32 %8, %8, 1, %16\n\tv_lshl_add_u32 %9, %9, 1, %16\n\tv_lshl_add_u32 %10, %10, 1, %16\n\tv_lshl_add_u32 %11, %11, 1, %16\n\tv_lshl_add_u32 %12, %12, 1, %16\n\tv_lshl_add_u32 %13, %13, 1, %16\n\tv_lshl_add_u32 %14, %14, 1, %16\n\tv_lshl_add_u32 %15, %15, 1, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 18) asm volatile("v_mad_u32_u24 %0, %0, %16, %16\n\tv_mad_u32_u24 %1, %1, %16, %16\n\tv_mad_u32_u24 %2, %2, %16, %16\n\tv_mad_u32_u24 %3, %3, %16, %16\n\tv_mad_u32_u24 %4, %4, %16, %16\n\tv_mad_u32_u24 %5, %5, %16, %16\n\tv_mad_u32_u24 %6, %6, %16, %16\n\tv_mad_u32_u24 %7, %7, %16, %16\n\tv_mad_u32_u24 %8, %8, %16, %16\n\tv_mad_u32_u24 %9, %9, %16, %16\n\tv_mad_u32_u24 %10, %10, %16, %16\n\tv_mad_u32_u24 %11, %11, %16, %16\n\tv_mad_u32_u24 %12, %12, %16, %16\n\tv_mad_u32_u24 %13, %13, %16, %16\n\tv_mad_u32_u24 %14, %14, %16, %16\n\tv_mad_u32_u24 %15, %15, %16, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 19) asm volatile("v_mul_lo_u32 %0, %0, %16\n\tv_mul_lo_u32 %1, %1, %16\n\tv_mul_lo_u32 %2, %2, %16\n\tv_mul_lo_u32 %3, %3, %16\n\tv_mul_lo_u32 %4, %4, %16\n\tv_mul_lo_u32 %5, %5, %16\n\tv_mul_lo_u32 %6, %6, %16\n\tv_mul_lo_u32 %7, %7, %16\n\tv_mul_lo_u32 %8, %8, %16\n\tv_mul_lo_u32 %9, %9, %16\n\tv_mul_lo_u32 %10, %10, %16\n\tv_mul_lo_u32 %11, %11, %16\n\tv_mul_lo_u32 %12, %12, %16\n\tv_mul_lo_u32 %13, %13, %16\n\tv_mul_lo_u32 %14, %14, %16\n\tv_mul_lo_u32 %15, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 20) asm volatile("v_alignbit_b32 %0, %0, %16, 3\n\tv_alignbit_b32 %1, %1, %16, 3\n\tv_alignbit_b32 %2, %2, %16, 3\n\tv_alignbit_b32 %3, %3, %16, 3\n\tv_alignbit_b32 %4, %4, %16, 3\n\tv_alignbit_b32 %5, %5, %16, 3\n\tv_alignbit_b32 %6, %6, %16, 3\n\tv_alignbit_b32 %7, %7, %16, 3\n\tv_alignbit_b32 %8, %8, %16, 3\n\tv_alignbit_b32 %9, %9, %16, 3\n\tv_alignbit_b32 %10, %10, %16, 3\n\tv_alignbit_b32 %11, %11, %16, 3\n\tv_alignbit_b32 %12, %12, %16, 3\n\tv_alignbit_b32 %13, %13, %16, 3\n\tv_alignbit_b32 %14, %14, %16, 3\n\tv_alignbit_b32 %15, %15, %16, 3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 21) asm volatile("v_perm_b32 %0, %0, %16, %16\n\tv_perm_b32 %1, %1, %16, %16\n\tv_perm_b32 %2, %2, %16, %16\n\tv_perm_b32 %3, %3, %16, %16\n\tv_perm_b32 %4, %4, %16, %16\n\tv_perm_b32 %5, %5, %16, %16\n\tv_perm_b32 %6, %6, %16, %16\n\tv_perm_b32 %7, %7, %16, %16\n\tv_perm_b32 %8, %8, %16, %16\n\tv_perm_b32 %9, %9, %16, %16\n\tv_perm_b32 %10, %10, %16, %16\n\tv_perm_b32 %11, %11, %16, %16\n\tv_perm_b32 %12, %12, %16, %16\n\tv_perm_b32 %13, %13, %16, %16\n\tv_perm_b32 %14, %14, %16, %16\n\tv_perm_b32 %15, %15, %16, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 22) asm volatile("v_ffbl_b32 %0, %0\n\tv_ffbl_b32 %1, %1\n\tv_ffbl_b32 %2, %2\n\tv_ffbl_b32 %3, %3\n\tv_ffbl_b32 %4, %4\n\tv_ffbl_b32 %5, %5\n\tv_ffbl_b32 %6, %6\n\tv_ffbl_b32 %7, %7\n\tv_ffbl_b32 %8, %8\n\tv_ffbl_b32 %9, %9\n\tv_ffbl_b32 %10, %10\n\tv_ffbl_b32 %11, %11\n\tv_ffbl_b32 %12, %12\n\tv_ffbl_b32 %13, %13\n\tv_ffbl_b32 %14, %14\n\tv_ffbl_b32 %15, %15" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 23) asm volatile("v_cndmask_b32 %0, %0, %16, vcc\n\tv_cndmask_b32 %1, %1, %16, vcc\n\tv_cndmask_b32 %2, %2, %16, vcc\n\tv_cndmask_b32 %3, %3, %16, vcc\n\tv_cndmask_b32 %4, %4, %16, vcc\n\tv_cndmask_b32 %5, %5, %16, vcc\n\tv_cndmask_b32 %6, %6, %16, vcc\n\tv_cndmask_b32 %7, %7, %16, vcc\n\tv_cndmask_b32 %8, %8, %16, vcc\n\tv_cndmask_b32 %9, %9, %16, vcc\n\tv_cndmask_b32 %10, %10, %16, vcc\n\tv_cndmask_b32 %11, %11, %16, vcc\n\tv_cndmask_b32 %12, %12, %16, vcc\n\tv_cndmask_b32 %13, %13, %16, vcc\n\tv_cndmask_b32 %14, %14, %16, vcc\n\tv_cndmask_b32 %15, %15, %16, vcc" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 24) asm volatile("v_cmp_eq_u32 vcc, %0, %16\n\tv_cmp_eq_u32 vcc, %1, %16\n\tv_cmp_eq_u32 vcc, %2, %16\n\tv_cmp_eq_u32 vcc, %3, %16\n\tv_cmp_eq_u32 vcc, %4, %16\n\tv_cmp_eq_u32 vcc, %5, %16\n\tv_cmp_eq_u32 vcc, %6, %16\n\tv_cmp_eq_u32 vcc, %7, %16\n\tv_cmp_eq_u32 vcc, %8, %16\n\tv_cmp_eq_u32 vcc, %9, %16\n\tv_cmp_eq_u32 vcc, %10, %16\n\tv_cmp_eq_u32 vcc, %11, %16\n\tv_cmp_eq_u32 vcc, %12, %16\n\tv_cmp_eq_u32 vcc, %13, %16\n\tv_cmp_eq_u32 vcc, %14, %16\n\tv_cmp_eq_u32 vcc, %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa) : "vcc");
    if constexpr (OP == 25) asm volatile("v_cmp_eq_u32 s[20:21], %0, %16\n\tv_cmp_eq_u32 s[20:21], %1, %16\n\tv_cmp_eq_u32 s[20:21], %2, %16\n\tv_cmp_eq_u32 s[20:21], %3, %16\n\tv_cmp_eq_u32 s[20:21], %4, %16\n\tv_cmp_eq_u32 s[20:21], %5, %16\n\tv_cmp_eq_u32 s[20:21], %6, %16\n\tv_cmp_eq_u32 s[20:21], %7, %16\n\tv_cmp_eq_u32 s[20:21], %8, %16\n\tv_cmp_eq_u32 s[20:21], %9, %16\n\tv_cmp_eq_u32 s[20:21], %10, %16\n\tv_cmp_eq_u32 s[20:21], %11, %16\n\tv_cmp_eq_u32 s[20:21], %12, %16\n\tv_cmp_eq_u32 s[20:21], %13, %16\n\tv_cmp_eq_u32 s[20:21], %14, %16\n\tv_cmp_eq_u32 s[20:21], %15, %16" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa) : "s20", "s21");
    if constexpr (OP == 26) asm volatile("v_mbcnt_lo_u32_b32 %0, %17, %0\n\tv_mbcnt_lo_u32_b32 %1, %17, %1\n\tv_mbcnt_lo_u32_b32 %2, %17, %2\n\tv_mbcnt_lo_u32_b32 %3, %17, %3\n\tv_mbcnt_lo_u32_b32 %4, %17, %4\n\tv_mbcnt_lo_u32_b32 %5, %17, %5\n\tv_mbcnt_lo_u32_b32 %6, %17, %6\n\tv_mbcnt_lo_u32_b32 %7, %17, %7\n\tv_mbcnt_lo_u32_b32 %8, %17, %8\n\tv_mbcnt_lo_u32_b32 %9, %17, %9\n\tv_mbcnt_lo_u32_b32 %10, %17, %10\n\tv_mbcnt_lo_u32_b32 %11, %17, %11\n\tv_mbcnt_lo_u32_b32 %12, %17, %12\n\tv_mbcnt_lo_u32_b32 %13, %17, %13\n\tv_mbcnt_lo_u32_b32 %14, %17, %14\n\tv_mbcnt_lo_u32_b32 %15, %17, %15" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 27) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s20, %1, 3\n\tv_readlane_b32 s20, %2, 3\n\tv_readlane_b32 s20, %3, 3\n\tv_readlane_b32 s20, %4, 3\n\tv_readlane_b32 s20, %5, 3\n\tv_readlane_b32 s20, %6, 3\n\tv_readlane_b32 s20, %7, 3\n\tv_readlane_b32 s20, %8, 3\n\tv_readlane_b32 s20, %9, 3\n\tv_readlane_b32 s20, %10, 3\n\tv_readlane_b32 s20, %11, 3\n\tv_readlane_b32 s20, %12, 3\n\tv_readlane_b32 s20, %13, 3\n\tv_readlane_b32 s20, %14, 3\n\tv_readlane_b32 s20, %15, 3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa) : "s20");
    if constexpr (OP == 28) asm volatile("v_readfirstlane_b32 s20, %0\n\tv_readfirstlane_b32 s20, %1\n\tv_readfirstlane_b32 s20, %2\n\tv_readfirstlane_b32 s20, %3\n\tv_readfirstlane_b32 s20, %4\n\tv_readfirstlane_b32 s20, %5\n\tv_readfirstlane_b32 s20, %6\n\tv_readfirstlane_b32 s20, %7\n\tv_readfirstlane_b32 s20, %8\n\tv_readfirstlane_b32 s20, %9\n\tv_readfirstlane_b32 s20, %10\n\tv_readfirstlane_b32 s20, %11\n\tv_readfirstlane_b32 s20, %12\n\tv_readfirstlane_b32 s20, %13\n\tv_readfirstlane_b32 s20, %14\n\tv_readfirstlane_b32 s20, %15" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa) : "s20");
    if constexpr (OP == 29) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %12, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %14, %14 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %15, %15 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 30) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %9, %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %10, %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %11, %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %12, %12, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %13, %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %14, %14, %14 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %15, %15, %15 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 31) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %4 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %5 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %6, %6 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %8, %8 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %9, %9 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %10, %10 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %11 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %12, %12 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %13, %13 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %14, %14 row_bcast:31 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %15, %15 row_bcast:31 row_mask:0xf bank_mask:0xf" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 32) asm volatile("ds_bpermute_b32 %0, %16, %0\n\tds_bpermute_b32 %1, %16, %1\n\tds_bpermute_b32 %2, %16, %2\n\tds_bpermute_b32 %3, %16, %3\n\tds_bpermute_b32 %4, %16, %4\n\tds_bpermute_b32 %5, %16, %5\n\tds_bpermute_b32 %6, %16, %6\n\tds_bpermute_b32 %7, %16, %7\n\tds_bpermute_b32 %8, %16, %8\n\tds_bpermute_b32 %9, %16, %9\n\tds_bpermute_b32 %10, %16, %10\n\tds_bpermute_b32 %11, %16, %11\n\tds_bpermute_b32 %12, %16, %12\n\tds_bpermute_b32 %13, %16, %13\n\tds_bpermute_b32 %14, %16, %14\n\tds_bpermute_b32 %15, %16, %15" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa) : "memory");
    if constexpr (OP == 33) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %1, %1 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %2, %2 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %3, %3 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %4, %4 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %5, %5 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %6, %6 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %7, %7 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %8, %8 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %9, %9 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %10, %10 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %11, %11 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %12, %12 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %13, %13 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %14, %14 offset:swizzle(SWAP,1)\n\tds_swizzle_b32 %15, %15 offset:swizzle(SWAP,1)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa) : "memory");
    if constexpr (OP == 34) asm volatile("v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\tv_lshlrev_b64 %4, 1, %4\n\tv_lshlrev_b64 %5, 1, %5\n\tv_lshlrev_b64 %6, 1, %6\n\tv_lshlrev_b64 %7, 1, %7\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\tv_lshlrev_b64 %4, 1, %4\n\tv_lshlrev_b64 %5, 1, %5\n\tv_lshlrev_b64 %6, 1, %6\n\tv_lshlrev_b64 %7, 1, %7" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 35) asm volatile("v_add_f64 %0, %0, %10\n\tv_add_f64 %1, %1, %10\n\tv_add_f64 %2, %2, %10\n\tv_add_f64 %3, %3, %10\n\tv_add_f64 %4, %4, %10\n\tv_add_f64 %5, %5, %10\n\tv_add_f64 %6, %6, %10\n\tv_add_f64 %7, %7, %10\n\tv_add_f64 %0, %0, %10\n\tv_add_f64 %1, %1, %10\n\tv_add_f64 %2, %2, %10\n\tv_add_f64 %3, %3, %10\n\tv_add_f64 %4, %4, %10\n\tv_add_f64 %5, %5, %10\n\tv_add_f64 %6, %6, %10\n\tv_add_f64 %7, %7, %10" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 36) asm volatile("v_mul_f64 %0, %0, %10\n\tv_mul_f64 %1, %1, %10\n\tv_mul_f64 %2, %2, %10\n\tv_mul_f64 %3, %3, %10\n\tv_mul_f64 %4, %4, %10\n\tv_mul_f64 %5, %5, %10\n\tv_mul_f64 %6, %6, %10\n\tv_mul_f64 %7, %7, %10\n\tv_mul_f64 %0, %0, %10\n\tv_mul_f64 %1, %1, %10\n\tv_mul_f64 %2, %2, %10\n\tv_mul_f64 %3, %3, %10\n\tv_mul_f64 %4, %4, %10\n\tv_mul_f64 %5, %5, %10\n\tv_mul_f64 %6, %6, %10\n\tv_mul_f64 %7, %7, %10" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 37) asm volatile("v_fma_f64 %0, %0, %10, %10\n\tv_fma_f64 %1, %1, %10, %10\n\tv_fma_f64 %2, %2, %10, %10\n\tv_fma_f64 %3, %3, %10, %10\n\tv_fma_f64 %4, %4, %10, %10\n\tv_fma_f64 %5, %5, %10, %10\n\tv_fma_f64 %6, %6, %10, %10\n\tv_fma_f64 %7, %7, %10, %10\n\tv_fma_f64 %0, %0, %10, %10\n\tv_fma_f64 %1, %1, %10, %10\n\tv_fma_f64 %2, %2, %10, %10\n\tv_fma_f64 %3, %3, %10, %10\n\tv_fma_f64 %4, %4, %10, %10\n\tv_fma_f64 %5, %5, %10, %10\n\tv_fma_f64 %6, %6, %10, %10\n\tv_fma_f64 %7, %7, %10, %10" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 38) asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\tv_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7\n\tv_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\tv_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 39) asm volatile("v_log_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_log_f32 %2, %2\n\tv_log_f32 %3, %3\n\tv_log_f32 %4, %4\n\tv_log_f32 %5, %5\n\tv_log_f32 %6, %6\n\tv_log_f32 %7, %7\n\tv_log_f32 %8, %8\n\tv_log_f32 %9, %9\n\tv_log_f32 %10, %10\n\tv_log_f32 %11, %11\n\tv_log_f32 %12, %12\n\tv_log_f32 %13, %13\n\tv_log_f32 %14, %14\n\tv_log_f32 %15, %15" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 40) asm volatile("s_and_b32 %0, %0, %9\n\ts_and_b32 %1, %1, %9\n\ts_and_b32 %2, %2, %9\n\ts_and_b32 %3, %3, %9\n\ts_and_b32 %4, %4, %9\n\ts_and_b32 %5, %5, %9\n\ts_and_b32 %6, %6, %9\n\ts_and_b32 %7, %7, %9\n\ts_and_b32 %0, %0, %9\n\ts_and_b32 %1, %1, %9\n\ts_and_b32 %2, %2, %9\n\ts_and_b32 %3, %3, %9\n\ts_and_b32 %4, %4, %9\n\ts_and_b32 %5, %5, %9\n\ts_and_b32 %6, %6, %9\n\ts_and_b32 %7, %7, %9" : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7]) : "v"(kk), "s"(sk), "v"(fa) : "scc");
    if constexpr (OP == 41) asm volatile("s_bcnt1_i32_b64 s20, %0\n\ts_bcnt1_i32_b64 s20, %1\n\ts_bcnt1_i32_b64 s20, %2\n\ts_bcnt1_i32_b64 s20, %3\n\ts_bcnt1_i32_b64 s20, %4\n\ts_bcnt1_i32_b64 s20, %5\n\ts_bcnt1_i32_b64 s20, %6\n\ts_bcnt1_i32_b64 s20, %7\n\ts_bcnt1_i32_b64 s20, %0\n\ts_bcnt1_i32_b64 s20, %1\n\ts_bcnt1_i32_b64 s20, %2\n\ts_bcnt1_i32_b64 s20, %3\n\ts_bcnt1_i32_b64 s20, %4\n\ts_bcnt1_i32_b64 s20, %5\n\ts_bcnt1_i32_b64 s20, %6\n\ts_bcnt1_i32_b64 s20, %7" : "+s"(q[0]), "+s"(q[1]), "+s"(q[2]), "+s"(q[3]), "+s"(q[4]), "+s"(q[5]), "+s"(q[6]), "+s"(q[7]) : "v"(kk), "s"(sk), "v"(fa) : "scc", "s20");
    if constexpr (OP == 42) asm volatile("s_ff1_i32_b64 s20, %0\n\ts_ff1_i32_b64 s20, %1\n\ts_ff1_i32_b64 s20, %2\n\ts_ff1_i32_b64 s20, %3\n\ts_ff1_i32_b64 s20, %4\n\ts_ff1_i32_b64 s20, %5\n\ts_ff1_i32_b64 s20, %6\n\ts_ff1_i32_b64 s20, %7\n\ts_ff1_i32_b64 s20, %0\n\ts_ff1_i32_b64 s20, %1\n\ts_ff1_i32_b64 s20, %2\n\ts_ff1_i32_b64 s20, %3\n\ts_ff1_i32_b64 s20, %4\n\ts_ff1_i32_b64 s20, %5\n\ts_ff1_i32_b64 s20, %6\n\ts_ff1_i32_b64 s20, %7" : "+s"(q[0]), "+s"(q[1]), "+s"(q[2]), "+s"(q[3]), "+s"(q[4]), "+s"(q[5]), "+s"(q[6]), "+s"(q[7]) : "v"(kk), "s"(sk), "v"(fa) : "s20");
    if constexpr (OP == 43) asm volatile("s_lshl_b64 %0, %0, 1\n\ts_lshl_b64 %1, %1, 1\n\ts_lshl_b64 %2, %2, 1\n\ts_lshl_b64 %3, %3, 1\n\ts_lshl_b64 %4, %4, 1\n\ts_lshl_b64 %5, %5, 1\n\ts_lshl_b64 %6, %6, 1\n\ts_lshl_b64 %7, %7, 1\n\ts_lshl_b64 %0, %0, 1\n\ts_lshl_b64 %1, %1, 1\n\ts_lshl_b64 %2, %2, 1\n\ts_lshl_b64 %3, %3, 1\n\ts_lshl_b64 %4, %4, 1\n\ts_lshl_b64 %5, %5, 1\n\ts_lshl_b64 %6, %6, 1\n\ts_lshl_b64 %7, %7, 1" : "+s"(q[0]), "+s"(q[1]), "+s"(q[2]), "+s"(q[3]), "+s"(q[4]), "+s"(q[5]), "+s"(q[6]), "+s"(q[7]) : "v"(kk), "s"(sk), "v"(fa) : "scc");
    if constexpr (OP == 44) asm volatile("s_mul_i32 %0, %0, %9\n\ts_mul_i32 %1, %1, %9\n\ts_mul_i32 %2, %2, %9\n\ts_mul_i32 %3, %3, %9\n\ts_mul_i32 %4, %4, %9\n\ts_mul_i32 %5, %5, %9\n\ts_mul_i32 %6, %6, %9\n\ts_mul_i32 %7, %7, %9\n\ts_mul_i32 %0, %0, %9\n\ts_mul_i32 %1, %1, %9\n\ts_mul_i32 %2, %2, %9\n\ts_mul_i32 %3, %3, %9\n\ts_mul_i32 %4, %4, %9\n\ts_mul_i32 %5, %5, %9\n\ts_mul_i32 %6, %6, %9\n\ts_mul_i32 %7, %7, %9" : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7]) : "v"(kk), "s"(sk), "v"(fa));
    if constexpr (OP == 45) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7]) : "v"(kk), "s"(sk), "v"(fa));
}

template <int OP>
__global__ __launch_bounds__(256) void op_kernel(WaveStamp* out, int iters, uint32_t kk_in, uint32_t sk_in) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t v[16], s[8];
    uint64_t q[8];
    double f[8];
    uint32_t kk = (kk_in | 1u) & 0xfcu;  // (a small multiple of 4: valid as a ds_bpermute address and a v_perm selector)
    uint32_t sk = sk_in | 1u;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 2654435761u + i * 40503u + kk_in;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[i] = __builtin_amdgcn_readfirstlane(kk_in * (i + 3) + blockIdx.x);
        q[i] = ((uint64_t)s[i] << 32) | (s[i] * 77u);
        f[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    }
    const double fa = 1.0 - 1e-12 * kk_in;
    lds[threadIdx.x] = kk_in;
    __syncthreads();
    uint64_t c0, c1, r0, r1;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
        op16<OP>(v, f, s, q, kk, sk, fa);
        op16<OP>(v, f, s, q, kk, sk, fa);
        op16<OP>(v, f, s, q, kk, sk, fa);
        op16<OP>(v, f, s, q, kk, sk, fa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    uint32_t sink = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sink ^= v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sink ^= s[i] ^ (uint32_t)q[i] ^ (uint32_t)__double_as_longlong(f[i]);
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    sink = __builtin_amdgcn_readfirstlane(sink) ^ (sink == 0x12345678u ? 1u : 0u);
    if (lane == 0) out[wave] = WaveStamp{c0, c1, r0, r1, hw, xcc, sink, 0};
}

template <int OP>
static void run_op(int W, int iters, int ncu, WaveStamp* d_out, std::vector<WaveStamp>& h, double* rate_out) {
    const int lds_bytes = (160 * 1024 / W) & ~255;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&op_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const int blocks = ncu * W;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    op_kernel<OP><<<blocks, 256, lds_bytes>>>(d_out, iters, 12345u, 777u);
    CK(hipEventRecord(e0));
    op_kernel<OP><<<blocks, 256, lds_bytes>>>(d_out, iters, 99991u, 4242u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = blocks * 4;
    h.resize(nw);
    CK(hipMemcpy(h.data(), d_out, sizeof(WaveStamp) * nw, hipMemcpyDeviceToHost));
    std::vector<double> clk(nw);
    for (int i = 0; i < nw; ++i) clk[i] = double(h[i].c1 - h[i].c0) / double(h[i].r1 - h[i].r0) * 100e6;
    std::sort(clk.begin(), clk.end());
    const double ghz = clk[nw / 2] / 1e9;
    // the SIMD's rate over the whole launch: W waves x 64 instructions x iterations / (launch time x in-kernel clock)
    *rate_out = double(W) * 64.0 * iters / (ms * 1e-3 * ghz * 1e9);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

template <int OP>
static void run_ops_from(int iters, int ncu, WaveStamp* d_out, std::vector<WaveStamp>& h) {
    if constexpr (OP < kNumOps) {
        double r1, r2, r4, r8;
        run_op<OP>(1, iters, ncu, d_out, h, &r1);
        run_op<OP>(2, iters, ncu, d_out, h, &r2);
        run_op<OP>(4, iters, ncu, d_out, h, &r4);
        run_op<OP>(8, iters, ncu, d_out, h, &r8);
        printf("op %-20s per SIMD per cycle at W=1/2/4/8: %.3f %.3f %.3f %.3f   cycles per instruction at W=1/8: %.2f %.2f\n", kOpName[OP], r1, r2, r4, r8,
               1.0 / r1, 1.0 / r8);
        fflush(stdout);
        run_ops_from<OP + 1>(iters, ncu, d_out, h);
    }
}

template <int STREAM>
static void run_one(int W, int iters, int ncu, WaveStamp* d_out, std::vector<WaveStamp>& h) {
    const int lds_bytes = (160 * 1024 / W) & ~255;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&calib_kernel<STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           lds_bytes));
    const int blocks = ncu * W;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // warm up (clocks), then the measured launch
    for (int rep = 0; rep < 3; ++rep) calib_kernel<STREAM><<<blocks, 256, lds_bytes>>>(d_out, iters, 12345u + rep, 777u);
    CK(hipEventRecord(e0));
    calib_kernel<STREAM><<<blocks, 256, lds_bytes>>>(d_out, iters, 99991u, 4242u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = blocks * 4;
    h.resize(nw);
    CK(hipMemcpy(h.data(), d_out, sizeof(WaveStamp) * nw, hipMemcpyDeviceToHost));
    std::vector<double> cyc(nw), clk(nw);
    std::map<uint64_t, int> per_simd;
    for (int i = 0; i < nw; ++i) {
        cyc[i] = double(h[i].c1 - h[i].c0);
        clk[i] = double(h[i].c1 - h[i].c0) / double(h[i].r1 - h[i].r0) * 100e6;
        // HW_ID (gfx9): simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]
        per_simd[(uint64_t(h[i].xcc_id & 0xf) << 32) | (h[i].hw_id & 0xff30u)]++;
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    int lo = 1 << 30, hi = 0;
    for (auto& kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
    const double med = cyc[nw / 2], p95 = cyc[size_t(nw * 0.95)];
    const double nv = double(kCount[STREAM][0]) * iters, ns = double(kCount[STREAM][1]) * iters,
                 nl = double(kCount[STREAM][2]) * iters;
    const double per_cycle = W * (nv + ns + nl) / med;
    // the SIMD's rate over the whole launch (the waves of a SIMD do not run in lockstep: the older ones are served first and
    // leave early, so a wave's own cycles say little): W waves x instructions / (launch time x in-kernel clock)
    const double launch_cycles = ms * 1e-3 * clk[nw / 2];
    (void)per_cycle;
    printf("%-9s W=%d  simds=%zu waves/simd=%d..%d  inst/wave=%.0f  wave cycles med=%.0f p95=%.0f  clock=%.3f GHz  launch=%.3f ms"
           "  | per SIMD per cycle over the launch: all %.3f  valu %.3f  salu %.3f  lds %.3f\n",
           kName[STREAM], W, per_simd.size(), lo, hi, nv + ns + nl, med, p95, clk[nw / 2] / 1e9, ms,
           W * (nv + ns + nl) / launch_cycles, W * nv / launch_cycles, W * ns / launch_cycles, W * nl / launch_cycles);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("# %s, %d CUs, clockRate %d kHz; %d loop iterations per wave\n", prop.gcnArchName, ncu, prop.clockRate, iters);
    WaveStamp* d_out = nullptr;
    CK(hipMalloc(&d_out, sizeof(WaveStamp) * size_t(ncu) * 8 * 4));
    std::vector<WaveStamp> h;
    const int Ws[] = {1, 2, 3, 4, 6, 8};
    for (int W : Ws) run_one<VALU>(W, iters, ncu, d_out, h);
    for (int W : Ws) run_one<VALU_DEP>(W, iters, ncu, d_out, h);
    for (int W : Ws) run_one<SALU>(W, iters, ncu, d_out, h);
    for (int W : Ws) run_one<VALU_SALU>(W, iters, ncu, d_out, h);
    for (int W : Ws) run_one<MIX_LDS>(W, iters, ncu, d_out, h);
    for (int W : Ws) run_one<LDS>(W, iters, ncu, d_out, h);
    for (int W : Ws) run_one<VALU_F64>(W, iters, ncu, d_out, h);
    run_ops_from<0>(iters / 2, ncu, d_out, h);
    CK(hipFree(d_out));
    return 0;
}
