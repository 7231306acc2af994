// kr_common.h — internal declarations shared by the host and device translation units.
#ifndef KR_COMMON_H
#define KR_COMMON_H

#include "krepp_amd.h"

#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace kr {

// per-thread error message behind kr_last_error()
int fail(int code, const std::string& msg);
void clear_error();

struct TreeNode {
  std::string label;   // as written in the Newick / reflist ("" if unlabelled)
  double blen;         // NaN if absent
  uint32_t parent;     // se of the parent, 0 for the root
  uint8_t kind;        // 1 leaf, 2 internal
};

// Tree with the reference's numbering: se = 1-based post-order index assigned while
// parsing (src/phytree.cpp:168-170,211-213); nodes[0] is the null node.
struct HostTree {
  std::vector<TreeNode> nodes;
  uint32_t nnodes() const { return (uint32_t)nodes.size() - 1; }
  std::string name(uint32_t se) const; // Node::get_name, src/phytree.hpp:134-145
};

bool parse_newick(const std::string& text, HostTree& out, std::string& err);
void balanced_tree(const std::vector<std::string>& names, HostTree& out);

// f(0) ... f(n-1) on a process-wide pool of sleeping threads (condition variables, no spinning: several callers
// -- the CLI's workers -- share the cores; OpenMP teams that busy-wait between regions cost 2x here).  The caller
// runs pieces too; returns when all are done.
void parallel_for(int n, const std::function<void(int)>& f);
int parallel_width(); // threads worth asking for (pool size + 1)
unsigned usable_cpus(); // hardware threads this process may run on, capped by the cgroup CPU quota

// MurmurHash3_x86_32 based 64-bit name hash (src/record.hpp:26-36)
uint64_t leaf_name_hash(const std::string& name);
uint64_t rehash64(uint64_t sh); // Subset::rehash, src/record.hpp:37-47

// ---- index build helpers shared by kr_build.cpp (CPU) and kr_minimizer.hip (GPU) ----
struct BuildCfg {
  uint32_t k, w, h, m, r;
  bool frac;
};
struct LshPositions {
  uint32_t k = 0, h = 0;
  std::vector<uint8_t> ppos, npos; // descending / ascending
  std::vector<uint8_t> pasc;       // ppos ascending
  uint32_t rix(uint64_t bp) const; // closed forms of LSHF::compute_hash / drop_ppos_lr
  uint32_t enc32(uint64_t bp) const;
};
bool make_positions(uint32_t k, uint32_t h, const uint8_t* ppos, LshPositions& out);
// HyperLogLog(12) estimate from 4096 registers (src/hyperloglog.hpp:113-135)
double hll12_estimate(const uint8_t* reg);
// row of a k-mer in a partial library, or -1 (src/rqseq.cpp:125-128)
inline int64_t build_row(uint32_t rix, const BuildCfg& c)
{
  uint32_t res = rix % c.m;
  if (c.frac ? res <= c.r : res == c.r) return c.frac ? (int64_t)(rix / c.m) * (c.r + 1) + res : (int64_t)(rix / c.m);
  return -1;
}
#if defined(__HIPCC__)
#define KR_HD __host__ __device__
#else
#define KR_HD
#endif
// round-half-even(|v| * 10^5) of the EXACT binary value of v, for 0 <= |v| < 1000: what printf("%.5f") prints, as an integer
// (glibc rounds the exact value in the current rounding mode: nearest, ties to even).  sc = v * 1e5 rounded to double, err = the
// rounding's exact residual (one fma: the error of a product is a double); fl = floor(sc) and fr = sc - fl are exact; since fr is a
// multiple of ulp(sc) and |err| <= ulp(sc) / 2, the sign of (fr - 1/2) decides unless it is zero, then err's, then parity.  Shared
// by the host formatter and the device one (kr_text_write_kernel); checked against snprintf on random values, on every j / 2^q tie
// and next to them (tests/test_host.py).  Returns false if v is outside the range (NaN, negative, >= 1000): the caller uses printf.
KR_HD inline bool fixed5_exact(double v, uint32_t* n_out)
{
  if (!(v >= 0.0 && v < 1000.0)) return false;
  const double sc = v * 100000.0;
  const double err = fma(v, 100000.0, -sc);
  const double fl = floor(sc), fr = sc - fl;
  uint32_t n = (uint32_t)fl;
  const double t = fr - 0.5; // exact: fr in [0, 1) is a multiple of ulp(sc) <= 2^-26, and so is 1/2
  if (t > 0.0 || (t == 0.0 && (err > 0.0 || (err == 0.0 && (n & 1u))))) ++n;
  if (n >= 100000000u) return false; // (v in [999.999995, 1000) rounds to "1000.00000": ten bytes, one more than the callers lay out)
  *n_out = n;
  return true;
}
// the digits of n = round(v * 10^5) as "%.5f" text: integer part, '.', five decimals; returns the length
KR_HD inline uint32_t fixed5_digits(uint32_t n, char* out)
{
  uint32_t ip = n / 100000u, fp = n - ip * 100000u;
  char tmp[4];
  uint32_t k = 0, o = 0;
  do {
    tmp[k++] = (char)('0' + ip % 10u);
    ip /= 10u;
  } while (ip);
  while (k) out[o++] = tmp[--k];
  out[o++] = '.';
  for (int q = 4; q >= 0; --q) {
    out[o + (uint32_t)q] = (char)('0' + fp % 10u);
    fp /= 10u;
  }
  return o + 5u;
}
// "%.5f" of a value without going through printf: scale, round half away in integers.  printf rounds the
// exact binary value; the two agree unless |v| * 1e5 lies within rounding noise of a tie (or v is out of the
// fast range, NaN, infinite), where printf decides.  Checked against printf on 20 M random values.
inline size_t fmt_fixed5(double v, char* out)
{
  const bool neg = std::signbit(v);
  const double d = neg ? -v : v;
  if (d >= 0.0 && d < 1000.0) {
    const double sc = d * 100000.0, fl = std::floor(sc), fr = sc - fl;
    if (std::fabs(fr - 0.5) > 1e-6) {
      const uint64_t n = (uint64_t)fl + (fr > 0.5 ? 1u : 0u);
      uint64_t ip = n / 100000u, fp = n % 100000u;
      char tmp[8];
      size_t k = 0, o = 0;
      if (neg) out[o++] = '-';
      do {
        tmp[k++] = (char)('0' + ip % 10u);
        ip /= 10u;
      } while (ip);
      while (k) out[o++] = tmp[--k];
      out[o++] = '.';
      for (int q = 4; q >= 0; --q) {
        out[o + q] = (char)('0' + fp % 10u);
        fp /= 10u;
      }
      return o + 5;
    }
  }
  return (size_t)snprintf(out, 64, "%.5f", v);
}

inline uint64_t fmix64(uint64_t v)
{ // xur64_hash, src/common.hpp:147-155
  v ^= v >> 33;
  v *= 0xff51afd7ed558ccdull;
  v ^= v >> 33;
  v *= 0xc4ceb9fe1a85ec53ull;
  v ^= v >> 33;
  return v;
}
// One contig on the CPU (RSeq::extract_mers): appends keys, adds the contig's HLL estimates.
void extract_contig_cpu(const uint8_t* seq, uint64_t len, const BuildCfg& c, const LshPositions& lsh,
                        std::vector<uint64_t>& keys, double& n1, double& n2);
// What the ring buffer of RSeq::extract_mers yields at the END of a contig whose last run of valid
// bases is shorter than w (src/rqseq.cpp:108-116: stale or zero slots take part): true if a
// minimizer is emitted there; `x` is its k-mer code, `z` its hash.
bool contig_end_special(const uint8_t* seq, uint64_t len, const BuildCfg& c, uint64_t& x, uint64_t& z);

// ---- `place` back end on the device (kr_device.hip), driven by kr_place.cpp ----
// The placement tree as flat arrays over its nodes q = 1..pn (q = post-order number, so the subtree of q is the
// range [lo[q], q]); idx_to_pt maps an index colour id (leaf se of the index tree) to its placement-tree node.
struct PlaceTreeArrays {
  uint32_t pn = 0, nidx = 0;
  const uint32_t* parent = nullptr; // [pn + 1]
  const uint32_t* eff = nullptr;    // [pn + 1] Node::get_eff_nchildren
  const uint8_t* elig = nullptr;    // [pn + 1] nchildren == eff_nchildren && nchildren != 1 (src/query.cpp:270)
  const uint32_t* lo = nullptr;     // [pn + 1] smallest node number in the subtree of q
  const uint32_t* idx_to_pt = nullptr; // [nidx + 1]
  const uint32_t* depth = nullptr;  // [pn + 1] number of ancestors of q
  // for the rows as text on the device (round 6; all null: no device text)
  const double* blen = nullptr;       // [pn + 1] branch length above q, NaN = none
  const uint32_t* card = nullptr;     // [pn + 1] leaves below q
  const char* labels = nullptr;       // node labels back to back
  const uint32_t* label_off = nullptr; // [pn + 2]
};
struct PlaceDeviceResult { // page-locked host arrays owned by the stream, valid until its next place call / batch
  uint32_t nreads = 0;
  const uint32_t* rd_c0 = nullptr;   // [nreads] first candidate slot of the read
  const uint32_t* rd_info = nullptr; // [nreads] bit 31: reported, bit 30: single placement, low 30 bits: candidates
  const uint32_t* c_se = nullptr;    // per KEPT candidate (chi-square below --chisq, not the root; a read's in emission order):
                                     // placement-tree node, d_llh, v_llh, chi-square
  const double *c_d = nullptr, *c_v = nullptr, *c_chisq = nullptr;
  uint64_t kept = 0;                 // kept candidates of this range (the next range's go behind them in the host arrays)
  bool overflow = false;             // the device ran out of candidate slots: take the host path for the batch
  uint32_t heavy_reads = 0;          // reads beyond the LDS arrays of kr_place_kernel, done by its second launch (upper bound)
  // rows as text written on the device (place_device_text_begin): the range's text in page-locked memory; text == nullptr: the
  // range has to be formatted from the candidates above (device text off, or a flag came up: `text_flags`)
  const char* text = nullptr;
  uint64_t text_len = 0;
  uint64_t text_flags = 0;
};
} // namespace kr
struct kr_stream;
namespace kr {
// For the batch last submitted on `s` (KR_TAP_ACCS) and waited for: ancestor accumulation (Minfo::add,
// src/query.hpp:139-152), candidate listing (src/query.cpp:268-281), Brent on the internal candidates and the
// chi-square of every candidate against the read's closest leaf, all on the device.
int place_on_device(kr_stream* s, const void* tree_tag, const PlaceTreeArrays& T, const uint32_t* read_len, uint32_t tau,
                    bool no_filter, double chisq, PlaceDeviceResult* out);
// The same by ranges of reads (round 5): begin once per batch (waits for the front end, sizes the workspaces), then per range
// launch (asynchronous) and finish (waits, copies the range's results back; `kept_base` = candidates earlier ranges left in the
// host arrays) -- the host's last phase of one range runs beside the kernels of the next (kr_place_stream).
int place_device_begin(kr_stream* s, const void* tree_tag, const PlaceTreeArrays& T, const uint32_t* read_len);
int place_device_launch(kr_stream* s, const PlaceTreeArrays& T, uint32_t r0, uint32_t n, uint32_t tau, bool no_filter, double chisq);
int place_device_finish(kr_stream* s, const PlaceTreeArrays& T, uint32_t r0, uint32_t n, uint32_t tau, bool no_filter, double chisq,
                        uint64_t kept_base, PlaceDeviceResult* out);
uint32_t place_stream_nreads(const kr_stream* s);
// Large host buffers the library hands to its caller (report text: tens of megabytes a batch) and its large temporaries: a block
// given back (kr_free / big_free) is kept -- up to a bound -- and handed out again, pages already touched.  glibc maps a block of
// more than 32 MB afresh on every malloc and unmaps it on free: 280 MB of page faults per kr_place_stream call, taken under the
// process's mmap lock by sixteen threads at once, were most of that call's host time (round 6).
void* big_alloc(size_t n);
void big_free(void* p);
// Error path of kr_place_stream: whatever place_device_launch queued for a later range (kernels, the copy into the page-locked
// counters) runs to its end before the stream is used again.  A PlaceDeviceResult is invalid after the next place_device_finish
// on its stream (the page-locked arrays it points to may have been renewed).
void place_device_abort(kr_stream* s);
// Ask for the rows of the batch's ranges as TEXT written on the device (kr_dev_place.inc): call after place_device_begin and before
// the first place_device_launch.  names: the batch's read ids; tabular 0 = jplace rows, 1 = tabular rows.  Off again at the next
// place_device_begin.
int place_device_text_begin(kr_stream* s, const PlaceTreeArrays& T, const char* const* names, uint32_t nreads, int tabular, bool multi);

} // namespace kr

#endif
