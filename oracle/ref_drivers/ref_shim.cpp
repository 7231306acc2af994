/*
 * ref_shim.cpp — thin extern "C" driver around the parts of the reference that
 * compile from their own sources with nothing but libc/libstdc++/zlib:
 *   src/hdhistllh.hpp   (likelihood functor; includes only <cmath>)
 *   src/MurmurHash3.cpp (stock MurmurHash3)
 *   src/kseq.h          (FASTA/FASTQ reader; needs zlib)
 *   src/hyperloglog.hpp (rho estimator used by the index builder)
 * The reference sources are compiled WHERE THEY LIE (-I/root/reference/src);
 * nothing is copied.  Everything else in the reference includes
 * parallel_hashmap/phmap.h via src/common.hpp:4 (an empty submodule) and is
 * therefore unbuildable here — see DESIGN.md.
 *
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/kr_oracle.cpp.
 */
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <zlib.h>

#include "hdhistllh.hpp"
#include "MurmurHash3.hpp"
#include "hyperloglog.hpp"
extern "C" {
#include "kseq.h"
}
KSEQ_INIT(gzFile, gzread)

extern "C" {

double ref_llh(uint32_t k, uint32_t h, uint32_t th, double* hist, double uc, double rho, double d)
{
  optimize::HDistHistLLH f(h, k, th);
  f.set_parameters(hist, uc, rho);
  return f(d);
}

uint32_t ref_murmur3_x86_32(const void* key, int len, uint32_t seed)
{
  uint32_t out = 0;
  MurmurHash3_x86_32(key, len, seed, &out);
  return out;
}

/* Parse a FASTA/FASTQ(.gz) file with the reference's kseq; names and sequences are
 * appended NUL-separated into the two buffers.  Returns the record count, or -1
 * if a buffer is too small.  *last_ret receives kseq_read's final return value. */
long ref_kseq_parse(const char* path, char* names, size_t names_cap, char* seqs, size_t seqs_cap, int* last_ret)
{
  gzFile f = gzopen(path, "rb");
  if (!f) return -2;
  kseq_t* ks = kseq_init(f);
  long n = 0;
  size_t no = 0, so = 0;
  int ret;
  while ((ret = kseq_read(ks)) >= 0) {
    if (no + ks->name.l + 1 > names_cap || so + ks->seq.l + 1 > seqs_cap) {
      n = -1;
      break;
    }
    memcpy(names + no, ks->name.s, ks->name.l + 1);
    no += ks->name.l + 1;
    memcpy(seqs + so, ks->seq.s, ks->seq.l + 1);
    so += ks->seq.l + 1;
    n++;
  }
  if (last_ret) *last_ret = ret;
  kseq_destroy(ks);
  gzclose(f);
  return n;
}

double ref_hll_estimate(const uint64_t* hashes, uint64_t n, uint32_t b)
{
  hll::HyperLogLog c((uint8_t)b);
  for (uint64_t i = 0; i < n; ++i) c.add((uint32_t)hashes[i]); /* add(const uint32_t): u64 truncates, as at src/rqseq.cpp:108-110 */
  return c.estimate();
}
}
