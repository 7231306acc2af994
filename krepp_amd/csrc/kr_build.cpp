// kr_build.cpp — CPU index construction (`krepp index`).  Stays on the CPU, as the
// north star asks; it exists because no index can be obtained any other way (the
// reference ships none).  Output files follow the reference's on-disk format
// byte for byte (src/krepp.cpp:18-29,206-246; src/table.cpp:77-83;
// src/record.cpp:213-219).
//
// Behaviour restated from:
//   set_nrows                         src/krepp.cpp:5-16
//   random LSH positions              src/lshf.cpp:126-147
//   minimizer extraction + rho        src/rqseq.cpp:51-144 (incl. the ring buffer that is
//                                     not reset at N / sequence ends, :67,108-114)
//   sort + unique per row             src/table.cpp:234-260
//   colour sets up the guide tree     src/table.cpp:182-232, src/record.cpp:5-107
//   compaction to colour ids          src/record.cpp:131-176
// Design differs from the reference (which materialises a row-vector table per tree
// node and unions them recursively): every genome yields a sorted list of
// (row, enc32) keys; equal keys are grouped across genomes and the colour of each
// group is folded up the guide tree.  The resulting leaf sets are identical; ids of
// colours that are not tree nodes depend on hash-map iteration order in the
// reference and on creation order here (set-level results unchanged).
#include "kr_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <sstream>
#include <sys/stat.h>
#include <unordered_map>
#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

inline unsigned code_of(unsigned char c)
{ // seq_nt4_table, src/common.cpp:10-14
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

} // namespace

namespace kr {

// HyperLogLog with b=12 as the reference instantiates it (src/rqseq.cpp:63-64,
// src/hyperloglog.hpp:98-135): index = top b bits of the 32-bit hash, rank =
// min(32-b, clz(hash << b)) + 1.
struct Hll {
  static constexpr int b = 12;
  std::vector<uint8_t> reg;
  Hll()
    : reg(1u << b, 0)
  {}
  void add(uint32_t hsh)
  {
    uint32_t ix = hsh >> (32 - b);
    uint32_t rest = hsh << b;
    int lz = rest ? __builtin_clz(rest) : 32;
    uint8_t rank = (uint8_t)(std::min(32 - b, lz) + 1);
    if (rank > reg[ix]) reg[ix] = rank;
  }
};

double hll12_estimate(const uint8_t* reg)
{
  const double mreg = 4096.0;
  const double alpha_mm = (0.7213 / (1.0 + 1.079 / mreg)) * mreg * mreg;
  double sum = 0.0;
  uint32_t zeros = 0;
  for (int i = 0; i < 4096; ++i) {
    sum += 1.0 / (double)(1u << reg[i]);
    zeros += reg[i] == 0;
  }
  double est = alpha_mm / sum;
  if (est <= 2.5 * mreg) {
    if (zeros) est = mreg * std::log(mreg / zeros);
  } else if (est > (1.0 / 30.0) * 4294967296.0) {
    est = -4294967296.0 * std::log(1.0 - est / 4294967296.0);
  }
  return est;
}

uint32_t LshPositions::rix(uint64_t bp) const
{
  uint32_t v = 0;
  for (uint32_t j = 0; j < h; ++j) v |= (uint32_t)((bp >> (2 * pasc[j])) & 3u) << (2 * j);
  return v;
}
uint32_t LshPositions::enc32(uint64_t bp) const
{
  uint32_t v = 0;
  for (uint32_t j = 0; j < k - h; ++j) {
    uint32_t c = (uint32_t)((bp >> (2 * npos[j])) & 3u);
    v |= (c & 1u) << j;
    v |= (c >> 1) << (16 + j);
  }
  return v;
}

bool make_positions(uint32_t k, uint32_t h, const uint8_t* ppos, LshPositions& lsh)
{
  lsh.k = k, lsh.h = h;
  lsh.ppos.assign(ppos, ppos + h);
  std::sort(lsh.ppos.begin(), lsh.ppos.end(), std::greater<uint8_t>());
  lsh.pasc.assign(lsh.ppos.rbegin(), lsh.ppos.rend());
  lsh.npos.clear();
  for (uint8_t p = 0; p < k; ++p)
    if (!std::count(lsh.ppos.begin(), lsh.ppos.end(), p)) lsh.npos.push_back(p);
  return lsh.pasc.size() == h && lsh.pasc.back() < k && lsh.npos.size() == k - h;
}

// RSeq::extract_mers (src/rqseq.cpp:51-144) for one contig, sdust off.
void extract_contig_cpu(const uint8_t* seq, uint64_t len, const BuildCfg& c, const LshPositions& lsh,
                        std::vector<uint64_t>& keys, double& n1, double& n2)
{
  uint32_t k = c.k, w = c.w;
  uint32_t ldiff = w > k ? w - k + 1 : 1;
  if (w <= k) w = k;
  Hll c1, c2;
  struct Slot {
    uint64_t x, z;
  };
  std::vector<Slot> win(ldiff, Slot{0, 0});
  uint64_t kix = 0;
  const uint64_t mask_bp = ~0ull >> ((32 - k) * 2);
  uint64_t bp = 0;
  uint32_t l = 0;
  for (uint64_t i = 0; i < len;) {
    unsigned cd = code_of(seq[i]);
    if (cd >= 4) {
      l = 0, i++;
      continue;
    }
    l++, i++;
    bp = (bp << 2) + cd; // compute/update_encoding give the same enc_bp once masked
    if (l < k) continue;
    uint64_t x = bp & mask_bp;
    Slot s{x, fmix64(x)};
    win[kix % ldiff] = s;
    c1.add((uint32_t)s.z);
    kix++;
    if ((l < w) && (i != len)) continue;
    // std::min_element: first minimum in slot order (src/rqseq.cpp:115-116)
    size_t best = 0;
    for (size_t q = 1; q < win.size(); ++q)
      if (win[q].z < win[best].z) best = q;
    const Slot& mn = win[best];
    c2.add((uint32_t)mn.z);
    int64_t row = build_row(lsh.rix(mn.x), c);
    if (row >= 0) keys.push_back(((uint64_t)row << 32) | lsh.enc32(mn.x));
  }
  n1 += hll12_estimate(c1.reg.data());
  n2 += hll12_estimate(c2.reg.data());
}

bool contig_end_special(const uint8_t* seq, uint64_t len, const BuildCfg& c, uint64_t& x, uint64_t& z)
{
  const uint32_t k = c.k, w = std::max(c.w, c.k), ldiff = w - k + 1;
  // length of the final run of valid bases
  uint64_t l = 0;
  while (l < len && code_of(seq[len - 1 - l]) < 4) ++l;
  if (l < k || l >= w) return false; // no k-mer at the end, or the regular window rule applies
  // the ring buffer holds the last `ldiff` k-mers computed anywhere in the contig (it is never
  // reset), and zeros where fewer than ldiff have been computed
  const uint64_t mask_bp = ~0ull >> ((32 - k) * 2);
  std::vector<uint64_t> xs; // most recent first
  uint64_t i = len;
  while (i > 0 && xs.size() < ldiff) {
    // k-mer ending at base i-1 is valid iff the k bases [i-k, i) are valid
    if (i >= k) {
      bool ok = true;
      uint64_t bp = 0;
      for (uint64_t q = i - k; q < i; ++q) {
        unsigned cd = code_of(seq[q]);
        if (cd >= 4) {
          ok = false;
          break;
        }
        bp = (bp << 2) + cd;
      }
      if (ok) xs.push_back(bp & mask_bp);
    }
    --i;
  }
  bool have_zero = xs.size() < ldiff;
  x = 0, z = 0;
  bool first = true;
  for (uint64_t v : xs) {
    uint64_t hz = fmix64(v);
    if (first || hz < z) x = v, z = hz, first = false;
  }
  if (have_zero && (first || 0 < z)) x = 0, z = 0; // a never-written slot {0,0,0} wins unless some hash is 0
  return true;
}

} // namespace kr

namespace {
using kr::BuildCfg;
using kr::LshPositions;
using kr::fmix64;

struct Genome {
  std::string name, path;
  std::vector<uint64_t> keys; // (row << 32) | enc32, sorted unique
  double n1 = 0, n2 = 0;      // HLL sums over contigs
  bool present = false;
};

// Record / Subset (src/record.cpp:5-107): colour sets identified by the wrapping sum
// of their leaves' 64-bit name hashes.
struct SubsetRec {
  uint64_t ch;   // one of the two parts (the larger one)
  uint32_t card;
  uint64_t nonce;
  uint32_t se;
};

struct Colours {
  std::unordered_map<uint64_t, SubsetRec> by_sh;
  std::vector<uint64_t> created; // non-node subsets in creation order

  // Record::add_subset (src/record.cpp:82-107) with check_subset_collision (:119-130)
  uint64_t add(uint64_t sh1, uint64_t sh2)
  {
    const SubsetRec& s1 = by_sh.at(sh1);
    const SubsetRec& s2 = by_sh.at(sh2);
    uint64_t ch = s1.card > s2.card ? sh1 : sh2;
    uint32_t card = s1.card + s2.card;
    uint64_t sh = sh1 + sh2, nonce = 0;
    for (;;) {
      auto it = by_sh.find(sh + nonce);
      if (it == by_sh.end()) break;
      const SubsetRec& s = it->second;
      bool collision = (s.ch == 0 || (sh + nonce) == 0) ? true : !(s.ch == sh1 || s.ch == sh2);
      if (!collision) return sh + nonce;
      uint64_t t = nonce;
      nonce = kr::rehash64(t * sh1 * sh2); // nonce++ is post-increment: the old value is hashed
    }
    sh += nonce;
    by_sh[sh] = SubsetRec{ch, card, nonce, 0};
    created.push_back(sh);
    return sh;
  }
};

struct BuildTree {
  kr::HostTree t;
  std::vector<std::vector<uint32_t>> kids; // by se
  std::vector<uint64_t> sh;                // by se
  std::vector<uint32_t> lo, hi;            // leaf-rank interval of each subtree
  std::vector<uint32_t> leaf_rank;         // by se (leaves)
};

// Colour of a key present in the leaves ranks[a..b) (sorted leaf ranks), restricted to
// the subtree of `se`.  Children are folded LAST to FIRST so that a clade present in
// all children lands on the tree node's own id (Record(tree) registers each node with
// ch = its first child, src/record.cpp:5-33).
uint64_t colour_of(const BuildTree& bt, Colours& col, uint32_t se, const uint32_t* ranks, size_t a, size_t b)
{
  if (bt.t.nodes[se].kind == 1) return bt.sh[se];
  if (b - a == (size_t)(bt.hi[se] - bt.lo[se])) return bt.sh[se]; // whole clade
  uint64_t acc = 0;
  bool have = false;
  const auto& ch = bt.kids[se];
  for (size_t ci = ch.size(); ci-- > 0;) {
    uint32_t c = ch[ci];
    const uint32_t* p0 = std::lower_bound(ranks + a, ranks + b, bt.lo[c]);
    const uint32_t* p1 = std::lower_bound(ranks + a, ranks + b, bt.hi[c]);
    if (p0 == p1) continue;
    uint64_t s = colour_of(bt, col, c, ranks, (size_t)(p0 - ranks), (size_t)(p1 - ranks));
    acc = have ? col.add(acc, s) : s;
    have = true;
  }
  return acc;
}

// the given positions, or LSHF::get_random_positions (src/lshf.cpp:126-147) with the thread-local mt19937 `gen`
bool choose_positions(const kr_build_params* bp, const BuildCfg& c, LshPositions& lsh)
{
  lsh.k = c.k, lsh.h = c.h;
  if (bp->ppos) {
    lsh.ppos.assign(bp->ppos, bp->ppos + c.h);
    std::sort(lsh.ppos.begin(), lsh.ppos.end(), std::greater<uint8_t>());
  } else {
    std::mt19937 gen;
    if (bp->seed) gen.seed(bp->seed);
    std::uniform_int_distribution<uint8_t> distrib(0, (uint8_t)(c.k - 1));
    while (lsh.ppos.size() < c.h) {
      uint8_t n = distrib(gen);
      if (!std::count(lsh.ppos.begin(), lsh.ppos.end(), n)) lsh.ppos.push_back(n);
    }
    std::sort(lsh.ppos.begin(), lsh.ppos.end(), std::greater<uint8_t>());
  }
  lsh.pasc.assign(lsh.ppos.rbegin(), lsh.ppos.rend());
  for (uint8_t p = 0; p < c.k; ++p)
    if (!std::count(lsh.ppos.begin(), lsh.ppos.end(), p)) lsh.npos.push_back(p);
  return lsh.pasc.size() == c.h && lsh.pasc.back() < c.k;
}

// BaseLSH::set_nrows (src/krepp.cpp:5-16)
uint32_t nrows_of(const BuildCfg& c)
{
  uint32_t hash_size = 1u << (2 * c.h), full_res = hash_size % c.m, nrows;
  if (c.frac) {
    nrows = (hash_size / c.m) * (c.r + 1);
    nrows = full_res > c.r ? nrows + (c.r + 1) : nrows + full_res;
  } else {
    nrows = hash_size / c.m;
    nrows = full_res > c.r ? nrows + 1 : nrows;
  }
  return nrows;
}

// validate_configuration (src/krepp.hpp:59-90)
const char* bad_configuration(const BuildCfg& c)
{
  if (c.w < c.k) return "The minimum minimizer window size (-w) is k (-k).";
  if (c.h < 3 || c.h > 15) return "The number of LSH positions (-h) must be in [3,15].";
  if (c.k > 31 || c.k < 19) return "The k-mer length (-k) must be in [19,31].";
  if (c.k - c.h > 16) return "For compact k-mer encodings, h must be >= k-16.";
  if (c.m == 0 || c.r >= c.m) return "need 0 <= r < m";
  return nullptr;
}

bool write_file(const std::string& path, const void* hdr, size_t hdr_bytes, const void* data, size_t bytes)
{
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = fwrite(hdr, 1, hdr_bytes, f) == hdr_bytes && (bytes == 0 || fwrite(data, 1, bytes, f) == bytes);
  return fclose(f) == 0 && ok;
}

} // namespace

extern "C" int kr_build_index(const char* input_tsv, const char* nwk_path, const char* out_dir,
                              const kr_build_params* bp)
{
  kr::clear_error();
  if (!input_tsv || !out_dir || !bp) return kr::fail(KR_ERR_ARG, "kr_build_index: null argument");
  BuildCfg c{bp->k, bp->w, bp->h, bp->m, bp->r, bp->frac != 0};
  if (const char* why = bad_configuration(c)) return kr::fail(KR_ERR_ARG, why);

  // input map: name \t path (src/krepp.cpp:147-162)
  std::vector<Genome> genomes;
  {
    std::ifstream in(input_tsv);
    if (!in.good()) return kr::fail(KR_ERR_IO, std::string("Error opening ") + input_tsv);
    std::string line;
    while (std::getline(in, line)) {
      size_t tab = line.find('\t');
      if (tab == std::string::npos) return kr::fail(KR_ERR_FORMAT, "Failed to read the reference name to path/URL mapping!");
      Genome g;
      g.name = line.substr(0, tab);
      size_t tab2 = line.find('\t', tab + 1);
      g.path = line.substr(tab + 1, tab2 == std::string::npos ? std::string::npos : tab2 - tab - 1);
      genomes.push_back(g);
    }
  }
  if (genomes.empty()) return kr::fail(KR_ERR_FORMAT, "empty input map");

  // guide tree (src/krepp.cpp:131-145)
  BuildTree bt;
  std::string nwk_text;
  if (nwk_path && *nwk_path) {
    std::ifstream tf(nwk_path);
    if (!tf.good()) return kr::fail(KR_ERR_IO, std::string("Error opening ") + nwk_path);
    std::stringstream ss;
    ss << tf.rdbuf();
    nwk_text = ss.str();
    std::string err;
    if (!kr::parse_newick(nwk_text, bt.t, err)) return kr::fail(KR_ERR_FORMAT, err);
  } else {
    std::vector<std::string> names;
    for (auto& g : genomes) names.push_back(g.name);
    kr::balanced_tree(names, bt.t);
  }
  uint32_t nn = bt.t.nnodes();
  bt.kids.assign(nn + 1, {});
  bt.sh.assign(nn + 1, 0);
  bt.lo.assign(nn + 1, 0);
  bt.hi.assign(nn + 1, 0);
  bt.leaf_rank.assign(nn + 1, 0);
  for (uint32_t se = 1; se <= nn; ++se)
    if (bt.t.nodes[se].parent) bt.kids[bt.t.nodes[se].parent].push_back(se);
  // post-order numbering => a subtree is a contiguous se interval; leaf ranks follow se order
  std::map<std::string, uint32_t> leaf_by_name;
  uint32_t nleaves = 0;
  for (uint32_t se = 1; se <= nn; ++se) {
    if (bt.t.nodes[se].kind == 1) {
      bt.leaf_rank[se] = nleaves;
      bt.lo[se] = nleaves;
      bt.hi[se] = ++nleaves;
      uint64_t s = kr::leaf_name_hash(bt.t.nodes[se].label);
      while (!s) s = kr::rehash64(0x9e3779b97f4a7c15ull + se); // src/phytree.cpp:207-209 rehashes an address
      bt.sh[se] = s;
      leaf_by_name[bt.t.nodes[se].label] = se;
    } else {
      uint32_t lo = 0xFFFFFFFFu, hi = 0;
      uint64_t s = 0;
      for (uint32_t cse : bt.kids[se]) {
        lo = std::min(lo, bt.lo[cse]);
        hi = std::max(hi, bt.hi[cse]);
        s += bt.sh[cse]; // Node::add_children, src/phytree.hpp:107-116
      }
      bt.lo[se] = lo, bt.hi[se] = hi, bt.sh[se] = s;
    }
  }
  Colours col;
  col.by_sh[0] = SubsetRec{0, 0, 0, 0};
  for (uint32_t se = 1; se <= nn; ++se) {
    uint64_t ch = bt.t.nodes[se].kind == 1 ? 0 : bt.sh[bt.kids[se].front()];
    if (col.by_sh.count(bt.sh[se]))
      return kr::fail(KR_ERR_FORMAT, "node-name hash collision in the guide tree (Record::check_tree_collision)");
    col.by_sh[bt.sh[se]] = SubsetRec{ch, bt.hi[se] - bt.lo[se], 0, se};
  }

  // LSH positions
  LshPositions lsh;
  if (!choose_positions(bp, c, lsh)) return kr::fail(KR_ERR_ARG, "bad LSH positions");
  const uint32_t nrows = nrows_of(c);

  // leaves: minimizers of every contig (DynHT::fill_table, src/table.cpp:247-260)
  int nthreads = bp->num_threads ? (int)bp->num_threads : 1;
  (void)nthreads;
  std::string first_err;
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
  for (int64_t gi = 0; gi < (int64_t)genomes.size(); ++gi) {
    Genome& g = genomes[gi];
    if (!leaf_by_name.count(g.name)) continue; // genome not in the tree: never visited by build_for_subtree
    kr_fastx* fx = nullptr;
    if (kr_fastx_open(g.path.c_str(), &fx) != KR_OK) {
#if defined(_OPENMP)
#pragma omp critical
#endif
      if (first_err.empty()) first_err = "Failed to open the file at " + g.path;
      continue;
    }
    kr_fastx_batch b;
    std::vector<uint8_t> all_bases; // gpu_minimizers: the whole genome goes to the device at once
    std::vector<uint64_t> all_offs{0};
    do {
      kr_fastx_next(fx, 1, &b); // one record at a time keeps contigs separate
      for (uint32_t q = 0; q < b.nreads; ++q) {
        uint64_t len = b.offsets[q + 1] - b.offsets[q];
        if (bp->gpu_minimizers) {
          all_bases.insert(all_bases.end(), b.bases + b.offsets[q], b.bases + b.offsets[q + 1]);
          all_offs.push_back(all_bases.size());
        } else if (len >= c.w) {
          kr::extract_contig_cpu(b.bases + b.offsets[q], len, c, lsh, g.keys, g.n1, g.n2); // RSeq::set_curr_seq: len >= w
        }
      }
    } while (b.more);
    kr_fastx_close(fx);
    if (bp->gpu_minimizers) {
      kr_build_params bq = *bp;
      bq.ppos = lsh.ppos.data();
      kr_minimizer_result mr;
      memset(&mr, 0, sizeof(mr));
      int mrc;
#if defined(_OPENMP)
#pragma omp critical(kr_gpu_minimizers)
#endif
      mrc = kr_minimizers_device(bp->device, &bq, all_bases.data(), all_offs.data(), (uint32_t)all_offs.size() - 1, &mr);
      if (mrc != KR_OK) {
#if defined(_OPENMP)
#pragma omp critical
#endif
        if (first_err.empty()) first_err = std::string("GPU minimizers: ") + kr_last_error();
        continue;
      }
      g.keys.assign(mr.keys, mr.keys + mr.nkeys);
      g.n1 = mr.n1, g.n2 = mr.n2;
      kr_minimizers_free(&mr);
    }
    std::sort(g.keys.begin(), g.keys.end());
    g.keys.erase(std::unique(g.keys.begin(), g.keys.end()), g.keys.end());
    g.present = true;
  }
  if (!first_err.empty()) return kr::fail(KR_ERR_IO, first_err);

  // group equal keys across genomes
  struct KL {
    uint64_t key;
    uint32_t rank;
  };
  std::vector<KL> all;
  {
    size_t tot = 0;
    for (auto& g : genomes) tot += g.keys.size();
    all.reserve(tot);
    for (auto& g : genomes) {
      if (!g.present) continue;
      uint32_t rank = bt.leaf_rank[leaf_by_name[g.name]];
      for (uint64_t key : g.keys) all.push_back(KL{key, rank});
      std::vector<uint64_t>().swap(g.keys);
    }
  }
  std::sort(all.begin(), all.end(), [](const KL& a, const KL& b) { return a.key != b.key ? a.key < b.key : a.rank < b.rank; });
  if (all.empty()) return kr::fail(KR_ERR_FORMAT, "No k-mers to index!");

  std::vector<uint64_t> inc(nrows, 0);
  std::vector<uint64_t> out_sh;
  std::vector<uint32_t> out_enc;
  std::vector<uint32_t> out_row;
  std::vector<uint32_t> ranks;
  uint32_t root_se = nn;
  for (size_t a = 0; a < all.size();) {
    size_t b = a;
    ranks.clear();
    while (b < all.size() && all[b].key == all[a].key) {
      if (ranks.empty() || ranks.back() != all[b].rank) ranks.push_back(all[b].rank);
      ++b;
    }
    uint64_t s = colour_of(bt, col, root_se, ranks.data(), 0, ranks.size());
    uint32_t row = (uint32_t)(all[a].key >> 32);
    if (row >= nrows) return kr::fail(KR_ERR_FORMAT, "row out of range");
    out_sh.push_back(s);
    out_enc.push_back((uint32_t)all[a].key);
    out_row.push_back(row);
    inc[row]++;
    a = b;
  }
  for (uint32_t rr = 1; rr < nrows; ++rr) inc[rr] += inc[rr - 1];

  // Record::make_compact + CRecord(record) (src/record.cpp:131-176)
  uint32_t next_se = nn + 1;
  for (uint64_t s : col.created) col.by_sh[s].se = next_se++;
  uint32_t nsubsets = next_se; // sh_to_se.size()+1 with the empty set at 0
  std::vector<uint32_t> pse((size_t)nsubsets * 2, 0);
  for (auto& kv : col.by_sh) {
    const SubsetRec& s = kv.second;
    if (kv.first == 0) continue;
    auto f1 = col.by_sh.find(s.ch);
    auto f2 = col.by_sh.find(kv.first - s.ch - s.nonce);
    pse[2 * (size_t)s.se] = f1 == col.by_sh.end() ? 0 : f1->second.se;
    pse[2 * (size_t)s.se + 1] = f2 == col.by_sh.end() ? 0 : f2->second.se; // operator[] default in the reference
  }
  std::vector<double> rho(nn + 1, 0.0);
  for (auto& g : genomes)
    if (g.present) rho[leaf_by_name[g.name]] = g.n1 > 0 ? g.n2 / g.n1 : 0.0; // RSeq::compute_rho, src/rqseq.hpp:79
  std::vector<uint32_t> cmer(out_enc.size() * 2);
  for (size_t i = 0; i < out_enc.size(); ++i) {
    cmer[2 * i] = out_enc[i];
    cmer[2 * i + 1] = col.by_sh[out_sh[i]].se;
  }

  // save_index (src/krepp.cpp:206-246)
  mkdir(out_dir, 0777);
  std::string sfx = "-m" + std::to_string(c.m) + "r" + std::to_string(c.r) + (c.frac ? "-frac" : "-no_frac");
  std::string dir = out_dir;
  uint64_t nk = out_enc.size();
  uint32_t nn1 = nn + 1;
  bool ok = write_file(dir + "/cmer" + sfx, &nk, 8, cmer.data(), cmer.size() * 4);
  ok = ok && write_file(dir + "/inc" + sfx, &nrows, 4, inc.data(), inc.size() * 8);
  {
    std::string blob;
    blob.append((const char*)&nn1, 4);
    blob.append((const char*)&nsubsets, 4);
    blob.append((const char*)pse.data(), pse.size() * 4);
    blob.append((const char*)rho.data(), rho.size() * 8);
    ok = ok && write_file(dir + "/crecord" + sfx, blob.data(), blob.size(), nullptr, 0);
  }
  {
    std::string names;
    for (auto& g : genomes) names += g.name + "\n";
    ok = ok && write_file(dir + "/reflist" + sfx, names.data(), names.size(), nullptr, 0);
  }
  if (!nwk_text.empty()) ok = ok && write_file(dir + "/tree" + sfx, nwk_text.data(), nwk_text.size(), nullptr, 0);
  {
    std::string md;
    uint8_t k8 = (uint8_t)c.k, w8 = (uint8_t)c.w, h8 = (uint8_t)c.h, f8 = c.frac ? 1 : 0;
    md.append((const char*)&k8, 1), md.append((const char*)&w8, 1), md.append((const char*)&h8, 1);
    md.append((const char*)&c.m, 4), md.append((const char*)&c.r, 4), md.append((const char*)&f8, 1);
    md.append((const char*)&nrows, 4);
    md.append((const char*)lsh.ppos.data(), lsh.ppos.size());
    md.append((const char*)lsh.npos.data(), lsh.npos.size());
    ok = ok && write_file(dir + "/metadata" + sfx, md.data(), md.size(), nullptr, 0);
  }
  {
    std::ostringstream info; // IndexMultiple::save_info (src/krepp.cpp:187-204); ignored by loaders
    info << "krepp version: v0.8.3-compatible (krepp-amd)\nseed: " << bp->seed << "\nk: " << c.k << "\nw: " << c.w
         << "\nh: " << c.h << "\nm: " << c.m << "\nfrac: " << (c.frac ? "true" : "false") << "\nnrows: " << nrows
         << "\ntotal_num_kmers: " << nk << "\n";
    std::string s = info.str();
    ok = ok && write_file(dir + "/metadata" + sfx + ".txt", s.data(), s.size(), nullptr, 0);
  }
  if (!ok) return kr::fail(KR_ERR_IO, std::string("failed to write index files under ") + out_dir);
  return KR_OK;
}

// ---------------------------------------------------------------------------
// Leaf stage on its own (CPU): the reference path for kr_minimizers_device.
// ---------------------------------------------------------------------------
namespace kr {
int check_minimizer_params(const kr_build_params* bp, BuildCfg& c, LshPositions& lsh)
{
  if (!bp || !bp->ppos) return fail(KR_ERR_ARG, "kr_minimizers: parameters with explicit ppos are required");
  c = BuildCfg{bp->k, bp->w, bp->h, bp->m, bp->r, bp->frac != 0};
  if (c.k > 31 || c.k < 3 || c.h == 0 || c.h >= c.k || c.k - c.h > 16 || c.h > 15 || c.m == 0 || c.r >= c.m || c.w > 255)
    return fail(KR_ERR_ARG, "kr_minimizers: unsupported k/w/h/m/r");
  if (!make_positions(c.k, c.h, bp->ppos, lsh)) return fail(KR_ERR_ARG, "kr_minimizers: bad LSH positions");
  return KR_OK;
}
int finish_minimizers(std::vector<uint64_t>& keys, double n1, double n2, kr_minimizer_result* out)
{
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  out->nkeys = keys.size();
  out->keys = (uint64_t*)malloc(std::max<size_t>(1, keys.size()) * 8);
  if (!out->keys) return fail(KR_ERR_NOMEM, "kr_minimizers: out of memory");
  if (!keys.empty()) memcpy(out->keys, keys.data(), keys.size() * 8);
  out->n1 = n1;
  out->n2 = n2;
  return KR_OK;
}
} // namespace kr

extern "C" int kr_minimizers_cpu(const kr_build_params* bp, const uint8_t* bases, const uint64_t* offsets, uint32_t ncontigs,
                                 kr_minimizer_result* out)
{
  kr::clear_error();
  if (!bases || !offsets || !out) return kr::fail(KR_ERR_ARG, "kr_minimizers_cpu: null argument");
  BuildCfg c;
  LshPositions lsh;
  int rc = kr::check_minimizer_params(bp, c, lsh);
  if (rc) return rc;
  std::vector<uint64_t> keys;
  double n1 = 0, n2 = 0;
  for (uint32_t q = 0; q < ncontigs; ++q) {
    uint64_t len = offsets[q + 1] - offsets[q];
    if (len >= c.w) kr::extract_contig_cpu(bases + offsets[q], len, c, lsh, keys, n1, n2);
  }
  return kr::finish_minimizers(keys, n1, n2, out);
}

extern "C" void kr_minimizers_free(kr_minimizer_result* r)
{
  if (!r) return;
  free(r->keys);
  r->keys = nullptr;
  r->nkeys = 0;
}

// `krepp sketch` (SketchSingle::create_sketch / save_sketch, src/krepp.cpp:110-129): the minimizers of ONE
// FASTA/FASTQ file as a table without colours.  SDynHT::fill_table (src/table.cpp:234-246) = extract_mers of
// every record of length >= w, rows sorted by code, duplicates removed; file = SFlatHT::save
// (src/table.cpp:34-40: nkmers u64, codes u32[], nrows u32, cumulative row ends u64[]) + save_configuration
// (src/krepp.cpp:18-29) + rho f64.
extern "C" int kr_build_sketch(const char* input_path, const char* out_path, const kr_build_params* bp)
{
  kr::clear_error();
  if (!input_path || !out_path || !bp) return kr::fail(KR_ERR_ARG, "kr_build_sketch: null argument");
  BuildCfg c{bp->k, bp->w, bp->h, bp->m, bp->r, bp->frac != 0};
  if (const char* why = bad_configuration(c)) return kr::fail(KR_ERR_ARG, why);
  LshPositions lsh;
  if (!choose_positions(bp, c, lsh)) return kr::fail(KR_ERR_ARG, "bad LSH positions");
  const uint32_t nrows = nrows_of(c);
  kr_fastx* fx = nullptr;
  if (kr_fastx_open(input_path, &fx) != KR_OK) return kr::fail(KR_ERR_IO, std::string("Failed to open the file at ") + input_path);
  std::vector<uint64_t> keys;
  double n1 = 0, n2 = 0;
  kr_fastx_batch b;
  do {
    if (kr_fastx_next(fx, 1u << 20, &b)) {
      kr_fastx_close(fx);
      return KR_ERR_IO;
    }
    for (uint32_t q = 0; q < b.nreads; ++q) {
      const uint64_t len = b.offsets[q + 1] - b.offsets[q];
      if (len >= c.w) kr::extract_contig_cpu(b.bases + b.offsets[q], len, c, lsh, keys, n1, n2); // RSeq::set_curr_seq
    }
  } while (b.more);
  kr_fastx_close(fx);
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  std::vector<uint32_t> enc(keys.size());
  std::vector<uint64_t> inc(nrows, 0);
  for (size_t i = 0; i < keys.size(); ++i) {
    const uint32_t row = (uint32_t)(keys[i] >> 32);
    if (row >= nrows) return kr::fail(KR_ERR_STATE, "kr_build_sketch: row out of range");
    enc[i] = (uint32_t)keys[i];
    inc[row]++;
  }
  for (uint32_t r = 1; r < nrows; ++r) inc[r] += inc[r - 1];
  const double rho = n1 > 0 ? n2 / n1 : 0.0; // RSeq::compute_rho, src/rqseq.hpp:79
  std::string blob;
  const uint64_t nk = enc.size();
  blob.append((const char*)&nk, 8);
  blob.append((const char*)enc.data(), enc.size() * 4);
  blob.append((const char*)&nrows, 4);
  blob.append((const char*)inc.data(), inc.size() * 8);
  const uint8_t k8 = (uint8_t)c.k, w8 = (uint8_t)c.w, h8 = (uint8_t)c.h, f8 = c.frac ? 1 : 0;
  blob.append((const char*)&k8, 1), blob.append((const char*)&w8, 1), blob.append((const char*)&h8, 1);
  blob.append((const char*)&c.m, 4), blob.append((const char*)&c.r, 4), blob.append((const char*)&f8, 1);
  blob.append((const char*)&nrows, 4);
  blob.append((const char*)lsh.ppos.data(), lsh.ppos.size());
  blob.append((const char*)lsh.npos.data(), lsh.npos.size());
  blob.append((const char*)&rho, 8);
  if (!write_file(out_path, blob.data(), blob.size(), nullptr, 0)) return kr::fail(KR_ERR_IO, "Failed to write the sketch!");
  fprintf(stderr, "Total number of k-mers included in the sketch: %llu\nSubsampling rate (rho) is: %g\n", (unsigned long long)nk, rho);
  return KR_OK;
}
