/*
 * kr_oracle.cpp — CPU restatement of krepp v0.8.3's `krepp dist` per-read path.
 *
 * TEST INFRASTRUCTURE ONLY (see kr_oracle.h).  Written from the behaviour of the
 * reference; nothing here is copied from it.  Citations are file:line into
 * /root/reference (bo1929/krepp, VERSION "v0.8.3", src/common.hpp:50).
 *
 * Deliberate differences from the reference, none of which changes a set-level
 * result:
 *   - per-read leaf maps iterate in ascending colour id (`se`); the reference
 *     iterates a hash map keyed by heap addresses (src/query.hpp:46,95), so its
 *     row order and its `<=` tie-breaks (src/query.cpp:110,123) are arbitrary.
 *   - out-of-range token / table accesses that are undefined behaviour in the
 *     reference (unlabelled root in src/phytree.cpp:173, bytes >= 128 in
 *     src/query.cpp:50) are given the obvious meaning (empty token, invalid base).
 *   - Brent's minimiser is restated from the published Boost.Math algorithm
 *     (boost/math/tools/minima.hpp); Boost is an absent submodule: PARITY UNPINNED.
 */
#include "kr_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <map>
#include <memory>
#include <queue>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#if defined(__BMI2__)
#include <immintrin.h>
#endif
#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// Primitives: src/common.cpp:10-18, src/common.hpp:147-243
// ---------------------------------------------------------------------------

// seq_nt4_table (src/common.cpp:10-14): A/a=0 C/c=1 G/g=2 T/t=3, everything else 4.
// The reference's table has 128 entries; bytes >= 128 index past it (UB) — treated as 4.
inline unsigned nt4(unsigned char c)
{
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

// nt4_lr_table (src/common.cpp:16): high bit of the code at bit 32, low bit at bit 0.
inline uint64_t nt4_lr(unsigned c) { return ((uint64_t)(c >> 1) << 32) | (uint64_t)(c & 1); }

// compute_encoding (src/common.hpp:225-235)
inline void compute_encoding(const char* s1, const char* s2, uint64_t& enc_lr, uint64_t& enc_bp)
{
  enc_lr = 0;
  enc_bp = 0;
  for (; s1 < s2; ++s1) {
    enc_lr <<= 1;
    enc_bp <<= 2;
    unsigned c = nt4((unsigned char)*s1);
    enc_bp += c;
    enc_lr += nt4_lr(c);
  }
}

// update_encoding (src/common.hpp:236-243)
inline void update_encoding(const char* s1, uint64_t& enc_lr, uint64_t& enc_bp)
{
  enc_lr <<= 1;
  enc_bp <<= 2;
  enc_lr &= 0xFFFFFFFEFFFFFFFEull;
  unsigned c = nt4((unsigned char)*s1);
  enc_bp += c;
  enc_lr += nt4_lr(c);
}

// revcomp_bp64 (src/common.hpp:177-186)
inline uint64_t revcomp_bp64(uint64_t x, uint32_t k)
{
  uint64_t res = ~x;
  res = ((res >> 2 & 0x3333333333333333ull) | (res & 0x3333333333333333ull) << 2);
  res = ((res >> 4 & 0x0F0F0F0F0F0F0F0Full) | (res & 0x0F0F0F0F0F0F0F0Full) << 4);
  res = ((res >> 8 & 0x00FF00FF00FF00FFull) | (res & 0x00FF00FF00FF00FFull) << 8);
  res = ((res >> 16 & 0x0000FFFF0000FFFFull) | (res & 0x0000FFFF0000FFFFull) << 16);
  res = ((res >> 32 & 0x00000000FFFFFFFFull) | (res & 0x00000000FFFFFFFFull) << 32);
  return res >> (2 * (32 - k));
}

// rmoddp_bp64 (src/common.hpp:188-197): keep the even bits, compact them.
inline uint64_t rmoddp_bp64(uint64_t x)
{
  x = x & 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
  x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
  x = (x | (x >> 16)) & 0x00000000ffffffffull;
  return x;
}

// conv_bp64_lr64 (src/common.hpp:223)
inline uint64_t conv_bp64_lr64(uint64_t x) { return (rmoddp_bp64(x >> 1) << 32) | rmoddp_bp64(x); }

// popcount_lr32 (src/common.hpp:175)
inline uint32_t popcount_lr32(uint32_t z) { return (uint32_t)__builtin_popcount((z | (z >> 16)) & 0x0000ffffu); }

// xur64_hash (src/common.hpp:147-155) == MurmurHash3 fmix64
inline uint64_t xur64_hash(uint64_t h)
{
  h ^= (h >> 33);
  h *= 0xff51afd7ed558ccdull;
  h ^= (h >> 33);
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= (h >> 33);
  return h;
}

// Bit extraction under a mask == x86 PEXT (src/lshf.cpp:62-69 use _pext_u64;
// src/common.hpp:245-256 extract_bits is the portable form).
inline uint64_t pext64(uint64_t x, uint64_t mask)
{
#if defined(__BMI2__)
  return _pext_u64(x, mask);
#else
  uint64_t res = 0;
  for (uint64_t bb = 1; mask != 0; bb += bb) {
    if (x & mask & (0 - mask)) res |= bb;
    mask &= (mask - 1);
  }
  return res;
#endif
}

// MurmurHash3_x86_32 — published algorithm (Austin Appleby, public domain); the
// reference calls it only on node names (src/record.hpp:26-36, seeds 0 and 1).
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t murmur3_x86_32(const void* key, int len, uint32_t seed)
{
  const uint8_t* data = (const uint8_t*)key;
  const int nblocks = len / 4;
  uint32_t h1 = seed;
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  for (int i = 0; i < nblocks; i++) {
    uint32_t k1;
    memcpy(&k1, data + 4 * i, 4);
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8; /* fallthrough */
    case 1:
      k1 ^= tail[0];
      k1 *= c1;
      k1 = rotl32(k1, 15);
      k1 *= c2;
      h1 ^= k1;
  }
  h1 ^= (uint32_t)len;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}

// Subset::get_singleton_sh (src/record.hpp:26-36): seed-0 hash in the high word.
uint64_t name_hash(const std::string& name)
{
  uint64_t a1 = murmur3_x86_32(name.data(), (int)name.size(), 0);
  uint64_t a2 = murmur3_x86_32(name.data(), (int)name.size(), 1);
  return (a1 << 32) | a2;
}

// ---------------------------------------------------------------------------
// LSH: src/lshf.cpp:12-69,149-157
// ---------------------------------------------------------------------------
struct Lsh {
  uint32_t k = 0, h = 0, m = 0;
  std::vector<uint8_t> ppos, npos; // ppos descending, npos ascending (src/lshf.cpp:126-147)
  uint64_t mask_hash_bp = 0, mask_drop_lr = 0;

  // set_lshf (src/lshf.cpp:39-54): only the two masks the query path uses.
  void set()
  {
    k = (uint32_t)(ppos.size() + npos.size());
    h = (uint32_t)ppos.size();
    mask_hash_bp = 0;
    mask_drop_lr = 0;
    for (int i = (int)npos.size() - 1; i >= 0; --i) mask_drop_lr += (0x0000000100000001ull << npos[i]);
    for (uint32_t i = 0; i < 16 - (k - h); ++i) mask_drop_lr += 0x1ull << (i + k);
    for (int i = (int)ppos.size() - 1; i >= 0; --i) mask_hash_bp += (0x3ull << (ppos[i] * 2));
  }
  // compute_hash (src/lshf.cpp:62)
  uint32_t compute_hash(uint64_t enc_bp) const { return (uint32_t)pext64(enc_bp, mask_hash_bp); }
  // drop_ppos_lr (src/lshf.cpp:64-69)
  uint32_t drop_ppos_lr(uint64_t enc_lr) const { return (uint32_t)pext64(enc_lr, mask_drop_lr); }
  bool compatible(const Lsh& o) const { return m == o.m && h == o.h && k == o.k && npos == o.npos && ppos == o.ppos; }
};

// ---------------------------------------------------------------------------
// Tree: src/phytree.cpp:84-253,394-404; src/phytree.hpp
// ---------------------------------------------------------------------------
struct TNode {
  std::string name;
  double blen = NAN;
  bool is_leaf = true;
  uint32_t se = 0;
  int parent = -1;
  std::vector<int> children;
  uint32_t card = 0;          // leaves below (Node::add_children, src/phytree.hpp:107-116)
  uint32_t eff_nchildren = 0; // children with a mapped leaf below (Tree::compute_eff_nchildren)
  bool is_taxon = false;      // Node::set_rank (src/phytree.hpp:90-94): taxa of a lineage tree and its root
  std::string rank;
};

struct Tree {
  std::vector<TNode> nodes;       // storage, arbitrary order
  std::vector<int> se_to_node{-1}; // se_to_node[0] = null (src/phytree.hpp:53)
  uint32_t nnodes = 0;
  int root = -1;
  size_t atter = 0;
  std::string err;

  int new_node()
  {
    nodes.emplace_back();
    return (int)nodes.size() - 1;
  }
  bool check_node(uint32_t se) const { return se <= nnodes; } // src/phytree.hpp:34
  int get_node(uint32_t se) const { return se < se_to_node.size() ? se_to_node[se] : -1; }

  // split_nwk (src/phytree.cpp:84-148)
  bool split_nwk(std::string nwk, std::vector<std::string>& el)
  {
    std::string buf;
    bool is_quoted = false, quote = false, quote_p = false, is_comment = false;
    if (nwk.empty()) {
      err = "Given Newick tree seems to be empty?!?.";
      return false;
    }
    if (nwk.back() == '\n') nwk.pop_back();
    if (nwk.empty() || nwk.back() != ';') {
      err = "Given Newick tree ends with a character other than ';'.";
      return false;
    }
    for (size_t i = 0; i < nwk.size(); i++) {
      char c = nwk[i];
      if (is_comment) {
        is_comment = is_comment != (c == ']');
        continue;
      }
      quote = (c == '\'' || c == '"');
      if (quote & quote_p) {
        is_quoted = false;
        buf += "'";
        continue;
      }
      quote_p = quote;
      if (quote) {
        is_quoted = (is_quoted != quote);
        continue;
      } else if (is_quoted) {
        is_comment = is_comment != (c == '[');
        if (!is_comment) buf += c;
      } else if (c == '(' || c == ')' || c == ':' || c == ',') {
        if (c != '(' && (i == 0 || nwk[i - 1] != '(')) {
          el.push_back(buf);
          buf.clear();
        }
        el.push_back(std::string(1, c));
      } else {
        if (c == '[' || c == ']') {
          err = "Given Newick tree contains an unquoted label or length with '[' or ']'.";
          return false;
        }
        if (c == ';') {
          if (i == nwk.size() - 1) break;
          err = "Given Newick tree contains an unexpected ';'.";
          return false;
        }
        if ((c == ' ' || c == '\n') && !buf.empty()) {
          err = "Given Newick tree contains an unquoted label or length with ' ' or newline.";
          return false;
        }
        buf += c;
      }
    }
    if (!buf.empty()) el.push_back(buf);
    return true;
  }

  const std::string& tok(const std::vector<std::string>& el, size_t i) const
  {
    static const std::string empty;
    return i < el.size() ? el[i] : empty;
  }

  void parse_label(int nd, const std::vector<std::string>& el)
  { // src/phytree.cpp:177-188 and 193-204 (same text in both branches)
    nodes[nd].name = "";
    nodes[nd].blen = NAN;
    if (atter < el.size() && tok(el, atter) != ",") {
      if (tok(el, atter) != ":") {
        nodes[nd].name = tok(el, atter);
        atter++;
      }
      if (tok(el, atter) == ":") {
        nodes[nd].blen = atof(tok(el, atter + 1).c_str());
        atter += 2;
      }
    }
  }

  // Node::parse (src/phytree.cpp:150-215): `se` is the 1-based post-order index.
  bool parse(int nd, const std::vector<std::string>& el)
  {
    if (atter >= el.size()) return true;
    if (el[atter] == "(") {
      while (true) {
        atter++;
        int child = new_node();
        nodes[child].parent = nd;
        if (!parse(child, el)) return false;
        nodes[nd].children.push_back(child);
        nodes[nd].is_leaf = false;
        if (tok(el, atter) == ",")
          continue;
        else
          break;
      }
      if (nodes[nd].children.size() == 1) {
        err = "A node has a single child in the backbone tree! Please suppress unifurcations.";
        return false;
      }
      nnodes++;
      nodes[nd].se = nnodes;
      se_to_node.push_back(nd);
      if (tok(el, atter) == ")") {
        atter++;
        if (atter < el.size() && el[atter] == ")") return true;
      }
      parse_label(nd, el);
    } else {
      parse_label(nd, el);
      nodes[nd].is_leaf = true;
      nnodes++;
      nodes[nd].se = nnodes;
      se_to_node.push_back(nd);
    }
    return true;
  }

  // Tree::load (src/phytree.cpp:394-404)
  bool load(const std::string& nwk)
  {
    std::vector<std::string> el;
    if (!split_nwk(nwk, el)) return false;
    nodes.reserve(el.size() + 2);
    root = new_node();
    atter = 0;
    nnodes = 0;
    return parse(root, el);
  }

  // Tree::parse_lineages (src/phytree.cpp:320-369): taxonomy as a tree.  One line per reference:
  // ID <tab> lineage, taxa separated by ';' (after "; " -> ";"), each "r__Name".  A taxon keeps the parent
  // it was first seen under; parentless nodes are hung below "root" at the end -- the reference does that in
  // hash-map order, here in order of first appearance.  Node::card is added to the parent when the child is
  // attached (Node::add_children, src/phytree.hpp:107-116), i.e. before a taxon has received its own
  // children: restated as is.  Numbering: post-order, children in attachment order (:261-298).
  static std::string strip_rank_prefixes(const std::string& t)
  { // std::regex_replace(taxon, std::regex(".__"), ""): every non-overlapping <any char but a line end>"__"
    std::string o;
    size_t i = 0;
    while (i < t.size()) {
      if (i + 2 < t.size() && t[i] != '\n' && t[i] != '\r' && t[i + 1] == '_' && t[i + 2] == '_')
        i += 3;
      else
        o += t[i++];
    }
    return o;
  }
  static std::string rank_of(const std::string& t)
  { // std::regex_replace(taxon, std::regex("__.*"), ""): '.' stops at a line end
    std::string o;
    size_t i = 0;
    while (i < t.size()) {
      if (i + 1 < t.size() && t[i] == '_' && t[i + 1] == '_') {
        i += 2;
        while (i < t.size() && t[i] != '\n' && t[i] != '\r') ++i;
      } else
        o += t[i++];
    }
    return o;
  }
  void attach(int child, int parent)
  { // Node::set_parent + add_children (src/phytree.hpp:95-116)
    if (parent < 0) return;
    nodes[child].parent = parent;
    nodes[parent].children.push_back(child);
    nodes[parent].card += nodes[child].card;
    nodes[parent].is_leaf = false;
  }
  void number_post_order(int nd)
  {
    for (int c : nodes[nd].children) number_post_order(c);
    nnodes++;
    nodes[nd].se = nnodes;
    se_to_node.push_back(nd);
  }
  bool parse_lineages(const std::string& text)
  {
    root = new_node();
    nodes[root].name = "root";
    nodes[root].is_leaf = true; // until something is attached
    nodes[root].is_taxon = true;
    nodes[root].rank = "root";
    nnodes = 0;
    std::map<std::string, int> taxon_to_node;
    std::vector<int> order; // creation order of the map's nodes
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
      { // std::regex_replace(line, std::regex("; "), ";"): one left-to-right pass
        std::string o;
        for (size_t q = 0; q < line.size(); ++q) {
          o += line[q];
          if (line[q] == ';' && q + 1 < line.size() && line[q + 1] == ' ') ++q;
        }
        line.swap(o);
      }
      size_t tab = line.find('\t');
      if (line.empty() || tab == std::string::npos || tab + 1 == line.size()) {
        err = "Failed to reference to lineage mapping!";
        return false;
      }
      std::string name = line.substr(0, tab), lineage = line.substr(tab + 1);
      size_t tab2 = lineage.find('\t');
      if (tab2 != std::string::npos) lineage.resize(tab2);
      std::istringstream lss(lineage);
      std::string taxon;
      int parent = -1;
      while (std::getline(lss, taxon, ';')) {
        std::string rank = rank_of(taxon);
        taxon = strip_rank_prefixes(taxon);
        if (taxon.empty()) continue;
        if (!taxon_to_node.count(taxon)) {
          int nd = new_node();
          nodes[nd].name = taxon;
          nodes[nd].is_leaf = false;
          nodes[nd].card = 0;
          attach(nd, parent);
          nodes[nd].is_taxon = true;
          nodes[nd].rank = rank;
          taxon_to_node[taxon] = nd;
          order.push_back(nd);
        }
        parent = taxon_to_node[taxon];
      }
      if (taxon_to_node.count(name)) {
        err = "The same reference appears more than once in the lineage file.";
        return false;
      }
      int nd = new_node();
      nodes[nd].name = name;
      nodes[nd].is_leaf = true;
      nodes[nd].card = 1;
      attach(nd, parent);
      taxon_to_node[name] = nd;
      order.push_back(nd);
    }
    for (int nd : order)
      if (nodes[nd].parent < 0) attach(nd, root);
    // A taxon left without children (only when a later taxon of its lineage already existed under another parent)
    // makes the reference's traversal walk off an empty child list (src/phytree.cpp:266-267); here it ends the path.
    for (auto& nd : nodes)
      if (nd.children.empty()) nd.is_leaf = true;
    number_post_order(root);
    return true;
  }

  // Node::generate_tree (src/phytree.cpp:217-253): balanced tree by recursive
  // halving, SECOND half first; every node numbered post-order.
  void generate(int nd, const std::vector<std::string>& names, size_t first, size_t last)
  {
    size_t n = last - first;
    if (n == 1) {
      nodes[nd].name = names[first];
      nodes[nd].blen = 1.0;
      nodes[nd].is_leaf = true;
      nnodes++;
      nodes[nd].se = nnodes;
      se_to_node.push_back(nd);
    } else {
      size_t half = first + n / 2;
      for (int pix = 0; pix < 2; ++pix) {
        int child = new_node();
        nodes[child].parent = nd;
        if (pix)
          generate(child, names, first, half);
        else
          generate(child, names, half, last);
        nodes[nd].children.push_back(child);
      }
      nodes[nd].blen = 1.0;
      nodes[nd].is_leaf = false;
      nodes[nd].name = "";
      nnodes++;
      nodes[nd].se = nnodes;
      se_to_node.push_back(nd);
    }
  }
  void generate_tree(const std::vector<std::string>& names)
  { // Tree::generate_tree (src/phytree.cpp:38-45)
    nodes.reserve(2 * names.size() + 2);
    root = new_node();
    nnodes = 0;
    generate(root, names, 0, names.size());
  }

  // post-order name sequence, for check_compatible (src/phytree.cpp:10-36)
  std::vector<std::string> postorder_names() const
  {
    std::vector<std::string> v;
    for (uint32_t se = 1; se <= nnodes; ++se) v.push_back(nodes[se_to_node[se]].name);
    return v;
  }
  // Node::get_name (src/phytree.hpp:134-145)
  std::string get_name(uint32_t se) const
  {
    int nd = get_node(se);
    if (nd < 0) return "";
    if (!nodes[nd].name.empty()) return nodes[nd].name;
    return std::to_string(se - 1);
  }
};

// ---------------------------------------------------------------------------
// Index: src/index.cpp:51-201, src/table.cpp:65-75, src/record.cpp:203-211,
//        src/krepp.cpp:66-108
// ---------------------------------------------------------------------------
struct Lib {
  std::string suffix;
  uint32_t r = 0;
  bool frac = false;
  uint8_t w = 0;
  uint32_t nrows_meta = 0;
  std::vector<uint64_t> inc;                        // cumulative bucket END offsets
  std::vector<std::pair<uint32_t, uint32_t>> cmer; // (enc32, se)
  uint32_t nnodes = 0, nsubsets = 0;
  std::vector<std::pair<uint32_t, uint32_t>> pse;
  std::vector<double> rho;
};

} // namespace

struct ko_index {
  Lsh lsh;
  Tree tree;
  bool wbackbone = false;
  bool have_lsh = false, have_tree = false;
  std::vector<Lib> libs;
  std::map<uint32_t, uint32_t> r_to_lib;       // residue -> library (src/index.hpp:41)
  std::map<uint32_t, uint32_t> r_to_numerator; // src/index.hpp:42
  std::vector<std::string> names;              // cached get_name per se
  std::vector<uint8_t> kind;                   // 0 null, 1 leaf, 2 internal
  // placement tree (`place`): the backbone itself, or a user tree the index leaves are mapped onto
  Tree ptree;
  bool have_ptree = false;
  std::vector<int> se_to_pnode; // index colour id of a leaf -> node of ptree, or -1
};

namespace {

template <typename T>
bool rd(std::ifstream& f, T& v)
{
  f.read(reinterpret_cast<char*>(&v), sizeof(T));
  return f.good();
}

bool read_file(const std::string& path, std::string& out)
{
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

// Index::load_partial_index (src/index.cpp:51-158)
bool load_partial_index(ko_index* ix, const std::string& dir, const std::string& suffix, std::string& err)
{
  std::ifstream md(dir + "/metadata" + suffix, std::ios::binary);
  if (!md.is_open()) {
    err = "Failed to open " + dir + "/metadata" + suffix;
    return false;
  }
  // layout written by BaseLSH::save_configuration (src/krepp.cpp:18-29)
  uint8_t k, w, h, fracb;
  uint32_t m, r, nrows;
  rd(md, k), rd(md, w), rd(md, h), rd(md, m), rd(md, r), rd(md, fracb), rd(md, nrows);
  if (!md.good() || h > k) {
    err = "Failed to read the metadata of a partial index!";
    return false;
  }
  Lsh lsh;
  lsh.m = m;
  lsh.ppos.resize(h);
  lsh.npos.resize(k - h);
  md.read(reinterpret_cast<char*>(lsh.ppos.data()), h);
  md.read(reinterpret_cast<char*>(lsh.npos.data()), k - h);
  if (!md.good()) {
    err = "Failed to read the metadata of a partial index!";
    return false;
  }
  lsh.set();
  if (ix->have_lsh && !ix->lsh.compatible(lsh)) { // src/lshf.cpp:159-180
    err = "Partial libraries have incompatible hash functions!";
    return false;
  }
  ix->lsh = lsh;
  ix->have_lsh = true;

  Lib lib;
  lib.suffix = suffix;
  lib.r = r;
  lib.frac = fracb != 0;
  lib.w = w;
  lib.nrows_meta = nrows;
  { // FlatHT::load (src/table.cpp:65-75)
    std::ifstream ms(dir + "/cmer" + suffix, std::ios::binary);
    std::ifstream is(dir + "/inc" + suffix, std::ios::binary);
    if (!ms.is_open() || !is.is_open()) {
      err = "Failed to open cmer/inc of " + suffix;
      return false;
    }
    uint64_t nkmers = 0;
    rd(ms, nkmers);
    lib.cmer.resize(nkmers);
    ms.read(reinterpret_cast<char*>(lib.cmer.data()), nkmers * 8);
    uint32_t nr = 0;
    rd(is, nr);
    lib.inc.resize(nr);
    is.read(reinterpret_cast<char*>(lib.inc.data()), (std::streamsize)nr * 8);
    if (!ms.good() || !is.good()) {
      err = "Failed to read the k-mer vector / offset array of a partial index!";
      return false;
    }
  }
  { // CRecord::load (src/record.cpp:203-211)
    std::ifstream cs(dir + "/crecord" + suffix, std::ios::binary);
    if (!cs.is_open()) {
      err = "Failed to open crecord" + suffix;
      return false;
    }
    rd(cs, lib.nnodes), rd(cs, lib.nsubsets);
    lib.pse.resize(lib.nsubsets);
    cs.read(reinterpret_cast<char*>(lib.pse.data()), (std::streamsize)lib.nsubsets * 8);
    lib.rho.resize(lib.nnodes);
    cs.read(reinterpret_cast<char*>(lib.rho.data()), (std::streamsize)lib.nnodes * 8);
    if (!cs.good()) {
      err = "Failed to read the color array of a partial index!";
      return false;
    }
  }
  uint32_t lix = (uint32_t)ix->libs.size();
  ix->libs.push_back(std::move(lib));
  // src/index.cpp:144-157
  if (fracb) {
    for (uint32_t q = 0; q <= r; ++q) {
      ix->r_to_lib[q] = lix;
      ix->r_to_numerator[q] = r + 1;
    }
  } else {
    ix->r_to_lib[r] = lix;
    ix->r_to_numerator[r] = 1;
  }
  return true;
}

// ---------------------------------------------------------------------------
// Likelihood: src/hdhistllh.hpp:51-96
// ---------------------------------------------------------------------------
struct Llh {
  uint32_t th = 0, k = 0, h = 0;
  double rho = 1, uc = 0;
  const double* mc = nullptr;
  std::vector<uint64_t> binom_k, binom_hnk;

  Llh() {}
  Llh(uint32_t h_, uint32_t k_, uint32_t th_)
    : th(th_), k(k_), h(h_)
  { // ctor src/hdhistllh.hpp:51-69 (uint64 arithmetic, truncating division)
    uint64_t vc = 1;
    uint32_t nh = k - h;
    binom_k.assign(k + 1, 0);
    binom_hnk.assign(th + 1, 0);
    binom_k[0] = 1;
    binom_hnk[0] = 0;
    for (uint32_t i = 0; i < k; ++i) binom_k[i + 1] = (binom_k[i] * (k - i)) / (i + 1);
    for (uint32_t i = 1; i <= th; ++i) {
      vc = (vc * (uint64_t)(nh - i + 1)) / i;
      binom_hnk[i] = (i <= k ? binom_k[i] : 0) - vc;
    }
  }
  void set_parameters(const double* mc_, double uc_, double rho_)
  { // src/hdhistllh.hpp:91-96
    mc = mc_;
    uc = uc_;
    rho = rho_;
  }
  // operator() src/hdhistllh.hpp:71-89 — same operation order.
  double operator()(double d) const
  {
    double sum = 0.0, lv_m = 0.0;
    double powdc = pow((1.0 - d), k);
    double logdn = log(1.0 - d);
    double logdp = log(d) - logdn;
    logdn *= k;
    double dratio = d / (1.0 - d);
    for (uint32_t x = 0; x <= k; ++x) {
      if (x <= th) {
        sum -= (logdn + x * logdp) * mc[x];
        lv_m += binom_hnk[x] * powdc;
      } else {
        lv_m += powdc * binom_k[x];
      }
      powdc *= dratio;
    }
    return sum - log(rho * lv_m + 1.0 - rho) * uc;
  }
};

// boost::math::tools::brent_find_minima(f, min, max, bits) as called at
// src/query.cpp:430 with (1e-10, 0.5, 16).  Boost is not in the tree: restated
// from the published algorithm.  PARITY UNPINNED.
template <class F>
std::pair<double, double> brent_find_minima(F& f, double min, double max, int bits, uint64_t* nevals)
{
  bits = std::min(53 / 2, bits);
  const double tolerance = ldexp(1.0, 1 - bits);
  double x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
  const double golden = 0.3819660f; // float literal widened, as in Boost
  x = w = v = max;
  fw = fv = fx = f(x);
  uint64_t ne = 1;
  delta2 = delta = 0;
  for (;;) {
    mid = (min + max) / 2;
    fract1 = tolerance * fabs(x) + tolerance / 4;
    fract2 = 2 * fract1;
    if (fabs(x - mid) <= (fract2 - (max - min) / 2)) break;
    if (fabs(delta2) > fract1) {
      double r = (x - w) * (fx - fv);
      double q = (x - v) * (fx - fw);
      double p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = fabs(q);
      double td = delta2;
      delta2 = delta;
      if ((fabs(p) >= fabs(q * td / 2)) || (p <= q * (min - x)) || (p >= q * (max - x))) {
        delta2 = (x >= mid) ? min - x : max - x;
        delta = golden * delta2;
      } else {
        delta = p / q;
        u = x + delta;
        if (((u - min) < fract2) || ((max - u) < fract2)) delta = (mid - x) < 0 ? -fabs(fract1) : fabs(fract1);
      }
    } else {
      delta2 = (x >= mid) ? min - x : max - x;
      delta = golden * delta2;
    }
    u = (fabs(delta) >= fract1) ? (x + delta) : (delta > 0 ? (x + fabs(fract1)) : (x - fabs(fract1)));
    fu = f(u);
    ne++;
    if (fu <= fx) {
      if (u >= x)
        min = x;
      else
        max = x;
      v = w;
      w = x;
      x = u;
      fv = fw;
      fw = fx;
      fx = fu;
    } else {
      if (u < x)
        min = u;
      else
        max = u;
      if ((fu <= fw) || (w == x)) {
        v = w;
        w = u;
        fv = fw;
        fw = fu;
      } else if ((fu <= fv) || (v == x) || (v == w)) {
        v = u;
        fv = fu;
      }
    }
  }
  if (nevals) *nevals += ne;
  return std::make_pair(x, fx);
}

// ---------------------------------------------------------------------------
// Minfo: src/query.hpp:100-228
// ---------------------------------------------------------------------------
struct Minfo {
  double nmers = 0, mismatch_count = 0, match_count = 0, rho = 0.0;
  uint32_t rmatch_count = 0, last_pos = 0, last_hdist = 0xFFFFFFFFu, hdist_min = 0xFFFFFFFFu;
  std::vector<double> hist;
  double chisq = NAN, v_llh = NAN, d_llh = DBL_MAX;
  double lwr = 1; // src/query.hpp:225
  uint32_t strand = 0;
  bool passed = false;

  Minfo() {}
  explicit Minfo(uint32_t th) { hist.assign(th + 1, 0.0); }
  Minfo(uint32_t th, uint32_t nmers_, double rho_)
    : nmers(nmers_), rho(rho_)
  { // src/query.hpp:116-123
    rmatch_count = 1;
    mismatch_count = nmers_;
    hist.assign(th + 1, 0.0);
  }
  // update_match (src/query.hpp:153-176): per read position only the minimum hd counts.
  void update_match(uint32_t pos, uint32_t hd)
  {
    if (last_hdist == 0xFFFFFFFFu || last_pos != pos) {
      match_count++;
      mismatch_count--;
      hist[hd]++;
      last_pos = pos;
      last_hdist = hd;
    } else if (last_hdist > hd) {
      hist[hd]++;
      hist[last_hdist]--;
      last_hdist = hd;
    }
    if (hd < hdist_min) hdist_min = hd;
  }
  // add (src/query.hpp:139-152): weighted accumulation of a descendant leaf into an ancestor
  void add(const Minfo& m, double denom)
  {
    mismatch_count = nmers ? mismatch_count : m.nmers;
    match_count += m.match_count * denom;
    mismatch_count -= m.match_count * denom;
    for (size_t x = 0; x < hist.size(); ++x) hist[x] = hist[x] + m.hist[x] * denom;
    hdist_min = std::min(hdist_min, m.hdist_min);
    nmers = std::max(nmers, m.nmers);
    rho = std::max(rho, m.rho);
    rmatch_count++;
  }
  // get_leq_tau (src/query.hpp:189-196)
  double get_leq_tau(uint32_t tau) const
  {
    double t = 0.0;
    for (uint32_t x = 0; x <= tau && x < hist.size(); ++x) t += hist[x];
    return t;
  }
  // jukes_cantor_dist (src/query.hpp:197; CJC = 4.0 / 3.0, src/query.hpp:11)
  double jukes_cantor_dist() const { return -0.75 * log(1 - 4.0 / 3.0 * d_llh); }
  // optimize_likelihood (src/query.cpp:426-433)
  void optimize_likelihood(Llh& f, ko_counters& c)
  {
    f.set_parameters(hist.data(), mismatch_count, rho);
    auto sol = brent_find_minima(f, 1e-10, 0.5, 16, &c.llh_evals);
    c.brent_runs++;
    d_llh = sol.first;
    v_llh = sol.second;
  }
  // likelihood_ratio (src/query.cpp:420-424)
  double likelihood_ratio(double d, Llh& f) const
  {
    f.set_parameters(hist.data(), mismatch_count, rho);
    return 2 * (f(d) - v_llh);
  }
};

// IMers (src/query.hpp:22-47): one per strand per read.
struct IMers {
  uint32_t enmers = 0, onmers = 0;
  uint32_t hdist_filt = 0xFFFFFFFFu;
  std::map<uint32_t, Minfo> leaf_to_minfo; // keyed by leaf se (reference: node pointer)
};

struct Worker {
  const ko_index* ix;
  ko_params p;
  Llh llh;
  uint64_t mask_bp, mask_lr;
  ko_counters c;
  std::vector<ko_row> rows;
  std::vector<ko_acc> accs;
  std::vector<ko_hit> hits;
  std::string text;
  uint32_t read_ix = 0;
  // place mode (IBatch::place_sequences, src/query.cpp:198-216)
  bool place_mode = false, tabular = false, has_previous = false;
  std::vector<ko_placement> placements;
  // --summarize (src/query.cpp:160-171): reference -> weighted read count
  bool summarize = false;
  std::map<uint32_t, double> node_to_wcount;
  std::map<uint32_t, double> pnode_to_wcount; // place --summarize: se of the placement-tree node -> count

  Worker(const ko_index* ix_, const ko_params& p_)
    : ix(ix_), p(p_)
  {
    memset(&c, 0, sizeof(c));
    uint32_t k = ix->lsh.k, h = ix->lsh.h;
    llh = Llh(h, k, p.hdist_th); // src/query.cpp:34
    uint64_t u64m = ~0ull;       // src/query.cpp:35-37
    mask_lr = ((u64m >> (64 - k)) << 32) + ((u64m << 32) >> (64 - k));
    mask_bp = u64m >> ((32 - k) * 2);
  }

  // Index::check_partial (src/index.hpp:27)
  bool check_partial(uint32_t rix) const { return ix->r_to_lib.count(rix % ix->lsh.m) != 0; }

  // IMers::add_matching_mer (src/query.cpp:352-390) with Index::bucket_indices
  // (src/index.cpp:160-168) and FlatHT::bucket_start/next (src/table.hpp:121-136).
  void add_matching_mer(IMers& im, uint32_t strand, uint32_t pos, uint32_t kpos, uint32_t rix, uint32_t enc_lr)
  {
    uint32_t m = ix->lsh.m;
    uint32_t rix_res = rix % m;
    uint32_t offset = rix / m;
    uint32_t numer = ix->r_to_numerator.at(rix_res);
    if (numer > 1) offset = offset * numer + rix_res;
    uint32_t lix = ix->r_to_lib.at(rix_res);
    const Lib& lib = ix->libs[lix];
    uint64_t b0 = offset ? (offset - 1 < lib.inc.size() ? lib.inc[offset - 1] : lib.cmer.size()) : 0;
    uint64_t b1 = offset < lib.inc.size() ? lib.inc[offset] : lib.cmer.size();
    c.probes++;
    c.bucket_entries += (b1 - b0);
    const Tree& tree = ix->tree;
    std::queue<uint32_t> se_q;
    for (uint64_t e = b0; e < b1; ++e) {
      uint32_t hd = popcount_lr32(lib.cmer[e].first ^ enc_lr);
      if (hd > p.hdist_th) continue;
      if (hd < im.hdist_filt) im.hdist_filt = hd;
      c.hits++;
      if (p.collect & 2u) {
        ko_hit ht;
        ht.read = read_ix, ht.strand = strand, ht.pos = pos, ht.kpos = kpos, ht.lib = lix, ht.hd = hd;
        ht.cmer_index = e, ht.enc = lib.cmer[e].first, ht.se = lib.cmer[e].second;
        hits.push_back(ht);
      }
      se_q.push(lib.cmer[e].second);
      while (!se_q.empty()) {
        uint32_t se = se_q.front();
        se_q.pop();
        if (tree.check_node(se)) {
          int kd = ix->kind[se]; // 0: Tree::get_node(se) == nullptr, 1: leaf, 2: internal
          if (kd == 0) {
            continue;
          } else if (kd == 1) {
            auto it = im.leaf_to_minfo.find(se);
            if (it == im.leaf_to_minfo.end()) {
              c.rho_reads++;
              it = im.leaf_to_minfo.emplace(se, Minfo(p.hdist_th, im.enmers, lib.rho[se])).first;
              it->second.strand = strand;
            }
            it->second.update_match(pos, hd);
            continue;
          }
        }
        c.pse_reads++;
        std::pair<uint32_t, uint32_t> pse = se < lib.pse.size() ? lib.pse[se] : std::make_pair(0u, 0u);
        se_q.push(pse.first);
        se_q.push(pse.second);
      }
    }
    im.onmers++;
  }

  // IBatch::search_mers (src/query.cpp:40-94), non-CANONICAL build (src/rqseq.hpp:9).
  uint32_t search_mers(const char* seq, uint64_t len, IMers& im_or, IMers& im_rc)
  {
    const Lsh& lsh = ix->lsh;
    uint32_t k = lsh.k;
    uint32_t onmers = 0;
    uint32_t i, l;
    uint64_t orenc64_bp = 0, orenc64_lr = 0, rcenc64_bp;
    for (i = l = 0; i < len;) {
      if (nt4((unsigned char)seq[i]) >= 4) {
        l = 0, i++;
        continue;
      }
      l++, i++;
      if (l < k) continue;
      if (l == k)
        compute_encoding(seq + i - k, seq + i, orenc64_lr, orenc64_bp);
      else
        update_encoding(seq + i - 1, orenc64_lr, orenc64_bp);
      orenc64_bp &= mask_bp;
      orenc64_lr &= mask_lr;
      rcenc64_bp = revcomp_bp64(orenc64_bp, k);
      onmers++;
      c.kmers_valid++;
      c.lsh_evals += 2;
      uint32_t orrix = lsh.compute_hash(orenc64_bp);
      if (check_partial(orrix)) add_matching_mer(im_or, 0, i - k, i - k, orrix, lsh.drop_ppos_lr(orenc64_lr));
      uint32_t rcrix = lsh.compute_hash(rcenc64_bp);
      if (check_partial(rcrix))
        add_matching_mer(im_rc, 1, (uint32_t)(len - i), i - k, rcrix, lsh.drop_ppos_lr(conv_bp64_lr64(rcenc64_bp)));
    }
    return onmers;
  }

  // fixed, precision-5 number as the reference's stringstream prints it (src/query.cpp:208-209)
  static std::string f5(double v)
  {
    char b[64];
    if (std::isnan(v)) return v < 0 || std::signbit(v) ? "-nan" : "nan";
    snprintf(b, sizeof(b), "%.5f", v);
    return b;
  }

  // IBatch::report_placement (src/query.cpp:218-333); macros PP_JPLACE_FIELDS / PP_TABULAR_FIELDS
  // (src/query.hpp:202-206).  Maps are keyed by placement-tree node id (the reference keys by pointer).
  bool report_placement(std::map<uint32_t, Minfo*>& node_to_minfo, uint32_t nd_closest, Minfo* mi_closest, const char* name)
  {
    const Tree& pt = ix->ptree;
    if (node_to_minfo.size() == 0 || !(p.no_filter || (mi_closest->get_leq_tau(p.tau) > 1.0))) return false;
    std::string id = name ? name : "";
    auto pnode = [&](uint32_t se) { return ix->se_to_pnode[se]; };
    auto en = [&](int nd) { return pt.nodes[nd].se - 1; };
    auto midpoint = [&](int nd) { return std::isnan(pt.nodes[nd].blen) ? 0.0 : pt.nodes[nd].blen / 2.0; };
    auto jplace_fields = [&](int nd, const Minfo& mi) {
      return "[" + std::to_string(en(nd)) + ", " + f5(mi.jukes_cantor_dist() - midpoint(nd)) + ", " + f5(midpoint(nd)) + ", " +
             f5(-mi.v_llh) + ", " + f5(mi.lwr) + ", " + f5(mi.d_llh) + "]";
    };
    auto tabular_fields = [&](int nd, const Minfo& mi) {
      const std::string& nm = pt.nodes[nd].name;
      return (nm.empty() ? std::string("NA") : nm) + "\t" + std::to_string(en(nd)) + "\t" + f5(mi.lwr) + "\t" + f5(mi.d_llh);
    };
    auto record = [&](int nd, const Minfo& mi) {
      ko_placement pl;
      pl.read = read_ix, pl.edge = en(nd), pl.lwr = mi.lwr, pl.d_llh = mi.d_llh, pl.v_llh = mi.v_llh;
      pl.pendant = mi.jukes_cantor_dist() - midpoint(nd), pl.distal = midpoint(nd);
      placements.push_back(pl);
    };
    int nd_pp = pnode(nd_closest);
    Minfo* mi_pp = mi_closest;
    mi_pp->chisq = 0;
    if (!tabular && !summarize) {
      if (has_previous) text += ",\n";
      text += "\t\t\t{\"n\" : [\"" + id + "\"], \"p\" : [";
    }
    if (node_to_minfo.size() == 1) {
      record(nd_pp, *mi_pp);
      if (summarize)
        pnode_to_wcount[pt.nodes[nd_pp].se] += 1.0;
      else if (tabular)
        text += id + "\t" + tabular_fields(nd_pp, *mi_pp) + "\n";
      else
        text += jplace_fields(nd_pp, *mi_pp) + "]}";
      return true;
    }
    // keyed by (se, node id): iteration in ascending edge number (the reference's order is arbitrary)
    std::map<std::pair<uint32_t, int>, Minfo*> pp_map_se;
    std::map<int, Minfo*> pp_map;
    std::vector<std::unique_ptr<Minfo>> owned;
    for (auto& kv : node_to_minfo) {
      int nd_curr = pnode(kv.first);
      Minfo* mi_curr = kv.second;
      pp_map[nd_curr] = mi_curr;
      double denom = 1.0;
      int nd_parent = nd_curr;
      while ((nd_parent = pt.nodes[nd_parent].parent) >= 0) {
        // src/query.cpp:257-262.  Keys of node_to_minfo are leaves, which never carry a rank: the first
        // branch cannot fire even on a lineage tree (kept for the record).
        if (pt.nodes[nd_parent].is_taxon && pt.nodes[nd_curr].is_taxon)
          denom = 1.0;
        else
          denom /= pt.nodes[nd_parent].eff_nchildren;
        if (!pp_map.count(nd_parent)) {
          owned.emplace_back(new Minfo(p.hdist_th));
          pp_map[nd_parent] = owned.back().get();
        }
        pp_map[nd_parent]->add(*mi_curr, denom);
      }
    }
    for (auto& kv : pp_map) pp_map_se[std::make_pair(pt.nodes[kv.first].se, kv.first)] = kv.second;
    std::vector<int> nd_v;
    for (auto& kv : pp_map_se) {
      int nd = kv.first.second;
      Minfo* mi = kv.second;
      uint32_t nch = (uint32_t)pt.nodes[nd].children.size();
      if (nch != pt.nodes[nd].eff_nchildren || nch == 1) continue;
      if (p.no_filter || (mi->get_leq_tau(p.tau) > 1.0)) {
        if (!pt.nodes[nd].is_leaf) mi->optimize_likelihood(llh, c);
        mi->chisq = mi_closest->likelihood_ratio(mi->d_llh, llh);
        c.llh_evals++;
        if ((mi->chisq < p.chisq) && pt.nodes[nd].parent >= 0) nd_v.push_back(nd);
      }
    }
    double total_lwr = 0;
    for (int nd : nd_v) {
      Minfo* mi = pp_map[nd];
      mi->lwr = exp(-mi->chisq / 2);
      total_lwr = total_lwr + mi->lwr;
    }
    if (p.multi) {
      for (size_t i = 0; i < nd_v.size(); ++i) {
        int nd = nd_v[i];
        Minfo* mi = pp_map[nd];
        mi->lwr = mi->lwr / total_lwr;
        record(nd, *mi);
        if (summarize) {
          pnode_to_wcount[pt.nodes[nd].se] += 1.0 / nd_v.size();
          continue;
        }
        if (i > 0 && !tabular) text += ",";
        if (tabular)
          text += id + "\t" + tabular_fields(nd, *mi) + "\n";
        else
          text += "\n\t\t\t\t" + jplace_fields(nd, *mi);
      }
      if (!tabular && !summarize) text += "]\n\t\t\t}";
    } else {
      if (nd_v.size() > 1) {
        std::stable_sort(nd_v.begin(), nd_v.end(), [&](int lhs, int rhs) {
          return (pt.nodes[lhs].card == pt.nodes[rhs].card) ? pp_map[lhs]->d_llh > pp_map[rhs]->d_llh
                                                            : pt.nodes[lhs].card < pt.nodes[rhs].card;
        });
      }
      if (nd_v.empty()) { // the reference dereferences nd_v.back() here (UB); nothing can be reported
        if (!tabular && !summarize) text += "]}";
        return true;
      }
      int nd = nd_v.back();
      Minfo* mi = pp_map[nd];
      mi->lwr = mi->lwr / total_lwr;
      record(nd, *mi);
      if (summarize)
        pnode_to_wcount[pt.nodes[nd].se] += 1.0;
      else if (tabular)
        text += id + "\t" + tabular_fields(nd, *mi) + "\n";
      else
        text += jplace_fields(nd, *mi) + "]}";
    }
    return true;
  }

  // One read: IBatch::estimate_distances body (src/query.cpp:141-156) =
  // search_mers + summarize_matches (:96-139) + report_distances (:158-196).
  void run_read(const char* seq, uint64_t len, const char* name, ko_readinfo& ri)
  {
    uint32_t k = ix->lsh.k;
    IMers im_or, im_rc;
    // IMers ctor (src/query.cpp:335-350)
    im_or.enmers = im_rc.enmers = len ? (uint32_t)(len - k + 1) : 0;
    c.reads++;
    c.bases += len;
    uint32_t onmers = search_mers(seq, len, im_or, im_rc);
    ri.onmers = onmers;
    ri.hdist_filt[0] = im_or.hdist_filt;
    ri.hdist_filt[1] = im_rc.hdist_filt;

    // summarize_matches (src/query.cpp:96-139)
    std::map<uint32_t, Minfo*> node_to_minfo;
    uint32_t nd_closest = 0; // 0 stands for tree->get_root()
    Minfo mi_root(p.hdist_th);
    Minfo* mi_closest = &mi_root;
    im_or.hdist_filt = 2 * im_or.hdist_filt + 1; // u32 wrap keeps "none" = max
    im_rc.hdist_filt = 2 * im_rc.hdist_filt + 1;
    for (auto& kv : im_or.leaf_to_minfo) {
      Minfo* mi = &kv.second;
      mi->mismatch_count = onmers - mi->match_count;
      if (mi->hdist_min > im_or.hdist_filt) continue;
      mi->passed = true;
      mi->optimize_likelihood(llh, c);
      if (mi->d_llh <= mi_closest->d_llh) {
        nd_closest = kv.first;
        mi_closest = mi;
      }
      node_to_minfo.emplace(kv.first, mi);
    }
    for (auto& kv : im_rc.leaf_to_minfo) {
      Minfo* mi = &kv.second;
      mi->mismatch_count = onmers - mi->match_count;
      if (mi->hdist_min > im_rc.hdist_filt) continue;
      mi->passed = true;
      mi->optimize_likelihood(llh, c);
      if (mi->d_llh <= mi_closest->d_llh) {
        nd_closest = kv.first;
        mi_closest = mi;
      }
      node_to_minfo[kv.first] = mi;
      auto io = im_or.leaf_to_minfo.find(kv.first);
      if (io != im_or.leaf_to_minfo.end()) {
        Minfo* mi_or = &io->second;
        if ((mi->d_llh > mi_or->d_llh) || ((mi->d_llh == mi_or->d_llh) && (mi->match_count < mi_or->match_count)))
          node_to_minfo[kv.first] = mi_or;
      }
    }
    if (nd_closest != 0) node_to_minfo[nd_closest] = mi_closest;

    if (p.collect & 1u) {
      for (IMers* im : {&im_or, &im_rc})
        for (auto& kv : im->leaf_to_minfo) {
          const Minfo& mi = kv.second;
          ko_acc a;
          memset(&a, 0, sizeof(a));
          a.read = read_ix, a.se = kv.first, a.strand = mi.strand;
          a.match_count = (uint32_t)mi.match_count, a.hdist_min = mi.hdist_min, a.passed = mi.passed;
          a.rho = mi.rho, a.d_llh = mi.d_llh, a.v_llh = mi.v_llh;
          for (uint32_t x = 0; x <= p.hdist_th && x < 17; ++x) a.hist[x] = (uint32_t)mi.hist[x];
          accs.push_back(a);
          c.accs++;
        }
    } else {
      c.accs += im_or.leaf_to_minfo.size() + im_rc.leaf_to_minfo.size();
    }

    if (place_mode) {
      uint32_t np0 = (uint32_t)placements.size();
      if (report_placement(node_to_minfo, nd_closest, mi_closest, name) && !tabular) has_previous = true;
      ri.nrows = (uint32_t)placements.size() - np0;
      return;
    }

    // report_distances (src/query.cpp:158-196), non-summarize branch.
    uint32_t nrows0 = (uint32_t)rows.size();
    bool want_text = (p.collect & 4u) != 0;
    char buf[64];
    auto emit = [&](uint32_t se, const Minfo* mi) {
      ko_row r;
      r.read = read_ix, r.se = se, r.strand = mi ? mi->strand : 0, r.match_count = mi ? (uint32_t)mi->match_count : 0;
      r.d_llh = mi ? mi->d_llh : NAN, r.v_llh = mi ? mi->v_llh : NAN, r.chisq = mi ? mi->chisq : NAN;
      rows.push_back(r);
      if (want_text) {
        text += name ? name : "";
        if (se) {
          // DISTANCE_FIELDS (src/query.hpp:210), std::fixed precision 5 (src/query.cpp:152-153)
          snprintf(buf, sizeof(buf), "%.5f", mi->d_llh);
          text += "\t" + ix->names[se] + "\t" + buf + "\n";
        } else {
          text += "\tNA\tNaN\n";
        }
      }
    };
    bool dmax_set = !std::isnan(p.dist_max);
    if (summarize) { // src/query.cpp:160-171: overrides --no-multi and --no-filter
      std::vector<uint32_t> nd_v;
      for (auto& kv : node_to_minfo) {
        Minfo* mi = kv.second;
        mi->chisq = mi_closest->likelihood_ratio(mi->d_llh, llh);
        c.llh_evals++;
        if (mi->chisq < p.chisq && (!dmax_set || mi->d_llh < p.dist_max)) nd_v.push_back(kv.first);
      }
      for (uint32_t nd : nd_v) node_to_wcount[nd] += 1.0 / nd_v.size();
      return;
    }
    if (node_to_minfo.empty() || (dmax_set && (mi_closest->d_llh > p.dist_max))) {
      emit(0, nullptr);
    } else if (p.multi) {
      if (p.no_filter) {
        for (auto& kv : node_to_minfo)
          if (!dmax_set || kv.second->d_llh < p.dist_max) emit(kv.first, kv.second);
      } else {
        for (auto& kv : node_to_minfo) {
          Minfo* mi = kv.second;
          mi->chisq = mi_closest->likelihood_ratio(mi->d_llh, llh);
          c.llh_evals++;
          if (mi->chisq < p.chisq && (!dmax_set || mi->d_llh < p.dist_max)) emit(kv.first, mi);
        }
      }
    } else {
      emit(nd_closest, mi_closest);
    }
    ri.nrows = (uint32_t)rows.size() - nrows0;
    c.rows += ri.nrows;
    if (p.collect & 8u) rows.resize(nrows0); // timing runs: the rows are made and counted, not kept (no serial merge of GBs afterwards)
  }
};

void add_counters(ko_counters& a, const ko_counters& b)
{
  uint64_t* pa = reinterpret_cast<uint64_t*>(&a);
  const uint64_t* pb = reinterpret_cast<const uint64_t*>(&b);
  for (size_t i = 0; i < sizeof(ko_counters) / 8; ++i) pa[i] += pb[i];
}

template <typename T>
T* dup_vec(const std::vector<T>& v)
{
  T* p = (T*)malloc(std::max<size_t>(1, v.size()) * sizeof(T));
  if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

void set_err(char* err, int errlen, const std::string& s)
{
  if (err && errlen > 0) {
    snprintf(err, (size_t)errlen, "%s", s.c_str());
  }
}

} // namespace

extern "C" {

// TargetIndex::load_index (src/krepp.cpp:66-108)
ko_index* ko_index_load(const char* dir, char* err, int errlen)
{
  static const std::set<std::string> lall{"cmer", "crecord", "inc", "metadata", "tree", "reflist"};
  std::map<std::string, std::set<std::string>> suffix_to_ltype;
  DIR* d = opendir(dir);
  if (!d) {
    set_err(err, errlen, std::string("cannot open index directory ") + dir);
    return nullptr;
  }
  while (dirent* e = readdir(d)) {
    std::string filename = e->d_name;
    if (filename == "." || filename == "..") continue;
    size_t pos1 = filename.find('-', 0);
    if (pos1 == std::string::npos) continue;
    size_t pos2 = filename.find('-', pos1 + 1);
    if (pos2 == std::string::npos) continue;
    std::string ltype = filename.substr(0, pos1);
    if (!lall.count(ltype)) continue;
    // std::filesystem::path::extension().empty(): no '.' after the first character
    size_t dot = filename.rfind('.');
    if (dot != std::string::npos && dot != 0) continue;
    suffix_to_ltype[filename.substr(pos1)].insert(ltype);
  }
  closedir(d);
  ko_index* ix = new ko_index();
  std::string e;
  std::vector<std::string> first_names;
  for (auto& kv : suffix_to_ltype) {
    const std::string& suffix = kv.first;
    const std::set<std::string>& lt = kv.second;
    bool base = lt.count("cmer") && lt.count("crecord") && lt.count("inc") && lt.count("metadata");
    bool wb = base && lt.count("tree");
    bool wob = base && lt.count("reflist");
    Tree t;
    if (wb) { // Index::load_partial_tree (src/index.cpp:29-49)
      std::string nwk;
      if (!read_file(std::string(dir) + "/tree" + suffix, nwk) || !t.load(nwk)) {
        set_err(err, errlen, "Failed to read the backbone tree of a partial index! " + t.err);
        delete ix;
        return nullptr;
      }
      ix->wbackbone = true;
    } else if (wob) { // Index::generate_partial_tree (src/index.cpp:3-27)
      std::ifstream rf(std::string(dir) + "/reflist" + suffix);
      std::vector<std::string> names;
      std::string name;
      while (std::getline(rf, name)) names.push_back(name);
      if (names.empty()) {
        set_err(err, errlen, "Unable to open reference list file for an index without a tree.");
        delete ix;
        return nullptr;
      }
      t.generate_tree(names);
      ix->wbackbone = false;
    } else {
      set_err(err, errlen, "There is a partial index with a missing file!");
      delete ix;
      return nullptr;
    }
    if (ix->have_tree) { // Tree::check_compatible (src/phytree.cpp:10-36)
      if (t.postorder_names() != first_names) {
        set_err(err, errlen, "Partial libraries are based on different trees!");
        delete ix;
        return nullptr;
      }
    } else {
      first_names = t.postorder_names();
      ix->tree = std::move(t);
      ix->have_tree = true;
    }
    if (!load_partial_index(ix, dir, suffix, e)) {
      set_err(err, errlen, e);
      delete ix;
      return nullptr;
    }
  }
  if (ix->libs.empty()) {
    set_err(err, errlen, "no partial index found in the directory");
    delete ix;
    return nullptr;
  }
  // Index::make_rho_partial (src/index.cpp:188-201)
  double ratio_m = (double)ix->r_to_lib.size() / (double)ix->lsh.m;
  for (Lib& lib : ix->libs)
    for (double& r : lib.rho) r *= ratio_m; // CRecord::apply_rho_coef (src/record.cpp:304-309)
  ix->names.assign(ix->tree.nnodes + 1, "");
  ix->kind.assign(ix->tree.nnodes + 1, 0);
  for (uint32_t se = 1; se <= ix->tree.nnodes; ++se) {
    ix->names[se] = ix->tree.get_name(se);
    int nd = ix->tree.get_node(se);
    ix->kind[se] = nd < 0 ? 0 : (ix->tree.nodes[nd].is_leaf ? 1 : 2);
  }
  return ix;
}

void ko_index_free(ko_index* ix) { delete ix; }

// Replace the bucket table of one library with caller-supplied arrays in the on-disk layout
// (inc: cumulative ends; cmer: interleaved enc32,se).  Used by bench.py for the synthetic
// HBM-resident index, whose table is generated on the GPU and never written to disk.
int ko_index_replace_table(ko_index* ix, uint32_t lib, const uint64_t* inc, uint32_t nrows, const uint32_t* cmer,
                           uint64_t nkmers)
{
  if (!ix || lib >= ix->libs.size() || !inc || !cmer) return -1;
  Lib& L = ix->libs[lib];
  L.inc.assign(inc, inc + nrows);
  L.cmer.resize(nkmers);
  memcpy(L.cmer.data(), cmer, nkmers * 8);
  return 0;
}

void ko_index_info(const ko_index* ix, ko_info* o)
{
  memset(o, 0, sizeof(*o));
  o->k = ix->lsh.k, o->h = ix->lsh.h, o->m = ix->lsh.m;
  o->w = ix->libs.empty() ? 0 : ix->libs[0].w;
  o->nlibs = (uint32_t)ix->libs.size();
  o->nresidues = (uint32_t)ix->r_to_lib.size();
  o->nnodes = ix->tree.nnodes;
  for (uint32_t se = 1; se <= ix->tree.nnodes; ++se) o->nleaves += ix->kind[se] == 1;
  for (const Lib& l : ix->libs) o->nkmers += l.cmer.size(), o->nrows += l.inc.size();
  o->wbackbone = ix->wbackbone;
}

const char* ko_node_name(const ko_index* ix, uint32_t se) { return se < ix->names.size() ? ix->names[se].c_str() : ""; }
int ko_node_kind(const ko_index* ix, uint32_t se) { return se < ix->kind.size() ? ix->kind[se] : -1; }
uint32_t ko_node_parent(const ko_index* ix, uint32_t se)
{
  int nd = ix->tree.get_node(se);
  if (nd < 0 || ix->tree.nodes[nd].parent < 0) return 0;
  return ix->tree.nodes[ix->tree.nodes[nd].parent].se;
}
double ko_node_blen(const ko_index* ix, uint32_t se)
{
  int nd = ix->tree.get_node(se);
  return nd < 0 ? NAN : ix->tree.nodes[nd].blen;
}
void ko_lsh_positions(const ko_index* ix, uint8_t* ppos, uint8_t* npos)
{
  memcpy(ppos, ix->lsh.ppos.data(), ix->lsh.ppos.size());
  memcpy(npos, ix->lsh.npos.data(), ix->lsh.npos.size());
}

uint32_t ko_front_end(const ko_index* ix, const char* seq, uint64_t len, uint32_t* kpos, uint8_t* strand,
                      uint64_t* enc_bp, uint64_t* enc_lr, uint32_t* rix, uint32_t* enc32, uint8_t* pass)
{ // the loop of src/query.cpp:48-93 with the probe replaced by a tap
  const Lsh& lsh = ix->lsh;
  uint32_t k = lsh.k, n = 0;
  uint64_t u64m = ~0ull;
  uint64_t mask_lr = ((u64m >> (64 - k)) << 32) + ((u64m << 32) >> (64 - k));
  uint64_t mask_bp = u64m >> ((32 - k) * 2);
  uint32_t i, l;
  uint64_t bp = 0, lr = 0;
  for (i = l = 0; i < len;) {
    if (nt4((unsigned char)seq[i]) >= 4) {
      l = 0, i++;
      continue;
    }
    l++, i++;
    if (l < k) continue;
    if (l == k)
      compute_encoding(seq + i - k, seq + i, lr, bp);
    else
      update_encoding(seq + i - 1, lr, bp);
    bp &= mask_bp;
    lr &= mask_lr;
    uint64_t rc = revcomp_bp64(bp, k);
    uint64_t rclr = conv_bp64_lr64(rc);
    kpos[n] = i - k, strand[n] = 0, enc_bp[n] = bp, enc_lr[n] = lr;
    rix[n] = lsh.compute_hash(bp), enc32[n] = lsh.drop_ppos_lr(lr);
    pass[n] = ix->r_to_lib.count(rix[n] % lsh.m) != 0;
    n++;
    kpos[n] = i - k, strand[n] = 1, enc_bp[n] = rc, enc_lr[n] = rclr;
    rix[n] = lsh.compute_hash(rc), enc32[n] = lsh.drop_ppos_lr(rclr);
    pass[n] = ix->r_to_lib.count(rix[n] % lsh.m) != 0;
    n++;
  }
  return n;
}

// QueryIndex::estimate_distances (src/krepp.cpp:347-394): one task per batch of
// RBATCH_SIZE*DSEQ_LEN bases (src/rqseq.hpp:10-11,139); here batches of 512 reads.
static int dist_batch_impl(const ko_index* ix, const char* bases, const uint64_t* offsets, const char* const* names,
                           uint32_t nreads, const ko_params* p, ko_result* out, bool summarize)
{
  memset(out, 0, sizeof(*out));
  if (p->hdist_th > 16) return -1;
  const uint32_t B = 512;
  uint32_t nbatch = (nreads + B - 1) / B;
  std::vector<Worker*> parts(nbatch, nullptr);
  std::vector<ko_readinfo> rinfo(nreads);
  int nt = p->num_threads ? (int)p->num_threads : 1;
  (void)nt;
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
#endif
  for (int64_t b = 0; b < (int64_t)nbatch; ++b) {
    Worker* w = new Worker(ix, *p);
    w->summarize = summarize;
    uint32_t r0 = (uint32_t)b * B, r1 = std::min(nreads, r0 + B);
    for (uint32_t r = r0; r < r1; ++r) {
      w->read_ix = r;
      memset(&rinfo[r], 0, sizeof(ko_readinfo));
      w->run_read(bases + offsets[r], offsets[r + 1] - offsets[r], names ? names[r] : nullptr, rinfo[r]);
    }
    parts[b] = w;
  }
  std::map<uint32_t, double> wcount; // QueryIndex::estimate_distances merge (src/krepp.cpp:374-378)
  double twcount = 0;
  std::vector<ko_row> rows;
  std::vector<ko_acc> accs;
  std::vector<ko_hit> hits;
  std::string text;
  for (Worker* w : parts) {
    rows.insert(rows.end(), w->rows.begin(), w->rows.end());
    accs.insert(accs.end(), w->accs.begin(), w->accs.end());
    hits.insert(hits.end(), w->hits.begin(), w->hits.end());
    text += w->text;
    for (auto& kv : w->node_to_wcount) {
      twcount += kv.second;
      wcount[kv.first] += kv.second;
    }
    add_counters(out->counters, w->c);
    delete w;
  }
  if (summarize) { // src/krepp.cpp:388-393, ascending colour id instead of hash-map order
    char b1[64], b2[64];
    for (auto& kv : wcount) {
      snprintf(b1, sizeof(b1), "%.5f", kv.second);
      snprintf(b2, sizeof(b2), "%.5f", kv.second / twcount);
      text += ix->names[kv.first] + "\t" + b1 + "\t" + b2 + "\n";
    }
  }
  out->nrows = rows.size(), out->naccs = accs.size(), out->nhits = hits.size();
  out->rows = dup_vec(rows), out->accs = dup_vec(accs), out->hits = dup_vec(hits);
  out->reads = dup_vec(rinfo);
  out->text_len = text.size();
  out->text = (char*)malloc(text.size() + 1);
  memcpy(out->text, text.c_str(), text.size() + 1);
  return 0;
}

int ko_dist_batch(const ko_index* ix, const char* bases, const uint64_t* offsets, const char* const* names, uint32_t nreads,
                  const ko_params* p, ko_result* out)
{
  return dist_batch_impl(ix, bases, offsets, names, nreads, p, out, false);
}

// `krepp dist --summarize` over the whole input given as one batch (src/krepp.cpp:374-393)
int ko_dist_summarize(const ko_index* ix, const char* bases, const uint64_t* offsets, uint32_t nreads, const ko_params* p,
                      ko_result* out)
{
  return dist_batch_impl(ix, bases, offsets, nullptr, nreads, p, out, true);
}

void ko_result_free(ko_result* r)
{
  free(r->rows), free(r->accs), free(r->hits), free(r->reads), free(r->text), free(r->placements);
  memset(r, 0, sizeof(*r));
}

double ko_llh(uint32_t k, uint32_t h, uint32_t th, const double* hist, double uc, double rho, double d)
{
  Llh f(h, k, th);
  f.set_parameters(hist, uc, rho);
  return f(d);
}

int ko_brent(uint32_t k, uint32_t h, uint32_t th, const double* hist, double uc, double rho, double* d_out,
             double* v_out)
{
  Llh f(h, k, th);
  f.set_parameters(hist, uc, rho);
  uint64_t ne = 0;
  auto sol = brent_find_minima(f, 1e-10, 0.5, 16, &ne);
  *d_out = sol.first;
  *v_out = sol.second;
  return (int)ne;
}

uint64_t ko_revcomp_bp64(uint64_t x, uint32_t k) { return revcomp_bp64(x, k); }
uint64_t ko_conv_bp64_lr64(uint64_t x) { return conv_bp64_lr64(x); }
uint32_t ko_murmur3_x86_32(const void* key, int len, uint32_t seed) { return murmur3_x86_32(key, len, seed); }
uint64_t ko_name_hash(const char* name) { return name_hash(name); }
uint64_t ko_xur64(uint64_t h) { return xur64_hash(h); }

} // extern "C"

namespace {
void map_to_qtree(ko_index* ix, Tree&& q);
void compute_card(Tree& t, int nd)
{
  TNode& n = t.nodes[nd];
  if (n.is_leaf) {
    n.card = 1;
    return;
  }
  n.card = 0;
  for (int c : n.children) {
    compute_card(t, c);
    n.card += t.nodes[c].card;
  }
}

// Tree::stream_nwk_jplace (src/phytree.cpp:47-66): Newick with {edge} numbers, lengths at precision 5
void nwk_jplace(const Tree& t, int nd, std::string& o)
{
  const TNode& n = t.nodes[nd];
  if (!n.is_leaf) {
    o += "(";
    for (size_t i = 0; i < n.children.size(); ++i) {
      nwk_jplace(t, n.children[i], o);
      if (i + 1 < n.children.size()) o += ",";
    }
    o += ")";
  }
  o += n.name; // Node::stream_nwk_entry (src/phytree.hpp:146-153)
  if (!std::isnan(n.blen)) {
    char b[64];
    snprintf(b, sizeof(b), ":%.5f", n.blen);
    o += b;
  }
  o += "{" + std::to_string(n.se - 1) + "}";
  if (nd == t.root) o += ";";
}
} // namespace

extern "C" {

int ko_index_set_placement_tree(ko_index* ix, const char* nwk_text, char* err, int errlen)
{
  uint32_t nn = ix->tree.nnodes;
  ix->se_to_pnode.assign(nn + 1, -1);
  if (!nwk_text) { // TargetIndex::ensure_backbone without -t (src/krepp.cpp:59-63)
    if (!ix->wbackbone) {
      set_err(err, errlen, "Given index lacks a tree and no backbone tree is provided...");
      return -1;
    }
    ix->ptree = ix->tree;
    for (uint32_t se = 1; se <= nn; ++se) {
      int nd = ix->tree.get_node(se);
      if (nd >= 0 && ix->tree.nodes[nd].is_leaf) ix->se_to_pnode[se] = nd;
      ix->kind[se] = nd < 0 ? 0 : (ix->tree.nodes[nd].is_leaf ? 1 : 2); // undo an earlier mapping onto another tree
    }
    for (auto& n : ix->ptree.nodes) n.eff_nchildren = (uint32_t)n.children.size(); // Node::add_children
  } else {
    Tree q;
    if (!q.load(nwk_text)) {
      set_err(err, errlen, q.err);
      return -1;
    }
    map_to_qtree(ix, std::move(q));
  }
  compute_card(ix->ptree, ix->ptree.root);
  ix->have_ptree = true;
  return 0;
}

// TargetIndex::read_lineages (src/krepp.cpp:37-46): the placement tree is the taxonomy of a lineage file
int ko_index_set_lineage_tree(ko_index* ix, const char* lineage_text, char* err, int errlen)
{
  ix->se_to_pnode.assign(ix->tree.nnodes + 1, -1);
  Tree q;
  if (!lineage_text || !q.parse_lineages(lineage_text)) {
    set_err(err, errlen, lineage_text ? q.err : "Error opening the lineage file");
    return -1;
  }
  map_to_qtree(ix, std::move(q)); // Node::card stays as parse_lineages left it
  ix->have_ptree = true;
  return 0;
}

} // extern "C"

namespace {
// Tree::map_to_qtree (src/phytree.cpp:421-450) + compute_eff_nchildren (:452-473)
void map_to_qtree(ko_index* ix, Tree&& q)
{
  const uint32_t nn = ix->tree.nnodes;
  {
    std::map<std::string, uint32_t> name_to_se;
    for (uint32_t se = 1; se <= nn; ++se) {
      int nd = ix->tree.get_node(se);
      if (nd >= 0 && ix->tree.nodes[nd].is_leaf) {
        name_to_se[ix->tree.nodes[nd].name] = se;
        ix->kind[se] = 0; // se_to_node[se] = nullptr until a query-tree leaf claims it
      }
    }
    for (size_t nd = 0; nd < q.nodes.size(); ++nd) {
      const TNode& n = q.nodes[nd];
      if (n.is_leaf && !n.name.empty()) {
        auto it = name_to_se.find(n.name);
        if (it != name_to_se.end()) {
          ix->se_to_pnode[it->second] = (int)nd;
          ix->kind[it->second] = 1;
        }
      }
    }
    std::vector<char> covered(q.nodes.size(), 0);
    for (uint32_t se = 1; se <= nn; ++se) {
      int a = ix->se_to_pnode[se];
      while (a >= 0 && !covered[a]) {
        covered[a] = 1;
        a = q.nodes[a].parent;
      }
    }
    for (auto& n : q.nodes) n.eff_nchildren = 0;
    for (size_t nd = 0; nd < q.nodes.size(); ++nd)
      if (covered[nd] && q.nodes[nd].parent >= 0) q.nodes[q.nodes[nd].parent].eff_nchildren++;
    ix->ptree = std::move(q);
  }
}
} // namespace

extern "C" {

// QueryIndex::place_sequences (src/krepp.cpp:434-504): batch texts joined with ",\n" (jplace) or
// concatenated (tabular); within a batch IBatch::place_sequences (src/query.cpp:198-216).
int ko_place_batch(const ko_index* ix, const char* bases, const uint64_t* offsets, const char* const* names, uint32_t nreads,
                   const ko_params* p, int tabular, ko_result* out)
{
  memset(out, 0, sizeof(*out));
  if (!ix->have_ptree || p->hdist_th > 16) return -1;
  const uint32_t B = 512;
  uint32_t nbatch = (nreads + B - 1) / B;
  std::vector<Worker*> parts(nbatch, nullptr);
  std::vector<ko_readinfo> rinfo(nreads);
  int nt = p->num_threads ? (int)p->num_threads : 1;
  (void)nt;
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
#endif
  for (int64_t b = 0; b < (int64_t)nbatch; ++b) {
    Worker* w = new Worker(ix, *p);
    w->place_mode = true;
    w->tabular = tabular != 0;
    uint32_t r0 = (uint32_t)b * B, r1 = std::min(nreads, r0 + B);
    for (uint32_t r = r0; r < r1; ++r) {
      w->read_ix = r;
      memset(&rinfo[r], 0, sizeof(ko_readinfo));
      w->run_read(bases + offsets[r], offsets[r + 1] - offsets[r], names ? names[r] : nullptr, rinfo[r]);
    }
    parts[b] = w;
  }
  std::vector<ko_placement> pls;
  std::string text;
  bool has_previous = false;
  for (Worker* w : parts) {
    pls.insert(pls.end(), w->placements.begin(), w->placements.end());
    if (tabular) {
      text += w->text;
    } else if (!w->text.empty()) {
      if (has_previous) text += ",\n";
      text += w->text;
      has_previous = true;
    }
    add_counters(out->counters, w->c);
    delete w;
  }
  out->nplacements = pls.size();
  out->placements = dup_vec(pls);
  out->reads = dup_vec(rinfo);
  out->text_len = text.size();
  out->text = (char*)malloc(text.size() + 1);
  memcpy(out->text, text.c_str(), text.size() + 1);
  return 0;
}

// place --summarize (src/query.cpp:232-233,297-298,322-323; src/krepp.cpp:466-471,493-497): per placement-tree
// node the number of reads placed there, a read with n placements counting 1/n for each.  Batches of 512 reads
// summed in input order; rows in ascending edge number (the reference's row order is a hash map's).
int ko_place_summarize(const ko_index* ix, const char* bases, const uint64_t* offsets, uint32_t nreads, const ko_params* p,
                       ko_result* out)
{
  memset(out, 0, sizeof(*out));
  if (!ix->have_ptree || p->hdist_th > 16) return -1;
  const uint32_t B = 512;
  std::map<uint32_t, double> total;
  double twcount = 0;
  for (uint32_t r0 = 0; r0 < nreads; r0 += B) {
    Worker w(ix, *p);
    w.place_mode = true;
    w.summarize = true;
    ko_readinfo ri;
    for (uint32_t r = r0; r < std::min(nreads, r0 + B); ++r) {
      w.read_ix = r;
      memset(&ri, 0, sizeof(ri));
      w.run_read(bases + offsets[r], offsets[r + 1] - offsets[r], nullptr, ri);
    }
    for (auto& kv : w.pnode_to_wcount) twcount += kv.second, total[kv.first] += kv.second;
    add_counters(out->counters, w.c);
  }
  std::string text;
  for (auto& kv : total) {
    const TNode& n = ix->ptree.nodes[ix->ptree.get_node(kv.first)];
    text += (n.name.empty() ? std::string("NA") : n.name) + "\t" + std::to_string(kv.first - 1) + "\t" + Worker::f5(kv.second) + "\t" +
            Worker::f5(kv.second / twcount) + "\n";
  }
  out->text_len = text.size();
  out->text = (char*)malloc(text.size() + 1);
  memcpy(out->text, text.c_str(), text.size() + 1);
  return 0;
}

char* ko_place_frame(const ko_index* ix, int which, int tabular, const char* invocation, uint64_t total_qseq)
{
  std::string o, tree;
  nwk_jplace(ix->ptree, ix->ptree.root, tree);
  std::string inv = invocation ? invocation : "";
  if (which == 0) {
    if (tabular == 2) // --summarize
      o = "# software: krepp\tversion: v0.8.3\tinvocation :" + inv + "\n# " + tree +
          "\nDISTAL_NODE\tEDGE_NUM\tWEIGHTED_COUNT\tSEQUENCE_ABUNDANCE\n";
    else if (tabular) // QueryIndex::header_preport (src/krepp.cpp:396-408)
      o = "# software: krepp\tversion: v0.8.3\tinvocation :" + inv + "\n# " + tree + "\nSEQ_ID\tDISTAL_NODE\tEDGE_NUM\tLWR\tDIST\n";
    else // begin_jplace (src/krepp.cpp:426-432)
      o = "{\n\t\"version\" : 3,\n\t\"fields\" : [\"edge_num\", \"pendant_length\", \"distal_length\", \"likelihood\", "
          "\"like_weight_ratio\", \"distance\"],\n\t\"placements\" : [\n";
  } else if (!tabular) { // end_jplace (src/krepp.cpp:410-424)
    o = "],\n\t\"metadata\" : {\n\t\t\"software\" : \"krepp\",\n\t\t\"version\" : \"v0.8.3\",\n\t\t\"repository\" : "
        "\"https://github.com/bo1929/krepp\",\n\t\t\"num_queries\" : \"" + std::to_string(total_qseq) + "\",\n\t\t\"invocation\" : \"" +
        inv + "\"\n\t},\n\t\"tree\" : \"" + tree + "\"\n}";
  }
  char* r = (char*)malloc(o.size() + 1);
  memcpy(r, o.c_str(), o.size() + 1);
  return r;
}

// ---------------------------------------------------------------------------
// `krepp seek` (src/seek.cpp:1-126, src/sketch.cpp:3-39, src/sketch.hpp:18-22): a single-reference sketch.
// ---------------------------------------------------------------------------
struct ko_sketch {
  uint64_t nkmers = 0;
  uint32_t nrows = 0, m = 0, r = 0, nrows_cfg = 0;
  uint8_t k = 0, w = 0, h = 0;
  bool frac = false;
  double rho = 0;
  std::vector<uint32_t> enc_v; // SFlatHT (src/table.cpp:24-33)
  std::vector<uint64_t> inc_v;
  Lsh lsh;
};

ko_sketch* ko_sketch_load(const char* path, char* err, int errlen)
{ // Sketch::load_full_sketch (src/sketch.cpp:3-24) + make_rho_partial (:26-33)
  std::ifstream f(path, std::ifstream::binary);
  std::unique_ptr<ko_sketch> sk(new ko_sketch());
  bool ok = f.good() && rd(f, sk->nkmers);
  if (ok) {
    sk->enc_v.resize(sk->nkmers);
    f.read((char*)sk->enc_v.data(), (std::streamsize)(sk->nkmers * 4));
    ok = f.good() && rd(f, sk->nrows);
  }
  if (ok) {
    sk->inc_v.resize(sk->nrows);
    f.read((char*)sk->inc_v.data(), (std::streamsize)(sk->nrows * 8ull));
    ok = f.good() && rd(f, sk->k) && rd(f, sk->w) && rd(f, sk->h) && rd(f, sk->m) && rd(f, sk->r) && rd(f, sk->frac) &&
         rd(f, sk->nrows_cfg);
  }
  if (ok && sk->h <= sk->k) {
    sk->lsh.ppos.resize(sk->h);
    sk->lsh.npos.resize(sk->k - sk->h);
    f.read((char*)sk->lsh.ppos.data(), sk->h);
    f.read((char*)sk->lsh.npos.data(), sk->k - sk->h);
    ok = f.good() && rd(f, sk->rho);
  } else
    ok = false;
  if (!ok) {
    set_err(err, errlen, "Failed to read the sketch file!");
    return nullptr;
  }
  sk->lsh.m = sk->m;
  sk->lsh.set();
  if (sk->frac)
    sk->rho *= ((double)sk->r + 1.0) / (double)sk->m;
  else
    sk->rho *= 1.0 / (double)sk->m;
  return sk.release();
}
void ko_sketch_free(ko_sketch* sk) { delete sk; }
void ko_sketch_info(const ko_sketch* sk, uint64_t* nkmers, uint32_t* nrows, uint32_t* k, uint32_t* w, uint32_t* h, uint32_t* m,
                    uint32_t* r, uint32_t* frac, double* rho)
{
  *nkmers = sk->nkmers, *nrows = sk->nrows, *k = sk->k, *w = sk->w, *h = sk->h, *m = sk->m, *r = sk->r, *frac = sk->frac;
  *rho = sk->rho;
}
const uint32_t* ko_sketch_codes(const ko_sketch* sk) { return sk->enc_v.data(); }
const uint64_t* ko_sketch_inc(const ko_sketch* sk) { return sk->inc_v.data(); }
void ko_sketch_positions(const ko_sketch* sk, uint8_t* ppos, uint8_t* npos)
{
  memcpy(ppos, sk->lsh.ppos.data(), sk->lsh.ppos.size());
  memcpy(npos, sk->lsh.npos.data(), sk->lsh.npos.size());
}

// SBatch::seek_sequences (src/seek.cpp:22-56) over all reads; out->text = `SEQ_ID\tDIST` / `SEQ_ID\tNaN` rows,
// out->rows: one row per read (se = 1 if a distance was reported, else 0).
int ko_seek_batch(const ko_sketch* sk, const char* bases, const uint64_t* offsets, const char* const* names, uint32_t nreads,
                  uint32_t hdist_th, ko_result* out)
{
  memset(out, 0, sizeof(*out));
  if (hdist_th > 16) return -1;
  const uint32_t k = sk->k;
  const Lsh& lsh = sk->lsh;
  Llh llhfunc(sk->h, k, hdist_th);
  const uint64_t u64m = std::numeric_limits<uint64_t>::max();
  const uint64_t mask_lr = ((u64m >> (64 - k)) << 32) + ((u64m << 32) >> (64 - k));
  const uint64_t mask_bp = u64m >> ((32 - k) * 2);
  struct SSummary { // src/seek.hpp:18-40
    double mismatch_count, match_count = 0, d_llh = NAN, v_llh = NAN;
    std::vector<double> hdisthist_v;
  };
  auto check_partial = [&](uint32_t rix) { // src/sketch.hpp:18-22
    const uint32_t rix_res = rix % sk->m;
    return (sk->frac && (rix_res <= sk->r)) || (rix_res == sk->r);
  };
  auto add_matching_mer = [&](SSummary& su, uint32_t rix, uint32_t enc_lr) { // src/seek.cpp:104-121, src/sketch.cpp:35-39
    const uint32_t rix_res = rix % sk->m;
    const uint32_t offset = sk->frac ? (rix / sk->m) * (sk->r + 1) + rix_res : rix / sk->m;
    uint64_t b0 = offset ? sk->inc_v[offset - 1] : 0, b1 = sk->inc_v[offset];
    uint32_t hdist_min = hdist_th + 1;
    for (; b0 < b1; ++b0) {
      const uint32_t hdist_curr = popcount_lr32(sk->enc_v[b0] ^ enc_lr);
      if (hdist_curr < hdist_min) hdist_min = hdist_curr;
    }
    if (hdist_min <= hdist_th) {
      su.mismatch_count--;
      su.match_count++;
      su.hdisthist_v[hdist_min]++;
    }
  };
  std::vector<ko_row> rows(nreads);
  std::string text;
  uint64_t ne = 0;
  for (uint32_t bix = 0; bix < nreads; ++bix) {
    const char* seq = bases + offsets[bix];
    const uint64_t len = offsets[bix + 1] - offsets[bix];
    uint64_t onmers = 0;
    SSummary or_summary, rc_summary;
    or_summary.mismatch_count = rc_summary.mismatch_count = (double)(len - k + 1);
    or_summary.hdisthist_v.assign(hdist_th + 1, 0.0);
    rc_summary.hdisthist_v.assign(hdist_th + 1, 0.0);
    { // SBatch::search_mers (src/seek.cpp:58-102), non-CANONICAL build
      uint32_t i, l;
      uint64_t orenc64_bp = 0, orenc64_lr = 0, rcenc64_bp;
      for (i = l = 0; i < len;) {
        if (nt4((unsigned char)seq[i]) >= 4) {
          l = 0, i++;
          continue;
        }
        l++, i++;
        if (l < k) continue;
        if (l == k)
          compute_encoding(seq + i - k, seq + i, orenc64_lr, orenc64_bp);
        else
          update_encoding(seq + i - 1, orenc64_lr, orenc64_bp);
        orenc64_bp &= mask_bp;
        orenc64_lr &= mask_lr;
        rcenc64_bp = revcomp_bp64(orenc64_bp, k);
        onmers++;
        const uint32_t orrix = lsh.compute_hash(orenc64_bp);
        if (check_partial(orrix)) add_matching_mer(or_summary, orrix, lsh.drop_ppos_lr(orenc64_lr));
        const uint32_t rcrix = lsh.compute_hash(rcenc64_bp);
        if (check_partial(rcrix)) add_matching_mer(rc_summary, rcrix, lsh.drop_ppos_lr(conv_bp64_lr64(rcenc64_bp)));
      }
    }
    ko_row& row = rows[bix];
    memset(&row, 0, sizeof(row));
    row.read = bix;
    const std::string id = names ? names[bix] : "";
    if (or_summary.match_count + rc_summary.match_count) {
      or_summary.mismatch_count = (double)onmers - or_summary.match_count;
      rc_summary.mismatch_count = (double)onmers - rc_summary.match_count;
      for (SSummary* su : {&or_summary, &rc_summary}) { // SSummary::optimize_likelihood (src/seek.cpp:122-128)
        llhfunc.set_parameters(su->hdisthist_v.data(), su->mismatch_count, sk->rho);
        auto sol = brent_find_minima(llhfunc, 1e-10, 0.5, 16, &ne);
        su->d_llh = sol.first, su->v_llh = sol.second;
      }
      const SSummary& best = or_summary.d_llh < rc_summary.d_llh ? or_summary : rc_summary;
      row.se = 1, row.d_llh = best.d_llh;
      text += id + "\t" + Worker::f5(best.d_llh) + "\n";
    } else {
      row.d_llh = NAN;
      text += id + "\tNaN\n";
    }
  }
  out->nrows = nreads;
  out->rows = dup_vec(rows);
  out->text_len = text.size();
  out->text = (char*)malloc(text.size() + 1);
  memcpy(out->text, text.c_str(), text.size() + 1);
  return 0;
}

} // extern "C"
