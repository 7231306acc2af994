// kr_place.cpp — back end of `krepp place` on top of the dist front end.
//
// Reference: IBatch::report_placement (src/query.cpp:218-333), Minfo::add / get_leq_tau /
// jukes_cantor_dist (src/query.hpp:139-152,189-197), macros PP_JPLACE_FIELDS / PP_TABULAR_FIELDS
// (src/query.hpp:202-206), jplace framing (src/krepp.cpp:396-432), placement tree set-up
// (src/krepp.cpp:48-64, src/phytree.cpp:421-473), lineage trees (src/krepp.cpp:37-46,
// src/phytree.cpp:320-369), --summarize (src/query.cpp:232-233,297-298,322-323; src/krepp.cpp:493-497).
//
// Split of work: the GPU has already produced, per read, one record per (leaf, strand) with its
// histogram, distance and likelihood (kr_scan/acc/llh/select kernels).  Here the host walks each leaf's
// ancestors accumulating weighted histograms (a few hundred adds per read), then ALL likelihood work
// of the batch — Brent on every candidate internal node, and the chi-square evaluation of every
// candidate against the read's closest leaf — goes back to the GPU as two kr_llh_batch launches.
#include "kr_common.h"

#include <algorithm>
#include <limits>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <chrono>
#include <functional>
#include <string>
#include <vector>

struct kr_place_tree {
  kr::HostTree t;                 // placement tree, se = post-order number, edge = se - 1
  std::vector<std::vector<uint32_t>> kids;
  std::vector<uint32_t> card, eff;
  std::vector<uint32_t> idx_to_pt; // index colour id of a leaf -> se in the placement tree (0 = absent)
  std::vector<uint8_t> kinds;     // node_kind array for kr_index_upload
  uint32_t root = 0;
  // flat arrays for the device back end (kr::place_on_device): parent, subtree start, candidate eligibility
  std::vector<uint32_t> parent_arr, lo, depth; // depth: number of ancestors (post-order numbering: parents after children)
  std::vector<uint8_t> elig;
  bool postorder = false; // every subtree is the node range [lo[q], q]: what the device kernel's ancestor listing needs
  // for the rows as text on the device (round 6): branch lengths, labels back to back
  std::vector<double> blen_arr;
  std::string label_blob;
  std::vector<uint32_t> label_off; // [pn + 2]
};

namespace {

std::string f5(double v)
{ // std::fixed, precision 5 (src/krepp.cpp:438-439)
  char b[64];
  if (std::isnan(v)) return std::signbit(v) ? "-nan" : "nan";
  snprintf(b, sizeof(b), "%.5f", v);
  return b;
}

void nwk_jplace(const kr_place_tree& pt, uint32_t se, std::string& o)
{ // Tree::stream_nwk_jplace (src/phytree.cpp:47-66)
  const kr::TreeNode& n = pt.t.nodes[se];
  if (n.kind == 2) {
    o += "(";
    for (size_t i = 0; i < pt.kids[se].size(); ++i) {
      nwk_jplace(pt, pt.kids[se][i], o);
      if (i + 1 < pt.kids[se].size()) o += ",";
    }
    o += ")";
  }
  o += n.label;
  if (!std::isnan(n.blen)) o += ":" + f5(n.blen);
  o += "{" + std::to_string(se - 1) + "}";
  if (se == pt.root) o += ";";
}

constexpr uint32_t kMaxNp = 17; // hdist_th <= 16

struct Acc { // the fields of Minfo that placement needs
  double nmers = 0, mismatch = 0, match = 0, rho = 0;
  double hist[kMaxNp];
  double d = 1.7976931348623157e308, v = NAN, chisq = NAN, lwr = 1;
  bool leaf = false;
  void zero(uint32_t np)
  {
    nmers = mismatch = match = rho = 0;
    for (uint32_t x = 0; x < np; ++x) hist[x] = 0;
    d = 1.7976931348623157e308, v = NAN, chisq = NAN, lwr = 1, leaf = false;
  }
  void add(const Acc& m, double denom, uint32_t np)
  { // Minfo::add (src/query.hpp:139-152)
    mismatch = nmers ? mismatch : m.nmers;
    match += m.match * denom;
    mismatch -= m.match * denom;
    for (uint32_t x = 0; x < np; ++x) hist[x] = hist[x] + m.hist[x] * denom;
    nmers = std::max(nmers, m.nmers);
    rho = std::max(rho, m.rho);
  }
  double leq_tau(uint32_t tau, uint32_t np) const
  {
    double s = 0;
    for (uint32_t x = 0; x <= tau && x < np; ++x) s += hist[x];
    return s;
  }
  double jc() const { return -0.75 * log(1 - 4.0 / 3.0 * d); }
};

struct Cand {
  uint32_t read, se; // placement-tree node
  Acc a;
  bool internal;
};

void map_leaves(const kr_host_index* hx, const kr_index_view& v, kr_place_tree& pt);
bool lineage_tree(const char* text, kr_place_tree& pt, std::string& err);
void finish_tree(kr_place_tree& pt, const kr_index_view& v, bool mapped, bool derive_card);

char* dup_text(const std::string& s, uint64_t* len)
{
  char* p = (char*)malloc(s.size() + 1);
  if (p) memcpy(p, s.c_str(), s.size() + 1);
  *len = s.size();
  return p;
}

} // namespace

// kr_place_stream batches of this process by back end (kr_place_counters): on the device, or sent whole to the host path
static std::atomic<uint64_t> g_place_device_batches{0}, g_place_host_batches{0}, g_place_heavy_reads{0};
std::atomic<uint64_t> g_place_text_device{0}, g_place_text_fallback{0};

extern "C" {

int kr_place_tree_create(const kr_host_index* hx, const char* nwk_text, kr_place_tree** out)
{
  kr::clear_error();
  if (!hx || !out) return kr::fail(KR_ERR_ARG, "kr_place_tree_create: null argument");
  kr_index_view v;
  kr_host_index_view(hx, &v);
  std::unique_ptr<kr_place_tree> pt(new kr_place_tree());
  const uint32_t nn = v.tree_nnodes;
  pt->idx_to_pt.assign(nn + 1, 0);
  pt->kinds.assign(v.node_kind, v.node_kind + nn + 1);
  if (!nwk_text) { // TargetIndex::ensure_backbone (src/krepp.cpp:59-63)
    if (!v.wbackbone) return kr::fail(KR_ERR_ARG, "Given index lacks a tree and no backbone tree is provided...");
    pt->t.nodes.resize(nn + 1);
    pt->t.nodes[0] = kr::TreeNode{"", NAN, 0, 0};
    for (uint32_t se = 1; se <= nn; ++se) {
      pt->t.nodes[se] = kr::TreeNode{kr_host_index_node_label(hx, se), kr_host_index_node_blen(hx, se),
                                     kr_host_index_node_parent(hx, se), v.node_kind[se]};
      if (v.node_kind[se] == 1) pt->idx_to_pt[se] = se;
    }
  } else { // Tree::map_to_qtree (src/phytree.cpp:421-450)
    std::string err;
    if (!kr::parse_newick(nwk_text, pt->t, err)) return kr::fail(KR_ERR_FORMAT, err);
    map_leaves(hx, v, *pt);
  }
  finish_tree(*pt, v, nwk_text != nullptr, true);
  *out = pt.release();
  return KR_OK;
}

int kr_place_tree_create_lineage(const kr_host_index* hx, const char* lineage_text, kr_place_tree** out)
{
  kr::clear_error();
  if (!hx || !lineage_text || !out) return kr::fail(KR_ERR_ARG, "kr_place_tree_create_lineage: null argument");
  kr_index_view v;
  kr_host_index_view(hx, &v);
  std::unique_ptr<kr_place_tree> pt(new kr_place_tree());
  std::string err;
  if (!lineage_tree(lineage_text, *pt, err)) return kr::fail(KR_ERR_FORMAT, err);
  map_leaves(hx, v, *pt);
  finish_tree(*pt, v, true, false);
  *out = pt.release();
  return KR_OK;
}

} // extern "C"

namespace {

// Tree::map_to_qtree (src/phytree.cpp:421-450): labelled leaves of the placement tree claim the index
// leaves of the same name; every other index leaf becomes a null node
void map_leaves(const kr_host_index* hx, const kr_index_view& v, kr_place_tree& pt)
{
  const uint32_t nn = v.tree_nnodes;
  pt.idx_to_pt.assign(nn + 1, 0);
  pt.kinds.assign(v.node_kind, v.node_kind + nn + 1);
  std::map<std::string, uint32_t> name_to_se;
  for (uint32_t se = 1; se <= nn; ++se)
    if (v.node_kind[se] == 1) {
      name_to_se[kr_host_index_node_name(hx, se)] = se;
      pt.kinds[se] = 0;
    }
  for (uint32_t q = 1; q <= pt.t.nnodes(); ++q) {
    const kr::TreeNode& n = pt.t.nodes[q];
    if (n.kind == 1 && !n.label.empty()) {
      auto it = name_to_se.find(n.label);
      if (it != name_to_se.end()) {
        pt.idx_to_pt[it->second] = q;
        pt.kinds[it->second] = 1;
      }
    }
  }
}

// Tree::parse_lineages (src/phytree.cpp:320-369).  A line is `ID <tab> lineage`; the lineage is a list of
// `r__Taxon` separated by ';' (a blank after the ';' is dropped first).  Taxa are nodes shared by name, each
// under the parent it was first seen with; what has no parent at the end hangs below a node "root" (the
// reference attaches those in hash-map order, this build in order of first appearance).  No branch lengths.
// Numbering is post-order with children in attachment order (src/phytree.cpp:261-298).  Node::card is summed
// into the parent at attachment time (src/phytree.hpp:107-116), so a taxon contributes the leaves attached to
// it SO FAR -- none when it is created; the quirk only feeds the --no-multi tie order and is kept.
bool lineage_tree(const char* text, kr_place_tree& pt, std::string& err)
{
  struct LNode {
    std::string name;
    int parent = -1;
    std::vector<int> kids;
    uint32_t card = 0;
  };
  std::vector<LNode> nd(1);
  nd[0].name = "root";
  std::map<std::string, int> by_name;
  auto hang = [&](int c, int par) {
    nd[c].parent = par;
    nd[par].kids.push_back(c);
    nd[par].card += nd[c].card;
  };
  auto line_end = [](char c) { return c == '\n' || c == '\r'; };
  const char* p = text;
  while (*p) {
    const char* e = strchr(p, '\n');
    std::string raw = e ? std::string(p, e) : std::string(p);
    p = e ? e + 1 : p + raw.size();
    std::string line; // regex "; " -> ";" in one pass
    for (size_t i = 0; i < raw.size(); ++i) {
      line += raw[i];
      if (raw[i] == ';' && i + 1 < raw.size() && raw[i + 1] == ' ') ++i;
    }
    // two std::getline(.., '\t'): the second fails when nothing follows the first tab
    const size_t t1 = line.find('\t');
    if (line.empty() || t1 == std::string::npos || t1 + 1 == line.size()) {
      err = "Failed to reference to lineage mapping!";
      return false;
    }
    const std::string id = line.substr(0, t1);
    size_t t2 = line.find('\t', t1 + 1);
    if (t2 == std::string::npos) t2 = line.size();
    int parent = -1;
    for (size_t a = t1 + 1; a < t2;) { // std::getline(lss, taxon, ';'): no piece after a trailing ';'
      size_t b = line.find(';', a);
      if (b == std::string::npos || b > t2) b = t2;
      // regex ".__" -> "": drop every <char>"__" (the char not a line end), left to right, non-overlapping
      std::string taxon;
      for (size_t i = a; i < b;) {
        if (i + 2 < b && !line_end(line[i]) && line[i + 1] == '_' && line[i + 2] == '_')
          i += 3;
        else
          taxon += line[i++];
      }
      a = b + 1;
      if (taxon.empty()) continue;
      auto it = by_name.find(taxon);
      if (it == by_name.end()) {
        nd.emplace_back();
        const int c = (int)nd.size() - 1;
        nd[c].name = taxon;
        if (parent >= 0) hang(c, parent);
        it = by_name.emplace(taxon, c).first;
      }
      parent = it->second;
    }
    if (by_name.count(id)) {
      err = "The same reference appears more than once in the lineage file.";
      return false;
    }
    nd.emplace_back();
    const int c = (int)nd.size() - 1;
    nd[c].name = id;
    nd[c].card = 1;
    if (parent >= 0) hang(c, parent);
    by_name.emplace(id, c);
  }
  if (nd.size() == 1) {
    err = "The lineage file is empty.";
    return false;
  }
  for (int c = 1; c < (int)nd.size(); ++c)
    if (nd[c].parent < 0) hang(c, 0);
  // post-order numbers without recursion (lineages can be long chains)
  std::vector<uint32_t> se(nd.size(), 0);
  std::vector<std::pair<int, size_t>> stack{{0, 0}};
  uint32_t next = 0;
  while (!stack.empty()) {
    auto& top = stack.back();
    if (top.second < nd[top.first].kids.size()) {
      const int c = nd[top.first].kids[top.second++];
      stack.push_back({c, 0});
    } else {
      se[top.first] = ++next;
      stack.pop_back();
    }
  }
  pt.t.nodes.assign(nd.size() + 1, kr::TreeNode{"", NAN, 0, 0});
  pt.card.assign(nd.size() + 1, 0);
  for (size_t c = 0; c < nd.size(); ++c) {
    kr::TreeNode& n = pt.t.nodes[se[c]];
    n.label = nd[c].name;
    n.blen = NAN;
    n.parent = nd[c].parent < 0 ? 0 : se[nd[c].parent];
    n.kind = nd[c].kids.empty() ? 1 : 2; // a taxon left childless (its name reused under another parent) ends the path: the
                                          // reference's traversal would walk off an empty child list there (src/phytree.cpp:266-267)
    pt.card[se[c]] = nd[c].card;
  }
  return true;
}

void finish_tree(kr_place_tree& ptr, const kr_index_view& v, bool mapped, bool derive_card)
{
  kr_place_tree* pt = &ptr;
  const uint32_t nn = v.tree_nnodes;
  const uint32_t pn = pt->t.nnodes();
  pt->root = pn; // post-order: the root is numbered last
  pt->kids.assign(pn + 1, {});
  for (uint32_t se = 1; se <= pn; ++se)
    if (pt->t.nodes[se].parent) pt->kids[pt->t.nodes[se].parent].push_back(se);
  if (derive_card) {
    pt->card.assign(pn + 1, 0);
    for (uint32_t se = 1; se <= pn; ++se) { // children precede parents
      if (pt->t.nodes[se].kind == 1) pt->card[se] = 1;
      if (pt->t.nodes[se].parent) pt->card[pt->t.nodes[se].parent] += pt->card[se];
    }
  }
  // eff_nchildren: Node::add_children counts every child; after map_to_qtree only children with a
  // mapped leaf below (Tree::compute_eff_nchildren, src/phytree.cpp:452-473)
  pt->eff.assign(pn + 1, 0);
  if (!mapped) {
    for (uint32_t se = 1; se <= pn; ++se) pt->eff[se] = (uint32_t)pt->kids[se].size();
  } else {
    std::vector<char> covered(pn + 1, 0);
    for (uint32_t se = 1; se <= nn; ++se) {
      uint32_t a = pt->idx_to_pt[se];
      while (a && !covered[a]) {
        covered[a] = 1;
        a = pt->t.nodes[a].parent;
      }
    }
    for (uint32_t se = 1; se <= pn; ++se)
      if (covered[se] && pt->t.nodes[se].parent) pt->eff[pt->t.nodes[se].parent]++;
  }
  // flat arrays for the device back end
  pt->parent_arr.assign(pn + 1, 0), pt->lo.assign(pn + 1, 0), pt->elig.assign(pn + 1, 0);
  std::vector<uint32_t> size(pn + 1, 1);
  pt->postorder = true;
  for (uint32_t se = 1; se <= pn; ++se) {
    const uint32_t par = pt->t.nodes[se].parent;
    pt->parent_arr[se] = par;
    pt->lo[se] = se;
    const uint32_t nch = (uint32_t)pt->kids[se].size();
    pt->elig[se] = (nch == pt->eff[se] && nch != 1) ? 1 : 0; // src/query.cpp:270
    if (par && par <= se) pt->postorder = false;
  }
  pt->depth.assign(pn + 1, 0);
  if (pt->postorder)
    for (uint32_t se = pn; se >= 1; --se) { // parents have the larger numbers: depths flow downwards
      const uint32_t par = pt->parent_arr[se];
      pt->depth[se] = par ? pt->depth[par] + 1u : 0u;
    }
  if (pt->postorder) {
    for (uint32_t se = 1; se <= pn; ++se) { // children precede parents: sizes and range starts flow upwards
      const uint32_t par = pt->parent_arr[se];
      if (par) size[par] += size[se], pt->lo[par] = std::min(pt->lo[par], pt->lo[se]);
    }
    for (uint32_t se = 1; se <= pn; ++se)
      if (se - pt->lo[se] + 1 != size[se]) pt->postorder = false; // the subtree is not a contiguous range of numbers
  }
  pt->blen_arr.assign(pn + 1, std::numeric_limits<double>::quiet_NaN());
  pt->label_off.assign(pn + 2, 0);
  pt->label_blob.clear();
  for (uint32_t se = 0; se <= pn; ++se) {
    if (se < pt->t.nodes.size()) pt->blen_arr[se] = pt->t.nodes[se].blen, pt->label_blob += pt->t.nodes[se].label;
    pt->label_off[se + 1] = (uint32_t)pt->label_blob.size();
  }
}

} // namespace


namespace {

struct ReadPlan {
  bool reported = false, single = false;
  size_t c0 = 0, c1 = 0; // candidate range
  int closest = -1;      // (host path) index into its closest-leaf list
  uint32_t closest_pt = 0;
};
struct CandLite { // what the last phase needs of a candidate
  uint32_t se;
  double d, v, chisq, lwr;
  double jc() const { return -0.75 * log(1 - 4.0 / 3.0 * d); }
};

  // Round 6: the text goes into a raw, growing byte buffer per thread -- room is checked once per number or name, bytes are
  // stored through a pointer -- instead of a std::string appended to character by character (25 appends per placement, each with
  // its capacity test: 13-27 ms per 200,000 reads of jplace on 16 threads, the longest phase of a kr_place_stream call on the
  // 1000-genome tree; profiles/round6_place_big_tree.txt)
  struct OutBuf {
    char* b = nullptr;
    size_t n = 0, cap = 0;
    bool oom = false;
    OutBuf() = default;
    OutBuf(const OutBuf&) = delete;
    OutBuf& operator=(const OutBuf&) = delete;
    ~OutBuf() { kr::big_free(b); }
    void need(size_t k)
    {
      if (n + k <= cap) return;
      const size_t nc = std::max(cap + cap / 2, n + k + 65536);
      char* nb = (char*)kr::big_alloc(nc);
      if (!nb) { // (keeps writing into what it has, from the start: the caller reports the failure)
        oom = true, n = 0;
        if (cap < k + 65536) {
          kr::big_free(b);
          b = (char*)malloc(k + 65536), cap = b ? k + 65536 : 0;
          if (!b) { // (not even that: nothing sensible is left to do -- the std::string this replaces would have thrown here)
            fprintf(stderr, "[ERROR] krepp place: out of memory\n");
            abort();
          }
        }
        return;
      }
      if (n) memcpy(nb, b, n);
      kr::big_free(b);
      b = nb, cap = nc;
    }
    void put(char c) { b[n++] = c; }
    void lit(const char* s_, size_t l) { memcpy(b + n, s_, l), n += l; }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    const char* data() const { return b; }
  };
  // two decimal digits at a time from a table (a division by 10 per digit was a third of a jplace row's time)
static const char kD2[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                            "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
inline uint32_t place_digits5(uint32_t nn, char* q)
{ // nn = round(|v| * 1e5) < 1e8: integer part, '.', five decimals
    const uint32_t ip = nn / 100000u, fp = nn - ip * 100000u;
    uint32_t o_ = 0;
    if (ip >= 100u) {
      q[o_++] = (char)('0' + ip / 100u);
      memcpy(q + o_, kD2 + 2u * (ip % 100u), 2), o_ += 2;
    } else if (ip >= 10u) {
      memcpy(q + o_, kD2 + 2u * ip, 2), o_ += 2;
    } else {
      q[o_++] = (char)('0' + ip);
    }
    q[o_++] = '.';
    const uint32_t a_ = fp / 1000u, b_ = fp - a_ * 1000u; // fp = a_ (2 digits) | b_ (3 digits)
    memcpy(q + o_, kD2 + 2u * a_, 2);
    q[o_ + 2] = (char)('0' + b_ / 100u);
    memcpy(q + o_ + 3, kD2 + 2u * (b_ % 100u), 2);
    return o_ + 5u;
}
inline void place_num(OutBuf& o, double v)
{ // "%.5f" (the reference's stream settings: std::fixed, 5 decimals)
    o.need(80);
    if (std::isnan(v)) {
      if (std::signbit(v)) o.put('-');
      o.lit("nan", 3);
      return;
    }
    uint32_t nn = 0;
    const double av = std::fabs(v);
    bool ok = false;
    if (av < 1000.0) {
      // round(av * 1e5) as printf rounds the exact binary value: away from a tie the scaled value's own fraction decides (its
      // rounding error is below 2^-27); within 1e-6 of a tie the exact routine does (kr_common.h: fma, parity)
      const double sc = av * 100000.0;
      const uint32_t fi = (uint32_t)sc; // (truncation = floor: sc >= 0)
      const double fr = sc - (double)fi;
      if (std::fabs(fr - 0.5) > 1e-6)
        nn = fi + (fr > 0.5 ? 1u : 0u), ok = true;
      else
        ok = kr::fixed5_exact(av, &nn);
      ok = ok && nn < 100000000u;
    }
    if (ok) {
      if (std::signbit(v)) o.put('-');
      o.n += place_digits5(nn, o.b + o.n);
    } else {
      o.n += kr::fmt_fixed5(v, o.b + o.n);
    }
}

// Last phase of report_placement (src/query.cpp:283-331) for a whole batch: chi-square filter, LWR, selection, text.
// `cand(i)` gives candidate i; a read's candidates [c0, c1) are in ascending node number.
// `src`: reported(r), single(r), fetch(r, buf) -> fills buf with the read's candidates in ascending node number.
template <typename Source>
int emit_placements(const kr_place_tree* pt, uint32_t nreads, const Source& src,
                    const char* const* names, const kr_params* p, int tabular, int* has_previous, char** text, uint64_t* len,
                    kr_placement** placements, uint64_t* nplacements, const std::function<void(const char*)>& lap, uint32_t r_begin = 0)
{ // reads [r_begin, r_begin + nreads) of the batch (a range of it: kr_place_stream works through a batch in ranges)
  const int nt = std::max(1, std::min(std::min(kr::parallel_width(), 32), (int)(nreads / 4096)));
  auto en = [&](uint32_t q) { return q - 1; };
  auto mid = [&](uint32_t q) { return std::isnan(pt->t.nodes[q].blen) ? 0.0 : pt->t.nodes[q].blen / 2.0; };
  auto unum = [](OutBuf& o, uint32_t v) { // decimal
    o.need(12);
    char t_[12];
    int k_ = 12;
    while (v >= 100u) {
      const uint32_t r_ = v % 100u;
      v /= 100u;
      k_ -= 2, memcpy(t_ + k_, kD2 + 2u * r_, 2);
    }
    if (v >= 10u)
      k_ -= 2, memcpy(t_ + k_, kD2 + 2u * v, 2);
    else
      t_[--k_] = (char)('0' + v);
    o.lit(t_ + k_, (size_t)(12 - k_));
  };
  const bool want_pl = placements && nplacements;
  auto jfields = [&](OutBuf& o, uint32_t q, const CandLite& a) {
    o.need(16);
    o.put('[');
    unum(o, en(q));
    o.need(4), o.lit(", ", 2), place_num(o, a.jc() - mid(q));
    o.need(4), o.lit(", ", 2), place_num(o, mid(q));
    o.need(4), o.lit(", ", 2), place_num(o, -a.v);
    o.need(4), o.lit(", ", 2), place_num(o, a.lwr);
    o.need(4), o.lit(", ", 2), place_num(o, a.d);
    o.need(4), o.put(']');
  };
  auto tfields = [&](OutBuf& o, uint32_t q, const CandLite& a) {
    const std::string& nm = pt->t.nodes[q].label;
    o.need(nm.size() + 8);
    if (nm.empty())
      o.lit("NA", 2);
    else
      o.lit(nm.data(), nm.size());
    o.put('\t');
    unum(o, en(q));
    o.need(4), o.put('\t'), place_num(o, a.lwr);
    o.need(4), o.put('\t'), place_num(o, a.d);
  };
  const bool jp = tabular == 0, tb = tabular == 1; // 2: --summarize, no text (the caller sums the placements)
  std::vector<OutBuf> part((size_t)nt);
  std::vector<std::vector<kr_placement>> ppls((size_t)nt);
  kr::parallel_for(nt, [&](int t) {
    const uint32_t ra = r_begin + (uint32_t)((uint64_t)nreads * t / nt), rb = r_begin + (uint32_t)((uint64_t)nreads * (t + 1) / nt);
    OutBuf& out = part[(size_t)t];
    std::vector<kr_placement>& pls = ppls[(size_t)t];
    bool prev = false; // within the piece; pieces are joined with the separator below
    auto record = [&](uint32_t r, uint32_t q, const CandLite& a) {
      if (!want_pl) return; // (--summarize and library callers ask for the placements; the text modes of the CLI do not)
      kr_placement x;
      x.read = r, x.edge = en(q), x.lwr = a.lwr, x.d_llh = a.d, x.v_llh = a.v, x.pendant = a.jc() - mid(q), x.distal = mid(q);
      pls.push_back(x);
    };
    std::vector<size_t> nd_v;
    std::vector<CandLite> cands; // the read's candidates
    out.need((size_t)(rb - ra) * (jp ? 224 : tb ? 96 : 0) + 256);
    if (want_pl) pls.reserve((size_t)(rb - ra) * 2);
    for (uint32_t r = ra; r < rb; ++r) {
      if (!src.reported(r)) continue;
      struct { bool single; size_t c0, c1; } pl{src.single(r), 0, 0};
      pl.c1 = src.fetch(r, cands);
      const char* id = names ? names[r] : "";
      const size_t idl = strlen(id);
      auto lit = [&](const char* s_) { const size_t l_ = strlen(s_); out.need(l_ + 8), out.lit(s_, l_); };
      auto tline = [&](const CandLite& c_) { // SEQ_ID \t fields \n
        out.need(idl + 8), out.lit(id, idl), out.put('\t');
        tfields(out, c_.se, c_);
        out.need(4), out.put('\n');
      };
      if (jp) {
        if (prev) lit(",\n");
        lit("\t\t\t{\"n\" : [\"");
        out.need(idl + 8), out.lit(id, idl);
        lit("\"], \"p\" : [");
        prev = true;
      }
      if (pl.single) {
        CandLite& c = cands[pl.c0];
        record(r, c.se, c);
        if (tb)
          tline(c);
        else if (jp)
          jfields(out, c.se, c), lit("]}");
        continue;
      }
      nd_v.clear();
      for (size_t i = pl.c0; i < pl.c1; ++i)
        if (cands[i].chisq < p->chisq && pt->t.nodes[cands[i].se].parent) nd_v.push_back(i);
      double total = 0;
      for (size_t i : nd_v) {
        cands[i].lwr = exp(-cands[i].chisq / 2);
        total = total + cands[i].lwr;
      }
      if (p->multi) {
        for (size_t j2 = 0; j2 < nd_v.size(); ++j2) {
          CandLite& c = cands[nd_v[j2]];
          c.lwr = c.lwr / total;
          record(r, c.se, c);
          if (j2 > 0 && jp) lit(",");
          if (tb)
            tline(c);
          else if (jp)
            lit("\n\t\t\t\t"), jfields(out, c.se, c);
        }
        if (jp) lit("]\n\t\t\t}");
      } else {
        if (nd_v.size() > 1)
          std::stable_sort(nd_v.begin(), nd_v.end(), [&](size_t l, size_t rr) {
            uint32_t cl_ = pt->card[cands[l].se], cr = pt->card[cands[rr].se];
            return cl_ == cr ? cands[l].d > cands[rr].d : cl_ < cr;
          });
        if (nd_v.empty()) {
          if (jp) lit("]}");
          continue;
        }
        CandLite& c = cands[nd_v.back()];
        c.lwr = c.lwr / total;
        record(r, c.se, c);
        if (tb)
          tline(c);
        else if (jp)
          jfields(out, c.se, c), lit("]}");
      }
    }
  });
  lap("D: filter + text");
  for (auto& pb_ : part)
    if (pb_.oom) return kr::fail(KR_ERR_NOMEM, "kr_place_batch: out of memory");
  // pieces joined in order, straight into the buffer the caller receives
  bool prev = *has_previous != 0;
  std::vector<size_t> at((size_t)nt + 1, 0), sep((size_t)nt, 0), pat((size_t)nt + 1, 0);
  for (int t = 0; t < nt; ++t) {
    if (!part[(size_t)t].empty()) {
      sep[(size_t)t] = (jp && prev) ? 2 : 0;
      if (jp) prev = true;
    }
    at[(size_t)t + 1] = at[(size_t)t] + sep[(size_t)t] + part[(size_t)t].size();
    pat[(size_t)t + 1] = pat[(size_t)t] + ppls[(size_t)t].size();
  }
  const size_t total_len = at[(size_t)nt], total_pl = pat[(size_t)nt];
  char* buf = (char*)kr::big_alloc(total_len + 1);
  if (!buf) return kr::fail(KR_ERR_NOMEM, "kr_place_batch: out of memory");
  kr_placement* pbuf = nullptr;
  if (placements && nplacements) {
    pbuf = (kr_placement*)kr::big_alloc(std::max<size_t>(1, total_pl) * sizeof(kr_placement));
    if (!pbuf) {
      kr::big_free(buf);
      return kr::fail(KR_ERR_NOMEM, "kr_place_batch: out of memory");
    }
  }
  kr::parallel_for(nt, [&](int t) {
    char* o = buf + at[(size_t)t];
    if (sep[(size_t)t]) o[0] = ',', o[1] = '\n', o += 2;
    memcpy(o, part[(size_t)t].data(), part[(size_t)t].size());
    if (pbuf && !ppls[(size_t)t].empty()) memcpy(pbuf + pat[(size_t)t], ppls[(size_t)t].data(), ppls[(size_t)t].size() * sizeof(kr_placement));
  });
  buf[total_len] = 0;
  lap("D: join");
  *has_previous = prev ? 1 : 0;
  *text = buf;
  *len = total_len;
  if (pbuf) *placements = pbuf, *nplacements = total_pl;
  return KR_OK;
}

} // namespace

extern "C" {

void kr_place_tree_free(kr_place_tree* pt) { delete pt; }
// tests: one number as the placement rows print it (place_num); returns the length
int kr_debug_place_fixed5(double v, char* out)
{
  OutBuf o;
  place_num(o, v);
  memcpy(out, o.b, o.n);
  out[o.n] = 0;
  return (int)o.n;
}
const uint8_t* kr_place_tree_kinds(const kr_place_tree* pt) { return pt ? pt->kinds.data() : nullptr; }

int kr_place_frame(const kr_place_tree* pt, int which, int tabular, const char* invocation, uint64_t total_qseq, char** text,
                   uint64_t* len)
{
  if (!pt || !text || !len) return kr::fail(KR_ERR_ARG, "kr_place_frame: null argument");
  std::string o, tree, inv = invocation ? invocation : "";
  nwk_jplace(*pt, pt->root, tree);
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
        "\"https://github.com/bo1929/krepp\",\n\t\t\"num_queries\" : \"" + std::to_string(total_qseq) + "\",\n\t\t\"invocation\" : \"" + inv +
        "\"\n\t},\n\t\"tree\" : \"" + tree + "\"\n}";
  }
  *text = dup_text(o, len);
  return *text ? KR_OK : kr::fail(KR_ERR_NOMEM, "kr_place_frame: out of memory");
}

int kr_place_batch(const kr_host_index* hx, const kr_index* dix, const kr_place_tree* pt, const kr_result_view* rv,
                   const uint64_t* offsets, const char* const* names, const kr_params* p, int tabular, int* has_previous,
                   char** text, uint64_t* len, kr_placement** placements, uint64_t* nplacements)
{
  kr::clear_error();
  if (!hx || !dix || !pt || !rv || !offsets || !p || !has_previous || !text || !len)
    return kr::fail(KR_ERR_ARG, "kr_place_batch: null argument");
  if (!rv->rec_hist) return kr::fail(KR_ERR_STATE, "kr_place_batch: the batch must be submitted with KR_TAP_ACCS");
  kr_index_view iv;
  kr_host_index_view(hx, &iv);
  const uint32_t np = p->hdist_th + 1, th = p->hdist_th, k = iv.k;
  const double* rho_tab = iv.libs[0].rho;

  const bool timing = getenv("KR_PLACE_TIMING") != nullptr;
  auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = wall();
  std::function<void(const char*)> lap = [&](const char* what) {
    if (!timing) return;
    const double now = wall();
    fprintf(stderr, "[place] %s %.1f ms\n", what, (now - t_mark) * 1e3);
    t_mark = now;
  };
  // ---- phase A (host, one OpenMP thread per range of reads): per read, the leaf set, the closest, ancestor
  //      accumulation, candidates.  Each thread keeps a small open-addressing table node -> accumulator that
  //      is reused from read to read (stamps instead of clearing).
  std::vector<ReadPlan> plan(rv->nreads);
  // one thread per ~4096 reads, at most 32: small batches do not pay for waking a large team
  const int nt = std::max(1, std::min(std::min(kr::parallel_width(), 32), (int)(rv->nreads / 4096)));
  struct ThreadOut {
    std::vector<Cand> cands;
    std::vector<Acc> closest;
  };
  std::vector<ThreadOut> tout((size_t)nt);
  auto ranged = [&](size_t n, const std::function<void(size_t, size_t)>& body) { // [0, n) in nt contiguous pieces
    kr::parallel_for(nt, [&](int t) { body(n * (size_t)t / (size_t)nt, n * ((size_t)t + 1) / (size_t)nt); });
  };
  kr::parallel_for(nt, [&](int t) {
    ThreadOut& T = tout[(size_t)t];
    const uint32_t ra = (uint32_t)((uint64_t)rv->nreads * t / nt), rb = (uint32_t)((uint64_t)rv->nreads * (t + 1) / nt);
    std::vector<uint32_t> tkey(256), tval(256), tstamp(256, 0);
    uint32_t stamp = 0;
    std::vector<Acc> pool;
    std::vector<std::pair<uint32_t, uint32_t>> entries, leaves; // (node, pool index)
    auto find_or_add = [&](uint32_t node) -> uint32_t {
      for (;;) {
        const uint32_t mask = (uint32_t)tkey.size() - 1;
        uint32_t hsl = (node * 2654435761u) & mask;
        while (tstamp[hsl] == stamp && tkey[hsl] != node) hsl = (hsl + 1) & mask;
        if (tstamp[hsl] == stamp) return tval[hsl];
        if (2 * (entries.size() + 1) > tkey.size()) { // grow, re-insert
          const size_t ncap = tkey.size() * 2;
          tkey.assign(ncap, 0), tval.assign(ncap, 0), tstamp.assign(ncap, 0);
          stamp = 1;
          for (auto& e : entries) {
            uint32_t q = (e.first * 2654435761u) & (uint32_t)(ncap - 1);
            while (tstamp[q] == stamp) q = (q + 1) & (uint32_t)(ncap - 1);
            tstamp[q] = stamp, tkey[q] = e.first, tval[q] = e.second;
          }
          continue;
        }
        tstamp[hsl] = stamp, tkey[hsl] = node, tval[hsl] = (uint32_t)pool.size();
        entries.emplace_back(node, (uint32_t)pool.size());
        pool.emplace_back();
        pool.back().zero(np);
        return tval[hsl];
      }
    };
    for (uint32_t r = ra; r < rb; ++r) {
      const uint32_t o = rv->read_off[r], n = rv->read_cnt[r];
      if (n == 0) continue;
      const double enmers = (double)((offsets[r + 1] - offsets[r]) >= k ? (offsets[r + 1] - offsets[r]) - k + 1 : 0);
      auto leaf_acc = [&](uint32_t i, Acc& a) {
        a.zero(np);
        a.leaf = true;
        double mc = 0;
        for (uint32_t x = 0; x < np; ++x) a.hist[x] = rv->rec_hist[(uint64_t)x * rv->rec_hist_stride + i], mc += a.hist[x];
        a.match = mc;
        a.mismatch = (double)rv->read_onmers[r] - mc; // src/query.cpp:104
        a.nmers = enmers;                              // IMers::enmers (src/query.cpp:335-350)
        a.rho = rho_tab[rv->rec_key[i] >> 1];
        a.d = rv->rec_d[i], a.v = rv->rec_v[i];
      };
      // closest: last record in (strand, se) order with d <= best (summarize_matches' `<=`, as in the oracle)
      double best = 1.7976931348623157e308;
      int cl = -1;
      for (int strand = 0; strand < 2; ++strand)
        for (uint32_t i = o; i < o + n; ++i)
          if (rv->rec_key[i] && (rv->rec_key[i] & 1u) == (uint32_t)strand && rv->rec_d[i] <= best) best = rv->rec_d[i], cl = (int)i;
      if (cl < 0) continue;
      ReadPlan& pl = plan[r];
      Acc closest;
      leaf_acc((uint32_t)cl, closest);
      pl.closest_pt = pt->idx_to_pt[rv->rec_key[cl] >> 1];
      if (!(p->no_filter || closest.leq_tau(p->tau, np) > 1.0)) continue; // src/query.cpp:220
      // node_to_minfo: the record chosen for each leaf (rec_sel under multi / no_filter / no dist-max)
      ++stamp;
      if (stamp == 0) std::fill(tstamp.begin(), tstamp.end(), 0u), stamp = 1;
      pool.clear(), entries.clear(), leaves.clear();
      for (uint32_t i = o; i < o + n; ++i)
        if (rv->rec_sel[i]) {
          const uint32_t q = pt->idx_to_pt[rv->rec_key[i] >> 1];
          if (!q) continue;
          const uint32_t slot = find_or_add(q);
          leaf_acc(i, pool[slot]);
        }
      if (entries.empty()) continue;
      pl.reported = true;
      pl.c0 = T.cands.size();
      closest.chisq = 0;
      pl.closest = (int)T.closest.size();
      T.closest.push_back(closest);
      if (entries.size() == 1) { // src/query.cpp:233-244
        pl.single = true;
        T.cands.push_back(Cand{r, pl.closest_pt, closest, false});
        pl.c1 = T.cands.size();
        continue;
      }
      leaves = entries;
      std::sort(leaves.begin(), leaves.end()); // ascending edge number: the order the sums are formed in
      for (auto& lf : leaves) { // src/query.cpp:250-267
        const Acc src = pool[lf.second];
        double denom = 1.0;
        uint32_t par = lf.first;
        while ((par = pt->t.nodes[par].parent)) {
          denom /= pt->eff[par];
          pool[find_or_add(par)].add(src, denom, np);
        }
      }
      std::sort(entries.begin(), entries.end());
      for (auto& e : entries) { // src/query.cpp:270-283 (likelihoods deferred to the GPU)
        const uint32_t q = e.first;
        const Acc& a = pool[e.second];
        const uint32_t nch = (uint32_t)pt->kids[q].size();
        if (nch != pt->eff[q] || nch == 1) continue;
        if (p->no_filter || a.leq_tau(p->tau, np) > 1.0) T.cands.push_back(Cand{r, q, a, !a.leaf});
      }
      pl.c1 = T.cands.size();
    }
  });
  lap("A: aggregation");
  // one candidate list in read order: pointers into the threads' vectors
  std::vector<Cand*> cands;
  std::vector<size_t> cbase((size_t)nt + 1, 0), lbase((size_t)nt + 1, 0);
  for (int t = 0; t < nt; ++t) cbase[(size_t)t + 1] = cbase[(size_t)t] + tout[(size_t)t].cands.size(), lbase[(size_t)t + 1] = lbase[(size_t)t] + tout[(size_t)t].closest.size();
  cands.resize(cbase[(size_t)nt]);
  std::vector<const Acc*> closest_acc(lbase[(size_t)nt]);
  kr::parallel_for(nt, [&](int t) {
    const uint32_t ra = (uint32_t)((uint64_t)rv->nreads * t / nt), rb = (uint32_t)((uint64_t)rv->nreads * (t + 1) / nt);
    const size_t cb = cbase[(size_t)t], lb = lbase[(size_t)t];
    for (uint32_t r = ra; r < rb; ++r)
      if (plan[r].reported) plan[r].c0 += cb, plan[r].c1 += cb, plan[r].closest += (int)lb;
    ThreadOut& T = tout[(size_t)t];
    for (size_t q = 0; q < T.cands.size(); ++q) cands[cb + q] = &T.cands[q];
    for (size_t q = 0; q < T.closest.size(); ++q) closest_acc[lb + q] = &T.closest[q];
  });
  lap("A: merge");
  // ---- phase B (GPU): Brent on the internal candidates
  {
    std::vector<size_t> which;
    for (size_t i = 0; i < cands.size(); ++i)
      if (cands[i]->internal) which.push_back(i);
    std::vector<double> hist(which.size() * np), uc(which.size()), rho(which.size()), d(which.size()), v(which.size());
    ranged(which.size(), [&](size_t j0, size_t j1) {
      for (size_t j = j0; j < j1; ++j) {
      const Acc& a = cands[which[j]]->a;
      for (uint32_t x = 0; x < np; ++x) hist[j * np + x] = a.hist[x];
      uc[j] = a.mismatch, rho[j] = a.rho;
    }
    });
    int rc = kr_llh_batch(dix, th, 0, which.size(), hist.data(), uc.data(), rho.data(), nullptr, d.data(), v.data());
    if (rc) return rc;
    ranged(which.size(), [&](size_t j0, size_t j1) {
      for (size_t j = j0; j < j1; ++j) cands[which[j]]->a.d = d[j], cands[which[j]]->a.v = v[j];
    });
  }
  lap("B: Brent on internal candidates");
  // ---- phase C (GPU): chisq = 2 * (f_closest(d_candidate) - v_closest)   (src/query.cpp:276, :420-424): one problem per
  //      read (its closest leaf), one evaluation per candidate
  {
    std::vector<size_t> which;
    for (size_t i = 0; i < cands.size(); ++i)
      if (!plan[cands[i]->read].single) which.push_back(i);
    const size_t nprob = closest_acc.size(), nev = which.size();
    std::unique_ptr<double[]> hist(new double[std::max<size_t>(1, nprob * np)]), uc(new double[std::max<size_t>(1, nprob)]),
      rho(new double[std::max<size_t>(1, nprob)]), din(new double[std::max<size_t>(1, nev)]), f(new double[std::max<size_t>(1, nev)]);
    std::unique_ptr<uint32_t[]> pidx(new uint32_t[std::max<size_t>(1, nev)]);
    ranged(nprob, [&](size_t q0, size_t q1) {
      for (size_t q = q0; q < q1; ++q) {
      const Acc& c = *closest_acc[q];
      for (uint32_t x = 0; x < np; ++x) hist[q * np + x] = c.hist[x];
      uc[q] = c.mismatch, rho[q] = c.rho;
    }
    });
    ranged(nev, [&](size_t j0, size_t j1) {
      for (size_t j = j0; j < j1; ++j) pidx[j] = (uint32_t)plan[cands[which[j]]->read].closest, din[j] = cands[which[j]]->a.d;
    });
    int rc = kr_llh_eval_indexed(dix, th, nprob, hist.get(), uc.get(), rho.get(), nev, pidx.get(), din.get(), f.get());
    if (rc) return rc;
    ranged(nev, [&](size_t j0, size_t j1) {
      for (size_t j = j0; j < j1; ++j) {
      Cand& c = *cands[which[j]];
      c.a.chisq = 2 * (f[j] - closest_acc[(size_t)plan[c.read].closest]->v);
    }
    });
  }
  lap("C: chi-square evaluations");
  // ---- phase D: chi-square filter, LWR, selection, text (shared with the device path)
  struct HostSource {
    const std::vector<ReadPlan>& plan;
    const std::vector<Cand*>& cands;
    bool reported(uint32_t r) const { return plan[r].reported; }
    bool single(uint32_t r) const { return plan[r].single; }
    size_t fetch(uint32_t r, std::vector<CandLite>& buf) const
    {
      const ReadPlan& pl = plan[r];
      buf.clear();
      for (size_t i = pl.c0; i < pl.c1; ++i) buf.push_back(CandLite{cands[i]->se, cands[i]->a.d, cands[i]->a.v, cands[i]->a.chisq, cands[i]->a.lwr});
      return buf.size();
    }
  } src{plan, cands};
  return emit_placements(pt, rv->nreads, src, names, p, tabular, has_previous, text, len, placements, nplacements, lap);
}

// As kr_place_batch, for the batch last submitted on `s` with KR_TAP_ACCS, which need not be collected: the tree
// aggregation, the Brent minimisations of the internal candidates and the chi-squares run on the device
// (kr_place_kernel) on the records where they lie; what comes back is the candidates (node, d, v, chi-square), and the
// host does the last phase only (filter, exp / LWR, Jukes-Cantor, text).  Bit-identical to kr_place_batch.  Reads
// beyond the kernel's LDS arrays (256 leaves / 1024 ancestors) are done by its second launch with the arrays in global
// scratch; only a placement tree whose numbering is not post-order (or a batch that runs out of candidate slots) sends
// the whole batch through kr_place_batch.
int kr_place_stream(const kr_host_index* hx, const kr_index* dix, const kr_place_tree* pt, kr_stream* s, uint32_t nreads,
                    const uint64_t* offsets, const char* const* names, const kr_params* p, int tabular, int* has_previous, char** text,
                    uint64_t* len, kr_placement** placements, uint64_t* nplacements)
{
  kr::clear_error();
  if (!hx || !dix || !pt || !s || !offsets || !p || !has_previous || !text || !len || !nreads)
    return kr::fail(KR_ERR_ARG, "kr_place_stream: null argument");
  auto host_path = [&]() -> int {
    g_place_host_batches.fetch_add(1, std::memory_order_relaxed);
    kr_result_view rv;
    int rc = kr_batch_collect(s, &rv);
    if (rc) return rc;
    if (rv.nreads != nreads) return kr::fail(KR_ERR_ARG, "kr_place_stream: nreads does not match the submitted batch");
    return kr_place_batch(hx, dix, pt, &rv, offsets, names, p, tabular, has_previous, text, len, placements, nplacements);
  };
  if (!pt->postorder || getenv("KR_PLACE_HOST")) return host_path();
  const bool timing = getenv("KR_PLACE_TIMING") != nullptr;
  auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = wall();
  std::function<void(const char*)> lap = [&](const char* what) {
    if (!timing) return;
    const double now = wall();
    fprintf(stderr, "[place/device] %s %.1f ms\n", what, (now - t_mark) * 1e3);
    t_mark = now;
  };
  const int nt = std::max(1, std::min(std::min(kr::parallel_width(), 32), (int)(nreads / 4096)));
  std::vector<uint32_t> read_len(nreads);
  kr::parallel_for(nt, [&](int t) {
    for (size_t r = (size_t)nreads * t / nt; r < (size_t)nreads * (t + 1) / nt; ++r) read_len[r] = (uint32_t)(offsets[r + 1] - offsets[r]);
  });
  kr::PlaceTreeArrays T;
  T.pn = pt->t.nnodes(), T.nidx = (uint32_t)pt->idx_to_pt.size() - 1;
  T.parent = pt->parent_arr.data(), T.eff = pt->eff.data(), T.elig = pt->elig.data(), T.lo = pt->lo.data(), T.idx_to_pt = pt->idx_to_pt.data(), T.depth = pt->depth.data();
  // The batch in RANGES of reads (round 5): while the host filters and formats the candidates of one range ("D", as long as the
  // device's part: 12-17 ms against 12 ms per 400,000 reads on a 1000-genome tree), the place kernels of the next are running.
  // Two ranges from 131,072 reads (400,000 reads, tabular, one call at a time: 8.0 M reads/s whole, 12.8 M in two ranges, 10.0 M in
  // four, 9.2 M in eight -- every range has its own launches, waits and copies; a 65,536-read batch of the CLI loses in any split:
  // profiles/round5_place_ranges.txt).  KR_PLACE_RANGES=1: the whole batch at once, as before.
  // Round 6: jplace and tabular rows are written on the DEVICE, where the placements are (kr_dev_place.inc), when the caller wants
  // text and no placement records (the CLI's text modes): the candidates stay in HBM, the range's text comes back as bytes, and
  // with no last phase on the host there is nothing for a second range to hide.  KR_PLACE_HOST_TEXT=1: the host's last phase.
  const bool dev_text = (tabular == 0 || tabular == 1) && !(placements && nplacements) && names && pt->card.size() == (size_t)T.pn + 1 && !getenv("KR_PLACE_HOST_TEXT");
  if (dev_text) T.blen = pt->blen_arr.data(), T.card = pt->card.data(), T.labels = pt->label_blob.data(), T.label_off = pt->label_off.data();
  uint32_t nranges = (nreads >= 131072u && !dev_text) ? 2u : 1u;
  if (const char* e = getenv("KR_PLACE_RANGES")) nranges = (uint32_t)std::max(1, std::min(16, atoi(e)));
  nranges = std::min<uint32_t>(nranges, std::max<uint32_t>(1u, nreads));
  int rc = kr::place_device_begin(s, pt, T, read_len.data());
  if (rc) return rc;
  if (kr::place_stream_nreads(s) != nreads) return kr::fail(KR_ERR_ARG, "kr_place_stream: nreads does not match the submitted batch");
  auto r_of = [&](uint32_t k) { return (uint32_t)((uint64_t)nreads * k / nranges); };
  if (dev_text && (rc = kr::place_device_text_begin(s, T, names, nreads, tabular, p->multi != 0))) return rc;
  lap("front end waited for, workspaces");
  if ((rc = kr::place_device_launch(s, T, 0, r_of(1), p->tau, p->no_filter != 0, p->chisq))) return rc;
  struct Piece { char* text = nullptr; uint64_t len = 0; kr_placement* pl = nullptr; uint64_t npl = 0; };
  std::vector<Piece> pieces(nranges);
  auto drop = [&]() {
    for (auto& pc : pieces) kr::big_free(pc.text), kr::big_free(pc.pl);
  };
  uint64_t kept_base = 0, heavy = 0;
  int prev = *has_previous;
  bool overflow = false;
  for (uint32_t k = 0; k < nranges && !rc; ++k) {
    const uint32_t r0 = r_of(k), r1 = r_of(k + 1);
    kr::PlaceDeviceResult res;
    rc = kr::place_device_finish(s, T, r0, r1 - r0, p->tau, p->no_filter != 0, p->chisq, kept_base, &res);
    if (rc) break;
    lap("A-C of a range: aggregation, Brent, chi-square on the device + copy back");
    if (res.overflow) {
      overflow = true;
      break;
    }
    heavy += res.heavy_reads;
    kept_base += res.kept;
    if (k + 1 < nranges) // the next range's kernels run beside this range's last phase
      if ((rc = kr::place_device_launch(s, T, r1, r_of(k + 2) - r1, p->tau, p->no_filter != 0, p->chisq))) break;
    if (dev_text) (res.text ? g_place_text_device : g_place_text_fallback).fetch_add(1, std::memory_order_relaxed);
    if (dev_text && !res.text && timing) fprintf(stderr, "[place/device] the device's text was not taken: flags %llu (1 a number out of range, 2 more text than the buffer holds, 4 more than 64 candidates kept in a read, 8 a rounding tie too close)\n", (unsigned long long)res.text_flags);
    if (res.text) { // the range's rows, written on the device
      const char* tp = res.text;
      uint64_t tl = res.text_len;
      if (tabular == 0 && tl >= 2) { // jplace: every reported read begins with ",\n"; the first one of a file loses it
        if (!prev) tp += 2, tl -= 2;
        prev = 1;
      }
      Piece& pcd = pieces[k];
      pcd.text = (char*)kr::big_alloc(tl + 1);
      if (!pcd.text) {
        rc = kr::fail(KR_ERR_NOMEM, "kr_place_stream: out of memory");
        break;
      }
      const int sl = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::min(kr::parallel_width(), 16), tl >> 20));
      kr::parallel_for(sl, [&](int q) {
        const uint64_t l0 = tl * (uint64_t)q / sl, l1 = tl * (uint64_t)(q + 1) / sl;
        if (l1 > l0) memcpy(pcd.text + l0, tp + l0, l1 - l0);
      });
      pcd.text[tl] = 0;
      pcd.len = tl;
      lap("D: the device's text taken over");
      continue;
    }
    // each read's candidates in ascending node number (the order the host path forms them in), straight from the
    // arrays the device wrote
    struct DeviceSource {
      const kr::PlaceDeviceResult& res;
      bool reported(uint32_t r) const { return (res.rd_info[r] >> 31) != 0; }
      bool single(uint32_t r) const { return ((res.rd_info[r] >> 30) & 1u) != 0; }
      size_t fetch(uint32_t r, std::vector<CandLite>& buf) const
      {
        const uint32_t n = res.rd_info[r] & 0x3FFFFFFFu, c0 = res.rd_c0[r];
        buf.clear();
        for (uint32_t i = 0; i < n; ++i) // (0x7FFFFFFF: node 0 of a single placement -- 0 itself marks an unused slot on the device)
          buf.push_back(CandLite{res.c_se[c0 + i] == 0x7FFFFFFFu ? 0u : res.c_se[c0 + i], res.c_d[c0 + i], res.c_v[c0 + i], res.c_chisq[c0 + i], 1.0});
        if (n > 1) std::sort(buf.begin(), buf.end(), [](const CandLite& a, const CandLite& b) { return a.se < b.se; });
        return n;
      }
    } src{res};
    Piece& pc = pieces[k];
    rc = emit_placements(pt, r1 - r0, src, names, p, tabular, &prev, &pc.text, &pc.len, placements ? &pc.pl : nullptr, nplacements ? &pc.npl : nullptr, lap, r0);
  }
  if (rc) {
    drop();
    kr::place_device_abort(s); // (the next range's kernels and its copy into the page-locked counters are in flight -- kr_batch_wait
                               //  returns at once here, the batch was waited for long ago: round-5 advice)
    return rc;
  }
  if (overflow) { // out of candidate slots beyond what a rerun can name: the host back end takes the whole batch
    drop();
    return host_path();
  }
  g_place_device_batches.fetch_add(1, std::memory_order_relaxed);
  g_place_heavy_reads.fetch_add(heavy, std::memory_order_relaxed);
  // the ranges' pieces joined in order
  if (nranges == 1) {
    *text = pieces[0].text, *len = pieces[0].len;
    if (placements && nplacements) *placements = pieces[0].pl, *nplacements = pieces[0].npl;
  } else {
    uint64_t tl = 0, tp = 0;
    for (auto& pc : pieces) tl += pc.len, tp += pc.npl;
    char* buf = (char*)kr::big_alloc(tl + 1);
    kr_placement* pb = (placements && nplacements) ? (kr_placement*)kr::big_alloc(std::max<uint64_t>(1, tp) * sizeof(kr_placement)) : nullptr;
    if (!buf || ((placements && nplacements) && !pb)) {
      kr::big_free(buf), kr::big_free(pb);
      drop();
      return kr::fail(KR_ERR_NOMEM, "kr_place_stream: out of memory");
    }
    std::vector<uint64_t> at(nranges + 1, 0), pat(nranges + 1, 0);
    for (uint32_t k = 0; k < nranges; ++k) at[k + 1] = at[k] + pieces[k].len, pat[k + 1] = pat[k] + pieces[k].npl;
    // (in slices, side by side: two threads copying 35 MB each into fresh pages took 13 ms of a 58 ms jplace call)
    const int sl = std::max(1, std::min(kr::parallel_width(), 16) / (int)nranges);
    kr::parallel_for((int)nranges * sl, [&](int q) {
      const size_t k = (size_t)(q / sl), part_ = (size_t)(q % sl);
      const uint64_t l0 = pieces[k].len * part_ / sl, l1 = pieces[k].len * (part_ + 1) / sl;
      if (l1 > l0) memcpy(buf + at[k] + l0, pieces[k].text + l0, l1 - l0);
      if (pb && part_ == 0 && pieces[k].npl) memcpy(pb + pat[k], pieces[k].pl, pieces[k].npl * sizeof(kr_placement));
    });
    buf[tl] = 0;
    drop();
    *text = buf, *len = tl;
    if (pb) *placements = pb, *nplacements = tp;
    lap("ranges joined");
  }
  *has_previous = prev;
  return KR_OK;
}

void kr_place_text_counters(uint64_t* device_ranges, uint64_t* fallback_ranges)
{ // ranges of reads whose rows were written on the device / that were formatted by the host although device text was asked for
  if (device_ranges) *device_ranges = g_place_text_device.load(std::memory_order_relaxed);
  if (fallback_ranges) *fallback_ranges = g_place_text_fallback.load(std::memory_order_relaxed);
}

void kr_place_counters(uint64_t* device_batches, uint64_t* host_batches, uint64_t* heavy_reads)
{
  if (device_batches) *device_batches = g_place_device_batches.load(std::memory_order_relaxed);
  if (host_batches) *host_batches = g_place_host_batches.load(std::memory_order_relaxed);
  if (heavy_reads) *heavy_reads = g_place_heavy_reads.load(std::memory_order_relaxed);
}

uint32_t kr_place_tree_nnodes(const kr_place_tree* pt) { return pt ? pt->t.nnodes() : 0; }

int kr_place_summary_add(const kr_place_tree* pt, const kr_placement* pls, uint64_t n, double* wcount, double* twcount)
{ // place --summarize: a read with m placements counts 1/m at each (src/query.cpp:232-233,297-298,322-323);
  // summed per 512 reads first, as the reference's batches are (src/krepp.cpp:466-471)
  if (!pt || (!pls && n) || !wcount || !twcount) return kr::fail(KR_ERR_ARG, "kr_place_summary_add: null argument");
  // the group's sums in a dense array + the list of nodes it touched, flushed in ascending node order: the additions a std::map
  // keyed by node made (until round 6: 60 ms per 880,000 placements, most of a `--summarize` call), in the same order
  const uint32_t nn_ = pt->t.nnodes();
  std::vector<double> part((size_t)nn_ + 2, 0.0);
  std::vector<uint8_t> seen((size_t)nn_ + 2, 0);
  std::vector<uint32_t> touched;
  auto flush = [&]() {
    std::sort(touched.begin(), touched.end());
    for (uint32_t k_ : touched) *twcount += part[k_], wcount[k_] += part[k_], part[k_] = 0.0, seen[k_] = 0;
    touched.clear();
  };
  for (uint64_t i = 0; i < n;) {
    uint64_t j = i;
    while (j < n && pls[j].read == pls[i].read) ++j;
    if (i && (pls[i].read >> 9) != (pls[i - 1].read >> 9)) flush();
    for (uint64_t q = i; q < j; ++q) {
      if (pls[q].edge + 1 > pt->t.nnodes()) return kr::fail(KR_ERR_ARG, "kr_place_summary_add: edge out of range");
      const uint32_t k_ = pls[q].edge + 1;
      if (!seen[k_]) seen[k_] = 1, touched.push_back(k_);
      part[k_] += 1.0 / (double)(j - i);
    }
    i = j;
  }
  flush();
  return KR_OK;
}

int kr_place_summary_text(const kr_place_tree* pt, const double* wcount, double twcount, char** text, uint64_t* len)
{ // rows of QueryIndex::place_sequences (src/krepp.cpp:493-497), ascending edge number
  if (!pt || !wcount || !text || !len) return kr::fail(KR_ERR_ARG, "kr_place_summary_text: null argument");
  std::string o;
  for (uint32_t se = 1; se <= pt->t.nnodes(); ++se)
    if (wcount[se] != 0) {
      const std::string& nm = pt->t.nodes[se].label;
      o += (nm.empty() ? std::string("NA") : nm) + "\t" + std::to_string(se - 1) + "\t" + f5(wcount[se]) + "\t" + f5(wcount[se] / twcount) + "\n";
    }
  *text = dup_text(o, len);
  return *text ? KR_OK : kr::fail(KR_ERR_NOMEM, "kr_place_summary_text: out of memory");
}

} // extern "C"
