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
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

struct kr_place_tree {
  kr::HostTree t;                 // placement tree, se = post-order number, edge = se - 1
  std::vector<std::vector<uint32_t>> kids;
  std::vector<uint32_t> card, eff;
  std::vector<uint32_t> idx_to_pt; // index colour id of a leaf -> se in the placement tree (0 = absent)
  std::vector<uint8_t> kinds;     // node_kind array for kr_index_upload
  uint32_t root = 0;
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

struct Acc { // the fields of Minfo that placement needs
  double nmers = 0, mismatch = 0, match = 0, rho = 0;
  std::vector<double> hist;
  double d = 1.7976931348623157e308, v = NAN, chisq = NAN, lwr = 1;
  bool leaf = false;
  void add(const Acc& m, double denom)
  { // Minfo::add (src/query.hpp:139-152)
    mismatch = nmers ? mismatch : m.nmers;
    match += m.match * denom;
    mismatch -= m.match * denom;
    for (size_t x = 0; x < hist.size(); ++x) hist[x] = hist[x] + m.hist[x] * denom;
    nmers = std::max(nmers, m.nmers);
    rho = std::max(rho, m.rho);
  }
  double leq_tau(uint32_t tau) const
  {
    double s = 0;
    for (uint32_t x = 0; x <= tau && x < hist.size(); ++x) s += hist[x];
    return s;
  }
  double jc() const { return -0.75 * log(1 - 4.0 / 3.0 * d); }
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
    n.kind = nd[c].kids.empty() ? 1 : 2;
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
}

} // namespace

extern "C" {

void kr_place_tree_free(kr_place_tree* pt) { delete pt; }
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

  // ---- phase A (host): per read, the leaf set, the closest, ancestor accumulation, candidates
  struct Cand {
    uint32_t read, se; // placement-tree node
    Acc a;
    bool internal;
  };
  struct ReadPlan {
    bool reported = false, single = false;
    size_t c0 = 0, c1 = 0; // candidate range
    Acc closest;
    uint32_t closest_pt = 0;
  };
  std::vector<ReadPlan> plan(rv->nreads);
  std::vector<Cand> cands;
  for (uint32_t r = 0; r < rv->nreads; ++r) {
    const uint32_t o = rv->read_off[r], n = rv->read_cnt[r];
    if (n == 0) continue;
    const double enmers = (double)((offsets[r + 1] - offsets[r]) >= k ? (offsets[r + 1] - offsets[r]) - k + 1 : 0);
    auto leaf_acc = [&](uint32_t i) {
      Acc a;
      a.leaf = true;
      a.hist.resize(np);
      double mc = 0;
      for (uint32_t x = 0; x < np; ++x) a.hist[x] = rv->rec_hist[(uint64_t)x * rv->rec_hist_stride + i], mc += a.hist[x];
      a.match = mc;
      a.mismatch = (double)rv->read_onmers[r] - mc; // src/query.cpp:104
      a.nmers = enmers;                              // IMers::enmers (src/query.cpp:335-350)
      a.rho = rho_tab[rv->rec_key[i] >> 1];
      a.d = rv->rec_d[i], a.v = rv->rec_v[i];
      return a;
    };
    // closest: last record in (strand, se) order with d <= best (summarize_matches' `<=`, as in the oracle)
    double best = 1.7976931348623157e308;
    int cl = -1;
    for (int strand = 0; strand < 2; ++strand)
      for (uint32_t i = o; i < o + n; ++i)
        if (rv->rec_key[i] && (rv->rec_key[i] & 1u) == (uint32_t)strand && rv->rec_d[i] <= best) best = rv->rec_d[i], cl = (int)i;
    if (cl < 0) continue;
    ReadPlan& pl = plan[r];
    pl.closest = leaf_acc((uint32_t)cl);
    pl.closest_pt = pt->idx_to_pt[rv->rec_key[cl] >> 1];
    if (!(p->no_filter || pl.closest.leq_tau(p->tau) > 1.0)) continue; // src/query.cpp:220
    // node_to_minfo: the record chosen for each leaf (rec_sel under multi / no_filter / no dist-max)
    std::map<uint32_t, Acc> pp; // keyed by placement-tree se: ascending edge order
    uint32_t nleaf = 0;
    for (uint32_t i = o; i < o + n; ++i)
      if (rv->rec_sel[i]) {
        uint32_t q = pt->idx_to_pt[rv->rec_key[i] >> 1];
        if (!q) continue;
        pp[q] = leaf_acc(i);
        nleaf++;
      }
    if (nleaf == 0) continue;
    pl.reported = true;
    pl.c0 = cands.size();
    pl.closest.chisq = 0;
    if (nleaf == 1) { // src/query.cpp:233-244
      pl.single = true;
      cands.push_back(Cand{r, pl.closest_pt, pl.closest, false});
      pl.c1 = cands.size();
      continue;
    }
    std::vector<uint32_t> leaves;
    for (auto& kv : pp) leaves.push_back(kv.first);
    for (uint32_t lf : leaves) { // src/query.cpp:250-267
      const Acc src = pp[lf];
      double denom = 1.0;
      uint32_t par = lf;
      while ((par = pt->t.nodes[par].parent)) {
        denom /= pt->eff[par];
        auto it = pp.find(par);
        if (it == pp.end()) {
          Acc z;
          z.hist.assign(np, 0.0);
          it = pp.emplace(par, z).first;
        }
        it->second.add(src, denom);
      }
    }
    for (auto& kv : pp) { // src/query.cpp:270-283 (likelihoods deferred to the GPU)
      uint32_t q = kv.first;
      uint32_t nch = (uint32_t)pt->kids[q].size();
      if (nch != pt->eff[q] || nch == 1) continue;
      if (p->no_filter || kv.second.leq_tau(p->tau) > 1.0) cands.push_back(Cand{r, q, kv.second, !kv.second.leaf});
    }
    pl.c1 = cands.size();
  }

  // ---- phase B (GPU): Brent on the internal candidates
  {
    std::vector<size_t> which;
    for (size_t i = 0; i < cands.size(); ++i)
      if (cands[i].internal) which.push_back(i);
    std::vector<double> hist(which.size() * np), uc(which.size()), rho(which.size()), d(which.size()), v(which.size());
    for (size_t j = 0; j < which.size(); ++j) {
      const Acc& a = cands[which[j]].a;
      for (uint32_t x = 0; x < np; ++x) hist[j * np + x] = a.hist[x];
      uc[j] = a.mismatch, rho[j] = a.rho;
    }
    int rc = kr_llh_batch(dix, th, 0, which.size(), hist.data(), uc.data(), rho.data(), nullptr, d.data(), v.data());
    if (rc) return rc;
    for (size_t j = 0; j < which.size(); ++j) cands[which[j]].a.d = d[j], cands[which[j]].a.v = v[j];
  }
  // ---- phase C (GPU): chisq = 2 * (f_closest(d_candidate) - v_closest)   (src/query.cpp:276, :420-424)
  {
    std::vector<size_t> which;
    for (size_t i = 0; i < cands.size(); ++i)
      if (!plan[cands[i].read].single) which.push_back(i);
    std::vector<double> hist(which.size() * np), uc(which.size()), rho(which.size()), din(which.size()), f(which.size());
    for (size_t j = 0; j < which.size(); ++j) {
      const Acc& c = plan[cands[which[j]].read].closest;
      for (uint32_t x = 0; x < np; ++x) hist[j * np + x] = c.hist[x];
      uc[j] = c.mismatch, rho[j] = c.rho, din[j] = cands[which[j]].a.d;
    }
    int rc = kr_llh_batch(dix, th, 1, which.size(), hist.data(), uc.data(), rho.data(), din.data(), nullptr, f.data());
    if (rc) return rc;
    for (size_t j = 0; j < which.size(); ++j) {
      Cand& c = cands[which[j]];
      c.a.chisq = 2 * (f[j] - plan[c.read].closest.v);
    }
  }

  // ---- phase D (host): candidate filter, LWR, text
  std::string out;
  std::vector<kr_placement> pls;
  bool prev = *has_previous != 0;
  auto en = [&](uint32_t q) { return q - 1; };
  auto mid = [&](uint32_t q) { return std::isnan(pt->t.nodes[q].blen) ? 0.0 : pt->t.nodes[q].blen / 2.0; };
  auto jfields = [&](uint32_t q, const Acc& a) {
    return "[" + std::to_string(en(q)) + ", " + f5(a.jc() - mid(q)) + ", " + f5(mid(q)) + ", " + f5(-a.v) + ", " + f5(a.lwr) + ", " +
           f5(a.d) + "]";
  };
  auto tfields = [&](uint32_t q, const Acc& a) {
    const std::string& nm = pt->t.nodes[q].label;
    return (nm.empty() ? std::string("NA") : nm) + "\t" + std::to_string(en(q)) + "\t" + f5(a.lwr) + "\t" + f5(a.d);
  };
  auto record = [&](uint32_t r, uint32_t q, const Acc& a) {
    kr_placement x;
    x.read = r, x.edge = en(q), x.lwr = a.lwr, x.d_llh = a.d, x.v_llh = a.v, x.pendant = a.jc() - mid(q), x.distal = mid(q);
    pls.push_back(x);
  };
  const bool jp = tabular == 0, tb = tabular == 1; // 2: --summarize, no text (the caller sums the placements)
  for (uint32_t r = 0; r < rv->nreads; ++r) {
    ReadPlan& pl = plan[r];
    if (!pl.reported) continue;
    std::string id = names ? names[r] : "";
    if (jp) {
      if (prev) out += ",\n";
      out += "\t\t\t{\"n\" : [\"" + id + "\"], \"p\" : [";
      prev = true;
    }
    if (pl.single) {
      Cand& c = cands[pl.c0];
      record(r, c.se, c.a);
      if (tb)
        out += id + "\t" + tfields(c.se, c.a) + "\n";
      else if (jp)
        out += jfields(c.se, c.a) + "]}";
      continue;
    }
    std::vector<size_t> nd_v;
    for (size_t i = pl.c0; i < pl.c1; ++i)
      if (cands[i].a.chisq < p->chisq && pt->t.nodes[cands[i].se].parent) nd_v.push_back(i);
    double total = 0;
    for (size_t i : nd_v) {
      cands[i].a.lwr = exp(-cands[i].a.chisq / 2);
      total = total + cands[i].a.lwr;
    }
    if (p->multi) {
      for (size_t j = 0; j < nd_v.size(); ++j) {
        Cand& c = cands[nd_v[j]];
        c.a.lwr = c.a.lwr / total;
        record(r, c.se, c.a);
        if (j > 0 && jp) out += ",";
        if (tb)
          out += id + "\t" + tfields(c.se, c.a) + "\n";
        else if (jp)
          out += "\n\t\t\t\t" + jfields(c.se, c.a);
      }
      if (jp) out += "]\n\t\t\t}";
    } else {
      if (nd_v.size() > 1)
        std::stable_sort(nd_v.begin(), nd_v.end(), [&](size_t l, size_t rr) {
          uint32_t cl_ = pt->card[cands[l].se], cr = pt->card[cands[rr].se];
          return cl_ == cr ? cands[l].a.d > cands[rr].a.d : cl_ < cr;
        });
      if (nd_v.empty()) {
        if (jp) out += "]}";
        continue;
      }
      Cand& c = cands[nd_v.back()];
      c.a.lwr = c.a.lwr / total;
      record(r, c.se, c.a);
      if (tb)
        out += id + "\t" + tfields(c.se, c.a) + "\n";
      else if (jp)
        out += jfields(c.se, c.a) + "]}";
    }
  }
  *has_previous = prev ? 1 : 0;
  *text = dup_text(out, len);
  if (!*text) return kr::fail(KR_ERR_NOMEM, "kr_place_batch: out of memory");
  if (placements && nplacements) {
    *nplacements = pls.size();
    *placements = (kr_placement*)malloc(std::max<size_t>(1, pls.size()) * sizeof(kr_placement));
    if (!pls.empty()) memcpy(*placements, pls.data(), pls.size() * sizeof(kr_placement));
  }
  return KR_OK;
}

uint32_t kr_place_tree_nnodes(const kr_place_tree* pt) { return pt ? pt->t.nnodes() : 0; }

int kr_place_summary_add(const kr_place_tree* pt, const kr_placement* pls, uint64_t n, double* wcount, double* twcount)
{ // place --summarize: a read with m placements counts 1/m at each (src/query.cpp:232-233,297-298,322-323);
  // summed per 512 reads first, as the reference's batches are (src/krepp.cpp:466-471)
  if (!pt || (!pls && n) || !wcount || !twcount) return kr::fail(KR_ERR_ARG, "kr_place_summary_add: null argument");
  std::map<uint32_t, double> part;
  auto flush = [&]() {
    for (auto& kv : part) *twcount += kv.second, wcount[kv.first] += kv.second;
    part.clear();
  };
  for (uint64_t i = 0; i < n;) {
    uint64_t j = i;
    while (j < n && pls[j].read == pls[i].read) ++j;
    if (i && (pls[i].read >> 9) != (pls[i - 1].read >> 9)) flush();
    for (uint64_t q = i; q < j; ++q) {
      if (pls[q].edge + 1 > pt->t.nnodes()) return kr::fail(KR_ERR_ARG, "kr_place_summary_add: edge out of range");
      part[pls[q].edge + 1] += 1.0 / (double)(j - i);
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
