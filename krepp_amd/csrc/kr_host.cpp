// kr_host.cpp — host side of the drop-in surface: on-disk index reader, backbone tree,
// FASTA/FASTQ batcher and the `dist` report writer.  CPU only; no HIP in this file.
//
// Reference behaviour followed (file:line into bo1929/krepp v0.8.3):
//   index directory discovery        src/krepp.cpp:66-108
//   metadata / inc / cmer / crecord  src/krepp.cpp:18-29, src/index.cpp:51-158,
//                                    src/table.cpp:65-75, src/record.cpp:203-211
//   rho scaling                      src/index.cpp:188-201
//   Newick numbering                 src/phytree.cpp:84-215,394-404
//   balanced tree from a reflist     src/phytree.cpp:217-253
//   FASTX records / batching         src/kseq.h:177-219, src/rqseq.cpp:180-197
//   report text                      src/query.cpp:152-196, src/query.hpp:210
#include "kr_common.h"

#include <sched.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <cctype>
#include <map>
#include <memory>
#include <set>
#include <condition_variable>
#include <deque>
#include <fcntl.h>
#include <mutex>
#include <omp.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <chrono>
#include <sys/mman.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace kr {

static thread_local std::string g_err;
int fail(int code, const std::string& msg)
{
  g_err = msg;
  return code;
}
void clear_error() { g_err.clear(); }

// ---------------------------------------------------------------------------
// MurmurHash3_x86_32 (public-domain algorithm by Austin Appleby), used by the
// reference only to hash node names (src/record.hpp:26-47).
// ---------------------------------------------------------------------------
static inline uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
static uint32_t mm3_32(const uint8_t* p, size_t n, uint32_t seed)
{
  uint32_t hsh = seed;
  size_t nb = n >> 2;
  for (size_t b = 0; b < nb; ++b) {
    uint32_t kk = (uint32_t)p[4 * b] | ((uint32_t)p[4 * b + 1] << 8) | ((uint32_t)p[4 * b + 2] << 16) |
                  ((uint32_t)p[4 * b + 3] << 24);
    kk = rol(kk * 0xcc9e2d51u, 15) * 0x1b873593u;
    hsh = rol(hsh ^ kk, 13) * 5u + 0xe6546b64u;
  }
  uint32_t kk = 0;
  size_t rem = n & 3, base = nb * 4;
  if (rem == 3) kk ^= (uint32_t)p[base + 2] << 16;
  if (rem >= 2) kk ^= (uint32_t)p[base + 1] << 8;
  if (rem >= 1) {
    kk ^= p[base];
    hsh ^= rol(kk * 0xcc9e2d51u, 15) * 0x1b873593u;
  }
  hsh ^= (uint32_t)n;
  hsh ^= hsh >> 16;
  hsh *= 0x85ebca6bu;
  hsh ^= hsh >> 13;
  hsh *= 0xc2b2ae35u;
  hsh ^= hsh >> 16;
  return hsh;
}
uint64_t leaf_name_hash(const std::string& name)
{
  const uint8_t* p = reinterpret_cast<const uint8_t*>(name.data());
  return ((uint64_t)mm3_32(p, name.size(), 0) << 32) | mm3_32(p, name.size(), 1);
}
uint64_t rehash64(uint64_t sh)
{
  uint8_t b[8];
  memcpy(b, &sh, 8);
  return ((uint64_t)mm3_32(b, 8, 0) << 32) | mm3_32(b, 8, 1);
}

// ---------------------------------------------------------------------------
// Tree
// ---------------------------------------------------------------------------
std::string HostTree::name(uint32_t se) const
{
  if (se == 0 || se >= nodes.size()) return "";
  return nodes[se].label.empty() ? std::to_string(se - 1) : nodes[se].label;
}

namespace {

// Token stream equivalent to Tree::split_nwk (src/phytree.cpp:84-148): structural
// characters are tokens of their own; a label/length token (possibly empty) is
// emitted in front of every ')' ':' ',' that does not directly follow '('.
bool tokenize_newick(std::string s, std::vector<std::string>& tk, std::string& err)
{
  if (s.empty()) {
    err = "Given Newick tree seems to be empty?!?.";
    return false;
  }
  if (s.back() == '\n') s.pop_back();
  if (s.empty() || s.back() != ';') {
    err = "Given Newick tree ends with a character other than ';'.";
    return false;
  }
  std::string cur;
  bool in_quote = false, prev_was_quote = false, in_comment = false;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (in_comment) {
      if (c == ']') in_comment = false;
      continue;
    }
    const bool is_q = (c == '\'' || c == '"');
    if (is_q && prev_was_quote) { // doubled quote = literal quote character
      in_quote = false;
      cur += "'";
      continue;
    }
    prev_was_quote = is_q;
    if (is_q) {
      in_quote = !in_quote;
      continue;
    }
    if (in_quote) {
      if (c == '[')
        in_comment = true;
      else
        cur += c;
      continue;
    }
    switch (c) {
      case '(':
        tk.emplace_back("(");
        break;
      case ')':
      case ':':
      case ',':
        if (i == 0 || s[i - 1] != '(') {
          tk.push_back(cur);
          cur.clear();
        }
        tk.emplace_back(1, c);
        break;
      case '[':
      case ']':
        err = "Given Newick tree contains an unquoted label or length with '[' or ']'.";
        return false;
      case ';':
        if (i + 1 == s.size()) {
          i = s.size();
          break;
        }
        err = "Given Newick tree contains an unquoted label or length with ';' (or several trees).";
        return false;
      default:
        if ((c == ' ' || c == '\n') && !cur.empty()) {
          err = "Given Newick tree contains an unquoted label or length with ' ' or newline.";
          return false;
        }
        cur += c;
    }
  }
  if (!cur.empty()) tk.push_back(cur);
  return true;
}

struct NewickParser {
  const std::vector<std::string>& tk;
  size_t at = 0;
  HostTree& t;
  std::string err;
  NewickParser(const std::vector<std::string>& tk_, HostTree& t_)
    : tk(tk_), t(t_)
  {}
  bool is(size_t i, const char* s) const { return i < tk.size() && tk[i] == s; }

  void label_and_length(TreeNode& nd)
  { // src/phytree.cpp:177-188 (internal) == :193-204 (leaf)
    nd.label.clear();
    nd.blen = NAN;
    if (at >= tk.size() || is(at, ",")) return;
    if (!is(at, ":")) nd.label = tk[at++];
    if (is(at, ":")) {
      nd.blen = at + 1 < tk.size() ? atof(tk[at + 1].c_str()) : 0.0;
      at += 2;
    }
  }

  // Returns the se given to the subtree root; children get their numbers first
  // (post-order), exactly as Node::parse assigns `se = ++nnodes` after its children.
  uint32_t subtree()
  {
    TreeNode nd;
    nd.parent = 0;
    if (is(at, "(")) {
      std::vector<uint32_t> kids;
      do {
        ++at;
        uint32_t c = subtree();
        if (!err.empty()) return 0;
        kids.push_back(c);
      } while (is(at, ","));
      if (kids.size() == 1) {
        err = "A node has a single child in the backbone tree! Please suppress unifurcations.";
        return 0;
      }
      nd.kind = 2;
      bool bare_close = false;
      if (is(at, ")")) {
        ++at;
        bare_close = is(at, ")"); // src/phytree.cpp:173-175: return before reading a label
      }
      if (!bare_close) label_and_length(nd);
      t.nodes.push_back(nd);
      uint32_t se = (uint32_t)t.nodes.size() - 1;
      for (uint32_t c : kids) t.nodes[c].parent = se;
      return se;
    }
    label_and_length(nd);
    nd.kind = 1;
    t.nodes.push_back(nd);
    return (uint32_t)t.nodes.size() - 1;
  }
};

void balanced_rec(const std::vector<std::string>& names, size_t lo, size_t hi, HostTree& t, uint32_t& self)
{
  TreeNode nd;
  nd.parent = 0;
  nd.blen = 1.0;
  if (hi - lo == 1) {
    nd.kind = 1;
    nd.label = names[lo];
    t.nodes.push_back(nd);
    self = (uint32_t)t.nodes.size() - 1;
    return;
  }
  // second half first (src/phytree.cpp:235-243)
  size_t mid = lo + (hi - lo) / 2;
  uint32_t a = 0, b = 0;
  balanced_rec(names, mid, hi, t, a);
  balanced_rec(names, lo, mid, t, b);
  nd.kind = 2;
  t.nodes.push_back(nd);
  self = (uint32_t)t.nodes.size() - 1;
  t.nodes[a].parent = self;
  t.nodes[b].parent = self;
}

} // namespace

bool parse_newick(const std::string& text, HostTree& out, std::string& err)
{
  std::vector<std::string> tk;
  if (!tokenize_newick(text, tk, err)) return false;
  out.nodes.clear();
  out.nodes.push_back(TreeNode{"", NAN, 0, 0});
  NewickParser p(tk, out);
  p.subtree();
  if (!p.err.empty()) {
    err = p.err;
    return false;
  }
  return true;
}

void balanced_tree(const std::vector<std::string>& names, HostTree& out)
{
  out.nodes.clear();
  out.nodes.push_back(TreeNode{"", NAN, 0, 0});
  uint32_t root = 0;
  if (!names.empty()) balanced_rec(names, 0, names.size(), out, root);
}

} // namespace kr

// ---------------------------------------------------------------------------
// kr_host_index
// ---------------------------------------------------------------------------
namespace kr {
// CPUs this process may actually use: the hardware threads it is allowed on, capped by the cgroup CPU quota (a container
// can see 256 hardware threads and be granted 16 CPUs' worth of time: a pool sized by the former is throttled by the
// scheduler in the middle of its work)
unsigned usable_cpus()
{
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
  auto quota = [&](const char* path_max, const char* path_quota, const char* path_period) {
    if (FILE* f = fopen(path_max, "r")) { // cgroup v2: "<quota|max> <period>"
      char q[64] = {0};
      long long per = 0;
      if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
        const long long qq = atoll(q);
        if (qq > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (qq + per - 1) / per));
      }
      fclose(f);
      return;
    }
    long long qq = -1, per = 0; // cgroup v1
    if (FILE* f = fopen(path_quota, "r")) {
      if (fscanf(f, "%lld", &qq) != 1) qq = -1;
      fclose(f);
    }
    if (FILE* f = fopen(path_period, "r")) {
      if (fscanf(f, "%lld", &per) != 1) per = 0;
      fclose(f);
    }
    if (qq > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (qq + per - 1) / per));
  };
  quota("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  return n;
}

namespace {
struct Pool {
  struct Call {
    const std::function<void(int)>* f;
    int n, next = 0, done = 0;
  };
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<Call*> calls; // calls with pieces left to hand out
  std::vector<std::thread> threads;
  bool stop = false;
  Pool()
  {
    const char* e = getenv("KR_HOST_THREADS");
    // the caller works too (parallel_width = pool + 1): one thread per usable CPU, at most 32 in all
    const unsigned cpus = kr::usable_cpus();
    unsigned nt = e ? (unsigned)atoi(e) : std::min(31u, std::max(1u, cpus > 1 ? cpus - 1 : 1u));
    for (unsigned t = 0; t < nt; ++t) threads.emplace_back([this] { run(); });
  }
  ~Pool()
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : threads) t.join();
  }
  // take one piece of the front call (lock held); returns false if there is none
  bool take(Call*& c, int& i)
  {
    while (!calls.empty() && calls.front()->next >= calls.front()->n) calls.pop_front();
    if (calls.empty()) return false;
    c = calls.front();
    i = c->next++;
    return true;
  }
  void finish(Call* c)
  {
    std::lock_guard<std::mutex> lk(mu);
    if (++c->done == c->n) cv_done.notify_all();
  }
  void run()
  {
    for (;;) {
      Call* c = nullptr;
      int i = 0;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stop || take(c, i); });
        if (!c) return;
      }
      (*c->f)(i);
      finish(c);
    }
  }
};
Pool& pool()
{
  static Pool p;
  return p;
}
} // namespace

int parallel_width() { return (int)pool().threads.size() + 1; }

void parallel_for(int n, const std::function<void(int)>& f)
{
  if (n <= 0) return;
  if (n == 1) {
    f(0);
    return;
  }
  Pool& P = pool();
  Pool::Call call{&f, n};
  {
    std::lock_guard<std::mutex> lk(P.mu);
    P.calls.push_back(&call);
  }
  P.cv_work.notify_all();
  for (;;) { // the caller works on its own call
    int i;
    {
      std::lock_guard<std::mutex> lk(P.mu);
      if (call.next >= call.n) break;
      i = call.next++;
    }
    f(i);
    P.finish(&call);
  }
  std::unique_lock<std::mutex> lk(P.mu);
  P.cv_done.wait(lk, [&] { return call.done == call.n; });
  for (auto it = P.calls.begin(); it != P.calls.end(); ++it)
    if (*it == &call) {
      P.calls.erase(it);
      break;
    }
}
} // namespace kr

struct kr_host_lib {
  std::string suffix;
  std::vector<uint64_t> inc;
  std::vector<uint32_t> cmer; // interleaved
  std::vector<uint32_t> pse;  // interleaved
  std::vector<double> rho;
  uint32_t nnodes = 0, nsubsets = 0, r = 0, frac = 0, w = 0, nrows_meta = 0;
};

struct kr_host_index {
  uint32_t k = 0, h = 0, m = 0;
  std::vector<uint8_t> ppos, npos;
  std::vector<kr_host_lib> libs;
  kr::HostTree tree;
  std::vector<std::string> names; // printable name per se
  std::vector<uint8_t> kind;
  bool wbackbone = false;
  std::vector<kr_lib_view> lib_views;
};

namespace {

bool slurp(const std::string& path, std::string& out)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)std::max(0L, n));
  size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
  fclose(f);
  return got == (size_t)std::max(0L, n);
}

template <typename T>
bool take(const std::string& buf, size_t& off, T& v)
{
  if (off + sizeof(T) > buf.size()) return false;
  memcpy(&v, buf.data() + off, sizeof(T));
  off += sizeof(T);
  return true;
}

template <typename T>
bool read_array_file(const std::string& path, size_t header_bytes, void* header, std::vector<T>& out,
                     uint64_t (*count_of)(const void*))
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = fread(header, 1, header_bytes, f) == header_bytes;
  if (ok) {
    uint64_t n = count_of(header);
    out.resize(n);
    ok = n == 0 || fread(out.data(), sizeof(T), n, f) == n;
  }
  fclose(f);
  return ok;
}

} // namespace

extern "C" {

int kr_host_index_load(const char* index_dir, kr_host_index** out)
{
  kr::clear_error();
  if (!index_dir || !out) return kr::fail(KR_ERR_ARG, "kr_host_index_load: null argument");
  *out = nullptr;
  static const std::set<std::string> known{"cmer", "crecord", "inc", "metadata", "tree", "reflist"};
  DIR* d = opendir(index_dir);
  if (!d) return kr::fail(KR_ERR_IO, std::string("cannot open index directory ") + index_dir);
  // group files by everything after the type: "-m<M>r<R>-frac" (src/krepp.cpp:72-87)
  std::map<std::string, std::set<std::string>> groups;
  while (dirent* e = readdir(d)) {
    std::string fn = e->d_name;
    size_t p1 = fn.find('-');
    if (p1 == std::string::npos) continue;
    size_t p2 = fn.find('-', p1 + 1);
    if (p2 == std::string::npos) continue;
    if (!known.count(fn.substr(0, p1))) continue;
    size_t dot = fn.rfind('.');
    if (dot != std::string::npos && dot != 0) continue; // files with an extension are ignored
    groups[fn.substr(p1)].insert(fn.substr(0, p1));
  }
  closedir(d);
  if (groups.empty()) return kr::fail(KR_ERR_IO, std::string("no partial index in ") + index_dir);

  std::unique_ptr<kr_host_index> hx(new kr_host_index());
  const std::string dir = index_dir;
  std::vector<std::string> first_order;
  std::set<uint32_t> residues;
  for (auto& g : groups) {
    const std::string& sfx = g.first;
    const std::set<std::string>& have = g.second;
    bool core = have.count("cmer") && have.count("crecord") && have.count("inc") && have.count("metadata");
    if (!core || !(have.count("tree") || have.count("reflist")))
      return kr::fail(KR_ERR_FORMAT, "There is a partial index with a missing file!");

    // --- tree (src/index.cpp:3-49)
    kr::HostTree t;
    if (have.count("tree")) {
      std::string nwk, err;
      if (!slurp(dir + "/tree" + sfx, nwk)) return kr::fail(KR_ERR_IO, "Failed to open " + dir + "/tree" + sfx);
      if (!kr::parse_newick(nwk, t, err)) return kr::fail(KR_ERR_FORMAT, err);
      hx->wbackbone = true;
    } else {
      std::string txt;
      if (!slurp(dir + "/reflist" + sfx, txt))
        return kr::fail(KR_ERR_IO, "Unable to open reference list file for an index without a tree.");
      std::vector<std::string> names;
      size_t a = 0;
      while (a < txt.size()) {
        size_t b = txt.find('\n', a);
        if (b == std::string::npos) b = txt.size();
        names.push_back(txt.substr(a, b - a));
        a = b + 1;
      }
      kr::balanced_tree(names, t);
      hx->wbackbone = false;
    }
    std::vector<std::string> order;
    for (uint32_t se = 1; se <= t.nnodes(); ++se) order.push_back(t.nodes[se].label);
    if (hx->libs.empty()) {
      hx->tree = t;
      first_order = order;
    } else if (order != first_order) { // Tree::check_compatible (src/phytree.cpp:10-36)
      return kr::fail(KR_ERR_FORMAT, "Partial libraries are based on different trees!");
    }

    // --- metadata (src/krepp.cpp:18-29; src/index.cpp:58-70)
    std::string md;
    if (!slurp(dir + "/metadata" + sfx, md)) return kr::fail(KR_ERR_IO, "Failed to open " + dir + "/metadata" + sfx);
    size_t off = 0;
    uint8_t k8, w8, h8, frac8;
    uint32_t m32, r32, nrows32;
    if (!(take(md, off, k8) && take(md, off, w8) && take(md, off, h8) && take(md, off, m32) && take(md, off, r32) &&
          take(md, off, frac8) && take(md, off, nrows32)) ||
        h8 > k8 || off + k8 > md.size())
      return kr::fail(KR_ERR_FORMAT, "Failed to read the metadata of a partial index!");
    std::vector<uint8_t> ppos(md.begin() + off, md.begin() + off + h8);
    std::vector<uint8_t> npos(md.begin() + off + h8, md.begin() + off + k8);
    if (k8 > 31 || k8 - h8 > 16 || m32 == 0)
      return kr::fail(KR_ERR_FORMAT, "unsupported k/h/m in metadata" + sfx);
    if (hx->libs.empty()) {
      hx->k = k8, hx->h = h8, hx->m = m32;
      hx->ppos = ppos, hx->npos = npos;
    } else if (hx->k != k8 || hx->h != h8 || hx->m != m32 || hx->ppos != ppos || hx->npos != npos) {
      return kr::fail(KR_ERR_FORMAT, "Partial libraries have incompatible hash functions!"); // src/lshf.cpp:159-180
    }

    kr_host_lib lib;
    lib.suffix = sfx;
    lib.r = r32, lib.frac = frac8 ? 1 : 0, lib.w = w8, lib.nrows_meta = nrows32;
    // --- cmer + inc (src/table.cpp:65-75)
    uint64_t nk = 0;
    if (!read_array_file<uint32_t>(dir + "/cmer" + sfx, 8, &nk, lib.cmer,
                                   [](const void* h) { return 2 * *(const uint64_t*)h; }))
      return kr::fail(KR_ERR_IO, "Failed to read the k-mer vector of a partial index!");
    uint32_t nr = 0;
    if (!read_array_file<uint64_t>(dir + "/inc" + sfx, 4, &nr, lib.inc,
                                   [](const void* h) { return (uint64_t) * (const uint32_t*)h; }))
      return kr::fail(KR_ERR_IO, "Failed to read the offset array of a partial index!");
    if (!lib.inc.empty() && lib.inc.back() > nk)
      return kr::fail(KR_ERR_FORMAT, "inc" + sfx + " points past the end of cmer" + sfx);
    // --- crecord (src/record.cpp:203-211)
    {
      std::string cr;
      if (!slurp(dir + "/crecord" + sfx, cr)) return kr::fail(KR_ERR_IO, "Failed to open " + dir + "/crecord" + sfx);
      size_t o = 0;
      if (!take(cr, o, lib.nnodes) || !take(cr, o, lib.nsubsets) ||
          cr.size() < o + (size_t)lib.nsubsets * 8 + (size_t)lib.nnodes * 8)
        return kr::fail(KR_ERR_FORMAT, "Failed to read the color array of a partial index!");
      lib.pse.resize((size_t)lib.nsubsets * 2);
      memcpy(lib.pse.data(), cr.data() + o, (size_t)lib.nsubsets * 8);
      o += (size_t)lib.nsubsets * 8;
      lib.rho.resize(lib.nnodes);
      memcpy(lib.rho.data(), cr.data() + o, (size_t)lib.nnodes * 8);
    }
    if (lib.nnodes != hx->tree.nnodes() + 1)
      return kr::fail(KR_ERR_FORMAT, "crecord" + sfx + " does not match the tree (nnodes)");
    // residues served by this library (src/index.cpp:144-157)
    if (lib.frac) {
      for (uint32_t q = 0; q <= lib.r; ++q) residues.insert(q);
    } else {
      residues.insert(lib.r);
    }
    hx->libs.push_back(std::move(lib));
  }
  // Index::make_rho_partial (src/index.cpp:188-201)
  const double coef = (double)residues.size() / (double)hx->m;
  for (auto& lib : hx->libs)
    for (double& v : lib.rho) v *= coef;

  uint32_t nn = hx->tree.nnodes();
  hx->names.resize(nn + 1);
  hx->kind.assign(nn + 1, 0);
  for (uint32_t se = 1; se <= nn; ++se) {
    hx->names[se] = hx->tree.name(se);
    hx->kind[se] = hx->tree.nodes[se].kind;
  }
  for (auto& lib : hx->libs) {
    kr_lib_view v;
    memset(&v, 0, sizeof(v));
    v.inc = lib.inc.data(), v.cmer = lib.cmer.data(), v.pse = lib.pse.data(), v.rho = lib.rho.data();
    v.nkmers = lib.cmer.size() / 2, v.nrows = (uint32_t)lib.inc.size(), v.nsubsets = lib.nsubsets;
    v.nnodes = lib.nnodes, v.r = lib.r, v.frac = lib.frac, v.w = lib.w;
    hx->lib_views.push_back(v);
  }
  *out = hx.release();
  return KR_OK;
}

// `krepp seek`: a sketch (Sketch::load_full_sketch, src/sketch.cpp:3-24; SFlatHT::load, src/table.cpp:24-33) is the
// table of ONE reference without colours.  It is presented as an index with a single library and a one-leaf
// tree -- every entry carries the colour of that leaf -- so that the device path of `dist` serves it unchanged:
// per strand the accumulator of the only leaf IS SSummary's histogram (per k-mer position the minimum
// Hamming distance over the bucket, src/seek.cpp:104-121).  rho is scaled as Sketch::make_rho_partial does
// (src/sketch.cpp:26-33).
int kr_host_sketch_load(const char* sketch_path, kr_host_index** out)
{
  kr::clear_error();
  if (!sketch_path || !out) return kr::fail(KR_ERR_ARG, "kr_host_sketch_load: null argument");
  *out = nullptr;
  std::string buf;
  if (!slurp(sketch_path, buf)) return kr::fail(KR_ERR_IO, std::string("Failed to read the sketch file! ") + sketch_path);
  const std::string bad = "Failed to read the sketch file!";
  size_t off = 0;
  uint64_t nkmers = 0;
  uint32_t nrows = 0, m = 0, r = 0, nrows2 = 0;
  uint8_t k = 0, w = 0, h = 0, frac = 0;
  if (!take(buf, off, nkmers) || nkmers > (buf.size() - off) / 4) return kr::fail(KR_ERR_FORMAT, bad);
  const size_t enc_off = off;
  off += (size_t)nkmers * 4;
  if (!take(buf, off, nrows) || nrows > (buf.size() - off) / 8) return kr::fail(KR_ERR_FORMAT, bad);
  const size_t inc_off = off;
  off += (size_t)nrows * 8;
  if (!take(buf, off, k) || !take(buf, off, w) || !take(buf, off, h) || !take(buf, off, m) || !take(buf, off, r) ||
      !take(buf, off, frac) || !take(buf, off, nrows2))
    return kr::fail(KR_ERR_FORMAT, bad);
  if (k < 1 || k > 32 || h < 1 || h > k || m == 0 || off + k + 8 > buf.size()) return kr::fail(KR_ERR_FORMAT, bad);
  std::unique_ptr<kr_host_index> hx(new kr_host_index());
  hx->k = k, hx->h = h, hx->m = m;
  hx->ppos.assign(buf.begin() + off, buf.begin() + off + h);
  hx->npos.assign(buf.begin() + off + h, buf.begin() + off + k);
  off += k;
  double rho = 0;
  take(buf, off, rho);
  rho *= frac ? ((double)r + 1.0) / (double)m : 1.0 / (double)m;
  kr_host_lib lib;
  lib.r = r, lib.frac = frac, lib.w = w, lib.nrows_meta = nrows2;
  lib.inc.resize(nrows);
  memcpy(lib.inc.data(), buf.data() + inc_off, (size_t)nrows * 8);
  if (nrows && lib.inc.back() != nkmers) return kr::fail(KR_ERR_FORMAT, bad);
  lib.cmer.resize((size_t)nkmers * 2);
  for (uint64_t i = 0; i < nkmers; ++i) {
    memcpy(&lib.cmer[2 * i], buf.data() + enc_off + 4 * i, 4);
    lib.cmer[2 * i + 1] = 1; // the colour of the only leaf
  }
  lib.nnodes = 2, lib.nsubsets = 2;
  lib.pse.assign(4, 0);
  lib.rho = {0.0, rho};
  hx->libs.push_back(std::move(lib));
  std::string label = sketch_path;
  const size_t slash = label.rfind('/');
  if (slash != std::string::npos) label = label.substr(slash + 1);
  hx->tree.nodes.resize(2);
  hx->tree.nodes[0] = kr::TreeNode{"", NAN, 0, 0};
  hx->tree.nodes[1] = kr::TreeNode{label, NAN, 0, 1};
  hx->names = {"", label};
  hx->kind = {0, 1};
  hx->wbackbone = false;
  {
    const kr_host_lib& L = hx->libs[0];
    kr_lib_view v;
    memset(&v, 0, sizeof(v));
    v.inc = L.inc.data(), v.cmer = L.cmer.data(), v.pse = L.pse.data(), v.rho = L.rho.data();
    v.nkmers = nkmers, v.nrows = nrows, v.nsubsets = L.nsubsets, v.nnodes = L.nnodes, v.r = r, v.frac = frac, v.w = w;
    hx->lib_views.push_back(v);
  }
  *out = hx.release();
  return KR_OK;
}

void kr_host_index_free(kr_host_index* h) { delete h; }

int kr_host_index_view(const kr_host_index* h, kr_index_view* v)
{
  if (!h || !v) return kr::fail(KR_ERR_ARG, "kr_host_index_view: null argument");
  memset(v, 0, sizeof(*v));
  v->k = h->k, v->h = h->h, v->m = h->m;
  v->ppos = h->ppos.data(), v->npos = h->npos.data();
  v->nlibs = (uint32_t)h->lib_views.size(), v->libs = h->lib_views.data();
  v->tree_nnodes = h->tree.nnodes(), v->node_kind = h->kind.data();
  v->wbackbone = h->wbackbone;
  return KR_OK;
}

const char* kr_host_index_node_name(const kr_host_index* h, uint32_t se)
{
  return (h && se < h->names.size()) ? h->names[se].c_str() : "";
}
const char* kr_host_index_node_label(const kr_host_index* h, uint32_t se)
{
  return (h && se < h->tree.nodes.size()) ? h->tree.nodes[se].label.c_str() : "";
}
uint32_t kr_host_index_node_parent(const kr_host_index* h, uint32_t se)
{
  return (h && se < h->tree.nodes.size()) ? h->tree.nodes[se].parent : 0;
}
double kr_host_index_node_blen(const kr_host_index* h, uint32_t se)
{
  return (h && se && se < h->tree.nodes.size()) ? h->tree.nodes[se].blen : NAN;
}

void kr_params_default(kr_params* p)
{ // src/krepp.hpp:206-221, src/krepp.cpp:635-644
  p->hdist_th = 4;
  p->tau = 2;
  p->chisq = 2.706;
  p->dist_max = NAN;
  p->multi = 1;
  p->no_filter = 1;
}

} // extern "C"

// ---------------------------------------------------------------------------
// FASTA/FASTQ batcher.  Record grammar is kseq's (src/kseq.h:177-219):
//  - a record starts at the next '>' or '@';
//  - the name runs to the first whitespace, the rest of the line is a comment;
//  - sequence bytes are every isgraph() byte up to the next '>', '+' or '@'
//    (wherever it occurs, not only at line starts);
//  - after '+' the rest of that line is skipped and quality bytes (33..127) are
//    read until as many as the sequence have been seen; one further byte is
//    consumed; a short quality string ends reading like EOF (return -2 is < 0,
//    src/rqseq.cpp:189).
// ---------------------------------------------------------------------------
namespace {
// One four-line FASTQ record at b[p .. end) with a clean sequence line (graphic characters only, none of the
// record markers) and a quality line of the same length -- the records for which kseq's character rules
// reduce to "four lines".  `mk` is a marker kseq has already consumed (0: none).  Ok: name = b[nb .. ne),
// sequence = b[s0 .. s0 + slen), the next record starts at `next`.  More: the record is not complete in the
// buffer.  No: anything else (FASTA, wrapped lines, '\r', odd quality ...).
enum class Clean { Ok, More, No };
inline Clean clean_record(const unsigned char* b, size_t p, size_t end, int mk, size_t& nb, size_t& ne, size_t& s0, size_t& slen,
                          size_t& next)
{
  if (mk == 0) {
    if (p >= end) return Clean::More;
    mk = b[p++];
  }
  if (mk != '@') return Clean::No;
  const unsigned char* nl1 = p < end ? (const unsigned char*)memchr(b + p, '\n', end - p) : nullptr;
  if (!nl1) return Clean::More;
  s0 = (size_t)(nl1 - b) + 1;
  const unsigned char* nl2 = s0 < end ? (const unsigned char*)memchr(b + s0, '\n', end - s0) : nullptr;
  if (!nl2) return Clean::More;
  slen = (size_t)(nl2 - b) - s0;
  const size_t plus = s0 + slen + 1;
  if (plus >= end) return Clean::More;
  if (b[plus] != '+') return Clean::No;
  // (the separator line is "+" and nothing else in nearly every file: look before calling memchr)
  const unsigned char* nl3 = (plus + 1 < end && b[plus + 1] == '\n') ? b + plus + 1 : (const unsigned char*)memchr(b + plus, '\n', end - plus);
  if (!nl3) return Clean::More;
  const size_t q0 = (size_t)(nl3 - b) + 1;
  if (q0 + slen >= end) return Clean::More; // the quality characters and the one character kseq reads past them
  unsigned bad = 0;
  size_t i = 0;
#if defined(__SSE2__)
  { // 16 characters a step (these two checks were 40 % of the parser's time as byte loops: 100 of 260 ns per record)
    const __m128i lo = _mm_set1_epi8(33), hi_s = _mm_set1_epi8(93), hi_q = _mm_set1_epi8(94);
    const __m128i c_gt = _mm_set1_epi8('>'), c_pl = _mm_set1_epi8('+'), c_at = _mm_set1_epi8('@');
    __m128i acc = _mm_setzero_si128();
    for (; i + 16 <= slen; i += 16) {
      const __m128i s = _mm_loadu_si128((const __m128i*)(b + s0 + i)), q = _mm_loadu_si128((const __m128i*)(b + q0 + i));
      // c - 33 > 93 (unsigned): what is left after a saturating subtraction of 93
      acc = _mm_or_si128(acc, _mm_subs_epu8(_mm_sub_epi8(s, lo), hi_s));
      acc = _mm_or_si128(acc, _mm_subs_epu8(_mm_sub_epi8(q, lo), hi_q));
      acc = _mm_or_si128(acc, _mm_or_si128(_mm_cmpeq_epi8(s, c_gt), _mm_or_si128(_mm_cmpeq_epi8(s, c_pl), _mm_cmpeq_epi8(s, c_at))));
    }
    bad = _mm_movemask_epi8(_mm_cmpeq_epi8(acc, _mm_setzero_si128())) != 0xFFFF;
  }
#endif
  for (; i < slen; ++i) {
    const unsigned c = b[s0 + i];
    bad |= (unsigned)(c - 33u > 93u) | (unsigned)(c == '>') | (unsigned)(c == '+') | (unsigned)(c == '@');
    bad |= (unsigned)((unsigned)b[q0 + i] - 33u > 94u);
  }
  if (bad) return Clean::No;
  nb = p;
  ne = p; // name: up to the first whitespace
  while (ne < s0 - 1 && !isspace(b[ne])) ++ne;
  next = q0 + slen + 1;
  return Clean::Ok;
}

// ---- parallel parsing of plain (uncompressed) FASTQ files -------------------------------------------------
// The file is cut into chunks at guessed record starts (a line starting with '@' whose second next line
// starts with '+'); a pool of threads parses the chunks with clean_record.  A chunk counts only if its parse
// starts at a verified record start (offset 0, or the exact end of the previous verified chunk) and consumes
// the chunk exactly; by induction every accepted record is the one kseq would have produced.  The first chunk
// that does not (other formats, a wrong guess, the tail of the file) hands its clean prefix over and the
// reader continues sequentially -- character by character where needed -- from that offset.
struct FqChunk {
  uint64_t a = 0, b = 0; // file range
  std::vector<uint8_t> bases;
  std::vector<uint64_t> offsets{0};
  std::string name_blob;
  std::vector<size_t> name_off;
  bool ok = false;       // consumed [a, b) exactly
  uint64_t clean_end = 0; // file offset where the clean prefix ends
  bool ready = false;
  bool ramp = false;     // one of the pool's short first chunks: a batch of its own however short (kr_fastx_next)
  // (chunks are recycled with their buffers: fresh multi-MB allocations per chunk cost more in page faults than the parse they hold)
  std::unique_ptr<unsigned char[]> raw;
  size_t raw_cap = 0;
  void reset(uint64_t a_, uint64_t b_)
  {
    a = a_, b = b_, ok = ready = false, clean_end = 0;
    bases.clear(), offsets.assign(1, 0), name_blob.clear(), name_off.clear();
  }
};

struct FqPool {
  int fd = -1;
  uint64_t size = 0, next_off = 0, chunk_bytes = 0;
  bool issued_all = false;
  size_t depth = 0;
  uint64_t issued = 0; // chunks handed to the pool so far
  std::deque<std::unique_ptr<FqChunk>> inflight; // file order
  std::vector<std::unique_ptr<FqChunk>> spare;   // handed out and taken back by the reader thread only
  std::deque<FqChunk*> todo;
  std::mutex mu;
  std::condition_variable cv_todo, cv_ready;
  bool stop = false;
  std::vector<std::thread> threads;

  static bool pread_all(int fd, unsigned char* dst, uint64_t off, size_t n)
  {
    while (n) {
      ssize_t k = pread(fd, dst, n, (off_t)off);
      if (k <= 0) return false;
      dst += k, off += (uint64_t)k, n -= (size_t)k;
    }
    return true;
  }
  // (compiled twice: the record checks are byte loops that AVX2 takes 32 at a time; the dispatch is the loader's, at run time)
  // the file mapped once (KR_FASTX_MMAP=0: pread into a buffer of the chunk's, as until round 5): a chunk is parsed where the page
  // cache holds it -- no copy of the 315 bytes of a record before the 160 that are kept
  const unsigned char* map = nullptr;
  bool open_map()
  {
    if (getenv("KR_FASTX_MMAP") && atoi(getenv("KR_FASTX_MMAP")) == 0) return false;
    void* m = mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return false;
    (void)madvise(m, (size_t)size, MADV_SEQUENTIAL);
    map = (const unsigned char*)m;
    return true;
  }
  void parse(FqChunk& c)
  {
    const size_t len = (size_t)(c.b - c.a);
    const unsigned char* src = nullptr;
    if (map) {
      src = map + c.a;
    } else {
      if (len > c.raw_cap || !c.raw) c.raw.reset(new unsigned char[len + len / 8 + 64]), c.raw_cap = len + len / 8 + 64; // not zero-filled
      if (pread_all(fd, c.raw.get(), c.a, len)) src = c.raw.get();
    }
    size_t p = 0;
    if (src) {
      c.bases.reserve(len / 2);
      size_t nb, ne, s0, slen, next;
      while (p < len && clean_record(src, p, len, 0, nb, ne, s0, slen, next) == Clean::Ok) {
        c.name_off.push_back(c.name_blob.size());
        c.name_blob.append((const char*)src + nb, ne - nb);
        c.name_blob.push_back('\0');
        c.bases.insert(c.bases.end(), src + s0, src + s0 + slen);
        c.offsets.push_back(c.bases.size());
        p = next;
      }
    }
    c.ok = p == len && len != 0;
    c.clean_end = c.a + p;
  }
  void work()
  {
    for (;;) {
      FqChunk* c = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_todo.wait(lk, [&] { return stop || !todo.empty(); });
        if (stop) return;
        c = todo.front();
        todo.pop_front();
      }
      parse(*c);
      {
        std::lock_guard<std::mutex> lk(mu);
        c->ready = true;
      }
      cv_ready.notify_all();
    }
  }
  // a record start at or after t: UINT64_MAX if none is found nearby
  uint64_t boundary(uint64_t t)
  {
    if (t >= size) return size;
    std::vector<unsigned char> w((size_t)std::min<uint64_t>(256u << 10, size - t));
    if (!pread_all(fd, w.data(), t, w.size())) return UINT64_MAX;
    const unsigned char *b = w.data(), *e = b + w.size();
    const unsigned char* nl = (const unsigned char*)memchr(b, '\n', w.size());
    for (int tries = 0; nl && tries < 16; ++tries) {
      const unsigned char* c = nl + 1;
      if (c >= e) break;
      const unsigned char* l1 = (const unsigned char*)memchr(c, '\n', (size_t)(e - c));
      if (!l1) break;
      const unsigned char* l2 = l1 + 1 < e ? (const unsigned char*)memchr(l1 + 1, '\n', (size_t)(e - l1 - 1)) : nullptr;
      if (!l2 || l2 + 1 >= e) break;
      if (*c == '@' && l2[1] == '+') return t + (uint64_t)(c - b);
      nl = l1;
    }
    return UINT64_MAX;
  }
  void issue()
  {
    while (!issued_all && inflight.size() < depth) {
      // the first chunks are short (an eighth, then half of a chunk): every thread starts at once, and with whole chunks the first
      // batch would reach the caller only when a whole chunk has been parsed by one thread (65 ms for 78 MB -- a fifth of the time
      // an 8 M-read file takes)
      const uint64_t nth = std::max<size_t>(1, depth / 2);
      const uint64_t cb = issued < nth ? std::max<uint64_t>(4096, chunk_bytes / 8) : (issued < 2 * nth ? std::max<uint64_t>(4096, chunk_bytes / 2) : chunk_bytes);
      ++issued;
      const uint64_t b = next_off >= size ? size : boundary(next_off + cb);
      if (next_off >= size || b == UINT64_MAX || b <= next_off) { // end of file, or no usable cut: the rest is sequential
        issued_all = true;
        break;
      }
      std::unique_ptr<FqChunk> c;
      if (!spare.empty()) c = std::move(spare.back()), spare.pop_back();
      else c.reset(new FqChunk());
      c->reset(next_off, b);
      c->ramp = cb < chunk_bytes;
      next_off = b;
      {
        std::lock_guard<std::mutex> lk(mu);
        todo.push_back(c.get());
      }
      inflight.push_back(std::move(c));
      cv_todo.notify_one();
    }
  }
  void shutdown()
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_todo.notify_all();
    for (auto& t : threads) t.join();
    threads.clear();
    inflight.clear();
    todo.clear();
    if (map) munmap((void*)map, (size_t)size), map = nullptr;
    if (fd >= 0) close(fd);
    fd = -1;
  }
};
} // namespace

#include "kr_pgz.inc"

// ---- block-gzipped input (BGZF: a gzip file made of independent members of <= 64 KB, each announcing its
// compressed size in a "BC" extra field and its inflated size in its trailer) ------------------------------------
// Ordinary gzip is one dependent stream and is inflated by one thread (zlib, below).  BGZF members are independent:
// the reader walks the member headers, hands runs of members to a few threads (raw inflate + CRC check each) and
// consumes the inflated runs in file order, like gzread would deliver them.
struct BgzfSource {
  struct Buf { // grows, never shrinks, not zero-filled: runs are recycled (fresh multi-MB allocations per run cost more
               // in page faults than the inflate they hold)
    std::unique_ptr<unsigned char[]> p;
    size_t cap = 0, len = 0;
    unsigned char* ensure(size_t n)
    {
      if (n > cap || !p) p.reset(new unsigned char[n + n / 4 + 64]), cap = n + n / 4 + 64; // never NULL: zlib rejects a NULL next_out even for an empty member
      len = n;
      return p.get();
    }
  };
  struct Run {
    uint64_t off = 0, bytes = 0; // compressed range
    Buf in, out;
    bool ok = false, ready = false;
    pgz::Records rec; // the records inside the run: out[h .. t) (kr_pgz.inc: find_records)
    size_t h = 0, t = 0;
  };
  pgz::Stitch st; // runs are handed out with their records (next_records) while st.parsed_mode; then as bytes (read)
  std::vector<std::unique_ptr<Run>> spare;
  int fd = -1;
  uint64_t size = 0, next_off = 0;
  bool issued_all = false, failed = false;
  uint64_t trunc_at = UINT64_MAX; // offset of a member that the file ends in the middle of
  size_t depth = 0;
  std::deque<std::unique_ptr<Run>> inflight;
  std::deque<Run*> todo;
  std::mutex mu;
  std::condition_variable cv_todo, cv_ready;
  bool stop = false;
  std::vector<std::thread> threads;
  std::unique_ptr<Run> cur;
  size_t cur_pos = 0;

  // size of the member at `off` (0: not a BGZF member / end of file)
  static uint32_t member_size(const unsigned char* h, size_t avail)
  {
    if (avail < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
    const uint32_t xlen = h[10] | (h[11] << 8);
    if (avail < 12 + (size_t)xlen) return 0;
    for (uint32_t p = 12; p + 4 <= 12 + xlen;) {
      const uint32_t slen = h[p + 2] | (h[p + 3] << 8);
      if (h[p] == 'B' && h[p + 1] == 'C' && slen == 2 && p + 6 <= 12 + xlen) return (uint32_t)(h[p + 4] | (h[p + 5] << 8)) + 1u;
      p += 4 + slen;
    }
    return 0;
  }
  static bool is_bgzf(int fd)
  {
    unsigned char h[64];
    const ssize_t n = pread(fd, h, sizeof(h), 0);
    return n >= 18 && member_size(h, (size_t)n) != 0;
  }
  void inflate_run(Run& r, z_stream& zs)
  {
    unsigned char* const in = r.in.ensure((size_t)r.bytes);
    const size_t in_len = (size_t)r.bytes;
    if (!FqPool::pread_all(fd, in, r.off, in_len)) return;
    // pass 1: member boundaries and inflated sizes
    std::vector<std::array<uint64_t, 3>> mem; // offset in `in`, compressed size, offset in `out`
    uint64_t p = 0, total = 0;
    while (p < in_len) {
      const uint32_t ms = member_size(in + p, in_len - p);
      if (ms < 28 || p + ms > in_len) return;
      const uint32_t xl = in[p + 10] | (in[p + 11] << 8);
      if (12ull + xl + 8 > ms) return; // the extra field claims more than the member holds: no room for a payload and a trailer
      const unsigned char* t = in + p + ms - 4;
      const uint32_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
      if (isize > 65536u) return; // BGZF members inflate to at most 64 KB: a trailer that claims more is damage
      mem.push_back({p, ms, total});
      total += isize;
      p += ms;
    }
    unsigned char* const outp = r.out.ensure((size_t)total);
    for (auto& m : mem) {
      const unsigned char* h = in + m[0];
      const uint32_t xlen = h[10] | (h[11] << 8), ms = (uint32_t)m[1];
      const unsigned char* t = h + ms - 8;
      const uint32_t crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
      const uint32_t isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
      if (inflateReset(&zs) != Z_OK) return;
      zs.next_in = const_cast<unsigned char*>(h + 12 + xlen);
      zs.avail_in = ms - 12 - xlen - 8;
      zs.next_out = outp + m[2];
      zs.avail_out = isize;
      const int rc = inflate(&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.avail_out != 0) return;
      if (crc32(crc32(0L, Z_NULL, 0), outp + m[2], isize) != crc) return;
    }
    if (st.parsed_mode.load(std::memory_order_relaxed)) pgz::find_records(outp, (size_t)total, r.off == 0, r.rec, r.h, r.t); // (the reader may clear the flag meanwhile: either is fine)
    else r.rec.clear(), r.h = r.t = (size_t)total;
    r.ok = true;
  }
  void work()
  {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return;
    for (;;) {
      Run* r = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_todo.wait(lk, [&] { return stop || !todo.empty(); });
        if (stop) break;
        r = todo.front();
        todo.pop_front();
      }
      try {
        inflate_run(*r, zs);
      } catch (const std::bad_alloc&) { // a run that cannot be held is a failed run (reported as a damaged file), not std::terminate
        r->ok = false;
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        r->ready = true;
      }
      cv_ready.notify_all();
    }
    inflateEnd(&zs);
  }
  void issue()
  { // walk the member headers: runs of about 1 MB of compressed data
    while (!issued_all && inflight.size() < depth) {
      uint64_t o = next_off;
      unsigned char h[18 + 256];
      while (o < size && o - next_off < (1u << 20)) {
        const ssize_t n = pread(fd, h, sizeof(h), (off_t)o);
        const uint32_t ms = n > 0 ? member_size(h, (size_t)n) : 0;
        if (ms == 0 || o + ms > size) { // hand over what is complete; then: a member cut short by the end of the file
          trunc_at = ms ? o : UINT64_MAX; // is an error, anything that is not a BGZF member (an ordinary gzip member
          issued_all = true;              // appended to the file, say) is zlib's from there on
          break;
        }
        o += ms;
      }
      if (o >= size) issued_all = true;
      if (o == next_off) break;
      std::unique_ptr<Run> r;
      if (!spare.empty()) r = std::move(spare.back()), spare.pop_back();
      else r.reset(new Run());
      r->ok = r->ready = false;
      r->off = next_off, r->bytes = o - next_off;
      next_off = o;
      {
        std::lock_guard<std::mutex> lk(mu);
        todo.push_back(r.get());
      }
      inflight.push_back(std::move(r));
      cv_todo.notify_one();
    }
  }
  // the next run into `cur`: 1, or what read() returns when there is none (0 at the end, -1 on a damaged file, -2: continue with zlib at next_off)
  int next_run()
  {
    if (cur) spare.push_back(std::move(cur)); // only this thread touches `spare`
    cur.reset();
    issue();
    if (inflight.empty()) {
      if (failed || trunc_at == next_off) return -1;
      return next_off < size ? -2 : 0;
    }
    cur = std::move(inflight.front());
    inflight.pop_front();
    cur_pos = 0;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_ready.wait(lk, [&] { return cur->ready; });
    }
    if (!cur->ok) {
      failed = true;
      cur.reset();
      return -1;
    }
    issue();
    return 1;
  }
  // Records mode: the next run's records appended to `dst`.  1: done, 0: no more records this way (the end of the block-gzipped
  // part, or bytes that are not clean four-line FASTQ: read() continues), -1: damaged file.
  int next_records(pgz::Records& dst)
  {
    if (!st.parsed_mode) return 0;
    const int k = next_run();
    if (k == -1) return -1;
    if (k != 1) { // (read() reports the same 0 / -2 again, behind the bytes the stitcher still holds)
      st.end_of_records();
      return 0;
    }
    const int rc = st.take(dst, cur->out.p.get(), cur->out.len, cur->h, cur->t, cur->rec);
    cur_pos = cur->out.len; // taken, as records or into the stitcher's byte queue
    return rc;
  }
  // like gzread: up to n bytes, 0 at the end, -1 on a damaged file
  int read(unsigned char* dst, unsigned n)
  {
    if (st.parsed_mode) st.end_of_records();
    for (;;) {
      if (const unsigned k = st.drain(dst, n)) return (int)k;
      if (cur && cur_pos < cur->out.len) {
        const unsigned k = (unsigned)std::min<size_t>(n, cur->out.len - cur_pos);
        memcpy(dst, cur->out.p.get() + cur_pos, k);
        cur_pos += k;
        return (int)k;
      }
      const int k = next_run();
      if (k != 1) return k;
    }
  }
  void shutdown()
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_todo.notify_all();
    for (auto& t : threads) t.join();
    threads.clear();
    if (fd >= 0) close(fd);
    fd = -1;
  }
};

struct kr_fastx_held { // the buffers of one batch, owned by the caller between kr_fastx_detach and kr_fastx_release
  std::vector<uint8_t> bases;
  std::vector<uint64_t> offsets;
  std::string name_blob;
  std::vector<size_t> name_off;
  std::vector<const char*> name_ptrs;
};

struct kr_fastx {
  gzFile f = nullptr;
  std::unique_ptr<BgzfSource> bgzf; // block-gzipped input: members inflated in parallel
  std::unique_ptr<pgz::Source> pgz; // ordinary gzip input: chunks of the one deflate stream inflated in parallel (kr_pgz.inc)
  bool bgzf_error = false, pgz_error = false;
  std::string path;
  std::unique_ptr<FqPool> pool; // plain files: chunks parsed in parallel while this is set
  uint64_t pool_chunks = 0;     // chunks accepted from the pool
  uint64_t bgzf_parsed = 0;     // (block-gzipped runs handed out with their records, once `bgzf` is gone)
  std::vector<unsigned char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  int last_char = 0;
  bool done = false;
  // current batch
  std::vector<uint8_t> bases;
  std::vector<uint64_t> offsets;
  std::string name_blob;
  std::vector<size_t> name_off;
  std::vector<const char*> name_ptrs;
  // batches taken over by the caller (kr_fastx_detach) come back here with their buffers (kr_fastx_release, any thread): the next
  // batch is built in them, so that a chunk's parse never starts in freshly mapped memory (see FqChunk)
  std::mutex held_mu;
  std::vector<kr_fastx_held*> recycled;

  bool records_mode() const { return (pgz && pgz->st.parsed_mode) || (bgzf && bgzf->st.parsed_mode); }

  // the block-gzipped part of the file has ended before the file has: what follows (an ordinary gzip member
  // appended with `cat`, say) is zlib's, from that offset on
  bool leave_bgzf()
  {
    const uint64_t off = bgzf->next_off;
    bgzf_parsed = bgzf->st.chunks_parsed;
    bgzf->shutdown();
    bgzf.reset();
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0 || lseek(fd, (off_t)off, SEEK_SET) < 0) {
      if (fd >= 0) close(fd);
      return false;
    }
    gzFile g = gzdopen(fd, "rb");
    if (!g) {
      close(fd);
      return false;
    }
    gzbuffer(g, 1 << 20);
    if (f) gzclose(f);
    f = g;
    return true;
  }

  // gzread has stopped: at the end of the data, or because the stream is damaged / cut short (kseq would end the input silently
  // there, src/kseq.h:92-101; a query file that lost its tail is reported instead, like a damaged block-gzipped one)
  void note_zlib_error()
  {
    if (pgz || bgzf || !f) return;
    int en = Z_OK;
    (void)gzerror(f, &en);
    if (en != Z_OK && en != Z_STREAM_END) pgz_error = true;
  }

  int getc()
  {
    if (pos >= end) {
      if (eof) return -1;
      int n = pgz ? pgz->read(buf.data(), (unsigned)buf.size())
                  : (bgzf ? bgzf->read(buf.data(), (unsigned)buf.size()) : gzread(f, buf.data(), (unsigned)buf.size()));
      if (pgz && n < 0) pgz_error = true;
      if (bgzf && n == -2) {
        if (leave_bgzf()) n = gzread(f, buf.data(), (unsigned)buf.size());
        else n = -1, bgzf_error = true;
      }
      if (bgzf && n < 0) bgzf_error = true;
      if (n <= 0) {
        note_zlib_error();
        eof = true;
        return -1;
      }
      pos = 0;
      end = (size_t)n;
    }
    return buf[pos++];
  }

  // Refill keeping the unread bytes: move them to the front, read behind them.
  void fill_keep()
  {
    if (eof) return;
    if (pos > 0) {
      memmove(buf.data(), buf.data() + pos, end - pos);
      end -= pos;
      pos = 0;
    }
    while (end < buf.size()) {
      int n = pgz ? pgz->read(buf.data() + end, (unsigned)(buf.size() - end))
                  : (bgzf ? bgzf->read(buf.data() + end, (unsigned)(buf.size() - end)) : gzread(f, buf.data() + end, (unsigned)(buf.size() - end)));
      if (pgz && n < 0) pgz_error = true;
      if (bgzf && n == -2) {
        if (leave_bgzf()) n = gzread(f, buf.data() + end, (unsigned)(buf.size() - end));
        else n = -1, bgzf_error = true;
      }
      if (bgzf && n < 0) bgzf_error = true;
      if (n <= 0) {
        note_zlib_error();
        eof = true;
        break;
      }
      end += (size_t)n;
    }
  }

  // The common case in bulk: a four-line FASTQ record that lies completely in the buffer, with a clean
  // sequence line (graphic characters only, none of the record markers) and a quality line of the same
  // length.  Returns false WITHOUT consuming anything when the record is anything else (FASTA, wrapped lines,
  // '\r', odd quality, end of input ...): next_record then parses it character by character with kseq's rules.
  bool fast_fastq(std::string& name, std::vector<uint8_t>& seq_out, long& slen_out)
  {
    for (int attempt = 0; attempt < 2; ++attempt) {
      size_t nb, ne, s0, slen, next;
      const Clean c = clean_record(buf.data(), pos, end, last_char, nb, ne, s0, slen, next);
      if (c == Clean::No) return false;
      if (c == Clean::Ok) {
        name.assign((const char*)buf.data() + nb, ne - nb);
        seq_out.insert(seq_out.end(), buf.data() + s0, buf.data() + s0 + slen);
        pos = next;
        last_char = 0;
        slen_out = (long)slen;
        return true;
      }
      if (attempt == 1 || eof || (pos == 0 && end == buf.size())) return false;
      fill_keep();
    }
    return false;
  }

  // returns sequence length, or <0 at end of input / truncated record
  long next_record(std::string& name, std::vector<uint8_t>& seq_out)
  {
    long fl = 0;
    if (fast_fastq(name, seq_out, fl)) return fl;
    int c;
    if (last_char == 0) {
      while ((c = getc()) != -1 && c != '>' && c != '@') {
      }
      if (c == -1) return -1;
      last_char = c;
    }
    name.clear();
    // name: up to first whitespace; a stream that ends right after the marker is EOF
    c = getc();
    if (c == -1) return -1;
    while (c != -1 && !isspace(c)) {
      name.push_back((char)c);
      c = getc();
    }
    if (c != -1 && c != '\n') {
      while ((c = getc()) != -1 && c != '\n') {
      }
    }
    size_t l0 = seq_out.size();
    while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
      if (isgraph(c)) seq_out.push_back((uint8_t)c);
    }
    if (c == '>' || c == '@') last_char = c;
    long slen = (long)(seq_out.size() - l0);
    if (c != '+') return slen; // FASTA (or last record)
    while ((c = getc()) != -1 && c != '\n') {
    }
    if (c == -1) {
      seq_out.resize(l0);
      return -2;
    }
    long ql = 0;
    while ((c = getc()) != -1 && ql < slen)
      if (c >= 33 && c <= 127) ql++;
    last_char = 0;
    if (ql != slen) {
      seq_out.resize(l0);
      return -2;
    }
    return slen;
  }
};

extern "C" {

int kr_fastx_open(const char* path, kr_fastx** out)
{
  kr::clear_error();
  if (!path || !out) return kr::fail(KR_ERR_ARG, "kr_fastx_open: null argument");
  gzFile f = gzopen(path, "rb");
  if (!f) return kr::fail(KR_ERR_IO, std::string("Failed to open the file at ") + path); // src/rqseq.cpp:174-176
  gzbuffer(f, 1 << 20);
  kr_fastx* r = new kr_fastx();
  r->f = f;
  r->path = path;
  r->buf.resize(4 << 20);
  // plain regular files of some size: parse in parallel (KR_FASTX_THREADS=0 turns it off; KR_FASTX_PAR_MIN = bytes)
  {
    const char* et = getenv("KR_FASTX_THREADS");
    const char* em = getenv("KR_FASTX_PAR_MIN");
    unsigned nt = et ? (unsigned)atoi(et) : std::min(12u, std::max(1u, kr::usable_cpus() * 3 / 4));
    const uint64_t par_min = em ? strtoull(em, nullptr, 10) : (32ull << 20);
    struct stat sb;
    const bool regular = nt && stat(path, &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size >= par_min; // never a pipe
    int fd = regular ? open(path, O_RDONLY) : -1;
    unsigned char magic[2] = {0, 0};
    if (fd >= 0 && BgzfSource::is_bgzf(fd)) {
      r->bgzf.reset(new BgzfSource());
      r->bgzf->fd = fd;
      r->bgzf->size = (uint64_t)sb.st_size;
      r->bgzf->depth = 2 * nt;
      for (unsigned t = 0; t < nt; ++t) r->bgzf->threads.emplace_back([p = r->bgzf.get()] { p->work(); });
    } else if (fd >= 0 && pgz::Source::eligible(fd, (uint64_t)sb.st_size) && !(getenv("KR_PGZ") && atoi(getenv("KR_PGZ")) == 0)) {
      // ordinary gzip: the deflate stream cut into chunks that are inflated in parallel (kr_pgz.inc; KR_PGZ=0: zlib's gzread alone)
      r->pgz.reset(new pgz::Source());
      r->pgz->fd = fd;
      r->pgz->size = (uint64_t)sb.st_size;
      r->pgz->depth = 2 * nt;
      if (const char* ec = getenv("KR_PGZ_CHUNK")) r->pgz->chunk_bytes = std::max<uint64_t>(65536, strtoull(ec, nullptr, 10));
      if (!r->pgz->open_map()) {
        r->pgz->shutdown(); // (closes fd)
        r->pgz.reset();
      } else {
        for (unsigned t = 0; t < nt; ++t) r->pgz->threads.emplace_back([p = r->pgz.get()] { p->work(); });
      }
    } else if (fd >= 0 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
      r->pool.reset(new FqPool());
      r->pool->fd = fd;
      r->pool->size = (uint64_t)sb.st_size;
      (void)r->pool->open_map();
      r->pool->depth = 2 * nt;
      for (unsigned t = 0; t < nt; ++t) r->pool->threads.emplace_back([p = r->pool.get()] { p->work(); });
    } else if (fd >= 0) {
      close(fd);
    }
  }
  *out = r;
  return KR_OK;
}

// QSeq::read_next_batch (src/rqseq.cpp:180-197): keep reading until the batch holds
// at least `min_bases` bases (reference: RBATCH_SIZE*DSEQ_LEN = 76,800) or input ends.
int kr_fastx_next(kr_fastx* r, uint64_t min_bases, kr_fastx_batch* out)
{
  if (!r || !out) return kr::fail(KR_ERR_ARG, "kr_fastx_next: null argument");
  if (r->bases.capacity() == 0) { // the last batch was detached: build this one in buffers that came back
    kr_fastx_held* h = nullptr;
    {
      std::lock_guard<std::mutex> lk(r->held_mu);
      if (!r->recycled.empty()) h = r->recycled.back(), r->recycled.pop_back();
    }
    if (h) {
      r->bases.swap(h->bases), r->offsets.swap(h->offsets), r->name_blob.swap(h->name_blob), r->name_off.swap(h->name_off), r->name_ptrs.swap(h->name_ptrs);
      delete h;
    }
  }
  r->bases.clear();
  r->offsets.assign(1, 0);
  r->name_blob.clear();
  r->name_off.clear();
  uint64_t bpc = 0;
  if (r->pool) { // one chunk of about 2 * min_bases bytes of input per batch (about min_bases bases of ordinary FASTQ), at most 96 MB:
                 // a larger batch is put together from several chunks (the first by swap, the others appended), so that a
                 // million-read batch is still parsed by the whole pool and not by one thread per batch
    FqPool& P = *r->pool;
    if (!P.chunk_bytes) {
      const char* cm = getenv("KR_FASTX_CHUNK_MAX"); // (tests: batches of many small chunks)
      P.chunk_bytes = std::min<uint64_t>(std::max<uint64_t>(4096, 2 * min_bases), cm ? std::max<uint64_t>(4096, strtoull(cm, nullptr, 10)) : (96ull << 20));
    }
    bool fall_back = false;
    uint64_t resume = 0;
    for (;;) {
      P.issue();
      resume = P.next_off; // where sequential parsing takes over if no chunk is left
      if (P.inflight.empty()) {
        fall_back = r->name_off.empty(); // (a batch in hand goes out first: the next call falls back)
        break;
      }
      std::unique_ptr<FqChunk> c = std::move(P.inflight.front());
      P.inflight.pop_front();
      {
        std::unique_lock<std::mutex> lk(P.mu);
        P.cv_ready.wait(lk, [&] { return c->ready; });
      }
      if (r->name_off.empty()) { // the batch's first chunk: its vectors become the batch's
        r->bases.swap(c->bases);
        r->offsets.swap(c->offsets);
        r->name_blob.swap(c->name_blob);
        r->name_off.swap(c->name_off);
      } else { // behind what the batch holds
        const uint64_t b0 = r->bases.size();
        const size_t n0 = r->name_blob.size();
        r->bases.insert(r->bases.end(), c->bases.begin(), c->bases.end());
        for (size_t i = 1; i < c->offsets.size(); ++i) r->offsets.push_back(b0 + c->offsets[i]);
        r->name_blob.append(c->name_blob);
        for (size_t o : c->name_off) r->name_off.push_back(n0 + o);
      }
      bpc = r->bases.size();
      if (!c->ok) {
        fall_back = true, resume = c->clean_end;
        break;
      }
      r->pool_chunks++;
      const bool ramp = c->ramp;
      if (P.spare.size() < 2 * P.depth) P.spare.push_back(std::move(c)); // (with the batch's previous vectors: their capacity serves the next chunk)
      P.issue();
      // One chunk is a batch: a chunk of 2 * min_bases bytes of ordinary FASTQ (2 L + 6 + name bytes per record of L bases) holds about
      // 0.94 * min_bases bases, and asking for all of min_bases took a second chunk for every steady batch -- appended by copy, and
      // the batch of 1.9 * min_bases then too large for the CLI to hand over whole (round-5 advice).  (A short first chunk goes
      // out as the short batch it is: the caller's workers start on it.)
      if (bpc >= min_bases - min_bases / 8 || ramp) break;
    }
    if (fall_back && resume >= P.size && P.size != 0) {
      // every byte of the file went through the pool: the input is exhausted.  The pool (its threads, and gigabytes of recycled
      // chunk buffers) is torn down by kr_fastx_close, not here: unmapping them took 0.1 s between the last batch and the end of
      // input -- on the caller's critical path
      r->done = true;
    } else if (fall_back) { // the rest of the input goes through the sequential reader
      P.shutdown();
      r->pool.reset();
      if (gzseek(r->f, (z_off_t)resume, SEEK_SET) < 0) return kr::fail(KR_ERR_IO, "kr_fastx_next: seek failed in " + r->path);
      r->pos = r->end = 0, r->eof = false, r->last_char = 0;
    }
    if (!r->name_off.empty()) bpc = std::max<uint64_t>(bpc, min_bases); // a batch is ready: hand it over as it is
  }
  if (r->records_mode()) { // gzip input, clean four-line FASTQ so far: the chunks come with their records parsed by the pool
    pgz::Records rec;
    rec.bases.swap(r->bases), rec.offsets.swap(r->offsets), rec.name_blob.swap(r->name_blob), rec.name_off.swap(r->name_off);
    while (bpc < min_bases) {
      const int st = r->pgz ? r->pgz->next_records(rec) : r->bgzf->next_records(rec);
      if (st < 0) (r->pgz ? r->pgz_error : r->bgzf_error) = true, r->done = true;
      if (st <= 0) break; // 0: the sequential parser continues below, on the bytes that follow
      bpc = rec.bases.size();
    }
    rec.bases.swap(r->bases), rec.offsets.swap(r->offsets), rec.name_blob.swap(r->name_blob), rec.name_off.swap(r->name_off);
  }
  std::string name;
  bool cont = false;
  while (!r->pool && !r->records_mode() && !r->done && bpc < min_bases) {
    long l = r->next_record(name, r->bases);
    cont = l >= 0;
    if (!cont) {
      r->done = true;
      break;
    }
    bpc += (uint64_t)l;
    r->offsets.push_back(r->bases.size());
    r->name_off.push_back(r->name_blob.size());
    r->name_blob += name;
    r->name_blob.push_back('\0');
  }
  if (r->bgzf_error) return kr::fail(KR_ERR_IO, "damaged block-gzipped file (a member does not inflate or fails its CRC): " + r->path);
  if (r->pgz_error && r->pgz && r->pgz->too_dense.load())
    return kr::fail(KR_ERR_IO, "a 4 MB piece of this gzip file inflates to more than 1 GB: more than the parallel reader keeps in memory; "
                                "run with KR_PGZ=0 (zlib's streaming reader): " + r->path);
  if (r->pgz_error) return kr::fail(KR_ERR_IO, "damaged gzip file (the stream does not inflate, or a member fails its CRC / length check): " + r->path);
  r->name_ptrs.resize(r->name_off.size());
  for (size_t i = 0; i < r->name_off.size(); ++i) r->name_ptrs[i] = r->name_blob.c_str() + r->name_off[i];
  out->bases = r->bases.data();
  out->offsets = r->offsets.data();
  out->names = r->name_ptrs.data();
  out->nreads = (uint32_t)r->name_off.size();
  out->more = r->done ? 0 : 1;
  return KR_OK;
}

// The batch kr_fastx_next just returned changes hands: its buffers leave the reader (the pointers of the kr_fastx_batch stay valid,
// nothing is copied) until kr_fastx_release gives them back -- from any thread, before kr_fastx_close.  What QSeq hands to IBatch by
// swap (src/query.cpp:32-33), for a consumer that keeps several batches in flight.
int kr_fastx_detach(kr_fastx* r, kr_fastx_held** out)
{
  if (!r || !out) return kr::fail(KR_ERR_ARG, "kr_fastx_detach: null argument");
  kr_fastx_held* h = new (std::nothrow) kr_fastx_held();
  if (!h) return kr::fail(KR_ERR_NOMEM, "kr_fastx_detach: out of memory");
  h->bases = std::move(r->bases), h->offsets = std::move(r->offsets), h->name_blob = std::move(r->name_blob);
  h->name_off = std::move(r->name_off), h->name_ptrs = std::move(r->name_ptrs);
  r->bases = std::vector<uint8_t>(), r->offsets = std::vector<uint64_t>(), r->name_blob = std::string(); // (moved-from: make the emptiness definite)
  r->name_off = std::vector<size_t>(), r->name_ptrs = std::vector<const char*>();
  // (a short blob lives inside the string object and moved with it: the pointer array -- same address as before -- is refreshed)
  for (size_t i = 0; i < h->name_off.size() && i < h->name_ptrs.size(); ++i) h->name_ptrs[i] = h->name_blob.c_str() + h->name_off[i];
  *out = h;
  return KR_OK;
}
void kr_fastx_release(kr_fastx* r, kr_fastx_held* h)
{
  if (!h) return;
  if (r) {
    std::lock_guard<std::mutex> lk(r->held_mu);
    if (r->recycled.size() < 16) {
      r->recycled.push_back(h);
      return;
    }
  }
  delete h;
}

uint64_t kr_fastx_parallel_chunks(const kr_fastx* r) { return r ? (r->pgz ? r->pgz->chunks_used : r->pool_chunks) : 0; }
// (tests) ordinary gzip input: chunks whose speculative start was used / thrown away, gaps inflated sequentially
void kr_fastx_pgz_stats(const kr_fastx* r, uint64_t* used, uint64_t* discarded, uint64_t* gaps)
{
  if (used) *used = !r ? 0 : r->pgz ? r->pgz->st.chunks_parsed : r->bgzf ? r->bgzf->st.chunks_parsed : r->bgzf_parsed;
  if (discarded) *discarded = r && r->pgz ? r->pgz->chunks_discarded : 0;
  if (gaps) *gaps = r && r->pgz ? r->pgz->gaps : 0;
}

void kr_fastx_close(kr_fastx* r)
{
  if (!r) return;
  for (kr_fastx_held* h : r->recycled) delete h;
  r->recycled.clear();
  if (r->pool) r->pool->shutdown();
  if (r->bgzf) r->bgzf->shutdown();
  if (r->pgz) r->pgz->shutdown();
  if (r->f) gzclose(r->f);
  delete r;
}

// report_distances (src/query.cpp:158-196), non-summarize branch.  The selection
// (multi / filter / dist-max / closest) was already made on the device: rec_sel.
using kr::fmt_fixed5;

int kr_format_dist(const kr_host_index* h, const kr_result_view* rv, const char* const* names, char** text,
                   uint64_t* len)
{
  if (!h || !rv || !text || !len) return kr::fail(KR_ERR_ARG, "kr_format_dist: null argument");
  // reads are cut into contiguous ranges, one string per range, joined in order
  const int nt = std::max(1, std::min(std::min(kr::parallel_width(), 16), (int)(rv->nreads / 4096)));
  std::vector<std::string> part((size_t)nt);
  kr::parallel_for(nt, [&](int t) {
    const uint32_t r0 = (uint32_t)((uint64_t)rv->nreads * t / nt), r1 = (uint32_t)((uint64_t)rv->nreads * (t + 1) / nt);
    std::string& s = part[(size_t)t];
    s.reserve((size_t)(r1 - r0) * 64);
    char num[64];
    for (uint32_t r = r0; r < r1; ++r) {
      const char* id = names ? names[r] : "";
      const size_t idl = strlen(id);
      const uint32_t o = rv->read_off[r], n = rv->read_cnt[r];
      for (uint32_t i = o; i < o + n; ++i) {
        if (!rv->rec_sel[i]) continue;
        const char* nm = kr_host_index_node_name(h, rv->rec_key[i] >> 1);
        const size_t nl = fmt_fixed5(rv->rec_dix ? rv->dist_list[rv->rec_dix[i]] : rv->rec_d[i], num); // (KR_ROWS_INDEXED: DIST through the batch's list)
        s.append(id, idl);
        s += '\t';
        s += nm;
        s += '\t';
        s.append(num, nl);
        s += '\n';
      }
      if (rv->read_na[r]) { // src/query.cpp:173-176
        s.append(id, idl);
        s += "\tNA\tNaN\n";
      }
    }
  });
  size_t total = 0;
  for (auto& s : part) total += s.size();
  char* p = (char*)malloc(total + 1);
  if (!p) return kr::fail(KR_ERR_NOMEM, "kr_format_dist: out of memory");
  size_t at = 0;
  for (auto& s : part) {
    memcpy(p + at, s.data(), s.size());
    at += s.size();
  }
  p[total] = 0;
  *text = p;
  *len = total;
  return KR_OK;
}

// SBatch::seek_sequences (src/seek.cpp:22-56).  `rv` is a batch of the `dist` path on a sketch index
// (kr_host_sketch_load), stream parameters multi = 1, no_filter = 1, no dist-max: one record per strand that
// matched.  A read with any match reports the smaller of the two strands' distances -- `d_or < d_rc ? d_or :
// d_rc` -- where a strand WITHOUT matches is still optimised, on an all-zero histogram with mismatch_count =
// onmers: those minimisations (one per distinct onmers of the batch) run on the GPU through kr_llh_batch.
int kr_format_seek(const kr_host_index* h, const kr_index* dix, const kr_result_view* rv, uint32_t hdist_th,
                   const char* const* names, char** text, uint64_t* len)
{
  if (!h || !dix || !rv || !text || !len) return kr::fail(KR_ERR_ARG, "kr_format_seek: null argument");
  if (h->libs.size() != 1 || h->libs[0].rho.size() != 2) return kr::fail(KR_ERR_ARG, "kr_format_seek: not a sketch index");
  const double rho = h->libs[0].rho[1];
  std::map<uint32_t, double> dzero; // onmers -> d of the empty strand
  for (uint32_t r = 0; r < rv->nreads; ++r) {
    const uint32_t o = rv->read_off[r], n = rv->read_cnt[r];
    uint32_t nrec = 0;
    for (uint32_t i = o; i < o + n; ++i) nrec += rv->rec_key[i] != 0;
    if (nrec == 1) dzero[rv->read_onmers[r]] = 0;
  }
  if (!dzero.empty()) {
    const size_t nz = dzero.size(), np = (size_t)hdist_th + 1;
    std::vector<double> hist(nz * np, 0.0), uc(nz), rh(nz, rho), d(nz), v(nz);
    size_t j = 0;
    for (auto& kv : dzero) uc[j++] = (double)kv.first;
    int rc = kr_llh_batch(dix, hdist_th, 0, nz, hist.data(), uc.data(), rh.data(), nullptr, d.data(), v.data());
    if (rc) return rc;
    j = 0;
    for (auto& kv : dzero) kv.second = d[j++];
  }
  std::string s;
  s.reserve((size_t)rv->nreads * 24);
  char num[64];
  for (uint32_t r = 0; r < rv->nreads; ++r) {
    const char* id = names ? names[r] : "";
    const uint32_t o = rv->read_off[r], n = rv->read_cnt[r];
    bool have[2] = {false, false};
    double ds[2] = {0, 0};
    for (uint32_t i = o; i < o + n; ++i)
      if (rv->rec_key[i]) have[rv->rec_key[i] & 1u] = true, ds[rv->rec_key[i] & 1u] = rv->rec_d[i];
    s += id;
    if (!have[0] && !have[1]) {
      s += "\tNaN\n";
      continue;
    }
    for (int q = 0; q < 2; ++q)
      if (!have[q]) ds[q] = dzero[rv->read_onmers[r]];
    const double dmin = ds[0] < ds[1] ? ds[0] : ds[1];
    const size_t nl = fmt_fixed5(dmin, num);
    s += '\t';
    s.append(num, nl);
    s += '\n';
  }
  char* p = (char*)malloc(s.size() + 1);
  if (!p) return kr::fail(KR_ERR_NOMEM, "kr_format_seek: out of memory");
  memcpy(p, s.c_str(), s.size() + 1);
  *text = p;
  *len = s.size();
  return KR_OK;
}

// tests: "%.5f" through the rounding the device formatter uses (fixed5_exact + fixed5_digits, kr_common.h); returns the length, 0 if
// the value is outside the exact range
uint32_t kr_debug_fixed5(double v, char* out)
{
  uint32_t n = 0;
  if (!kr::fixed5_exact(v, &n)) return 0;
  const uint32_t l = kr::fixed5_digits(n, out);
  out[l] = 0;
  return l;
}

void kr_free(void* p) { kr::big_free(p); }

const char* kr_last_error(void) { return kr::g_err.c_str(); }
const char* kr_version(void) { return "krepp-amd 0.1.0 (mirrors krepp v0.8.3 dist)"; }

} // extern "C"

// ---- kr::big_alloc / big_free (kr_common.h)
namespace {
std::mutex g_big_mu;
std::unordered_map<void*, size_t> g_big_live;            // blocks handed out by big_alloc: their capacities
std::vector<std::pair<void*, size_t>> g_big_cache;        // blocks given back, kept for the next big_alloc
size_t g_big_cached = 0;
constexpr size_t kBigMin = 1u << 20, kBigCacheBytes = 3ull << 30, kBigCacheBlocks = 64;
} // namespace
void* kr::big_alloc(size_t n)
{
  if (n < kBigMin) return malloc(n);
  {
    std::lock_guard<std::mutex> lk(g_big_mu);
    size_t best = g_big_cache.size();
    for (size_t i = 0; i < g_big_cache.size(); ++i)
      if (g_big_cache[i].second >= n && g_big_cache[i].second <= 2 * n + (8u << 20) && (best == g_big_cache.size() || g_big_cache[i].second < g_big_cache[best].second)) best = i;
    if (best != g_big_cache.size()) {
      const auto blk = g_big_cache[best];
      g_big_cache[best] = g_big_cache.back();
      g_big_cache.pop_back();
      g_big_cached -= blk.second;
      g_big_live[blk.first] = blk.second;
      return blk.first;
    }
  }
  const size_t cap = n + n / 8 + 4096;
  void* p = malloc(cap);
  if (!p) return nullptr;
  std::lock_guard<std::mutex> lk(g_big_mu);
  g_big_live[p] = cap;
  return p;
}
void kr::big_free(void* p)
{
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_big_mu);
    auto it = g_big_live.find(p);
    if (it != g_big_live.end()) {
      const size_t cap = it->second;
      g_big_live.erase(it);
      if (g_big_cached + cap <= kBigCacheBytes && g_big_cache.size() < kBigCacheBlocks) {
        g_big_cache.emplace_back(p, cap);
        g_big_cached += cap;
        return;
      }
    }
  }
  free(p);
}
