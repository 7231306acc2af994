// krepp_main.cpp — the `krepp` command line for the sub-commands this repository
// implements: `dist` (MI355X) and `index` (CPU).  Option names, defaults, header lines
// and error conventions follow the reference CLI (src/krepp.cpp:508-712,
// src/krepp.hpp:206-221); everything else of the reference CLI (place, seek, sketch,
// inspect) is out of scope here and reported as such.
//
// `dist` pipeline: reader thread (gz/FASTX, src/rqseq.cpp:180-197) -> one worker per GPU,
// each with its own replica of the index and its own kr_stream -> writer emitting batches
// in INPUT order (the reference emits in task-completion order, src/krepp.cpp:368-383).
#include "krepp_amd.h"

#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <malloc.h>
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#define KREPP_VERSION "v0.8.3"

[[noreturn]] static void error_exit(const std::string& msg)
{ // src/common.cpp:20-24
  fprintf(stderr, "[ERROR] %s\n", msg.c_str());
  exit(EXIT_FAILURE);
}

struct Args {
  std::string sub;
  std::map<std::string, std::string> opt;
  std::map<std::string, bool> flag;
  bool has(const std::string& k) const { return opt.count(k) != 0; }
  std::string get(const std::string& k, const std::string& d = "") const
  {
    auto it = opt.find(k);
    return it == opt.end() ? d : it->second;
  }
};

static const std::map<std::string, std::string> kAlias = {
  {"-i", "--index-dir"}, {"-q", "--query"}, {"-o", "--output-path"}, {"-t", "--nwk-file"}, {"-k", "--kmer-len"},
  {"-w", "--win-len"}, {"-h", "--num-positions"}, {"-m", "--modulo-lsh"}, {"-r", "--residue-lsh"}, {"-l", "--lineage-file"},
};
static const char* kFlags[] = {"--multi", "--filter", "--summarize", "--frac", "--verbose", "--gpu-minimizers", "--tabular"};

// options as in the reference's CLI (src/krepp.cpp:516-675); the texts are this build's own
static void print_help(const std::string& sub)
{
  const char* query_opts =
    "  -q, --query PATH           query FASTA/FASTQ file (gzip accepted)\n"
    "  -o, --output-path PATH     write the report here [stdout]\n"
    "      --hdist-th N           largest Hamming distance at which a k-mer matches [4]\n";
  const char* index_query_opts =
    "  -i, --index-dir DIR        index directory (as written by `krepp index`, this build's or the reference's)\n"
    "      --chisq X              chi-square threshold of the likelihood-ratio filter [2.706]\n"
    "      --dist-max X           report only distances below X (1e-8 .. 0.33)\n"
    "      --multi / --no-multi   report every reference / edge that passes, or only the best [multi]\n"
    "      --summarize            weighted read counts per reference / edge instead of per-read rows\n"
    "      --gpus N, --device D   GPUs to use (index replicated on each), first device [1, 0]\n";
  const char* build_opts =
    "  -k, --kmer-len N           k-mer length, 19..31\n"
    "  -w, --win-len N            minimizer window (>= k) [k+6]\n"
    "  -h, --num-positions N      LSH positions [k-16]\n"
    "  -m, --modulo-lsh N         modulo partitioning the LSH space [4]\n"
    "  -r, --residue-lsh N        keep k-mers with LSH(x) mod m == r [1]\n"
    "      --frac / --no-frac     keep k-mers with LSH(x) mod m <= r [frac]\n"
    "      --seed N               seed of the random LSH positions\n";
  printf("krepp (MI355X build, mirrors krepp v0.8.3): dist | place | seek | index | sketch\n");
  if (sub.empty() || sub == "dist")
    printf("\nkrepp dist -i DIR -q READS: distances of every read to the references it matches\n%s%s"
           "      --filter / --no-filter keep only references not significantly worse than the closest [no-filter]\n",
           query_opts, index_query_opts);
  if (sub.empty() || sub == "place")
    printf("\nkrepp place -i DIR -q READS: placements on the backbone tree (jplace)\n%s%s"
           "  -t, --nwk-file PATH        rooted placement tree (overrides the index's backbone)\n"
           "  -l, --lineage-file PATH    place on the taxonomy of a Greengenes/GTDB style lineage file\n"
           "      --tau N                highest Hamming distance counted by the placement threshold [2]\n"
           "      --filter / --no-filter [filter]\n"
           "      --tabular              tab-separated rows instead of jplace\n",
           query_opts, index_query_opts);
  if (sub.empty() || sub == "seek")
    printf("\nkrepp seek -i SKETCH -q READS: distance of every read to the one sketched reference\n"
           "  -i, --sketch-path PATH     sketch file (as written by `krepp sketch`)\n%s", query_opts);
  if (sub.empty() || sub == "index")
    printf("\nkrepp index -i MAP.tsv -o DIR: build an index (reference ID <tab> FASTA path per line)\n"
           "  -t, --nwk-file PATH        rooted guide tree (default: a balanced tree over the IDs)\n%s"
           "      --num-threads N        CPU threads of the builder [1]\n"
           "      --gpu-minimizers       window minimizers of the genomes on the GPU (identical output)\n",
           build_opts);
  if (sub.empty() || sub == "sketch")
    printf("\nkrepp sketch -i GENOME.fa -o SKETCH: sketch of a single FASTA/FASTQ file (default k 26)\n%s", build_opts);
  printf("\nEnvironment knobs and what differs from the reference: INTEGRATION.md\n");
}

static Args parse(int argc, char** argv)
{
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string t = argv[i];
    if (t == "--help" || (t == "-h" && a.sub.empty())) {
      print_help(a.sub);
      exit(0);
    }
    if (t[0] != '-') {
      if (a.sub.empty()) {
        a.sub = t;
        continue;
      }
      error_exit("Unexpected argument: " + t);
    }
    auto al = kAlias.find(t);
    if (al != kAlias.end()) t = al->second;
    bool is_flag = false;
    for (const char* f : kFlags) {
      std::string pos = f, neg = std::string("--no-") + (f + 2);
      if (t == pos || t == neg) {
        a.flag[pos] = (t == pos);
        is_flag = true;
      }
    }
    if (is_flag) continue;
    if (i + 1 >= argc) error_exit("Option " + t + " requires a value");
    a.opt[t] = argv[++i];
  }
  return a;
}

struct Job {
  uint64_t seq = 0;
  // the batch as views: into a batch the reader handed over whole (kr_fastx_detach: nothing copied, given back by the worker
  // when the batch is done) or into the copies below (a reader batch cut into several jobs)
  const uint8_t* bases = nullptr;
  const uint64_t* offsets = nullptr;  // [n + 1], offsets[0] is the job's first base within `bases`
  const char* const* names = nullptr; // [n] NUL-terminated, back to back in one buffer, in order
  size_t n = 0;
  const char* blob = nullptr;         // = names[0]: the names' buffer
  size_t blob_bytes = 0;
  kr_fastx_held* held = nullptr;
  std::vector<uint8_t> own_bases;
  std::vector<uint64_t> own_offsets;
  std::string own_blob;
  std::vector<const char*> own_names;
  std::string text;
  std::vector<kr_placement> pls; // place --summarize: the placements, `read` = global read number
  uint64_t first_read = 0;
  bool done = false;
};

// `dist` and `place` share everything up to the per-batch back end (src/krepp.cpp:347-394, 434-504)
// mode 0 `dist`, 1 `place`, 2 `seek` (a sketch file served as a one-leaf index: src/krepp.cpp:321-345, src/seek.cpp)
static int run_query(const Args& a, const std::string& invocation, int mode)
{
  const bool place = mode == 1, seek = mode == 2;
  const std::string index_arg = a.has("--index-dir") ? a.get("--index-dir") : a.get("--sketch-path");
  if (!a.has("--query") || index_arg.empty()) error_exit("dist/place/seek require -q/--query and -i (index directory or sketch file)");
  if (a.has("--lineage-file") && !place) error_exit("-l/--lineage-file is an option of `place`");
  const bool summarize = !seek && a.flag.count("--summarize") && a.flag.at("--summarize");
  kr_params p;
  kr_params_default(&p);
  if (a.has("--hdist-th")) p.hdist_th = (uint32_t)atoi(a.get("--hdist-th").c_str());
  if (a.has("--chisq")) p.chisq = atof(a.get("--chisq").c_str());
  if (a.has("--dist-max")) {
    p.dist_max = atof(a.get("--dist-max").c_str());
    if (!(p.dist_max >= 1e-8 && p.dist_max <= 0.33)) error_exit("--dist-max: value not in range [1e-08, 0.33]");
  }
  if (a.flag.count("--multi")) p.multi = a.flag.at("--multi");
  if (place) p.no_filter = 0; // filter defaults to on for place (src/krepp.cpp:612-615)
  if (summarize && !place) p.no_filter = 0, p.multi = 1; // "Overrides --no-multi and --no-filter" (src/krepp.cpp:669-672, src/query.cpp:160-171)
  if (a.flag.count("--filter")) p.no_filter = !a.flag.at("--filter");
  if (a.has("--tau")) p.tau = (uint32_t)atoi(a.get("--tau").c_str());
  if (place && p.hdist_th < p.tau) error_exit("The threshold tau must be less than HD threshold --hdist-th!");
  const bool tabular_flag = a.flag.count("--tabular") && a.flag.at("--tabular");
  // place: 0 jplace, 1 tabular, 2 summary (--summarize wins over --tabular: src/krepp.cpp:401-405,441,467,493)
  const int tabular = (place && summarize) ? 2 : (tabular_flag ? 1 : 0);
  int ngpus = a.has("--gpus") ? atoi(a.get("--gpus").c_str()) : 1;
  int dev0 = a.has("--device") ? atoi(a.get("--device").c_str()) : 0;
  if (ngpus < 1) ngpus = 1;

  FILE* out = stdout;
  if (a.has("--output-path")) {
    out = fopen(a.get("--output-path").c_str(), "w");
    if (!out) error_exit("Failed to open " + a.get("--output-path"));
  }
  fprintf(stderr, "Loading the index and initializing...\n");
  kr_host_index* hx = nullptr;
  if (seek ? kr_host_sketch_load(index_arg.c_str(), &hx) : kr_host_index_load(index_arg.c_str(), &hx)) error_exit(kr_last_error());
  kr_index_view view;
  kr_host_index_view(hx, &view);
  kr_place_tree* ptree = nullptr;
  if (place) {
    auto slurp = [&](const std::string& path) {
      std::string text;
      FILE* tf = fopen(path.c_str(), "rb");
      if (!tf) error_exit("Error opening " + path);
      char buf[65536];
      size_t n;
      while ((n = fread(buf, 1, sizeof(buf), tf)) > 0) text.append(buf, n);
      fclose(tf);
      return text;
    };
    if (a.has("--lineage-file")) { // src/krepp.cpp:742-744: -l takes precedence over -t and the backbone
      if (kr_place_tree_create_lineage(hx, slurp(a.get("--lineage-file")).c_str(), &ptree)) error_exit(kr_last_error());
      fprintf(stderr, "Placing given sequences on the taxonomic lineage...\n");
    } else {
      std::string nwk_text;
      if (a.has("--nwk-file")) nwk_text = slurp(a.get("--nwk-file"));
      if (kr_place_tree_create(hx, a.has("--nwk-file") ? nwk_text.c_str() : nullptr, &ptree)) error_exit(kr_last_error());
      fprintf(stderr, "Placing given sequences on the backbone tree...\n");
    }
    view.node_kind = kr_place_tree_kinds(ptree); // leaves absent from the placement tree become null nodes
  }
  // the index crosses PCIe once; the other GPUs receive it by RCCL broadcast over xGMI (load time only)
  std::vector<kr_index*> dix(ngpus, nullptr);
  if (kr_index_upload(&view, dev0, KR_VIEW_HOST, &dix[0])) error_exit(kr_last_error());
  if (ngpus > 1) {
    std::vector<int> devs;
    for (int g = 1; g < ngpus; ++g) devs.push_back(dev0 + g);
    // RCCL may print a version banner on stdout when its first communicator is made; stdout is where the report rows go
    // (src/krepp.cpp:372-382).  This process is still single-threaded here (the workers start below), so pointing file
    // descriptor 1 at stderr for the duration of the call is safe -- the application's decision, not the library's.
    fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) (void)dup2(2, 1);
    const int brc = kr_index_broadcast(dix[0], ngpus - 1, devs.data(), dix.data() + 1);
    if (saved_out >= 0) {
      fflush(stdout);
      (void)dup2(saved_out, 1);
      close(saved_out);
    }
    if (brc) error_exit(kr_last_error());
  }
  // "Loading the index and initializing..." (src/krepp.cpp:756-757) ends HERE, when the index is in device memory.  The elapsed
  // time reported at the end is everything after it -- the workers' streams and page-locked buffers, then the batch loop
  // (estimate_distances(), src/krepp.cpp:347-394,759-762: the reference's clock covers all it does after the index load, so this
  // one does too since round 6; until then it began when the streams were ready) -- taken when the last row has been written and
  // before anything is torn down, as the reference takes it before its destructors run.  The reader opens the query file and
  // parses the first batches while the streams come up.
  auto t_init = std::chrono::steady_clock::now();
  const auto t0 = t_init;
  auto t_loop = t_init; // when the workers were ready ([timing] only)
  if (seek) { // QuerySketch::header_dreport (src/krepp.cpp:305-309)
    fprintf(out, "# software: krepp\tversion: " KREPP_VERSION "\tinvocation :%s\nSEQ_ID\tDIST\n", invocation.c_str());
  } else if (!place) { // header (src/krepp.cpp:311-319)
    fprintf(out, "# software: krepp\tversion: " KREPP_VERSION "\tinvocation :%s\n%s\n", invocation.c_str(),
            summarize ? "REFERENCE_NAME\tWEIGHTED_COUNT\tSEQUENCE_ABUNDANCE" : "SEQ_ID\tREFERENCE_NAME\tDIST");
  } else { // jplace opening or tabular header (src/krepp.cpp:440-447)
    char* t = nullptr;
    uint64_t l = 0;
    if (kr_place_frame(ptree, 0, tabular, invocation.c_str(), 0, &t, &l)) error_exit(kr_last_error());
    fwrite(t, 1, l, out);
    kr_free(t);
  }
  // the device front end of `place` keeps every chosen leaf: the place options act in the back end
  kr_params pfront;
  kr_params_default(&pfront);
  pfront.hdist_th = p.hdist_th;

  // reads per batch: the kernels' efficiency grows with the batch (launch tails amortise, likelihood problems repeat): 262,144 reads
  // for `dist` on a large input, 65,536 for `place` / `seek` (more state per read) and for inputs of a few batches anyway
  uint32_t max_reads_default = 1u << 16;
  if (!place && !seek) {
    struct stat sb;
    if (stat(a.get("--query").c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size >= (64ull << 20)) max_reads_default = 1u << 18;
  }
  const uint32_t max_reads = getenv("KR_CLI_BATCH_READS") ? (uint32_t)std::max(1, atoi(getenv("KR_CLI_BATCH_READS"))) : max_reads_default;
  const uint64_t batch_bases = (uint64_t)max_reads * 150, max_bases = batch_bases * 4;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<Job*> todo;
  std::map<uint64_t, Job*> finished;
  bool eof = false;
  uint64_t total_batches = 0;
  std::string worker_err;
  // KR_CLI_TIMING=1: seconds spent parsing, on the device (submit + collect), formatting and writing
  const bool timing = getenv("KR_CLI_TIMING") != nullptr;
  std::atomic<uint64_t> ns_parse{0}, ns_job{0}, ns_dev{0}, ns_fmt{0}, ns_write{0};
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point t) {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count();
  };
  // --summarize: reference -> weighted read count (src/krepp.cpp:374-378), by colour id; every worker sums its batches in an array
  // of its own (264 M rows per 8 M reads on a 1000-genome index: a shared map under a mutex was the whole run) and adds it here
  // when it ends -- the reference adds per batch under `omp critical`, in whatever order the tasks finish
  std::vector<double> wcount(summarize && !place ? (size_t)view.tree_nnodes + 2 : 0, 0.0);
  double twcount = 0;
  // Plain `dist` rows are formatted on the device (kr_batch_submit_text / kr_batch_collect_text: the text of a batch arrives as
  // bytes in the stream's page-locked buffer) and written by the WORKER itself, straight from that buffer, when the batch's turn
  // has come: `next_seq` is the batch the output is waiting for (advanced by the writer thread, which still orders the batches).
  // KR_CLI_HOST_TEXT=1: the host formatter (kr_format_dist) as before round 5.
  const bool dev_text = !place && !seek && !summarize && !getenv("KR_CLI_HOST_TEXT");
  uint64_t next_seq = 0;
  // A regular output file: a batch's place in it is known as soon as the batches before it know their lengths, so the workers
  // pwrite() their texts side by side (one thread copies ~10 GB/s into the page cache: 13 M reads/s of 33 rows each); anything
  // else (a pipe, a terminal) is written in turn.  KR_CLI_SERIAL_WRITE=1: in turn always.
  bool out_seekable = false;
  off_t file_off = 0;
  if (dev_text && out != stdout && !getenv("KR_CLI_SERIAL_WRITE")) {
    struct stat sb;
    fflush(out);
    if (fstat(fileno(out), &sb) == 0 && S_ISREG(sb.st_mode) && (file_off = ftello(out)) >= 0) out_seekable = true;
  }

  // KR_CLI_TIMING: when things happened, in seconds since the query phase began
  auto at = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  std::atomic<int> worker_ids{0};
  int workers_ready = 0;
  std::vector<kr_stream*> streams_to_free;
  kr_fastx* fx = nullptr; // (opened before the workers start; they hand batches back to it)
  auto worker = [&](int g) {
    const int wid = worker_ids++;
    double t_ready = 0, t_first = -1, t_last = 0;
    uint64_t njobs = 0;
    kr_stream* st = nullptr;
    uint64_t max_records = (uint64_t)max_reads * (place ? 128 : 64);
    if (const char* e = getenv("KR_DEBUG_CLI_RECORDS")) max_records = strtoull(e, nullptr, 10); // tests: force the split-and-retry path
    // (four times the reads of a batch: a batch of contigs is submitted as tiles of 128 k-mer positions, include/krepp_amd.h)
    if (kr_stream_create(dix[g], (place || seek) ? &pfront : &p, 4 * max_reads, max_bases, max_records, &st)) {
      std::lock_guard<std::mutex> lk(mu);
      worker_err = kr_last_error();
      ++workers_ready;
      cv_done.notify_all();
      return;
    }
    std::vector<double> wc_local(wcount.size(), 0.0); // --summarize: this worker's sums
    double tw_local = 0;
    bool text_on = false;
    if (dev_text) { // room for 1.5 KB of rows per read (33 rows of ~28 bytes on a 1000-genome index) and 64 bytes of id; a batch with
                    // more is split like one with too many records (KR_ERR_CAPACITY below)
      const char* tpr = getenv("KR_CLI_TEXT_PER_READ"); // (tests: a buffer that is too small)
      const uint64_t tb = tpr ? std::max<uint64_t>(4096, (uint64_t)max_reads * strtoull(tpr, nullptr, 10)) : std::max<uint64_t>(32ull << 20, (uint64_t)max_reads * 1536ull);
      if (kr_stream_text_enable(st, hx, tb, std::max<uint64_t>(1ull << 20, (uint64_t)max_reads * 64)) == 0)
        text_on = true;
      else
        fprintf(stderr, "[krepp_amd] report text on the host: %s\n", kr_last_error());
    }
    t_ready = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_init).count();
    {
      std::lock_guard<std::mutex> lk(mu);
      ++workers_ready;
    }
    cv_done.notify_all();
    // reads per submit this worker currently trusts: halved when a submit overflows a device buffer, doubled again
    // after a run of successes -- data with hundreds of rows per read settle at a piece size instead of failing a
    // full-size submit (and every intermediate size) for every batch
    size_t piece = SIZE_MAX;
    int streak = 0;
    for (;;) {
      Job* j = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return !todo.empty() || eof; });
        if (todo.empty()) break;
        j = todo.front();
        todo.pop_front();
      }
      cv_work.notify_all(); // the reader may be waiting for queue space
      const char* const* nm = j->names;
      const size_t job_n = j->n;             // (`j` may be gone once its rows have taken their place in the output)
      kr_fastx_held* const job_held = j->held;
      std::string text;
      std::vector<kr_placement> pls;
      bool my_turn = false; // this worker holds the output (plain `dist` with device text: rows are written as they arrive)
      bool handed_over = false; // ... and has passed it on already (the job's only piece took its place in the file; `j` is gone)
      auto emit = [&](const char* p, size_t n, bool whole_job) {
        off_t at_off = -1;
        {
          std::unique_lock<std::mutex> lk(mu);
          if (!my_turn) cv_done.wait(lk, [&] { return next_seq == j->seq || !worker_err.empty(); });
          my_turn = true;
          if (!worker_err.empty()) return;
          if (out_seekable) {
            at_off = file_off;
            file_off += (off_t)n;
            if (whole_job) { // the next batch may take its place while this one is being copied
              j->done = true;
              finished[j->seq] = j;
              handed_over = true;
            }
          }
        }
        if (handed_over) cv_done.notify_all();
        auto t_w = now();
        if (at_off >= 0) {
          // A batch's text is hundreds of megabytes (262,144 reads x 33 rows x 28 bytes), and one thread copies 2.5-5 GB/s into the
          // page cache: with two workers the 38 GB of a 50 M-read run were written at 5 GB/s and the run was that (round 6,
          // profiles/round6_cli_syn1000_50m.txt).  The text goes out in slices of at least 16 MB, side by side (KR_CLI_WRITE_THREADS,
          // default 4 per worker).
          const int fd = fileno(out);
          auto write_slice = [&](size_t a, size_t b) {
            size_t done = a;
            while (done < b) {
              const ssize_t w = pwrite(fd, p + done, b - done, at_off + (off_t)done);
              if (w < 0) {
                if (errno == EINTR) continue;
                std::lock_guard<std::mutex> lk(mu);
                worker_err = std::string("cannot write the output: ") + strerror(errno);
                break;
              }
              done += (size_t)w;
            }
          };
          static const int wt_env = getenv("KR_CLI_WRITE_THREADS") ? std::max(1, atoi(getenv("KR_CLI_WRITE_THREADS"))) : 4;
          const size_t nsl = std::max<size_t>(1, std::min<size_t>((size_t)wt_env, n >> 24));
          if (nsl == 1) {
            write_slice(0, n);
          } else {
            std::vector<std::thread> ws_;
            for (size_t q = 1; q < nsl; ++q) ws_.emplace_back(write_slice, n * q / nsl, n * (q + 1) / nsl);
            write_slice(0, n / nsl);
            for (auto& t_ : ws_) t_.join();
          }
        } else if (n) {
          fwrite(p, 1, n, out);
        }
        ns_write += since(t_w);
      };
      // reads [lo, hi) of the job; a batch that overflows a device-side buffer (KR_ERR_CAPACITY: unusually many
      // table hits or records per read) is resubmitted in halves, as include/krepp_amd.h prescribes
      std::function<int(size_t, size_t)> run = [&](size_t lo, size_t hi) -> int {
        if (hi - lo > piece && hi - lo > 1) { // known to be too much for one submit
          const size_t mid = lo + (hi - lo) / 2;
          const int rc0 = run(lo, mid);
          return rc0 ? rc0 : run(mid, hi);
        }
        std::vector<uint64_t> offs(hi - lo + 1);
        for (size_t i = lo; i <= hi; ++i) offs[i - lo] = j->offsets[i] - j->offsets[lo];
        kr_result_view rv;
        char* txt = nullptr;
        uint64_t len = 0;
        auto t_dev = now();
        // dist rows / the summary need (leaf, selected, DIST) only: nothing else is copied back; place keeps its records
        // on the device; seek also reads the k-mer counts
        int rc;
        const char* dtext = nullptr;
        uint64_t dlen = 0;
        bool on_device = false;
        if (text_on) {
          std::vector<uint32_t> id_off(hi - lo + 1);
          for (size_t i = lo; i < hi; ++i) id_off[i - lo] = (uint32_t)(nm[i] - j->blob);
          id_off[hi - lo] = hi < j->n ? (uint32_t)(nm[hi] - j->blob) : (uint32_t)j->blob_bytes;
          rc = kr_batch_submit_text(st, j->bases + j->offsets[lo], offs.data(), (uint32_t)(hi - lo), KR_BASES_HOST, j->blob,
                                    id_off.data(), 1);
          if (!rc) rc = kr_batch_collect_text(st, &dtext, &dlen);
          on_device = rc == 0;
          if (rc == KR_ERR_UNSUPPORTED) rc = kr_batch_collect(st, &rv); // (a tiled batch: its rows as record slots, formatted below)
        } else {
          rc = kr_batch_submit(st, j->bases + j->offsets[lo], offs.data(), (uint32_t)(hi - lo),
                               KR_BASES_HOST | (place ? KR_TAP_ACCS : (seek ? 0u : KR_ROWS_ONLY)));
          if (!rc) rc = place ? kr_batch_wait(st) : kr_batch_collect(st, &rv); // place: the records stay on the device
        }
        ns_dev += since(t_dev);
        auto t_fmt = now();
        if (rc == KR_ERR_CAPACITY && hi - lo > 1) {
          piece = std::min(piece, (hi - lo + 1) / 2);
          streak = 0;
          const size_t mid = lo + (hi - lo) / 2;
          rc = run(lo, mid);
          return rc ? rc : run(mid, hi);
        }
        if (!rc && seek) rc = kr_format_seek(hx, dix[g], &rv, p.hdist_th, nm + lo, &txt, &len);
        if (!rc && on_device) {
          emit(dtext, dlen, lo == 0 && hi == job_n);
          if (piece != SIZE_MAX && ++streak >= 16) piece = piece > SIZE_MAX / 2 ? SIZE_MAX : piece * 2, streak = 0; // (as below: grow back after a run of successes)
          return 0;
        }
        if (!rc && !place && !seek && !summarize) rc = kr_format_dist(hx, &rv, nm + lo, &txt, &len);
        if (!rc && summarize && !place) { // each read shares one unit among the references it keeps (src/query.cpp:168-170)
          for (uint32_t r = 0; r < rv.nreads; ++r) {
            uint32_t o = rv.read_off[r], n = rv.read_cnt[r], ns = 0;
            for (uint32_t i = o; i < o + n; ++i) ns += rv.rec_sel[i];
            const double w = 1.0 / ns;
            for (uint32_t i = o; i < o + n; ++i)
              if (rv.rec_sel[i]) {
                const uint32_t se = rv.rec_key[i] >> 1;
                if (se < wc_local.size()) wc_local[se] += w;
                tw_local += w;
              }
          }
        }
        if (!rc && place) {
          int prev = (!tabular && !text.empty()) ? 1 : 0; // pieces of a batch are joined here, batches by the writer (src/krepp.cpp:474-484)
          kr_placement* pp = nullptr;
          uint64_t npp = 0;
          rc = kr_place_stream(hx, dix[g], ptree, st, (uint32_t)(hi - lo), offs.data(), nm + lo, &p, tabular, &prev, &txt, &len,
                               tabular == 2 ? &pp : nullptr, tabular == 2 ? &npp : nullptr);
          for (uint64_t i = 0; i < npp; ++i) {
            pp[i].read = (uint32_t)(j->first_read + lo + pp[i].read);
            pls.push_back(pp[i]);
          }
          kr_free(pp);
        }
        if (!rc && txt && dev_text)
          emit(txt, len, false); // (in order with the pieces the device wrote)
        else if (!rc && txt)
          text.append(txt, len);
        kr_free(txt);
        ns_fmt += since(t_fmt);
        if (!rc && piece != SIZE_MAX && ++streak >= 16) piece = piece > SIZE_MAX / 2 ? SIZE_MAX : piece * 2, streak = 0;
        return rc;
      };
      const int rc = run(0, job_n);
      if (job_held) kr_fastx_release(fx, job_held); // (the batch's buffers go back to the reader)
      t_last = at(), ++njobs;
      if (t_first < 0) t_first = t_last;
      if (handed_over) continue; // (its rows are in the file; the writer thread has the job)
      std::lock_guard<std::mutex> lk(mu);
      if (rc) {
        worker_err = kr_last_error();
      } else {
        j->text.swap(text);
        j->pls.swap(pls);
      }
      j->done = true;
      finished[j->seq] = j;
      cv_done.notify_all();
    }
    if (timing) fprintf(stderr, "[timing] worker %d: stream ready %.3f s into the initialisation, first batch done at %.3f s, last of %llu at %.3f s\n", wid, t_ready, t_first, (unsigned long long)njobs, t_last);
    std::lock_guard<std::mutex> lk(mu);
    for (size_t q = 0; q < wc_local.size(); ++q) wcount[q] += wc_local[q];
    twcount += tw_local;
    streams_to_free.push_back(st); // (torn down with the index, after the elapsed time has been taken)
  };
  // two workers (each with its own stream) per GPU: one formats its rows while the other's batch is on the device.  (`place` on a
  // 1000-genome tree, 8 M reads, 65,536-read batches: 8.2 M reads/s with two workers, 7.6 M with three -- `scripts/time_cli_place_big.py`;
  // with 400,000-read batches through the C ABI a third host thread does pay: DESIGN.md 3.4)
  const int wpg = getenv("KR_CLI_WORKERS_PER_GPU") ? std::max(1, atoi(getenv("KR_CLI_WORKERS_PER_GPU"))) : 2;
  const int nworkers = ngpus * wpg;
  if (kr_fastx_open(a.get("--query").c_str(), &fx)) error_exit(kr_last_error()); // (before the workers: they hand batches back to it)
  std::vector<std::thread> workers;
  for (int w = 0; w < nworkers; ++w) workers.emplace_back(worker, w % ngpus);

  bool jplace_prev = false;
  // place --summarize: counts per placement-tree node, summed by the writer in input order; placements of a
  // 512-read group that straddles two jobs wait in `carry` (the groups are the reference's batches)
  std::vector<double> pwcount(place ? kr_place_tree_nnodes(ptree) + 1 : 0, 0.0);
  double ptwcount = 0;
  std::vector<kr_placement> carry;
  std::thread writer([&] {
    uint64_t next = 0;
    for (;;) {
      Job* j = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return finished.count(next) || (eof && next >= total_batches) || !worker_err.empty(); });
        if (!worker_err.empty()) return;
        auto it = finished.find(next);
        if (it == finished.end()) return;
        j = it->second;
        finished.erase(it);
      }
      if (tabular == 2) {
        const uint32_t end_group = (uint32_t)(j->first_read + j->n) >> 9;
        carry.insert(carry.end(), j->pls.begin(), j->pls.end());
        size_t cut = carry.size();
        if ((j->first_read + j->n) & 511u)
          while (cut > 0 && (carry[cut - 1].read >> 9) == end_group) --cut;
        if (kr_place_summary_add(ptree, carry.data(), cut, pwcount.data(), &ptwcount)) {
          std::lock_guard<std::mutex> lk(mu);
          worker_err = kr_last_error();
          return;
        }
        carry.erase(carry.begin(), carry.begin() + cut);
      } else if (place && !tabular) {
        if (!j->text.empty()) {
          if (jplace_prev) fputs(",\n", out);
          fwrite(j->text.data(), 1, j->text.size(), out);
          jplace_prev = true;
        }
      } else {
        auto t_w = now();
        fwrite(j->text.data(), 1, j->text.size(), out);
        ns_write += since(t_w);
      }
      delete j;
      ++next;
      {
        std::lock_guard<std::mutex> lk(mu);
        next_seq = next;
      }
      cv_done.notify_all();
      cv_work.notify_all();
    }
  });

  if (seek) fprintf(stderr, "Seeking query sequences in the sketch...\n");
  if (!place && !seek) fprintf(stderr, "Estimating distances between given sequences and references...\n");
  std::thread ready_watch([&] { // ([timing]: when the last worker had its stream)
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return workers_ready == nworkers; });
    t_loop = std::chrono::steady_clock::now();
  });
  uint64_t nbatches = 0, nreads_total = 0;
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!worker_err.empty()) break; // (a worker without a stream, or one whose batch failed: reported below)
    }
    kr_fastx_batch b;
    auto t_parse = now();
    if (kr_fastx_next(fx, batch_bases, &b)) error_exit(kr_last_error());
    ns_parse += since(t_parse);
    auto t_job = now();
    // split over-long batches so that they fit the stream limits
    uint32_t r0 = 0;
    while (r0 < b.nreads) {
      uint32_t r1 = r0;
      while (r1 < b.nreads && r1 - r0 < max_reads && b.offsets[r1 + 1] - b.offsets[r0] <= max_bases) ++r1;
      if (r1 == r0) error_exit("A query sequence is longer than the supported maximum per batch");
      Job* j = new Job();
      j->n = r1 - r0;
      if (r0 == 0 && r1 == b.nreads && !getenv("KR_CLI_COPY_BATCHES")) {
        // the whole batch is one job: it changes hands as it is (QSeq gives IBatch its vectors by swap, src/query.cpp:32-33); the
        // reader thread copies nothing and the worker gives the buffers back when the batch is done
        if (kr_fastx_detach(fx, &j->held)) error_exit(kr_last_error());
        j->bases = b.bases, j->offsets = b.offsets, j->names = b.names;
      } else {
        j->own_bases.assign(b.bases + b.offsets[r0], b.bases + b.offsets[r1]);
        j->own_offsets.resize(r1 - r0 + 1);
        for (uint32_t i = r0; i <= r1; ++i) j->own_offsets[i - r0] = b.offsets[i] - b.offsets[r0];
        // the reader keeps a batch's names back to back in one buffer, in order (include/krepp_amd.h)
        const char* first = b.names[r0];
        const char* last = b.names[r1 - 1];
        j->own_blob.assign(first, (size_t)(last - first) + strlen(last) + 1);
        j->own_names.resize(r1 - r0);
        for (uint32_t i = r0; i < r1; ++i) j->own_names[i - r0] = j->own_blob.data() + (b.names[i] - first);
        j->bases = j->own_bases.data(), j->offsets = j->own_offsets.data(), j->names = j->own_names.data();
      }
      j->blob = j->names[0];
      j->blob_bytes = (size_t)(j->names[j->n - 1] - j->names[0]) + strlen(j->names[j->n - 1]) + 1;
      j->first_read = nreads_total;
      nreads_total += r1 - r0;
      {
        std::unique_lock<std::mutex> lk(mu);
        j->seq = nbatches++;
        // bound the number of batches in flight
        cv_work.wait(lk, [&] { return todo.size() < (size_t)(2 * nworkers) || !worker_err.empty(); });
        todo.push_back(j);
      }
      cv_work.notify_all();
      r0 = r1;
    }
    ns_job += since(t_job);
    if (!b.more) break;
  }
  if (timing) fprintf(stderr, "[timing] reader: end of input at %.3f s (%llu batches)\n", at(), (unsigned long long)nbatches);
  {
    std::lock_guard<std::mutex> lk(mu);
    eof = true;
    total_batches = nbatches;
  }
  cv_work.notify_all();
  cv_done.notify_all();
  for (auto& w : workers) w.join();
  cv_done.notify_all();
  ready_watch.join();
  writer.join();
  if (!worker_err.empty()) error_exit(worker_err);
  if (summarize && !place) // src/krepp.cpp:388-393 (ascending colour id instead of hash-map order)
    for (uint32_t se = 0; se < wcount.size(); ++se)
      if (wcount[se] != 0) fprintf(out, "%s\t%.5f\t%.5f\n", kr_host_index_node_name(hx, se), wcount[se], wcount[se] / twcount);
  if (tabular == 2) { // src/krepp.cpp:493-497
    char* t = nullptr;
    uint64_t l = 0;
    if (kr_place_summary_add(ptree, carry.data(), carry.size(), pwcount.data(), &ptwcount) ||
        kr_place_summary_text(ptree, pwcount.data(), ptwcount, &t, &l))
      error_exit(kr_last_error());
    fwrite(t, 1, l, out);
    kr_free(t);
  }
  if (place) {
    char* t = nullptr;
    uint64_t l = 0;
    if (kr_place_frame(ptree, 1, tabular, invocation.c_str(), nreads_total, &t, &l)) error_exit(kr_last_error());
    fwrite(t, 1, l, out);
    kr_free(t);
    kr_place_tree_free(ptree);
  }
  if (out_seekable) {
    fflush(out);
    (void)fseeko(out, file_off, SEEK_SET);
  }
  if (out != stdout) fclose(out);
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, place ? "Done placing queries, elapsed: %g sec (%.0f reads/s on %d GPU(s))\n" : seek ? "Done seeking query sequences, elapsed: %g sec (%.0f reads/s on %d GPU(s))\n" : "Done estimating distances, elapsed: %g sec (%.0f reads/s on %d GPU(s))\n", sec,
          sec > 0 ? nreads_total / sec : 0.0, ngpus);
  if (timing)
    fprintf(stderr, "[timing] parse %.3f s, job hand-over (incl. waiting for queue space) %.3f s, device %.3f s, format %.3f s, write %.3f s; of the elapsed time, until the last worker had its stream and page-locked buffers: %.3f s (the reader parses meanwhile)\n",
            ns_parse / 1e9, ns_job / 1e9, ns_dev / 1e9, ns_fmt / 1e9, ns_write / 1e9, std::chrono::duration<double>(t_loop - t_init).count());
  fprintf(stderr, "Total number of sequences queried: %llu\n", (unsigned long long)nreads_total);
  // Everything is written.  Unmapping 19 GB of index, 10 GB of host tables and the page-locked buffers one by one takes 0.8-1.2 s
  // that nobody is waiting for: the process ends here and the kernel reclaims the lot (KR_CLI_CLEAN_EXIT=1: free everything in
  // order -- leak checkers, tests of the tear-down path).
  if (!getenv("KR_CLI_CLEAN_EXIT")) {
    fflush(stdout);
    fflush(stderr);
    _exit(0);
  }
  auto t_down = now();
  kr_fastx_close(fx); // (the reader's thread pool and chunk buffers)
  for (kr_stream* st : streams_to_free) kr_stream_destroy(st);
  for (auto* d : dix) kr_index_free(d);
  kr_host_index_free(hx);
  if (timing) fprintf(stderr, "[timing] tear-down %.3f s\n", since(t_down) / 1e9);
  return 0;
}

static int run_index(const Args& a)
{
  // the reference's `index` takes -i,--input-file (name<TAB>path map) and -o,--index-dir
  // (src/krepp.cpp:563-566); the shared short-option table maps -i/-o to the dist names.
  std::string input, outdir;
  if (a.has("--input-file")) {
    input = a.get("--input-file");
    outdir = a.has("--index-dir") ? a.get("--index-dir") : a.get("--output-path");
  } else {
    input = a.get("--index-dir");
    outdir = a.get("--output-path");
  }
  if (input.empty() || outdir.empty()) error_exit("index requires -i/--input-file and -o/--index-dir");
  kr_build_params bp;
  memset(&bp, 0, sizeof(bp));
  bp.k = 29, bp.w = 35, bp.h = 13, bp.m = 4, bp.r = 1, bp.frac = 1; // src/krepp.hpp:47-58
  if (a.has("--kmer-len")) bp.k = (uint32_t)atoi(a.get("--kmer-len").c_str());
  bool w_given = a.has("--win-len");
  if (w_given) bp.w = (uint32_t)atoi(a.get("--win-len").c_str());
  if (a.has("--num-positions")) bp.h = (uint32_t)atoi(a.get("--num-positions").c_str());
  if (!w_given) { // src/krepp.cpp:579-582
    bp.w = bp.k + 6;
    if (!a.has("--num-positions")) bp.h = bp.k - 16;
  }
  if (a.has("--modulo-lsh")) bp.m = (uint32_t)atoi(a.get("--modulo-lsh").c_str());
  if (a.has("--residue-lsh")) bp.r = (uint32_t)atoi(a.get("--residue-lsh").c_str());
  if (a.flag.count("--frac")) bp.frac = a.flag.at("--frac");
  bp.num_threads = a.has("--num-threads") ? (uint32_t)atoi(a.get("--num-threads").c_str()) : 1;
  bp.seed = a.has("--seed") ? (uint32_t)atoi(a.get("--seed").c_str()) : 0;
  bp.gpu_minimizers = a.flag.count("--gpu-minimizers") && a.flag.at("--gpu-minimizers");
  bp.device = a.has("--device") ? atoi(a.get("--device").c_str()) : 0;
  fprintf(stderr, "Building the index...\n");
  std::string nwk = a.get("--nwk-file");
  if (kr_build_index(input.c_str(), nwk.empty() ? nullptr : nwk.c_str(), outdir.c_str(), &bp)) error_exit(kr_last_error());
  fprintf(stderr, "Done converting & saving\n");
  return 0;
}

static int run_sketch(const Args& a)
{ // SketchSingle (src/krepp.cpp:516-541): -i,--input-file  -o,--output-path; defaults k 26, w k+6, h k-16, m 4, r 1, frac
  const std::string input = a.has("--input-file") ? a.get("--input-file") : a.get("--index-dir");
  const std::string outp = a.get("--output-path");
  if (input.empty() || outp.empty()) error_exit("sketch requires -i/--input-file and -o/--output-path");
  kr_build_params bp;
  memset(&bp, 0, sizeof(bp));
  bp.k = 26, bp.m = 4, bp.r = 1, bp.frac = 1;
  if (a.has("--kmer-len")) bp.k = (uint32_t)atoi(a.get("--kmer-len").c_str());
  bp.w = bp.k + 6, bp.h = bp.k - 16; // src/krepp.cpp:533-536 (and the defaults [k+6], [k-16])
  if (a.has("--win-len")) {
    bp.w = (uint32_t)atoi(a.get("--win-len").c_str());
    bp.h = a.has("--num-positions") ? (uint32_t)atoi(a.get("--num-positions").c_str()) : 10; // h keeps its default
  }
  if (a.has("--modulo-lsh")) bp.m = (uint32_t)atoi(a.get("--modulo-lsh").c_str());
  if (a.has("--residue-lsh")) bp.r = (uint32_t)atoi(a.get("--residue-lsh").c_str());
  if (a.flag.count("--frac")) bp.frac = a.flag.at("--frac");
  bp.seed = a.has("--seed") ? (uint32_t)atoi(a.get("--seed").c_str()) : 0;
  fprintf(stderr, "Initializing the sketch...\n");
  if (kr_build_sketch(input.c_str(), outp.c_str(), &bp)) error_exit(kr_last_error());
  fprintf(stderr, "Done sketching & saving\n");
  return 0;
}

int main(int argc, char** argv)
{
  // more hardware queues than the runtime's default of 4: with two workers per GPU (each a kr_stream with its own HIP
  // streams) every stream after the third would share the last queue, and copies queued there turn into blit kernels
  // behind the other worker's kernels instead of SDMA transfers beside them.  Before the first HIP call.
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  // batches allocate and free tens of MB of rows and text over and over: keep that memory in the heap instead of
  // mapping and unmapping it each time (page faults and mmap locking showed up as 50 ms stalls per batch)
  if (!getenv("KR_CLI_DEFAULT_MALLOC")) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, -1);
  }
  fprintf(stderr, "krepp version: " KREPP_VERSION " (krepp-amd, MI355X)\n"); // PRINT_VERSION, src/common.hpp:51
  Args a = parse(argc, argv);
  std::string invocation;
  for (int i = 0; i < argc; ++i) invocation += std::string(argv[i]) + (i + 1 < argc ? " " : "");
  std::time_t now = std::time(nullptr);
  fprintf(stderr, "Invocation: %s\n%s", invocation.c_str(), std::ctime(&now));
  int rc;
  if (a.sub == "dist")
    rc = run_query(a, invocation, 0);
  else if (a.sub == "place")
    rc = run_query(a, invocation, 1);
  else if (a.sub == "seek")
    rc = run_query(a, invocation, 2);
  else if (a.sub == "index")
    rc = run_index(a);
  else if (a.sub == "sketch")
    rc = run_sketch(a);
  else if (a.sub == "inspect")
    error_exit("sub-command `inspect` is outside the scope of this build (dist/place/seek/index/sketch)");
  else
    error_exit("A subcommand is required (dist | place | seek | index | sketch)");
  now = std::time(nullptr);
  fprintf(stderr, "%s", std::ctime(&now));
  return rc;
}
