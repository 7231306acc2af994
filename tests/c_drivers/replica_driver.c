/* replica_driver.c -- test driver in plain C over include/krepp_amd.h (built by tests/test_gpu_rccl_cli.py with gcc).
 *
 *   replica_driver INDEX_DIR READS.fq
 *
 * A process WITHOUT python or torch: load the index, upload it to device 0, replicate it through kr_index_broadcast onto
 * the same device (a one-rank RCCL communicator, the library dlopen()s librccl.so.1 itself), free the original, run the
 * reads on the REPLICA and print the `dist` rows.  Also prints which librccl the process mapped (from /proc/self/maps).
 * Mirrors what `krepp dist --gpus N` does for the GPUs after the first (src/krepp.cpp:92-106 is the reference's load loop).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "krepp_amd.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    int rc__ = (call);                                                     \
    if (rc__) {                                                            \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, kr_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)


int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  kr_host_index* hx = NULL;
  kr_index_view view;
  kr_index *root = NULL, *rep = NULL;
  CHECK(kr_host_index_load(argv[1], &hx));
  CHECK(kr_host_index_view(hx, &view));
  CHECK(kr_index_upload(&view, 0, KR_VIEW_HOST, &root));
  int dev = 0;
  /* RCCL may print a version banner on stdout when its first communicator is made; the library leaves file descriptors
   * alone, so the application (single-threaded here) points fd 1 at stderr around the call, like krepp_main.cpp does */
  fflush(stdout);
  int saved_out = dup(1);
  if (saved_out >= 0) dup2(2, 1);
  int brc = kr_index_broadcast(root, 1, &dev, &rep);
  if (saved_out >= 0) {
    fflush(stdout);
    dup2(saved_out, 1);
    close(saved_out);
  }
  CHECK(brc);
  if (kr_index_device_bytes(rep) != kr_index_device_bytes(root)) {
    fprintf(stderr, "replica size differs\n");
    return 1;
  }
  kr_index_free(root); /* only the replica answers from here on */
  {
    FILE* m = fopen("/proc/self/maps", "r");
    char line[4096];
    int found = 0;
    while (m && fgets(line, sizeof line, m)) {
      char* p = strstr(line, "librccl");
      if (p && !found) {
        char* path = strchr(line, '/');
        if (path) {
          path[strcspn(path, "\n")] = 0;
          fprintf(stderr, "rccl: %s\n", path);
          found = 1;
        }
      }
      if (strstr(line, "libtorch") || strstr(line, "libpython")) {
        fprintf(stderr, "unexpected: %s", line);
        return 1;
      }
    }
    if (m) fclose(m);
    if (!found) {
      fprintf(stderr, "librccl is not mapped\n");
      return 1;
    }
  }
  kr_fastx* fx = NULL;
  CHECK(kr_fastx_open(argv[2], &fx));
  kr_params p;
  kr_params_default(&p);
  kr_stream* st = NULL;
  CHECK(kr_stream_create(rep, &p, 1u << 16, 1u << 26, 0, &st));
  for (;;) {
    kr_fastx_batch b;
    CHECK(kr_fastx_next(fx, 76800, &b));
    if (b.nreads) {
      kr_result_view rv;
      char* text = NULL;
      uint64_t len = 0;
      CHECK(kr_batch_submit(st, b.bases, b.offsets, b.nreads, KR_BASES_HOST));
      CHECK(kr_batch_collect(st, &rv));
      CHECK(kr_format_dist(hx, &rv, b.names, &text, &len));
      fwrite(text, 1, len, stdout);
      kr_free(text);
    }
    if (!b.more) break;
  }
  kr_stream_destroy(st);
  kr_fastx_close(fx);
  kr_index_free(rep);
  kr_host_index_free(hx);
  return 0;
}
