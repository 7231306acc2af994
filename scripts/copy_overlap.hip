// Experiment: does a host<->device copy overlap a kernel that fills every wave slot of the chip (what the persistent
// scan kernel does)?  hipMemcpyAsync (the runtime picks a blit kernel or an SDMA engine) against
// hsa_amd_memory_async_copy (always an SDMA engine).   build: hipcc --offload-arch=gfx950 -O2 copy_overlap.hip -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void spin(unsigned long long cycles, int* sink)
{
  asm volatile("v_mov_b32 v127, 0" ::: "v127"); // 128 VGPRs: 4 waves per SIMD fill the register file
  unsigned long long t0 = __builtin_readcyclecounter();
  int x = 0;
  while (__builtin_readcyclecounter() - t0 < cycles) x++;
  if (x == -1) *sink = x;
}
static hsa_agent_t g_gpu, g_cpu;
static bool have_gpu = false, have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*)
{
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) g_gpu = a, have_gpu = true;
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) g_cpu = a, have_cpu = true;
  return HSA_STATUS_SUCCESS;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  const size_t N = 430ull << 20;
  void *d, *h;
  int* sink;
  CK(hipMalloc(&d, N));
  CK(hipHostMalloc(&h, N, hipHostMallocDefault));
  CK(hipMalloc((void**)&sink, 4));
  CK(hipMemset(d, 1, N));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  hipStream_t sk, sc;
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  hsa_init();
  hsa_iterate_agents(agent_cb, nullptr);
  hsa_signal_t sig;
  hsa_signal_create(1, 0, nullptr, &sig);
  const unsigned long long cyc = 30000000ull; // ~15 ms of s_memtime ticks at ~2 GHz (the measured duration is printed)
  auto kernel = [&](unsigned long long c) { hipLaunchKernelGGL(spin, dim3(p.multiProcessorCount * 4), dim3(256), 0, sk, c, sink); };
  // calibrate
  kernel(cyc);
  CK(hipStreamSynchronize(sk));
  double t = now();
  kernel(cyc);
  CK(hipStreamSynchronize(sk));
  const double tk = now() - t;
  printf("kernel alone: %.2f ms (grid %d x 256, 128 VGPRs)\n", tk * 1e3, p.multiProcessorCount * 4);
  for (int dir = 0; dir < 2; ++dir) {
    const char* dn = dir ? "H2D" : "D2H";
    auto hipcopy = [&]() { return dir ? hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, sc) : hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, sc); };
    auto hsacopy = [&]() {
      hsa_signal_store_relaxed(sig, 1);
      return dir ? hsa_amd_memory_async_copy(d, g_gpu, h, g_cpu, N, 0, nullptr, sig) : hsa_amd_memory_async_copy(h, g_cpu, d, g_gpu, N, 0, nullptr, sig);
    };
    auto hsawait = [&]() { while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) {} };
    CK(hipcopy());
    CK(hipStreamSynchronize(sc));
    t = now();
    CK(hipcopy());
    CK(hipStreamSynchronize(sc));
    printf("%s hipMemcpyAsync alone: %.2f ms (%.1f GB/s)\n", dn, (now() - t) * 1e3, N / (now() - t) / 1e9);
    t = now();
    kernel(cyc);
    CK(hipcopy());
    CK(hipStreamSynchronize(sc));
    const double tc = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("%s hipMemcpyAsync beside the kernel: copy done after %.2f ms, both after %.2f ms\n", dn, tc * 1e3, (now() - t) * 1e3);
    hsa_status_t hs = hsacopy();
    if (hs != HSA_STATUS_SUCCESS) { printf("hsa_amd_memory_async_copy failed: %d\n", (int)hs); continue; }
    hsawait();
    t = now();
    hsacopy();
    hsawait();
    printf("%s hsa_amd_memory_async_copy alone: %.2f ms (%.1f GB/s)\n", dn, (now() - t) * 1e3, N / (now() - t) / 1e9);
    t = now();
    kernel(cyc);
    hsacopy();
    hsawait();
    const double tc2 = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("%s hsa_amd_memory_async_copy beside the kernel: copy done after %.2f ms, both after %.2f ms\n", dn, tc2 * 1e3, (now() - t) * 1e3);
  }
  // ---- the library's pattern: the copy follows kernels of its own batch while another batch's kernel fills the chip
  hipStream_t s3;
  CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  auto small_kernel = [&](hipStream_t st) { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, 100000ull, sink); };
  {
    small_kernel(sc);
    CK(hipStreamSynchronize(sc));
    t = now();
    kernel(cyc);
    CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, sc));
    CK(hipStreamSynchronize(sc));
    const double tc = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("A: D2H on a stream that ran a kernel before (synchronised): copy done after %.2f ms, both after %.2f ms\n", tc * 1e3, (now() - t) * 1e3);
  }
  {
    small_kernel(sc);
    CK(hipStreamSynchronize(sc));
    t = now();
    kernel(cyc);
    CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s3));
    CK(hipStreamSynchronize(s3));
    const double tc = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("B: D2H on a stream that never runs kernels: copy done after %.2f ms, both after %.2f ms\n", tc * 1e3, (now() - t) * 1e3);
  }
  {
    t = now();
    small_kernel(sc);
    CK(hipEventRecord(ev, sc));
    kernel(cyc);
    CK(hipStreamWaitEvent(s3, ev, 0));
    CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s3));
    CK(hipStreamSynchronize(s3));
    const double tc = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("C: D2H on a copy-only stream behind an event of the kernel stream (no host sync): copy done after %.2f ms, both after %.2f ms\n", tc * 1e3, (now() - t) * 1e3);
  }
  {
    t = now();
    small_kernel(sc);
    kernel(cyc);
    CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, sc));
    CK(hipStreamSynchronize(sc));
    const double tc = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("D: D2H right behind a kernel on the same stream (no host sync): copy done after %.2f ms, both after %.2f ms\n", tc * 1e3, (now() - t) * 1e3);
  }
  {
    t = now();
    CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, sc));
    small_kernel(sc);
    kernel(cyc);
    CK(hipStreamSynchronize(sc));
    const double tc = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("E: H2D then a kernel on the same stream, big kernel submitted after: copy+small done after %.2f ms, both after %.2f ms\n", tc * 1e3, (now() - t) * 1e3);
  }
  {
    t = now();
    kernel(cyc);
    CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, sc));
    small_kernel(sc);
    CK(hipStreamSynchronize(sc));
    const double tc = now() - t;
    CK(hipStreamSynchronize(sk));
    printf("F: big kernel running, then H2D + small kernel on another stream (which ran kernels before): done after %.2f ms, both after %.2f ms\n", tc * 1e3, (now() - t) * 1e3);
  }
  return 0;
}
