// GPU utilisation sampler for MI355X nodes: the AMD-SMI counterpart of the reference's only native
// component (NVML/NVML.cpp:47-88, started by gpu.sh:7).  Same contract for the harness that reads
// its output (<job>_gpu.txt):
//   - one line per device per tick:  "H:M:S:ms  Device i: <name>  GPU Util: <u>  Mem Util: <m> Mem Usage: <bytes>"
//     (local wall-clock time, un-padded fields, utilisation in percent, VRAM in use in bytes);
//   - 6 ticks per second by default (166667 us period), unbuffered stdout;
//   - SIGINT (also SIGTERM) ends the loop cleanly.
// Differences, deliberate: the period is kept against a monotonic clock (the reference subtracts
// only tv_usec and drifts when a tick crosses a second boundary); a device that fails one query
// prints zeros for that field instead of ending the whole sampler; "Mem Util" is the memory-
// controller (UMC) activity AMD-SMI reports, the nearest equivalent of NVML's memory utilisation.
//
//   usage: amdsmi_sampler [--hz F] [--count N]
#include <amd_smi/amdsmi.h>

#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include <sys/time.h>
#include <unistd.h>

namespace {

volatile std::sig_atomic_t g_run = 1;
void on_signal(int) { g_run = 0; }

struct Gpu {
  amdsmi_processor_handle h;
  std::string name;
};

std::vector<Gpu> discover() {
  std::vector<Gpu> out;
  uint32_t ns = 0;
  if (amdsmi_get_socket_handles(&ns, nullptr) != AMDSMI_STATUS_SUCCESS || ns == 0) return out;
  std::vector<amdsmi_socket_handle> sockets(ns);
  if (amdsmi_get_socket_handles(&ns, sockets.data()) != AMDSMI_STATUS_SUCCESS) return out;
  for (uint32_t s = 0; s < ns; ++s) {
    uint32_t np = 0;
    if (amdsmi_get_processor_handles(sockets[s], &np, nullptr) != AMDSMI_STATUS_SUCCESS || np == 0) continue;
    std::vector<amdsmi_processor_handle> procs(np);
    if (amdsmi_get_processor_handles(sockets[s], &np, procs.data()) != AMDSMI_STATUS_SUCCESS) continue;
    for (uint32_t p = 0; p < np; ++p) {
      processor_type_t type;
      if (amdsmi_get_processor_type(procs[p], &type) != AMDSMI_STATUS_SUCCESS || type != AMDSMI_PROCESSOR_TYPE_AMD_GPU) continue;
      Gpu g;
      g.h = procs[p];
      amdsmi_asic_info_t asic;
      std::memset(&asic, 0, sizeof(asic));
      g.name = amdsmi_get_gpu_asic_info(procs[p], &asic) == AMDSMI_STATUS_SUCCESS && asic.market_name[0] ? asic.market_name : "AMD GPU";
      out.push_back(g);
    }
  }
  return out;
}

long long now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000;
}

}  // namespace

int main(int argc, char** argv) {
  double hz = 6.0;
  long long count = -1;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--hz") && i + 1 < argc) hz = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--count") && i + 1 < argc) count = std::atoll(argv[++i]);
    else {
      std::fprintf(stderr, "usage: %s [--hz F] [--count N]\n", argv[0]);
      return 64;
    }
  }
  if (!(hz > 0.0)) hz = 6.0;
  const long long period = (long long)(1e6 / hz + 0.5);  // 166667 us at 6 Hz

  std::signal(SIGINT, on_signal);
  std::signal(SIGTERM, on_signal);
  setvbuf(stdout, nullptr, _IONBF, 0);

  if (amdsmi_init(AMDSMI_INIT_AMD_GPUS) != AMDSMI_STATUS_SUCCESS) return 1;
  std::vector<Gpu> gpus = discover();
  if (gpus.empty()) {
    amdsmi_shut_down();
    return 2;
  }

  long long next = now_us();
  while (g_run && count != 0) {
    timeval tv;
    gettimeofday(&tv, nullptr);
    tm lt;
    localtime_r(&tv.tv_sec, &lt);
    const int ms = (int)(tv.tv_usec / 1000);
    for (size_t i = 0; i < gpus.size(); ++i) {
      amdsmi_engine_usage_t use;
      std::memset(&use, 0, sizeof(use));
      if (amdsmi_get_gpu_activity(gpus[i].h, &use) != AMDSMI_STATUS_SUCCESS) use.gfx_activity = use.umc_activity = 0;
      uint64_t used = 0;
      if (amdsmi_get_gpu_memory_usage(gpus[i].h, AMDSMI_MEM_TYPE_VRAM, &used) != AMDSMI_STATUS_SUCCESS) used = 0;
      std::printf("%d:%d:%d:%d  Device %zu: %s  GPU Util: %u  Mem Util: %u Mem Usage: %lli\n ", lt.tm_hour, lt.tm_min,
                  lt.tm_sec, ms, i, gpus[i].name.c_str(), use.gfx_activity, use.umc_activity, (long long)used);
    }
    if (count > 0) --count;
    next += period;
    const long long wait = next - now_us();
    if (wait > 0) usleep((useconds_t)wait);
    else next = now_us();  // fell behind (slow query): do not burst to catch up
  }
  amdsmi_shut_down();
  return 0;
}
