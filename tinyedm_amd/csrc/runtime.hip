// Per-device state of the library: the zero page.
//
// Several kernels point LDS-DMA loads of padding rows / out-of-range channels at a page of zeros instead of
// branching around them.  The page is allocated ONCE per device by edm_init() -- never lazily inside a launch
// function, so every compute entry point stays allocation-free, sync-free and graph-capturable from its first
// call (a cold call under stream capture used to hipMalloc) -- and is looked up per device, under a mutex.
#include "common.h"
#include <cstdlib>
#include <dirent.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace {
constexpr int MAX_DEV = 64;
constexpr size_t ZERO_BYTES = 4096;   // >= the longest zero row any kernel reads (conv_igemm4: Cin*2 + 64 bytes)
void* g_zero[MAX_DEV] = {};
std::mutex g_mu;
}  // namespace

// hipGraph replay on ROCm 7.2: with the runtime's default AQL-packet-capture path the first replay after a
// hipStreamSynchronize / hipDeviceSynchronize runs graph nodes with clobbered kernel arguments (measured round 2,
// see tinyedm_amd/_runtime_env.py).  The runtime reads DEBUG_CLR_GRAPH_PACKET_CAPTURE at its
// initialisation, so the library's load-time constructor sets it to 0 when the host has not chosen a value -- and
// records whether that can still have taken effect: FAIL CLOSED, the answer is yes only when the variable was already
// "0" when the library was loaded, or the process had not initialised the HIP runtime yet (it did not hold /dev/kfd
// open: runtime initialisation opens it).
namespace {
int g_graph_safe = 0;
bool kfd_open() {
  DIR* d = opendir("/proc/self/fd");
  if (!d) return true;  // cannot tell: assume the runtime is live
  bool live = false;
  char path[64], target[64];
  while (struct dirent* e = readdir(d)) {
    if (e->d_name[0] == '.') continue;
    snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
    const ssize_t n = readlink(path, target, sizeof target - 1);
    if (n > 0) {
      target[n] = 0;
      if (strcmp(target, "/dev/kfd") == 0) { live = true; break; }
    }
  }
  closedir(d);
  return live;
}
__attribute__((constructor)) void edm_runtime_ctor() {
  const char* e = getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE");
  if (e) {
    g_graph_safe = strcmp(e, "0") == 0;
  } else if (!kfd_open()) {
    (void)setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0", /*overwrite=*/0);
    g_graph_safe = 1;
  }
}
}  // namespace

// 1 when the graph-safe runtime setting was in place before the HIP runtime initialised (as far as the library can
// witness: see the constructor above), 0 otherwise.
extern "C" int edm_graph_replay_safe(void) { return g_graph_safe; }

// Allocate the device-side constants of `device` (idempotent, thread-safe).  Must be called once per device before
// the first kernel entry point, outside any stream capture; tinyedm_amd._lib does so when it binds the library.
extern "C" int edm_init(int device) {
  EDM_REQUIRE(device >= 0 && device < MAX_DEV, "edm_init: bad device %d", device);
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_zero[device]) return EDM_OK;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  void* p = nullptr;
  bool ok = hipSetDevice(device) == hipSuccess && hipMalloc(&p, ZERO_BYTES) == hipSuccess &&
            hipMemset(p, 0, ZERO_BYTES) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (prev >= 0) (void)hipSetDevice(prev);
  if (!ok) {
    edm_set_error("edm_init: cannot allocate the zero page on device %d", device);
    return EDM_ERR_LAUNCH;
  }
  g_zero[device] = p;
  return EDM_OK;
}

// The current device's zero page (4096 bytes), or NULL when edm_init() has not run for it.
extern "C" const void* edm_zero_page(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  return g_zero[dev];
}
