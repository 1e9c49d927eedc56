// Guard-page device allocator for torch.cuda.memory.CUDAPluggableAllocator (debugging aid, not product code).
//
// Every allocation gets its OWN virtual-address reservation (HIP virtual memory management API) in which only the pages that hold the
// tensor are mapped; the tensor is placed so that its last byte (rounded up to 16 bytes) is the last mapped byte (CVH_GUARD_MODE=end, the
// default) or so that its first byte is the first mapped byte (CVH_GUARD_MODE=begin).  A kernel that reads or writes past the end (before
// the beginning) of ANY operand therefore raises a GPU memory-access fault at the launch that does it instead of silently touching a
// neighbour in the caching allocator's segment.  free() waits for the device, unmaps the pages and NEVER reuses the address range, so a
// later launch through a stale pointer faults as well (use-after-free).
//
//   hipcc -O1 -shared -fPIC -o tools/_build/libguard_alloc.so tools/guard_alloc.cpp
//   CVH_GUARD_ALLOC=tools/_build/libguard_alloc.so AMD_SERIALIZE_KERNEL=3 CVH_TRACE_CALLS=1 python -m pytest tests/test_kernels_gpu.py -m gpu -v
// (tests/conftest.py installs it before the first device allocation.)  hipGraph captures are not supported under it.
#include <hip/hip_runtime.h>

#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Rec {
  void* va;
  size_t reserved, mapped;
  hipMemGenericAllocationHandle_t handle;
};
std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;
size_t g_gran = 0;
long long g_allocs = 0, g_bytes = 0;

// every allocation ever made (the address ranges are never reused): dumped when the process aborts, which is what the HSA runtime does on a
// GPU memory fault after printing the faulting address — tools/guard_run.py looks the address up in the dump
struct Hist {
  void* va;
  void* ptr;
  size_t reserved, size;
  long long seq;
  bool live;
};
std::vector<Hist> g_hist;

void dump_history(int) {
  const char* path = getenv("CVH_GUARD_DUMP");
  FILE* f = path ? fopen(path, "w") : nullptr;
  if (f) {
    for (const Hist& h : g_hist)
      fprintf(f, "%lld %p %zu %p %zu %d\n", h.seq, h.ptr, h.size, h.va, h.reserved, h.live ? 1 : 0);
    fclose(f);
  }
  signal(SIGABRT, SIG_DFL);
}

#define GA_CHECK(x)                                                                                 \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      fprintf(stderr, "[guard_alloc] %s -> %s (allocs %lld, %lld MB mapped)\n", #x, hipGetErrorString(e_), g_allocs, g_bytes >> 20); \
      fflush(stderr);                                                                               \
      abort();                                                                                      \
    }                                                                                               \
  } while (0)
}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  if (!g_gran) {
    GA_CHECK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
    fprintf(stderr, "[guard_alloc] granularity %zu bytes, mode %s\n", g_gran, getenv("CVH_GUARD_MODE") ? getenv("CVH_GUARD_MODE") : "end");
    signal(SIGABRT, dump_history);
    g_hist.reserve(1 << 20);
  }
  static const bool at_begin = getenv("CVH_GUARD_MODE") && !strcmp(getenv("CVH_GUARD_MODE"), "begin");
  const size_t need = size > 0 ? (size_t)size : 16;
  const size_t mapped = (need + g_gran - 1) / g_gran * g_gran;
  Rec r;
  r.mapped = mapped;
  r.reserved = mapped + 2 * g_gran;  // one unmapped granule on either side
  GA_CHECK(hipMemAddressReserve(&r.va, r.reserved, g_gran, nullptr, 0));
  GA_CHECK(hipMemCreate(&r.handle, mapped, &prop, 0));
  char* base = static_cast<char*>(r.va) + g_gran;
  GA_CHECK(hipMemMap(base, mapped, 0, r.handle, 0));
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  GA_CHECK(hipMemSetAccess(base, mapped, &acc, 1));
  void* p = at_begin ? base : base + (mapped - (need + 15) / 16 * 16);
  g_live[p] = r;
  g_hist.push_back(Hist{r.va, p, r.reserved, need, g_allocs, true});
  ++g_allocs;
  g_bytes += (long long)mapped;
  return p;
}

extern "C" void guard_free(void* ptr, ssize_t, int, hipStream_t) {
  if (!ptr) return;
  GA_CHECK(hipDeviceSynchronize());  // the pluggable allocator frees when the tensor dies, not when the stream is done with it
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_live.find(ptr);
  if (it == g_live.end()) {
    fprintf(stderr, "[guard_alloc] free of an unknown pointer %p\n", ptr);
    return;
  }
  Rec r = it->second;
  g_live.erase(it);
  for (size_t i = g_hist.size(); i-- > 0;)
    if (g_hist[i].ptr == ptr) {
      g_hist[i].live = false;
      break;
    }
  GA_CHECK(hipMemUnmap(static_cast<char*>(r.va) + g_gran, r.mapped));
  GA_CHECK(hipMemRelease(r.handle));
  g_bytes -= (long long)r.mapped;
  // the reservation is kept for the life of the process: the range is never handed out again, so a stale pointer keeps faulting
}
