// Debug allocator for torch.cuda.memory.CUDAPluggableAllocator: every allocation is mapped with the HIP virtual-memory
// API so that it ENDS at the end of its mapping and is followed by unmapped address space (a guard): a kernel that
// reads or writes >= 256 bytes past the end of any tensor faults right there, in eager mode, where the library call
// that did it can be named (tools/debug/guard_step.py).  Nothing is ever unmapped (a freed tensor may still be in use by
// queued kernels); meant for a few steps.
//   hipcc -shared -fPIC -O1 tools/debug/guard_alloc.cpp -o tools/debug/libguard_alloc.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <sys/types.h>

static size_t g_gran = 0;
static size_t g_total = 0;

extern "C" void *guard_malloc(ssize_t size, int device, hipStream_t) {
  if (size <= 0) return nullptr;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  if (!g_gran) {
    if (hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !g_gran) {
      fprintf(stderr, "guard_alloc: no allocation granularity\n");
      abort();
    }
    fprintf(stderr, "guard_alloc: granularity %zu\n", g_gran);
  }
  const size_t need = ((size_t)size + 255) & ~(size_t)255;           // torch expects >= 256-byte aligned blocks
  const size_t mapped = (need + g_gran - 1) / g_gran * g_gran;
  void *va = nullptr;
  hipMemGenericAllocationHandle_t h;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  static const bool head = getenv("GUARD_HEAD") != nullptr;     // guard BEFORE the tensor instead (catches under-runs)
  bool ok = hipMemAddressReserve(&va, mapped + 2 * g_gran, g_gran, nullptr, 0) == hipSuccess;   // unmapped granule on both sides
  if (ok) va = (char *)va + g_gran;
  if (!ok || hipMemCreate(&h, mapped, &prop, 0) != hipSuccess || hipMemMap(va, mapped, 0, h, 0) != hipSuccess ||
      hipMemSetAccess(va, mapped, &acc, 1) != hipSuccess) {
    fprintf(stderr, "guard_alloc: mapping %zu bytes failed (%s), %zu mapped so far\n", mapped, hipGetErrorString(hipGetLastError()), g_total);
    abort();
  }
  g_total += mapped;
  return head ? va : (char *)va + (mapped - need);
}

extern "C" void guard_free(void *, ssize_t, int, hipStream_t) {}
