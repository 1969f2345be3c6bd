// Device buffers for the output profiles, built with the HIP virtual-memory API from 1 GB physical allocations (plus one
// tail allocation).  Measured with the solve kernels' store pattern (tools/vmm_bw.hip, profiles/r01/vmm_bw.log): arrays
// backed by 1 GB physical chunks are written at 6.9 TB/s where 2-128 MB chunks or plain hipMalloc memory in the same
// region give 6.7-6.8 TB/s.  Which physical region the driver hands out still decides between the fast and the slow mode
// (DESIGN.md section 3.1); batched.Plan(placement="auto") times candidates of either kind.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <vector>

#include "crt1d_hip.h"

namespace {

struct Buffer {
  size_t size;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::mutex g_mu;
std::map<void*, Buffer> g_buffers;

constexpr size_t CHUNK = 1ull << 30;

void release(void* va, Buffer& b, size_t mapped) {
  if (mapped) (void)hipMemUnmap(va, mapped);
  for (auto h : b.handles) (void)hipMemRelease(h);
  (void)hipMemAddressFree(va, b.size);
}

}  // namespace

extern "C" {

int crt_hip_buffer_alloc(size_t bytes, void** ptr) {
  if (!ptr || bytes == 0) return CRT_ERR_BAD_ARG;
  *ptr = nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CRT_ERR_LAUNCH;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0)
    return CRT_ERR_UNSUPPORTED;
  if (gran < (2u << 20)) gran = 2u << 20;  // keep 2 MB alignment of every mapping
  Buffer b;
  b.size = ((bytes + gran - 1) / gran) * gran;
  void* va = nullptr;
  if (hipMemAddressReserve(&va, b.size, 0, nullptr, 0) != hipSuccess) return CRT_ERR_WORKSPACE;
  size_t off = 0;
  while (off < b.size) {
    const size_t c = b.size - off < CHUNK ? b.size - off : CHUNK;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, c, &prop, 0) != hipSuccess) {
      release(va, b, off);
      return CRT_ERR_WORKSPACE;
    }
    b.handles.push_back(h);
    if (hipMemMap(static_cast<char*>(va) + off, c, 0, h, 0) != hipSuccess) {
      release(va, b, off);
      return CRT_ERR_WORKSPACE;
    }
    off += c;
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(va, b.size, &acc, 1) != hipSuccess) {
    release(va, b, b.size);
    return CRT_ERR_WORKSPACE;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_buffers[va] = std::move(b);
  *ptr = va;
  return CRT_OK;
}

int crt_hip_buffer_free(void* ptr) {
  Buffer b;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_buffers.find(ptr);
    if (it == g_buffers.end()) return CRT_ERR_BAD_ARG;
    b = std::move(it->second);
    g_buffers.erase(it);
  }
  release(ptr, b, b.size);
  return CRT_OK;
}

}  // extern "C"
