// Device buffers for the output profiles with a deterministic placement in HBM.
//
// Finding (round 2; tools/chunk_probe.hip, matrix_probe.hip, class_probe.hip, remap_probe.hip, hazard_probe.hip, logs under
// profiles/r02/placement/): on MI355X every piece of physical memory belongs to one of (up to) three CLASSES.  The solve
// kernels' store pattern -- one workgroup per column streaming T levels of every output array -- runs at ~5.5 TB/s when all the
// memory being written at one time belongs to ONE class, whether that is one array, two or eight, and at 6.9-7.0 TB/s when it
// is spread over two or three classes (50/50: 6.95, 25/75: 6.65, 1/7: 6.2).  The class is a property of the physical memory
// (it follows the allocation handle through re-mappings, not the virtual address), it is constant over runs of 2-16 GB in
// allocation order, and which run comes first differs from process to process: that is the "placement lottery" of round 1
// (hipMalloc'ed output arrays sit back to back in one run about half of the time).  The most likely cause is the three ranks of
// a 12-high HBM3E stack; nothing here depends on that interpretation.
//
// So this allocator does not search, it builds: memory is taken from the driver in 512 MB physical chunks (HIP virtual-memory
// API), each new chunk is CLASSIFIED once by timing the store pattern into it together with a reference chunk of each known class
// (same class: slow, different class: fast; ~1 ms per probe, footprint 1 GB so that the 256 MB Infinity Cache cannot hide it),
// and the arrays of a set are laid out so that chunk i of array a has class (a + i) mod K: at every moment of a kernel that
// sweeps all arrays of the set in step, the chunks being written are spread evenly over the classes.
//
// HIP virtual-memory hazards measured on ROCm 7.2 (tools/hazard_probe.hip) that shape the code:
//  * after hipMemUnmap + hipMemMap of ANOTHER handle at a virtual address that was mapped before, kernels and hipMemcpy keep
//    using the OLD physical memory.  Therefore no virtual range is ever mapped twice: ranges of freed buffers stay reserved
//    (never handed back with hipMemAddressFree, which could return them from a later hipMemAddressReserve).  Costs address
//    space only (47-bit space; a 72 GB set can be allocated ~1800 times per process).
//  * one handle can be mapped at two addresses at once; a chunk keeps the "home" mapping it was classified through and is
//    mapped a second time into the array it serves.
// Chunks of freed buffers go back to a per-device pool (still classified) and are reused -- up to a RETENTION CAP (default 8 GB per
// device, CRT1D_POOL_RETAIN_MB): what a freed buffer brings beyond the cap goes straight back to the driver (most common class first),
// so that dropping a plan with a 70 GB output set does not leave 70 GB where torch's caching allocator, an RCCL buffer or another
// process cannot get at it.  crt_hip_buffer_trim releases the rest.
// Exploration (allocating and classifying more chunks than the request needs, to find the other classes) is bounded: at most half of
// the memory that would stay free after the request, at most 96 GB, and it is abandoned for good on a device that still shows ONE class
// after 80 GB (another GPU model or partition mode: nothing to balance there).  Probes run on a private non-blocking stream, so busy
// torch streams do not distort their timing; they synchronise the host, so a Plan must be constructed outside stream capture.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "crt1d_hip.h"

namespace {

constexpr size_t CHUNK = 512ull << 20;
constexpr int NCLS = 3;           // classes 0..2; 3 = ambiguous (a chunk that straddles two runs)
constexpr int P_NZ = 60, P_NB = 300, P_T = 8;
constexpr int P_COLS = (int)(CHUNK / ((size_t)P_NZ * P_NB * 8));  // 3728 columns of 144 000 B fill a chunk
constexpr double SLOW_BELOW = 1.09;  // pair rate / single rate: same class ~1.00-1.02, different classes ~1.22-1.27
constexpr double FAST_ABOVE = 1.15;

typedef double d2 __attribute__((ext_vector_type(2)));

// the store pattern of k_pipe / k_tri_pipe: workgroup = column, T levels of every array per round, 16 B per lane
__global__ __launch_bounds__(512) void k_probe_cols(double* o0, double* o1, int na) {
  const long long base = (long long)blockIdx.x * P_NZ * P_NB;
  for (int j0 = 0; j0 < P_NZ; j0 += P_T) {
    const int n2 = min(P_T, P_NZ - j0) * P_NB / 2;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
      d2 v;
      v.x = 0.0;
      v.y = 0.0;
      reinterpret_cast<d2*>(o0 + base + (long long)j0 * P_NB)[i] = v;
      if (na > 1) reinterpret_cast<d2*>(o1 + base + (long long)j0 * P_NB)[i] = v;
    }
  }
}

struct Chunk {
  hipMemGenericAllocationHandle_t h;
  char* home;  // the mapping it was classified through (kept for the life of the chunk)
  int cls;
};

struct Pool {
  std::vector<Chunk> free_chunks;
  Chunk ref[NCLS];
  int nref = 0;
  double single_ms = 0.0;  // the pattern into one chunk alone
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t stream = nullptr;  // private, non-blocking: the probes
  bool one_class = false;        // exploration found a single class in 80 GB: never explore again on this device
  long long explored = 0;        // chunks classified by exploration so far (statistics / the rule above)
  // statistics (crt_hip_buffer_stats)
  long long created = 0, released = 0, probes = 0;
  double probe_ms_total = 0.0;
};

struct Buffer {
  size_t size;    // bytes reserved (multiple of CHUNK)
  int dev;
  std::vector<Chunk> chunks;
};

std::mutex g_mu;
std::map<int, Pool> g_pools;
std::map<void*, Buffer> g_buffers;

hipMemAllocationProp dev_prop(int dev) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  return prop;
}

bool set_access(void* va, size_t size, int dev) {
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  return hipMemSetAccess(va, size, &acc, 1) == hipSuccess;
}

// unmap a range for good: the reservation is deliberately kept (see the header comment)
void retire_range(void* va, size_t size) { (void)hipMemUnmap(va, size); }

void release_chunk(Pool& p, Chunk& c) {
  if (c.home) retire_range(c.home, CHUNK);
  (void)hipMemRelease(c.h);
  c.home = nullptr;
  ++p.released;
}

bool new_chunk(Pool& p, int dev, Chunk& c) {
  hipMemAllocationProp prop = dev_prop(dev);
  if (hipMemCreate(&c.h, CHUNK, &prop, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  void* va = nullptr;
  if (hipMemAddressReserve(&va, CHUNK, 0, nullptr, 0) != hipSuccess) {
    (void)hipMemRelease(c.h);
    return false;
  }
  if (hipMemMap(va, CHUNK, 0, c.h, 0) != hipSuccess || !set_access(va, CHUNK, dev)) {
    (void)hipMemRelease(c.h);
    return false;
  }
  c.home = static_cast<char*>(va);
  c.cls = NCLS;
  ++p.created;
  return true;
}

// average time of the store pattern into one or two chunks (the pool's private stream; the caller holds g_mu)
double probe_ms(Pool& p, char* a, char* b) {
  if (!p.e0) {
    if (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess) return -1.0;
    if (hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess) p.stream = nullptr;  // (null stream as a last resort)
  }
  const int na = b ? 2 : 1, reps = 3;
  hipStream_t st = p.stream;
  hipLaunchKernelGGL(k_probe_cols, dim3(P_COLS), dim3(512), 0, st, reinterpret_cast<double*>(a), reinterpret_cast<double*>(b ? b : a), na);
  (void)hipEventRecord(p.e0, st);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(k_probe_cols, dim3(P_COLS), dim3(512), 0, st, reinterpret_cast<double*>(a), reinterpret_cast<double*>(b ? b : a), na);
  (void)hipEventRecord(p.e1, st);
  if (hipEventSynchronize(p.e1) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) return -1.0;
  ++p.probes;
  p.probe_ms_total += ms;
  return ms / reps;
}

// class of a new chunk: the reference it is SLOW with.  A chunk that is fast with every known reference founds a new class and
// becomes its (dedicated, never handed out) reference: returns false then, the chunk is consumed.
bool classify(Pool& p, Chunk& c) {
  if (p.nref == 0) {
    (void)probe_ms(p, c.home, nullptr);  // first launches of the process: clocks and code object warm up
    p.single_ms = probe_ms(p, c.home, nullptr);
    c.cls = 0;
    p.ref[p.nref++] = c;
    return false;
  }
  bool all_fast = true;
  for (int r = 0; r < p.nref; ++r) {
    const double t = probe_ms(p, p.ref[r].home, c.home);
    if (t <= 0 || p.single_ms <= 0) {
      all_fast = false;
      break;
    }
    const double ratio = 2.0 * p.single_ms / t;  // pair rate / single rate
    if (ratio < SLOW_BELOW) {
      c.cls = r;
      return true;
    }
    if (ratio < FAST_ABOVE) all_fast = false;
  }
  if (all_fast && p.nref < NCLS) {
    c.cls = p.nref;
    p.ref[p.nref++] = c;
    return false;
  }
  c.cls = NCLS;  // ambiguous: usable, but counted in no class
  return true;
}

char cls_letter(int c) { return c < NCLS ? "XYZ"[c] : '?'; }

long long g_retain = -1;  // chunks a pool may keep; -1 = not set yet (CRT1D_POOL_RETAIN_MB or 8 GB); guarded by g_mu
size_t retain_chunks() {
  if (g_retain < 0) {
    const char* e = getenv("CRT1D_POOL_RETAIN_MB");
    const long long mb = e ? atoll(e) : 8192;
    g_retain = (mb < 0 ? 0 : mb) / 512;
  }
  return (size_t)g_retain;
}

// keep at most `keep` free chunks in the pool: release from the most common class first (the rare ones are what the next
// request will be short of), ambiguous chunks before everything else
void cap_pool(Pool& p, size_t keep) {
  while (p.free_chunks.size() > keep) {
    size_t cnt[NCLS + 1] = {0, 0, 0, 0};
    for (auto& c : p.free_chunks) ++cnt[c.cls <= NCLS ? c.cls : NCLS];
    int worst = NCLS;
    if (cnt[NCLS] == 0) {
      worst = 0;
      for (int c = 1; c < NCLS; ++c)
        if (cnt[c] > cnt[worst]) worst = c;
    }
    for (size_t k = p.free_chunks.size(); k-- > 0;)
      if (p.free_chunks[k].cls == worst) {
        release_chunk(p, p.free_chunks[k]);
        p.free_chunks.erase(p.free_chunks.begin() + k);
        break;
      }
  }
}

}  // namespace

extern "C" {

int crt_hip_buffer_alloc_set(int32_t n, const size_t* bytes, void** ptrs) {
  if (n <= 0 || n > 16 || !bytes || !ptrs) return CRT_ERR_BAD_ARG;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CRT_ERR_LAUNCH;
  std::vector<size_t> m(n);
  size_t need = 0, rows = 0;
  for (int a = 0; a < n; ++a) {
    if (bytes[a] == 0) return CRT_ERR_BAD_ARG;
    ptrs[a] = nullptr;
    m[a] = (bytes[a] + CHUNK - 1) / CHUNK;
    need += m[a];
    rows = std::max(rows, m[a]);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  Pool& p = g_pools[dev];
  // ---- gather chunks: the pool's free ones, then new ones until the selection can be balanced (or the cap is reached)
  std::vector<Chunk> have;
  have.swap(p.free_chunks);
  auto count = [&](int c) { return (size_t)std::count_if(have.begin(), have.end(), [c](const Chunk& k) { return k.cls == c; }); };
  auto balanced = [&]() {
    if (have.size() < need) return false;
    if (need < 2) return true;
    size_t most = 0, classified = 0;
    for (int c = 0; c < NCLS; ++c) {
      const size_t k = count(c);
      classified += k;
      most = std::max(most, k);
    }
    // the classes other than the largest one can supply half of the selection (ambiguous chunks count for nothing)
    return classified - most >= (need + 1) / 2;
  };
  // Gathering: first `need` chunks, whatever they are.  If they cannot be balanced (a process often starts inside a run of
  // 10-35 GB of one class), keep allocating and classifying chunks -- holding everything, so that the driver has to move on to
  // other memory -- until the other classes can supply half of the request or the exploration budget (memory held at one time)
  // is spent; ~1.7 ms per chunk (create + map + probe), i.e. ~0.15 s for 40 GB.  (Large unmapped "spacer" allocations do not
  // move the driver's cursor for 512 MB requests -- they are served from other free blocks -- and cost 0.3 s each: tried, dropped.)
  // Surplus chunks are released before returning, rarest classes kept.
  size_t free0 = 0, tot0 = 0;
  if (hipMemGetInfo(&free0, &tot0) != hipSuccess) free0 = need * CHUNK + (22ull << 30);
  free0 += have.size() * CHUNK;  // what the pool already holds counts as available to this request
  const size_t hard = free0 > (6ull << 30) ? free0 - (6ull << 30) : 0;  // never take the last 6 GB of the device
  // exploration: at most half of what would stay free after the request, at most 96 GB, nothing on a one-class device
  const size_t spare = hard > need * CHUNK ? hard - need * CHUNK : 0;
  const size_t extra = p.one_class ? 0 : std::min<size_t>(spare / 2, 96ull << 30);
  const size_t budget = std::min(hard, need * CHUNK + extra);
  // chunks (80 GB) classified with a single class seen.  (64 chunks at first: inside the 2-35 GB run lengths of one class that round 2
  // measured -- one bench process in forty started in such a run, gave up and ran its set at the single-class rate, 0.65 instead of 0.83.)
  constexpr long long ONE_CLASS_AFTER = 160;
  bool oom = false;
  while (have.size() < need || !balanced()) {
    if ((have.size() + 1) * CHUNK > (have.size() < need ? hard : budget)) {
      oom = have.size() < need;
      break;
    }
    if (have.size() >= need && p.nref <= 1 && p.explored >= ONE_CLASS_AFTER) {
      p.one_class = true;  // remembered: later requests take their chunks and go
      break;
    }
    Chunk c;
    if (!new_chunk(p, dev, c)) {
      oom = have.size() < need;
      break;
    }
    if (have.size() >= need) ++p.explored;
    if (classify(p, c)) have.push_back(c);
  }
  if (have.size() < need) {
    // not enough memory for the request: give everything gathered back to the driver (the caller will fall back to another allocator,
    // which needs that memory)
    for (auto& c : have) release_chunk(p, c);
    return oom ? CRT_ERR_WORKSPACE : CRT_ERR_LAUNCH;
  }
  // ---- how many chunks of each class to use: as even as the supply allows
  size_t avail[NCLS + 1], sel[NCLS + 1] = {0, 0, 0, 0};
  for (int c = 0; c <= NCLS; ++c) avail[c] = count(c);
  for (size_t k = 0; k < need; ++k) {  // water-filling: always take from the class used least so far that still has supply
    int best = -1;
    for (int c = 0; c < NCLS; ++c)
      if (sel[c] < avail[c] && (best < 0 || sel[c] < sel[best])) best = c;
    if (best < 0) best = NCLS;  // only ambiguous chunks left
    ++sel[best];
  }
  // ---- interleave: slots in row-major order (chunk index i, array a); error diffusion over the classes, rotated per row
  std::vector<std::vector<Chunk>> per(n);
  double credit[NCLS + 1] = {0, 0, 0, 0};
  size_t used[NCLS + 1] = {0, 0, 0, 0};
  auto take = [&](int c) {
    for (size_t k = 0; k < have.size(); ++k)
      if (have[k].cls == c) {
        Chunk ch = have[k];
        have.erase(have.begin() + k);
        return ch;
      }
    return Chunk{};  // unreachable: sel[] never exceeds the supply
  };
  for (size_t i = 0; i < rows; ++i)
    for (int a = 0; a < n; ++a) {
      if (i >= m[a]) continue;
      int best = -1;
      for (int c = 0; c <= NCLS; ++c) {
        credit[c] += (double)sel[c] / (double)need;
        if (used[c] < sel[c] && (best < 0 || credit[c] > credit[best] + 1e-12)) best = c;
      }
      credit[best] -= 1.0;
      ++used[best];
      per[a].push_back(take(best));
    }
  // ---- unselected chunks: newly gathered surplus goes back to the driver (keep a few for the next small request)
  {  // (the spares kept are the rarest classes: they are what the next request will be short of)
    std::stable_sort(have.begin(), have.end(), [&](const Chunk& x, const Chunk& y) {
      const size_t cx = x.cls < NCLS ? avail[x.cls] : (size_t)-1, cy = y.cls < NCLS ? avail[y.cls] : (size_t)-1;
      return cx < cy;
    });
    const size_t keep = std::max<size_t>(6, retain_chunks());  // spares survive so that the next request need not explore again
    while (have.size() > keep) {
      release_chunk(p, have.back());
      have.pop_back();
    }
  }
  p.free_chunks.swap(have);
  // ---- map every array into a fresh virtual range (second mapping of its chunks)
  for (int a = 0; a < n; ++a) {
    const size_t size = m[a] * CHUNK;
    void* va = nullptr;
    bool ok = hipMemAddressReserve(&va, size, 0, nullptr, 0) == hipSuccess;
    size_t mapped = 0;
    for (size_t i = 0; ok && i < m[a]; ++i) {
      ok = hipMemMap(static_cast<char*>(va) + i * CHUNK, CHUNK, 0, per[a][i].h, 0) == hipSuccess;
      if (ok) mapped += CHUNK;
    }
    ok = ok && set_access(va, size, dev);
    if (!ok) {  // undo everything and hand the memory back: the caller falls back to another allocator
      if (va && mapped) retire_range(va, mapped);
      for (int b = a; b < n; ++b)
        for (auto& c : per[b]) release_chunk(p, c);
      for (int b = 0; b < a; ++b) {
        auto it = g_buffers.find(ptrs[b]);
        if (it != g_buffers.end()) {
          retire_range(ptrs[b], it->second.size);
          for (auto& c : it->second.chunks) release_chunk(p, c);
          g_buffers.erase(it);
        }
        ptrs[b] = nullptr;
      }
      for (auto& c : p.free_chunks) release_chunk(p, c);
      p.free_chunks.clear();
      (void)hipGetLastError();
      return CRT_ERR_WORKSPACE;
    }
    Buffer b;
    b.size = size;
    b.dev = dev;
    b.chunks = std::move(per[a]);
    g_buffers[va] = std::move(b);
    ptrs[a] = va;
  }
  return CRT_OK;
}

int crt_hip_buffer_alloc(size_t bytes, void** ptr) {
  if (!ptr || bytes == 0) return CRT_ERR_BAD_ARG;
  return crt_hip_buffer_alloc_set(1, &bytes, ptr);
}

int crt_hip_buffer_free(void* ptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_buffers.find(ptr);
  if (it == g_buffers.end()) return CRT_ERR_BAD_ARG;
  Buffer b = std::move(it->second);
  g_buffers.erase(it);
  retire_range(ptr, b.size);
  Pool& p = g_pools[b.dev];
  for (auto& c : b.chunks) p.free_chunks.push_back(c);
  cap_pool(p, retain_chunks());  // beyond the retention cap the memory goes back to the driver now
  return CRT_OK;
}

int crt_hip_buffer_trim(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CRT_ERR_LAUNCH;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_pools.find(dev);
  if (it == g_pools.end()) return CRT_OK;
  for (auto& c : it->second.free_chunks) release_chunk(it->second, c);
  it->second.free_chunks.clear();
  return CRT_OK;
}

int crt_hip_buffer_set_retain(size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_retain = (long long)(bytes / CHUNK);
  for (auto& kv : g_pools) cap_pool(kv.second, (size_t)g_retain);
  return CRT_OK;
}

int crt_hip_buffer_describe(const void* ptr, char* buf, size_t n) {
  if (!buf || n == 0) return CRT_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_buffers.find(const_cast<void*>(ptr));
  if (it == g_buffers.end()) return CRT_ERR_BAD_ARG;
  size_t k = 0;
  for (auto& c : it->second.chunks)
    if (k + 1 < n) buf[k++] = cls_letter(c.cls);
  buf[k] = 0;
  return CRT_OK;
}

int crt_hip_buffer_stats(int64_t* out6) {
  if (!out6) return CRT_ERR_BAD_ARG;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CRT_ERR_LAUNCH;
  std::lock_guard<std::mutex> lk(g_mu);
  Pool& p = g_pools[dev];
  out6[0] = p.created;
  out6[1] = p.released;
  out6[2] = p.probes;
  out6[3] = (int64_t)(p.probe_ms_total * 1000.0);  // microseconds of GPU time spent classifying
  out6[4] = (int64_t)p.free_chunks.size();
  out6[5] = p.nref;
  return CRT_OK;
}

}  // extern "C"
