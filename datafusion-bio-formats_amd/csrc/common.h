// common.h -- shared host-side definitions (Arrow C Data Interface structs, HIP RAII, errors).
#pragma once
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <stdint.h>
#include <stdlib.h>
#include <exception>
#include <stdexcept>
#include <string>

// ---- Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html) ----
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
extern "C" {
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
}
#endif

namespace bioscan {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
  if (e != hipSuccess) {
    throw Error(std::string("HIP error in ") + what + " (" + file + ":" + std::to_string(line) + "): " + hipGetErrorString(e));
  }
}
#define HIP_CHECK(x) ::bioscan::hip_check((x), #x, __FILE__, __LINE__)

// what bioscan_last_error() returns on this thread (set by every entry point that fails)
void set_last_error(const std::string& msg);

// Environment knobs, read ONCE per process (first use).  None of them changes what a scan computes:
//   BIOSCAN_DEBUG=1            diagnostics on stderr (K1 pass counters, record-chain rounds)
//   BIOSCAN_LAPS=1             host wall-clock laps of execute() on stderr
//   BIOSCAN_K1_WAVES_PER_CU=n  persistent-grid size of K1 (default: the occupancy API's answer)
//   BIOSCAN_K1=3               K1 as of r03 (inflate_v3.hip) for every member; default 4 = inflate_v4.hip, v3 only for its retries
//   BIOSCAN_HOST_POOL_GB=x     cap of the recycled host result blocks (default 8; read at every release)
//   BIOSCAN_CHUNK_MEMBERS=n    BGZF members per pipeline chunk of a host stream (default 16384)
//   BIOSCAN_CHUNK_MEMBERS_DEVICE=n  the same for bioscan_execute_device (default 1048576; 65536 with BIOSCAN_LOOKAHEAD=1)
//   BIOSCAN_LOOKAHEAD / BIOSCAN_K1_ONESHOT / BIOSCAN_LA_PRIORITY / BIOSCAN_LA_HEAD  the look-ahead inflate (engine.cpp: BamExecState)
struct EnvKnobs {
  bool debug = false, laps = false;
  int k1_waves_per_cu = 0;
  int k1_version = 4;     // BIOSCAN_K1=3: the r02/r03 kernel (inflate_v3.hip; always the wide-table fallback of v4) for every member
  double host_pool_gb = 8.0;    // idle host result blocks kept for reuse (a shared host tolerates 8 GB; r03 kept 64)
  double dev_pool_gb = 200.0;   // BIOSCAN_DEV_POOL_GB: cap of the cached (idle) device blocks
  uint32_t chunk_members = 16384;          // BGZF members per pipeline chunk of a host stream (~1.3 GB of Arrow buffers for short reads)
  uint32_t chunk_members_device = 1u << 20;  // device-resident execution keeps every chunk in HBM anyway: large chunks, short K1 tails
  bool chunk_members_device_set = false;
  uint32_t chunk_members_lookahead = 65536;  // ... with the look-ahead inflate on: small enough to pipeline
  int lookahead = 0;      // BIOSCAN_LOOKAHEAD=1: inflate of chunk c + 1 overlapped with the other stages of chunk c (measured: no gain)
  int k1_preheaders = 0;  // BIOSCAN_K1_PREHEADERS=1: K0 (inflate_headers.hip) parses the first block header of every member ahead of K1 (measured: time-neutral)
  int k1_oneshot = 1;     // BIOSCAN_K1_ONESHOT=0: look-ahead launches keep K1's persistent grid
  int k1_bounded_wpw = 4; // BIOSCAN_K1_BOUNDED_WPW=1: one-wave workgroups in bounded launches
  int k1_slots_pct = 200; // BIOSCAN_K1_SLOTS_PCT: scratch strides of a bounded context, percent of what the device holds at once
  int k1_per_wave = 1;    // BIOSCAN_K1_PER_WAVE: members a wave of a bounded K1 launch decodes before it retires
  int la_priority = 1;    // BIOSCAN_LA_PRIORITY=0: K1's stream and the stage stream at equal priority
  uint64_t la_head = 1u << 20;  // BIOSCAN_LA_HEAD: bytes in front of a look-ahead chunk for the record cut by the previous chunk
};
const EnvKnobs& env_knobs();

// Size-keyed cache of large device allocations: a scan allocates the same column / scratch sizes for
// every partition and every step, and hipMalloc / hipFree of tens of GB per scan is driver work that
// does not belong on the hot path.  Blocks >= 1 MiB are returned here instead of to the driver and
// handed out again on an exact size match; dev_pool_trim() releases everything (provider close).
void* dev_pool_alloc(size_t bytes);
void dev_pool_free(void* p, size_t bytes, int device);
int dev_pool_device();  // current HIP device (recorded at allocation)
void dev_pool_trim();

// device buffer with value semantics off (move only)
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int dev = 0;  // device the block lives on
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), dev(o.dev) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { reset(); p = o.p; n = o.n; dev = o.dev; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { reset(); }
  void alloc(size_t count) {
    reset();
    n = count;
    size_t bytes = (count ? count : 1) * sizeof(T);
    dev = dev_pool_device();
    p = (T*)dev_pool_alloc(bytes);
  }
  void reset() {
    if (p) { dev_pool_free(p, (n ? n : 1) * sizeof(T), std::uncaught_exceptions() > 0 ? -1 : dev); p = nullptr; n = 0; }
  }
  size_t bytes() const { return n * sizeof(T); }
};

// host buffer: pinned when the driver grants it, pageable otherwise (8 ranks pinning 16 GB each can exceed the
// node's locked-memory budget; a pageable source only makes the one-off upload slower)
// Cache of host blocks (size classes with <= 25 % slack) for result buffers.  They are plain malloc'd memory: pinning
// gigabytes per execute cost far more than the PCIe copy itself (5.2 GB of Arrow buffers: 1.1-1.6 s with fresh pinned
// allocations against ~0.15 s of copy time), a fresh pageable block takes the copy at 17 GB/s and a recycled one --
// its pages already touched -- at the full link rate.
void* host_pool_alloc(size_t bytes, size_t* cap, bool* pinned, bool want_pinned = false);
void host_pool_free(void* p, size_t cap, bool pinned);
void host_pool_trim();

struct HostBuf {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  bool pinned = false, pooled = false, mapped = false;
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  HostBuf(HostBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap), pinned(o.pinned), pooled(o.pooled), mapped(o.mapped) { o.p = nullptr; o.n = 0; }
  HostBuf& operator=(HostBuf&& o) noexcept {
    if (this != &o) { reset(); p = o.p; n = o.n; cap = o.cap; pinned = o.pinned; pooled = o.pooled; mapped = o.mapped; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~HostBuf() { reset(); }
  // use_pool = false for one-off blocks that should go back to the system at once (the file image before upload)
  void alloc(size_t bytes, bool use_pool = true, bool want_pinned = false) {
    reset();
    n = bytes;
    pooled = use_pool;
    if (use_pool) {
      p = (uint8_t*)host_pool_alloc(bytes ? bytes : 1, &cap, &pinned, want_pinned);
      return;
    }
    cap = bytes ? bytes : 1;  // the file image: read once, uploaded once -- pinned memory is the faster path for that (3.3 s against 6.2 s for 16.5 GB)
    if (hipHostMalloc((void**)&p, cap, hipHostMallocDefault) == hipSuccess) { pinned = true; return; }
    (void)hipGetLastError();
    p = (uint8_t*)malloc(cap);
    pinned = false;
    if (!p) throw Error("out of host memory");
  }
  // read-only view of a file (no copy: the pages come straight from the page cache and are uploaded from there)
  bool map_file(int fd, size_t bytes) {
    reset();
    void* m = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return false;
    p = (uint8_t*)m; n = bytes; cap = bytes; mapped = true; pooled = false; pinned = false;
    return true;
  }
  void reset() {
    if (!p) return;
    if (mapped) { munmap(p, cap); p = nullptr; n = 0; cap = 0; mapped = false; return; }
    if (pooled) host_pool_free(p, cap, pinned);
    else if (pinned) (void)hipHostFree(p);
    else free(p);
    p = nullptr; n = 0; cap = 0;
  }
};

}  // namespace bioscan
