// engine.cpp -- BamTableProvider / BamExec / stream mirror over the HIP kernels + the C ABI.
//
// Mirrors (names and argument meaning) bio-format-bam/src/table_provider.rs:381-529 (new),
// :941-962 (supports_filters_pushdown), :964-1115 (scan) and bio-format-bam/src/physical_exec.rs
// :108-172 (execute), :371-598 (sequential scan), :864-1372 (indexed scan).  There is NO CPU
// decode path in this library: every inflate / record walk / field extract runs on the GPU and
// the library refuses to open a file when no HIP device is usable.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <thread>
#include <deque>
#include <condition_variable>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

#include "bam_host.h"
#include "common.h"
#include "kernels.h"
#include "bgzf_source.h"
#include "vcf_api.h"

using namespace bioscan;

static thread_local std::string g_err;

// ---- device allocation cache (see common.h) -------------------------------------------------------
namespace bioscan {
void set_last_error(const std::string& msg) { g_err = msg; }
const EnvKnobs& env_knobs() {
  static const EnvKnobs k = [] {
    EnvKnobs v;
    if (const char* e = getenv("BIOSCAN_DEBUG")) v.debug = e[0] && e[0] != '0';
    if (const char* e = getenv("BIOSCAN_LAPS")) v.laps = e[0] && e[0] != '0';
    if (const char* e = getenv("BIOSCAN_K1_WAVES_PER_CU")) v.k1_waves_per_cu = atoi(e);
    if (const char* e = getenv("BIOSCAN_HOST_POOL_GB")) v.host_pool_gb = atof(e);
    if (const char* e = getenv("BIOSCAN_DEV_POOL_GB")) v.dev_pool_gb = atof(e);
    if (const char* e = getenv("BIOSCAN_CHUNK_MEMBERS")) v.chunk_members = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("BIOSCAN_CHUNK_MEMBERS_DEVICE")) { v.chunk_members_device = (uint32_t)std::max(1, atoi(e)); v.chunk_members_device_set = true; }
    if (const char* e = getenv("BIOSCAN_LOOKAHEAD")) v.lookahead = atoi(e);
    if (const char* e = getenv("BIOSCAN_K1_ONESHOT")) v.k1_oneshot = atoi(e);
    if (const char* e = getenv("BIOSCAN_K1_PREHEADERS")) v.k1_preheaders = atoi(e);
    if (const char* e = getenv("BIOSCAN_K1")) v.k1_version = atoi(e) == 3 ? 3 : 4;
    if (v.k1_version == 4 && !getenv("BIOSCAN_K1_PREHEADERS")) v.k1_preheaders = 1;  // v4 runs 16 waves per CU: the serial header parse is worth taking out of it (-7 %)
    if (const char* e = getenv("BIOSCAN_K1_PER_WAVE")) v.k1_per_wave = atoi(e);
    if (const char* e = getenv("BIOSCAN_K1_BOUNDED_WPW")) v.k1_bounded_wpw = atoi(e);
    if (const char* e = getenv("BIOSCAN_K1_SLOTS_PCT")) v.k1_slots_pct = atoi(e);
    if (const char* e = getenv("BIOSCAN_LA_PRIORITY")) v.la_priority = atoi(e);
    if (const char* e = getenv("BIOSCAN_LA_HEAD")) v.la_head = (uint64_t)std::max(0ll, atoll(e));
    return v;
  }();
  return k;
}
// Cached (idle) device blocks, keyed by (device, size class).  Most sizes of a scan are data dependent (records of a chunk,
// column totals, selected rows), so blocks are handed out by SIZE CLASS (<= 25 % slack below 64 MiB, <= 6 % above), not by exact
// size: a partition's buffers are reused by the next partition and the next step although no two sizes repeat.  The cap
// is per device (BIOSCAN_DEV_POOL_GB, at most 60 % of the device's memory); a block that does not fit evicts the least
// recently cached blocks of ITS device first, and an allocation the driver refuses gives that device's idle blocks back
// and is tried again -- other devices' caches, and the executes running on them, are left alone.
static std::mutex g_pool_mu;
struct PoolBlock { void* p; uint64_t seq; };
static std::multimap<std::pair<int, size_t>, PoolBlock> g_pool;                  // (device, class bytes) -> idle block
static std::map<int, std::map<uint64_t, std::multimap<std::pair<int, size_t>, PoolBlock>::iterator>> g_pool_lru;  // device -> age order
static std::map<int, size_t> g_pool_bytes, g_pool_cap;
static uint64_t g_pool_seq = 0;
static int cur_device() { int d = 0; (void)hipGetDevice(&d); return d; }
static size_t dev_class(size_t bytes) {
  if (bytes <= 256) return 256;
  const size_t top = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));  // highest power of two <= bytes
  const size_t step = bytes >= ((size_t)64 << 20) ? top >> 4 : top >> 2;
  return (bytes + step - 1) / step * step;
}
static size_t dev_pool_cap(int device) {  // g_pool_mu held
  auto it = g_pool_cap.find(device);
  if (it != g_pool_cap.end()) return it->second;
  size_t cap = (size_t)(env_knobs().dev_pool_gb * (double)(1ull << 30));
  int prev = 0;
  (void)hipGetDevice(&prev);
  size_t fr = 0, total = 0;
  if (hipSetDevice(device) == hipSuccess && hipMemGetInfo(&fr, &total) == hipSuccess && total) cap = std::min(cap, total / 10 * 6);
  (void)hipSetDevice(prev);
  g_pool_cap[device] = cap;
  return cap;
}
static void pool_evict_oldest(int device, size_t need_room) {  // g_pool_mu held: frees idle blocks of `device` until `need_room` fits
  auto& lru = g_pool_lru[device];
  const size_t cap = dev_pool_cap(device);
  while (!lru.empty() && g_pool_bytes[device] + need_room > cap) {
    auto it = lru.begin()->second;
    g_pool_bytes[device] -= it->first.second;
    (void)hipFree(it->second.p);  // (waits for the device; only when the cache is over its cap)
    g_pool.erase(it);
    lru.erase(lru.begin());
  }
}
static void pool_trim_device(int device) {  // g_pool_mu held
  auto& lru = g_pool_lru[device];
  for (auto& kv : lru) { (void)hipFree(kv.second->second.p); g_pool.erase(kv.second); }
  lru.clear();
  g_pool_bytes[device] = 0;
}
void* dev_pool_alloc(size_t bytes) {
  const size_t cls = dev_class(bytes);
  const int dev = cur_device();
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool.find({dev, cls});
    if (it != g_pool.end()) {
      void* p = it->second.p;
      g_pool_lru[dev].erase(it->second.seq);
      g_pool.erase(it);
      g_pool_bytes[dev] -= cls;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cls);
  if (e != hipSuccess) {  // out of memory: give this device's idle blocks back and retry once
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      pool_trim_device(dev);
    }
    e = hipMalloc(&p, cls);
  }
  HIP_CHECK(e);
  return p;
}
void dev_pool_free(void* p, size_t bytes, int device) {
  // `device` is the one the block was allocated on (recorded by DevBuf): a block is freed from Arrow release
  // callbacks and destructors on threads whose current device may be anything.
  // device < 0: released while an exception unwinds -- kernels that read the block may still be in flight, so it goes
  // back to the driver (hipFree waits for the device) instead of into the cache where another stream could pick it up
  const size_t cls = dev_class(bytes);
  if (device >= 0) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (cls <= dev_pool_cap(device)) {
      pool_evict_oldest(device, cls);
      auto it = g_pool.emplace(std::make_pair(device, cls), PoolBlock{p, ++g_pool_seq});
      g_pool_lru[device][it->second.seq] = it;
      g_pool_bytes[device] += cls;
      return;
    }
  }
  (void)hipFree(p);  // larger than the whole cap (or unwinding): back to the driver
}
int dev_pool_device() { return cur_device(); }
void dev_pool_trim() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto& kv : g_pool) (void)hipFree(kv.second.p);  // hipFree takes a pointer of any device
  g_pool.clear();
  g_pool_lru.clear();
  g_pool_bytes.clear();
}
// ---- host block cache ----
static std::mutex g_hpool_mu;
static std::multimap<size_t, void*> g_hpool;  // class size -> host block whose pages have been touched
static size_t g_hpool_bytes = 0;
static size_t host_class(size_t bytes) {
  if (bytes <= 4096) return 4096;
  size_t top = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));  // highest power of two <= bytes
  const size_t step = top >> 2;                                                  // four classes per octave
  return (bytes + step - 1) / step * step;
}
// (read at every free, not once: the cap is the one knob a long-lived host process may want to move while it runs, and
// bench.py reports its end-to-end leg under two caps from one process)
static size_t host_pool_limit() {
  double gb = env_knobs().host_pool_gb;
  if (const char* e = getenv("BIOSCAN_HOST_POOL_GB")) gb = atof(e);
  return (size_t)(gb * (double)(1ull << 30));
}
// Two kinds of cached host blocks.  Pageable ones serve one-shot copies (a fresh pageable block takes a copy at 17 GB/s and
// a recycled one -- its pages already touched -- at the link rate, tools/experiments/d2h_paths.cpp).  Pinned ones serve the
// chunk pipeline of a stream: hipMemcpyAsync only overlaps with the next chunk's kernels when the destination is pinned,
// and pinning costs ~0.15 s per GB once, which the cache amortises (a stream recycles two or three chunk-sized blocks).
static std::multimap<size_t, void*> g_hpool_pinned;
void* host_pool_alloc(size_t bytes, size_t* cap, bool* pinned, bool want_pinned) {
  const size_t c = host_class(bytes);
  {
    std::lock_guard<std::mutex> lk(g_hpool_mu);
    auto& pool = want_pinned ? g_hpool_pinned : g_hpool;
    auto it = pool.find(c);
    if (it != pool.end()) {
      void* p = it->second;
      pool.erase(it);
      g_hpool_bytes -= c;
      *cap = c; *pinned = want_pinned;
      return p;
    }
  }
  void* p = nullptr;
  if (want_pinned) {
    if (hipHostMalloc(&p, c, hipHostMallocDefault) == hipSuccess) { *cap = c; *pinned = true; return p; }
    (void)hipGetLastError();  // the locked-memory budget is exhausted: a pageable block still works, the copy just blocks
  }
  p = malloc(c);
  if (!p) {
    host_pool_trim();
    p = malloc(c);
  }
  if (!p) throw Error("out of host memory");
  *cap = c; *pinned = false;
  return p;
}
void host_pool_free(void* p, size_t cap, bool pinned) {
  {
    std::lock_guard<std::mutex> lk(g_hpool_mu);
    if (g_hpool_bytes + cap <= host_pool_limit()) {
      (pinned ? g_hpool_pinned : g_hpool).emplace(cap, p);
      g_hpool_bytes += cap;
      return;
    }
  }
  if (pinned) (void)hipHostFree(p); else free(p);
}
void host_pool_trim() {
  std::lock_guard<std::mutex> lk(g_hpool_mu);
  for (auto& kv : g_hpool) free(kv.second);
  g_hpool.clear();
  for (auto& kv : g_hpool_pinned) (void)hipHostFree(kv.second);
  g_hpool_pinned.clear();
  g_hpool_bytes = 0;
}
}  // namespace bioscan

namespace {

// -------------------------------------------------------------------------------------------------
struct StageTimer {
  hipEvent_t a, b;
  hipStream_t st;
  explicit StageTimer(hipStream_t s) : st(s) {
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
  }
  ~StageTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
  void start() { HIP_CHECK(hipEventRecord(a, st)); }
  double stop() {
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
  }
};

// members [b_lo, b_hi) + the record-aligned window of their inflated bytes that belongs to the caller
struct DecodeRange {
  uint32_t b_lo = 0, b_hi = 0;
  uint64_t first_rel = 0, stop_rel = 0;  // relative to the range's first inflated byte
};

// -------------------------------------------------------------------------------------------------
// Provider: the file resident in HBM, its header, schema and index.  Immutable once opened -- every execute() owns
// its stream, scratch and buffers (BamExecState below), as every execute of the reference owns its reader
// (bio-format-bam/src/physical_exec.rs:878-881).
// -------------------------------------------------------------------------------------------------
struct Provider : BgzfSource {
  int kind = 0;              // 0 = BAM, 1 = FASTQ
  int fq_compression = 0;    // FASTQ: 0 = none, 1 = BGZF
  bool fq_has_gzi = false;
  std::vector<std::pair<uint64_t, uint64_t>> gzi;  // (compressed, uncompressed) per block boundary
  bool zero_based = true;
  bool binary_cigar = false;
  std::vector<std::string> tag_fields;
  bool has_tag_fields = false;
  uint32_t chunk_members = 0;  // 0 = default

  BamHeader hdr;
  std::vector<FieldDef> fields;  // full schema
  std::vector<std::pair<std::string, std::string>> metadata;

  bool has_index = false;
  std::string index_error;  // the companion index was found but is not a BAI this reader takes (e.g. a .csi): what the first indexed execute says
  std::string index_path;
  Bai bai;

  DecodeRange whole_file() const {
    DecodeRange r;
    r.b_lo = 0; r.b_hi = n_blocks();
    r.first_rel = hdr.first_record_offset; r.stop_rel = ulen;
    return r;
  }

  // The image of `dev` covering members [m_lo, m_hi), with the reference-name table beside it.  This is the only
  // provider state an execute waits for; everything else it touches is its own.
  std::shared_ptr<DeviceImage> device_image(int dev, uint32_t m_lo, uint32_t m_hi) {
    auto img = image_for(dev, m_lo, m_hi);
    std::lock_guard<std::mutex> lk(ref_mu);
    if (img->d_ref_name_off.p) return img;
    HIP_CHECK(hipSetDevice(dev));
    std::vector<uint32_t> off{0}, len;
    std::string blob;
    for (auto& n : hdr.ref_names) {
      blob += n;
      off.push_back((uint32_t)blob.size());
      len.push_back((uint32_t)n.size());
    }
    img->d_ref_names.alloc(blob.size() + 8);  // bam_rows.hip reads names 8 bytes at a time
    img->d_ref_name_len.alloc(std::max<size_t>(len.size(), 1));
    DevBuf<uint32_t> o(off.size());
    if (!blob.empty()) HIP_CHECK(hipMemcpy(img->d_ref_names.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(o.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    if (!len.empty()) HIP_CHECK(hipMemcpy(img->d_ref_name_len.p, len.data(), len.size() * 4, hipMemcpyHostToDevice));
    img->d_ref_name_off = std::move(o);  // published last: its pointer is the "ready" flag read above
    return img;
  }
  std::mutex ref_mu;
};

// One step of a partition: the records of `range` that pass `sel` (reference: one region query, one unmapped-tail scan,
// the no-coor scan, or the sequential scan of the whole file).
struct WorkItem {
  DecodeRange range;
  RowSelect sel{};
  // Further regions served by the same decode: consecutive mapped regions of a partition whose records follow each other
  // in the file (ascending, disjoint) and whose member ranges touch.  Region-major order is then file order, so one pass
  // over the union of their members with the OR of their predicates returns the same rows in the same order as one pass
  // per region -- without re-inflating the boundary members and without one small K1 launch per region.
  std::vector<RowSelect> more;
  // [begin, end) pairs of absolute inflated offsets: the BAI chunks of `sel` and of every entry of `more` (RowSelect::ch_lo / ch_n)
  std::vector<uint64_t> chunk_tab;
};

// -------------------------------------------------------------------------------------------------
// Plan
// -------------------------------------------------------------------------------------------------
struct Plan {
  Provider* prov = nullptr;
  bool has_projection = false;
  std::vector<int32_t> projection;
  std::vector<FieldDef> out_fields;
  int64_t limit = -1;
  bool indexed = false;  // partition_assignments: Some(..)
  bool empty = false;    // EmptyExec
  std::vector<PartitionAssignment> assignments;
  std::vector<Filter> residual;
  int fq_strategy = 0;  // FASTQ: 0 sequential, 1 BGZF block ranges, 2 plain byte ranges
  std::vector<std::pair<uint64_t, uint64_t>> fq_parts;  // (start, end); end = ~0 for open-ended
  // work list of each partition (BAI chunk queries + block lookups), computed on first use: re-planning it on every
  // execute cost 1-3 ms of host time per partition
  mutable std::mutex work_mu;
  mutable std::vector<std::pair<bool, std::vector<WorkItem>>> work_cache;
  std::vector<int32_t> part_device;  // HIP device of each partition (bioscan_scan_devices); empty = the provider's device
  int device_of(int partition) const { return part_device.empty() ? prov->device : part_device[(size_t)partition]; }
  int n_partitions() const {
    if (prov && prov->kind == 1) return fq_strategy == 0 ? 1 : (int)fq_parts.size();
    return empty ? 0 : (indexed ? (int)assignments.size() : 1);
  }
};

// -------------------------------------------------------------------------------------------------
// Result columns.  A Result holds the Arrow buffers of ONE CHUNK of a partition's rows.  `phase` rows of its first
// batch were produced by the previous chunk: batch b covers rows [max(0, b*bs - phase), (b+1)*bs - phase).
// -------------------------------------------------------------------------------------------------
struct Column {
  FieldDef fd;
  uint64_t n_rows = 0;
  // device
  DevBuf<uint8_t> d_values;   // fixed: 4*n ; var: bytes ; list: child bytes
  DevBuf<uint64_t> d_off64;   // n+1 (var / list)
  DevBuf<int32_t> d_off32;    // nb*(bs+1)
  DevBuf<uint64_t> d_valid;   // ceil(n/64) words
  DevBuf<uint32_t> d_len;     // scratch lengths
  DevBuf<uint64_t> d_base;    // per batch: first byte / element (copied to h_batch_base)
  uint64_t total_bytes = 0;   // var: bytes; list: elements
  // host
  HostBuf h_values, h_off32, h_valid;
  HostBuf h_batch_base_buf;            // per batch: first byte / element -- PINNED: a D2H copy into pageable memory blocks the
                                       // calling thread until everything queued before it on the copy stream is done, which
                                       // serialised a chunk's kernels behind the previous chunk's whole transfer
  const uint64_t* h_batch_base = nullptr;
  bool is_var() const { return fd.kind == AK_UTF8 || fd.kind == AK_BINARY; }
  bool is_list() const { return fd.kind >= AK_LIST_INT8; }
  uint32_t list_elem_bytes() const {
    switch (fd.kind) {
      case AK_LIST_INT8: case AK_LIST_UINT8: return 1;
      case AK_LIST_INT16: case AK_LIST_UINT16: return 2;
      default: return 4;
    }
  }
};

struct Result {
  uint64_t n_rows = 0;
  uint32_t batch_size = 8192;
  uint32_t phase = 0;
  std::vector<Column> cols;
  bool on_host = false;
  int device = 0;
  hipEvent_t copied = nullptr;  // recorded behind the chunk's D2H copies (copy stream); null once waited for
  bioscan_scan_stats stats{};
  uint64_t n_batches() const { return n_rows ? (n_rows + phase + batch_size - 1) / batch_size : 0; }
  uint64_t batch_row0(uint64_t b) const { return b ? b * batch_size - phase : 0; }
  uint64_t batch_rows(uint64_t b) const { return std::min<uint64_t>((b + 1) * (uint64_t)batch_size - phase, n_rows) - batch_row0(b); }
  ~Result() {
    if (copied) {
      int prev = 0;
      (void)hipGetDevice(&prev);
      (void)hipSetDevice(device);
      (void)hipEventSynchronize(copied);  // the copies write into host blocks that go back to the cache right after this
      (void)hipEventDestroy(copied);
      (void)hipSetDevice(prev);
    }
  }
};

static int list_elem_code(ArrowKind k) { return (int)k - (int)AK_LIST_INT8; }

static uint32_t read_err(DevBuf<uint32_t>& err, hipStream_t st) {
  uint32_t e = 0;
  HIP_CHECK(hipMemcpyAsync(&e, err.p, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return e;
}
static void throw_extract_err(uint32_t e) {
  switch (e) {
    case 0: return;
    case 2: throw Error("BAM read error: reference sequence id out of range");
    case 3: throw Error("BAM read error: invalid CIGAR op code");
    case 4: throw Error("BAM read error: malformed optional field");
    case 5: throw Error("BAM read error: duplicate optional field tag in one record");
    case 6: throw Error("Arrow error: tag value type mismatch");
    case 7: throw Error("Arrow error: tag value does not fit the column type");
    case 8: throw Error("BAM read error: invalid record (variable-length fields exceed block_size)");
    default: throw Error("BAM read error: device error " + std::to_string(e));
  }
}

// Build device-side residual filter program (record_filter.rs semantics on BamRecordFields)
static bool build_terms(const Plan& plan, std::vector<FilterTerm>* terms) {
  // returns false when some term can never pass (NULL literal in a comparison)
  for (auto& f : plan.residual) {
    int field;
    if (f.column == "chrom") field = 0;
    else if (f.column == "start") field = 1;
    else if (f.column == "end") field = 2;
    else if (f.column == "mapping_quality") field = 3;
    else if (f.column == "flags") field = 4;
    else continue;  // field not known to BamRecordFields -> passes
    FilterTerm t{};
    t.field = field;
    t.op = f.op;
    auto num = [&](const Literal& l, double* v) {
      if (l.kind == BIOSCAN_LIT_INT) { *v = (double)l.i; return true; }
      if (l.kind == BIOSCAN_LIT_FLOAT) { *v = l.f; return true; }
      return false;
    };
    auto chrom_idx = [&](const std::string& s) {
      for (size_t i = 0; i < plan.prov->hdr.ref_names.size(); i++)
        if (plan.prov->hdr.ref_names[i] == s) return (double)i;
      return -2.0;
    };
    if (f.op <= BIOSCAN_OP_GE) {
      if (f.values.size() != 1) continue;
      const Literal& l = f.values[0];
      if (l.kind == BIOSCAN_LIT_NULL) return false;
      if (field == 0) {
        if (l.kind != BIOSCAN_LIT_STR) continue;
        if (f.op != BIOSCAN_OP_EQ && f.op != BIOSCAN_OP_NE) continue;
        t.vals[0] = chrom_idx(l.s);
      } else {
        if (!num(l, &t.vals[0])) continue;
      }
      t.n_vals = 1;
    } else if (f.op == BIOSCAN_OP_BETWEEN || f.op == BIOSCAN_OP_NOT_BETWEEN) {
      if (f.values.size() != 2) continue;
      if (f.values[0].kind == BIOSCAN_LIT_NULL || f.values[1].kind == BIOSCAN_LIT_NULL) return false;
      if (field == 0) continue;  // string field has no u32/f32/f64 accessor -> passes
      if (!num(f.values[0], &t.vals[0]) || !num(f.values[1], &t.vals[1])) continue;
      t.n_vals = 2;
    } else {
      // eight literals per term; a longer list continues in the following terms (`more`)
      int n = 0;
      for (auto& l : f.values) {
        if (n == 8) { t.n_vals = 8; t.more = 1; terms->push_back(t); t.has_null = 0; t.more = 0; n = 0; }
        if (field == 0) {
          if (l.kind == BIOSCAN_LIT_NULL) t.has_null = 1;
          else if (l.kind == BIOSCAN_LIT_STR) t.vals[n++] = chrom_idx(l.s);
        } else {
          double v;
          if (num(l, &v)) t.vals[n++] = v; else t.has_null = 1;
        }
      }
      t.n_vals = n;
    }
    terms->push_back(t);
  }
  return true;
}

static uint64_t voff_to_uoff(const Provider& p, uint64_t voff) {
  uint64_t c = voff >> 16;
  auto it = std::lower_bound(p.blk_coff.begin(), p.blk_coff.end(), c);
  if (it == p.blk_coff.end() || *it != c) throw Error("BGZF seek failed: virtual offset does not address a block start");
  size_t b = it - p.blk_coff.begin();
  return p.blk_uoff[b] + (voff & 0xFFFF);
}
static size_t block_of_coff(const Provider& p, uint64_t c) {
  auto it = std::lower_bound(p.blk_coff.begin(), p.blk_coff.end(), c);
  if (it == p.blk_coff.end() || *it != c) throw Error("BGZF seek failed: virtual offset does not address a block start");
  return (size_t)(it - p.blk_coff.begin());
}


// -------------------------------------------------------------------------------------------------
// Work list of a partition (bio-format-bam/src/physical_exec.rs:864-1258): regions in assignment order.  A mapped
// region decodes the members spanned by noodles' merged chunk list of its query (reg2bins, linear-index floor); the
// no-coor scan starts behind the last placed record; the unmapped tail of a reference starts at that reference's last
// chunk end and runs until the reference changes (the executor stops it there).
// -------------------------------------------------------------------------------------------------
static DecodeRange range_from_voffs(const Provider& p, uint64_t lo, uint64_t hi, bool to_eof) {
  DecodeRange d;
  d.b_lo = (uint32_t)block_of_coff(p, lo >> 16);
  const uint64_t base = p.blk_uoff[d.b_lo];
  d.first_rel = (lo & 0xFFFF);
  if (to_eof) {
    d.b_hi = p.n_blocks();
    d.stop_rel = p.ulen - base;
  } else {
    const size_t be = block_of_coff(p, hi >> 16);
    if ((hi & 0xFFFF) == 0) { d.b_hi = (uint32_t)be; d.stop_rel = p.blk_uoff[be] - base; }
    else { d.b_hi = (uint32_t)be + 1; d.stop_rel = p.blk_uoff[be] + (hi & 0xFFFF) - base; }
  }
  return d;
}

static std::vector<WorkItem> build_work_uncached(const Plan& plan, int partition, size_t n_terms) {
  const Provider& p = *plan.prov;
  std::vector<WorkItem> items;
  if (!plan.indexed) {
    WorkItem w;
    w.range = p.whole_file();
    w.sel.mode = 0;
    items.push_back(w);
    return items;
  }
  auto ref_index = [&](const std::string& name) -> long {
    for (size_t i = 0; i < p.hdr.ref_names.size(); i++) if (p.hdr.ref_names[i] == name) return (long)i;
    return -1;
  };
  for (auto& r : plan.assignments[partition].regions) {
    WorkItem w;
    w.sel.zero_based = p.zero_based ? 1 : 0;
    w.sel.n_terms = (int32_t)n_terms;
    if (r.unmapped_tail) {
      uint64_t seek = 0;
      bool have = false;
      if (r.chrom == "*") {
        if (!(p.bai.has_no_coor && p.bai.n_no_coor > 0)) continue;
        w.sel.mode = 3;
        for (auto& rf : p.bai.refs)
          for (auto& b : rf.bins)
            for (auto& c : b.second) { seek = std::max(seek, c.second); have = true; }
      } else {
        const long ref = ref_index(r.chrom);
        if (ref < 0) throw Error("Reference '" + r.chrom + "' not found in BAM header");
        if ((size_t)ref >= p.bai.refs.size()) throw Error("Reference index " + std::to_string(ref) + " not found in BAI index");
        w.sel.mode = 2;
        w.sel.ref = (int32_t)ref;
        for (auto& b : p.bai.refs[ref].bins) for (auto& c : b.second) { seek = have ? std::max(seek, c.second) : c.second; have = true; }
        if (!have)
          for (auto& rf : p.bai.refs) if (!rf.intervals.empty()) { seek = have ? std::max(seek, rf.intervals.back()) : rf.intervals.back(); have = true; }
      }
      // Where the scan can end at the latest: a reference's records (placed-unmapped ones included) all lie in front of the
      // first record of any later reference, and the index knows where that is.  Without this bound the decode range of
      // every tail ran to the end of the file (one chunk = up to 2^20 members inflated to find a handful of reads, and a
      // device image that spans the rest of the file); the scan itself still stops where the reference changes.
      uint64_t bound = 0;
      if (w.sel.mode == 2) {
        for (size_t r2 = (size_t)w.sel.ref + 1; r2 < p.bai.refs.size() && !bound; r2++) {
          uint64_t first = ~0ull;
          for (auto& b2 : p.bai.refs[r2].bins) for (auto& c : b2.second) first = std::min(first, c.first);
          if (first != ~0ull) bound = first;
        }
      }
      if (have && seek && bound) {
        // the usual case is bound == seek: the reference's last chunk ends where the next reference begins and there is no
        // tail at all (the reference's reader would run to the end of the file looking for it and return nothing)
        if (bound <= seek) continue;
        w.range = range_from_voffs(p, seek, bound, false);
      } else {
        w.range = (have && seek) ? range_from_voffs(p, seek, 0, true) : p.whole_file();
      }
    } else {
      const long ref = ref_index(r.chrom);
      if (ref < 0) throw Error("BAM region query failed: region reference sequence does not exist in reference sequences: " + r.chrom);
      auto chunks = bai_query_chunks(p.bai, (size_t)ref, r.has_start, r.start, r.has_end, r.end);
      if (r.has_end && r.end >= 1 && (size_t)ref < p.bai.refs.size()) {
        // The BAI query has no upper bound: the bins of the coarser levels that overlap the region hold reads up to 64 Mb
        // (512 Mb) behind its end -- single-read chunks the reference seeks to, inflates and filters out one by one, and
        // which here would stretch the decode span over every member in between (config 2 cut into 8 partitions: 7.6 % more
        // members than the file has).  The first record of the next non-empty LEAF bin behind the region's last 16 kb
        // window starts behind the region's end, and so does every record after it (the file is sorted by start): chunks
        // from there on cannot hold a row of the answer.
        const BaiRef& br = p.bai.refs[(size_t)ref];
        const uint64_t w_end = (r.end - 1) >> 14;
        auto it = br.bins.upper_bound((uint32_t)std::min<uint64_t>(4681 + w_end, 37448));
        if (4681 + w_end < 37448 && it != br.bins.end() && it->first < 37449 && !it->second.empty()) {
          uint64_t V = ~0ull;
          for (auto& c : it->second) V = std::min(V, c.first);
          std::vector<std::pair<uint64_t, uint64_t>> kept;
          for (auto& c : chunks) if (c.first < V) kept.push_back({c.first, std::min(c.second, V)});
          chunks.swap(kept);
        }
      }
      if (chunks.empty()) continue;  // nothing indexed for the region
      uint64_t lo = ~0ull, hi = 0;
      for (auto& c : chunks) { lo = std::min(lo, c.first); hi = std::max(hi, c.second); }
      w.range = range_from_voffs(p, lo, hi, false);
      // the decode covers the span of the chunks; the answer only holds records that start inside one of them
      for (auto& c : chunks) {
        w.chunk_tab.push_back(p.blk_uoff[block_of_coff(p, c.first >> 16)] + (c.first & 0xFFFF));
        w.chunk_tab.push_back(p.blk_uoff[block_of_coff(p, c.second >> 16)] + (c.second & 0xFFFF));
      }
      w.sel.ch_lo = 0;
      w.sel.ch_n = (uint32_t)chunks.size();
      w.sel.mode = 1;
      w.sel.ref = (int32_t)ref;
      w.sel.start1 = r.has_start ? (int64_t)r.start : 0;
      w.sel.end1 = r.has_end ? (int64_t)r.end : INT64_MAX;
      w.sel.q_start1 = r.has_start ? (int64_t)r.start : 1;
    }
    // merge into the previous item when this region continues it in file order
    if (w.sel.mode == 1 && !items.empty() && items.back().sel.mode == 1) {
      WorkItem& a = items.back();
      const RowSelect& last = a.more.empty() ? a.sel : a.more.back();
      const bool after = w.sel.ref > last.ref || (w.sel.ref == last.ref && last.end1 != INT64_MAX && w.sel.start1 > last.end1);
      const bool touches = w.range.b_lo <= a.range.b_hi + 64 && w.range.b_lo >= a.range.b_lo && w.range.b_hi >= a.range.b_hi;
      if (after && touches) {
        const uint64_t shift = p.blk_uoff[w.range.b_lo] - p.blk_uoff[a.range.b_lo];
        a.range.stop_rel = std::max(a.range.stop_rel, w.range.stop_rel + shift);
        a.range.b_hi = w.range.b_hi;
        w.sel.ch_lo = (uint32_t)(a.chunk_tab.size() / 2);
        a.chunk_tab.insert(a.chunk_tab.end(), w.chunk_tab.begin(), w.chunk_tab.end());
        a.more.push_back(w.sel);
        continue;
      }
    }
    items.push_back(w);
  }
  return items;
}
static std::vector<WorkItem> build_work(const Plan& plan, int partition, size_t n_terms) {
  std::lock_guard<std::mutex> lk(plan.work_mu);
  const size_t np = (size_t)std::max(plan.n_partitions(), 1);
  if (plan.work_cache.size() != np) plan.work_cache.assign(np, {false, {}});
  auto& slot = plan.work_cache[(size_t)partition];
  if (!slot.first) { slot.second = build_work_uncached(plan, partition, n_terms); slot.first = true; }
  return slot.second;
}

// members spanned by a work list (what has to be resident on the device that executes it)
static void work_span(const std::vector<WorkItem>& items, uint32_t* m_lo, uint32_t* m_hi) {
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  for (auto& w : items) { lo = std::min(lo, w.range.b_lo); hi = std::max(hi, w.range.b_hi); }
  if (lo > hi) { lo = 0; hi = 0; }
  *m_lo = lo; *m_hi = hi;
}

// -------------------------------------------------------------------------------------------------
// D2H of a chunk's Arrow buffers.  The copies run on `copy_st` behind an event of the compute stream, into pinned
// blocks, so they overlap with the next chunk's kernels; finish_copy() waits for them and drops the device buffers.
// -------------------------------------------------------------------------------------------------
static void start_copy_to_host(Result& res, hipStream_t st, hipStream_t copy_st) {
  const uint64_t n = res.n_rows, nb = res.n_batches(), nwords = (n + 63) / 64;
  const uint32_t batch_size = res.batch_size;
  for (auto& col : res.cols)
    if (col.d_off32.p && n && !col.d_base.p) {  // (the two-pass extract has written the bases of the core columns already)
      col.d_base.alloc(nb);
      launch_batch_bases(col.d_off64.p, nb, batch_size, res.phase, col.d_base.p, st);
    }
  hipEvent_t ready;
  HIP_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(ready, st));
  HIP_CHECK(hipStreamWaitEvent(copy_st, ready, 0));
  (void)hipEventDestroy(ready);
  for (auto& col : res.cols) {
    if (col.d_values.p && n) {
      uint64_t bytes = col.is_var() ? col.total_bytes : col.is_list() ? col.total_bytes * col.list_elem_bytes() : n * 4;
      col.h_values.alloc(std::max<uint64_t>(bytes, 1), true, true);
      if (bytes) HIP_CHECK(hipMemcpyAsync(col.h_values.p, col.d_values.p, bytes, hipMemcpyDeviceToHost, copy_st));
    }
    if (col.d_off32.p && n) {
      uint64_t bytes = nb * ((uint64_t)batch_size + 1) * 4;
      col.h_off32.alloc(bytes, true, true);
      HIP_CHECK(hipMemcpyAsync(col.h_off32.p, col.d_off32.p, bytes, hipMemcpyDeviceToHost, copy_st));
      col.h_batch_base_buf.alloc(std::max<uint64_t>(nb, 1) * 8, true, true);
      col.h_batch_base = (const uint64_t*)col.h_batch_base_buf.p;
      HIP_CHECK(hipMemcpyAsync(col.h_batch_base_buf.p, col.d_base.p, nb * 8, hipMemcpyDeviceToHost, copy_st));
    }
    if (col.d_valid.p && n) {
      col.h_valid.alloc(nwords * 8 + 8, true, true);
      HIP_CHECK(hipMemcpyAsync(col.h_valid.p, col.d_valid.p, nwords * 8, hipMemcpyDeviceToHost, copy_st));
    }
  }
  HIP_CHECK(hipEventCreateWithFlags(&res.copied, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(res.copied, copy_st));
}
static void finish_copy(Result& res) {
  if (res.copied) {
    HIP_CHECK(hipEventSynchronize(res.copied));
    (void)hipEventDestroy(res.copied);
    res.copied = nullptr;
  }
  for (auto& col : res.cols) {
    col.d_values.reset(); col.d_off64.reset(); col.d_off32.reset(); col.d_valid.reset(); col.d_base.reset();
  }
  res.on_host = true;
}
static void copy_result_to_host(Result& res, hipStream_t st) {  // one-shot form (FASTQ): same stream, waited for at once
  start_copy_to_host(res, st, st);
  finish_copy(res);
}

// -------------------------------------------------------------------------------------------------
// BamExecState: the chunk pipeline of ONE execute() -- the analogue of the reference's per-partition reader + builders
// (bio-format-bam/src/physical_exec.rs:857-862: one batch in flight per partition).  The partition's work items are
// streamed in chunks of `chunk_members` BGZF members: inflate + CRC32, record chain (the record that straddles the end
// of a chunk is carried over as bytes), row selection, field extract into the chunk's own Arrow buffers, D2H on a second
// stream while the next chunk runs.  HBM footprint: two inflated chunk buffers + one chunk of Arrow buffers per result
// in flight -- O(chunk), independent of the file size.
// -------------------------------------------------------------------------------------------------
// what a Stream polls: the next chunk of rows of a partition (never an empty one), nullptr at its end
struct ChunkProducer {
  bioscan_scan_stats total{};
  virtual std::shared_ptr<Result> next_chunk() = 0;
  virtual ~ChunkProducer() {}
};
struct BamExecState : ChunkProducer {
  const Plan& plan;
  Provider& p;
  const uint32_t batch_size;
  const bool to_host;
  const uint32_t chunk_members;
  // A host stream starts with small chunks and doubles them up to chunk_members: the first batch leaves after ~2048 members'
  // worth of work and PCIe time instead of a whole chunk's (the reference hands out its first batch after 8192 records);
  // only when the chunk size was not set by the caller (tests drive exact chunk sizes).
  const bool ramp;
  uint32_t chunks_done = 0;
  uint32_t item_chunks = 0;  // chunks of the current work item so far
  bool item_is_tail = false; // an unmapped-tail scan: usually over within the first member
  uint32_t chunk_len(uint32_t k) const {
    uint64_t c = chunk_members;
    if (ramp) c = std::min<uint64_t>(c, 2048ull << std::min<uint32_t>(k, 16));
    // a tail scan ends where the reference changes: 4, 16, 64, ... members instead of a whole chunk to find that out
    if (item_is_tail) c = std::min<uint64_t>(c, 4ull << std::min<uint32_t>(2 * (item_chunks + (k - chunks_done)), 24));
    return (uint32_t)std::max<uint64_t>(c, 1);
  }
  std::shared_ptr<DeviceImage> img;  // this partition's device: resident members + tables + reference names
  K1Ctx k1;
  hipStream_t st = nullptr, copy_st = nullptr;
  // Look-ahead inflate: K1 (vector-issue bound) of chunk c + 1 runs on the context's own low-priority stream while the
  // HBM-bound stages of chunk c (CRC32, record chain, row selection, extract) run on `st`; K1 is launched one wave per
  // member (one-shot) so that retiring waves leave room on every CU for those stages.  The inflate destination of a buffer
  // starts `la_head` bytes in: the record cut by the previous chunk's end is copied in front of it afterwards.
  const bool la;
  bool own_st = false;
  struct Pending { bool valid = false; uint32_t m0 = 0, m1 = 0; hipEvent_t t0 = nullptr, t1 = nullptr; } pend[2];
  DevBuf<uint32_t> la_status[2];
  uint64_t data_off[2] = {0, 0};  // where the inflated chunk starts inside ubuf[k]
  std::vector<WorkItem> items;
  size_t item = 0;
  // the region predicates and the BAI chunk table of a region item on the device: constant for the item, uploaded when its
  // first chunk asks for them (r03 re-uploaded both from pageable memory for every chunk)
  DevBuf<RowSelect> d_sels;
  DevBuf<uint64_t> d_chunks;
  size_t sels_item = (size_t)-1;
  int n_sels = 0;
  // position inside the current item
  bool item_open = false;
  uint32_t next_member = 0;
  uint64_t skip = 0;        // bytes of the next chunk in front of its first record (BAM header / chunk offset of the range start)
  uint64_t consumed = 0;    // inflated bytes of the item's range handed to chunks so far
  uint64_t carry_len = 0;   // bytes of a record cut by the previous chunk's end, at the head of ubuf[cur]
  bool tail_seen = false, tail_done = false;
  DevBuf<uint8_t> ubuf[2];
  int cur = 0;
  uint64_t rows_emitted = 0;
  bool satisfiable = true;
  std::vector<FilterTerm> terms;
  DevBuf<FilterTerm> d_terms;
  DevBuf<uint16_t> d_tags;
  std::chrono::steady_clock::time_point wall0 = std::chrono::steady_clock::now();

  BamExecState(const Plan& pl, int partition, uint32_t bs, bool host)
      : plan(pl), p(*pl.prov), batch_size(bs), to_host(host),
        chunk_members(pl.prov->chunk_members ? pl.prov->chunk_members : host ? env_knobs().chunk_members
                      : (env_knobs().lookahead && !env_knobs().chunk_members_device_set) ? env_knobs().chunk_members_lookahead : env_knobs().chunk_members_device),
        ramp(host && !pl.prov->chunk_members), la(env_knobs().lookahead != 0) {
    satisfiable = build_terms(plan, &terms);
    items = build_work(plan, partition, terms.size());
    if (!satisfiable) items.clear();
    uint32_t m_lo = 0, m_hi = 0;
    work_span(items, &m_lo, &m_hi);
    img = p.device_image(plan.device_of(partition), m_lo, m_hi);
    HIP_CHECK(hipSetDevice(img->device));
    if (la) {
      int least = 0, greatest = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
      const bool prio = env_knobs().la_priority != 0;
      HIP_CHECK(hipStreamCreateWithPriority(&k1.stream, hipStreamNonBlocking, prio ? least : 0));
      HIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio ? greatest : 0));
      own_st = true;
      for (auto& q : pend) { HIP_CHECK(hipEventCreate(&q.t0)); HIP_CHECK(hipEventCreate(&q.t1)); }
    }
    p.init_ctx(k1, *img, std::min<uint32_t>(chunk_members, std::max<uint32_t>(m_hi - m_lo, 1)), la && env_knobs().k1_oneshot != 0);
    if (!la) st = k1.stream;
    else HIP_CHECK(hipStreamSynchronize(k1.stream));  // (the slot flags are zeroed on K1's stream)
    // (the D2H copies of a chunk are blit kernels on this ROCm -- no SDMA transfer shows in a memory-copy trace -- with a tiny
    // footprint: they run beside K1's persistent grid at the full link rate; stream priorities and a smaller K1 grid were
    // measured and change nothing)
    if (to_host) HIP_CHECK(hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking));
    d_terms.alloc(std::max<size_t>(terms.size(), 1));
    if (!terms.empty()) HIP_CHECK(hipMemcpyAsync(d_terms.p, terms.data(), terms.size() * sizeof(FilterTerm), hipMemcpyHostToDevice, st));
    const int nt = (int)p.tag_fields.size();
    if (nt) {
      std::vector<uint16_t> tg(nt);
      for (int k = 0; k < nt; k++) tg[k] = (uint16_t)((uint8_t)p.tag_fields[k][0] | ((uint16_t)(uint8_t)p.tag_fields[k][1] << 8));
      d_tags.alloc(nt);
      HIP_CHECK(hipMemcpyAsync(d_tags.p, tg.data(), nt * 2, hipMemcpyHostToDevice, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));  // the staging vectors above go out of scope
  }
  ~BamExecState() override {
    // every stream of this execute is idle before any of its buffers (members declared after `k1` are destroyed first)
    // goes back to the shared pool -- also when an exception ended a chunk between a launch and its sync
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (img) (void)hipSetDevice(img->device);
    if (k1.stream) (void)hipStreamSynchronize(k1.stream);
    if (st && st != k1.stream) (void)hipStreamSynchronize(st);
    if (copy_st) {
      (void)hipStreamSynchronize(copy_st);
      (void)hipStreamDestroy(copy_st);
    }
    if (own_st && st) (void)hipStreamDestroy(st);
    for (auto& q : pend) {
      if (q.t0) (void)hipEventDestroy(q.t0);
      if (q.t1) (void)hipEventDestroy(q.t1);
    }
    (void)hipSetDevice(prev);
  }
  bool dead = false;  // an error ended a chunk: the stream must not be polled again

  // look-ahead: K1 of members [m0, m1) into buffer k (on K1's stream; returns at once)
  void launch_ahead(int k, uint32_t m0, uint32_t m1) {
    const uint64_t bytes = p.blk_uoff[m1] - p.blk_uoff[m0];
    const uint64_t head = env_knobs().la_head;
    if (ubuf[k].n < head + bytes + 64) ubuf[k].alloc(head + bytes + 64);
    data_off[k] = head;
    if (la_status[k].n < m1 - m0) la_status[k].alloc(std::max<uint32_t>(m1 - m0, 1));
    Pending& q = pend[k];
    HIP_CHECK(hipEventRecord(q.t0, k1.stream));
    p.launch_inflate_to(k1, *img, ubuf[k].p + head, m1 - m0, m0, la_status[k].p);
    HIP_CHECK(hipEventRecord(q.t1, k1.stream));
    q.valid = true; q.m0 = m0; q.m1 = m1;
  }

  // Next chunk of rows of the partition (never an empty one), or nullptr when the partition is exhausted.
  std::shared_ptr<Result> next_chunk() override {
    if (dead) throw Error("the stream ended with an error");
    HIP_CHECK(hipSetDevice(img->device));
    struct Guard { bool* d; bool ok = false; ~Guard() { if (!ok) *d = true; } } guard{&dead};
    auto r = next_chunk_impl();
    guard.ok = true;
    return r;
  }
  std::shared_ptr<Result> next_chunk_impl() {
    for (;;) {
      if (!item_open) {
        if (item >= items.size()) return nullptr;
        const WorkItem& w = items[item];
        next_member = w.range.b_lo;
        skip = w.range.first_rel;
        consumed = 0;
        carry_len = 0;
        tail_seen = false;
        tail_done = false;
        item_chunks = 0;
        item_is_tail = w.sel.mode == 2;
        item_open = true;
        if (w.range.b_hi <= w.range.b_lo || w.range.first_rel >= w.range.stop_rel) { item_open = false; item++; continue; }
      }
      auto res = run_chunk(items[item]);
      if (!item_open) item++;
      if (res && res->n_rows) return res;
    }
  }

  std::shared_ptr<Result> run_chunk(const WorkItem& w) {
    StageTimer t(st);
    bioscan_scan_stats s{};
    // ---- members of this chunk; the item ends with the member that holds stop_rel ----
    const uint32_t m0 = next_member, m1 = std::min<uint32_t>(w.range.b_hi, m0 + chunk_len(chunks_done));
    const uint64_t range_u0 = p.blk_uoff[w.range.b_lo];
    const uint64_t chunk_bytes = p.blk_uoff[m1] - p.blk_uoff[m0];
    const bool last = m1 == w.range.b_hi || consumed + chunk_bytes >= w.range.stop_rel;
    const uint64_t take = last ? w.range.stop_rel - consumed : chunk_bytes;  // bytes of the chunk inside the record window
    uint8_t* u = ubuf[cur].p;
    s.n_blocks = m1 - m0;
    s.compressed_bytes = p.blk_coff[m1] - p.blk_coff[m0];
    s.inflated_bytes = chunk_bytes;
    if (!la) {
      if (ubuf[cur].n < carry_len + chunk_bytes + 64) {
        DevBuf<uint8_t> g(carry_len + chunk_bytes + 64);
        if (carry_len) HIP_CHECK(hipMemcpyAsync(g.p, ubuf[cur].p, carry_len, hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        ubuf[cur] = std::move(g);
        u = ubuf[cur].p;
      }
      t.start();
      p.launch_inflate(k1, *img, u + carry_len, m1 - m0, m0);
      s.ms_inflate = t.stop();
      p.report_k1_debug(k1.ctr.p, m1 - m0);
      // CRC32 validation (noodles-bgzf checks every block)
      t.start();
      p.launch_crc(k1, *img, u + carry_len, m1 - m0, m0);
      s.ms_crc = t.stop();
      p.check_inflate_status(k1, m0, m1 - m0);
    } else {
      // this chunk's inflate is in flight since the previous chunk (or starts now: first chunk of an item); the next
      // chunk's is queued behind it before this chunk's other stages begin
      if (!(pend[cur].valid && pend[cur].m0 == m0 && pend[cur].m1 == m1)) {
        if (pend[cur].valid) { HIP_CHECK(hipStreamSynchronize(k1.stream)); pend[cur].valid = false; }
        if (carry_len) throw Error("internal: a carried record without a look-ahead buffer");
        launch_ahead(cur, m0, m1);
      }
      if (!last) {
        const uint32_t m2 = std::min<uint32_t>(w.range.b_hi, m1 + chunk_len(chunks_done + 1));
        launch_ahead(cur ^ 1, m1, m2);
      }
      HIP_CHECK(hipStreamWaitEvent(st, pend[cur].t1, 0));
      u = ubuf[cur].p + data_off[cur] - carry_len;
      t.start();
      p.launch_crc_on(*img, u + carry_len, m1 - m0, m0, la_status[cur].p, st);
      s.ms_crc = t.stop();   // (waits for this chunk's inflate as well: the events of K1's stream are complete now)
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, pend[cur].t0, pend[cur].t1));
      s.ms_inflate = ms;
      pend[cur].valid = false;
      p.report_k1_debug(k1.ctr.p, m1 - m0);
      p.check_inflate_status_on(la_status[cur].p, st, m0, m1 - m0);
    }
    (void)range_u0;

    // ---- record chain over [0, L): records starting in [first_rec, L); a record cut by L is carried ----
    t.start();
    const uint64_t L = carry_len + take;
    uint64_t n_rec = 0, end_of_records = L;
    DevBuf<uint64_t> rec_off;
    if (skip >= take + carry_len) {
      skip -= take;  // still inside the BAM header
    } else {
      const uint64_t first_rec = skip;
      skip = 0;
      const uint64_t nseg = std::max<uint64_t>((L + SEG_BYTES - 1) / SEG_BYTES, 1);
      DevBuf<uint64_t> entry(nseg), exit_(nseg), base(nseg + 1), tmp(scan_tmp_elems(nseg));
      DevBuf<uint32_t> count(nseg), dirty(nseg), ctr(2), starts(nseg * (uint64_t)SEG_SLOT);
      HIP_CHECK(hipMemsetAsync(ctr.p, 0, 8, st));
      HIP_CHECK(hipMemsetAsync(dirty.p, 0, nseg * 4, st));
      ChainBuffers cb{entry.p, exit_.p, count.p, dirty.p, ctr.p, ctr.p + 1, starts.p};
      const int partial = last ? 0 : 1;
      launch_seg_guess(u, L, first_rec, nseg, (int32_t)p.hdr.ref_names.size(), cb, st);
      launch_seg_walk(u, L, nseg, cb, 0, partial, st);
      for (int iter = 0;; iter++) {
        s.chain_iterations = (uint64_t)iter + 1;
        HIP_CHECK(hipMemsetAsync(ctr.p, 0, 4, st));
        launch_seg_verify(L, first_rec, nseg, cb, st);
        uint32_t nfix = 0;
        HIP_CHECK(hipMemcpyAsync(&nfix, ctr.p, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (env_knobs().debug) fprintf(stderr, "[bioscan] record chain verify round %d: %u segment(s) corrected of %llu\n", iter, nfix, (unsigned long long)nseg);
        if (nfix == 0) break;
        if ((uint64_t)iter > nseg + 2) throw Error("record boundary scan did not converge");
        launch_seg_walk(u, L, nseg, cb, 1, partial, st);
      }
      launch_exclusive_scan_u32_to_u64(count.p, base.p, nseg, tmp.p, st);
      DevBuf<unsigned long long> lastx(2);
      launch_last_exit(exit_.p, nseg, lastx.p, st);
      uint64_t total_rec = 0;
      unsigned long long lx[2] = {0, 0};
      HIP_CHECK(hipMemcpyAsync(&total_rec, base.p + nseg, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(lx, lastx.p, 16, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      n_rec = total_rec;
      rec_off.alloc(n_rec + 1);
      launch_seg_gather(nseg, cb, base.p, rec_off.p, st);
      uint32_t errf = 0;
      HIP_CHECK(hipMemcpyAsync(&errf, ctr.p + 1, 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      const uint64_t lastv = lx[1];
      if (errf || lastv == SEG_BAD) throw Error("BAM read error: truncated or corrupt record (invalid block_size)");
      if (lastv != SEG_NONE && (lastv & SEG_PARTIAL)) end_of_records = lastv & ~SEG_PARTIAL;
      else if (lastv == SEG_NONE) end_of_records = first_rec < L ? first_rec : L;  // no complete record: everything is carried
      else end_of_records = lastv;
      if (last && first_rec < L && end_of_records != L) throw Error("BAM read error: unexpected end of record stream");
      if (end_of_records > L) throw Error("BAM read error: truncated or corrupt record (invalid block_size)");
    }
    s.ms_chain = t.stop();
    s.n_records = n_rec;

    // ---- rows of this chunk ----
    t.start();
    DevBuf<uint64_t> rows_owned;
    const uint64_t* rows = rec_off.p;
    uint64_t n_rows = n_rec;
    bool stop_item = last;
    if ((w.sel.mode == 1 || w.sel.mode == 3) && n_rec) {
      // region / no-coor items: one pass over the records decides every region of the decode (no key table)
      if (sels_item != item) {
        std::vector<RowSelect> sels;
        sels.push_back(w.sel);
        for (auto& extra : w.more) sels.push_back(extra);
        n_sels = (int)sels.size();
        d_sels.alloc(sels.size());
        d_chunks.alloc(std::max<size_t>(w.chunk_tab.size(), 1));
        HIP_CHECK(hipMemcpyAsync(d_sels.p, sels.data(), sels.size() * sizeof(RowSelect), hipMemcpyHostToDevice, st));
        if (!w.chunk_tab.empty()) HIP_CHECK(hipMemcpyAsync(d_chunks.p, w.chunk_tab.data(), w.chunk_tab.size() * 8, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));   // (the staging vector goes out of scope)
        sels_item = item;
      }
      DevBuf<uint32_t> kerr(1), keep(n_rec);
      DevBuf<uint64_t> kscan(n_rec + 1), tmp(scan_tmp_elems(n_rec));
      HIP_CHECK(hipMemsetAsync(kerr.p, 0, 4, st));
      // u[0] is the first carried byte: the bytes of member m0 start carry_len further on
      launch_row_flags_rec(u, rec_off.p, n_rec, d_sels.p, n_sels, d_terms.p, keep.p, kerr.p, st,
                           w.chunk_tab.empty() ? nullptr : d_chunks.p, p.blk_uoff[m0] - carry_len);
      launch_exclusive_scan_u32_to_u64(keep.p, kscan.p, n_rec, tmp.p, st);
      uint64_t tsel = 0;
      uint32_t e8 = 0;
      HIP_CHECK(hipMemcpyAsync(&tsel, kscan.p + n_rec, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(&e8, kerr.p, 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      throw_extract_err(e8);
      n_rows = tsel;
      rows_owned.alloc(std::max<uint64_t>(n_rows, 1));
      if (n_rows) launch_compact_rows(rec_off.p, keep.p, kscan.p, n_rec, rows_owned.p, 0, st);
      HIP_CHECK(hipStreamSynchronize(st));  // keep / kscan are released at the end of this scope
      rows = rows_owned.p;
    } else if (w.sel.mode != 0 && n_rec) {
      DevBuf<int32_t> k_refid(n_rec), k_pos(n_rec), k_end1(n_rec);
      DevBuf<uint32_t> k_fm(n_rec);
      RecKeys rk{k_refid.p, k_pos.p, k_end1.p, k_fm.p};
      StageTimer tk(st);
      tk.start();
      DevBuf<uint32_t> kerr(1);
      HIP_CHECK(hipMemsetAsync(kerr.p, 0, 4, st));
      launch_rec_keys(u, rec_off.p, n_rec, rk, kerr.p, st);
      s.ms_keys = tk.stop();
      throw_extract_err(read_err(kerr, st));
      RowSelect sel = w.sel;
      bool any = true;
      if (sel.mode == 2) {
        // unmapped tail of a reference (physical_exec.rs:1036-1135): from the first record of the reference at / after the
        // seek position until the reference changes; chunks before the first such record hold nothing, the chunk in which
        // the reference changes is the last one
        DevBuf<unsigned long long> d_idx(1);
        unsigned long long h = n_rec;
        uint64_t first = 0;
        if (!tail_seen) {
          HIP_CHECK(hipMemcpyAsync(d_idx.p, &h, 8, hipMemcpyHostToDevice, st));
          launch_find_first(k_refid.p, n_rec, 0, sel.ref, 1, d_idx.p, st);
          HIP_CHECK(hipMemcpyAsync(&h, d_idx.p, 8, hipMemcpyDeviceToHost, st));
          HIP_CHECK(hipStreamSynchronize(st));
          first = h;
          if (first < n_rec) tail_seen = true; else any = false;
        }
        uint64_t lastr = n_rec;
        if (tail_seen && any) {
          h = n_rec;
          HIP_CHECK(hipMemcpyAsync(d_idx.p, &h, 8, hipMemcpyHostToDevice, st));
          launch_find_first(k_refid.p, n_rec, first, sel.ref, 0, d_idx.p, st);
          HIP_CHECK(hipMemcpyAsync(&h, d_idx.p, 8, hipMemcpyDeviceToHost, st));
          HIP_CHECK(hipStreamSynchronize(st));
          lastr = h;
          if (lastr < n_rec) stop_item = true;  // the reference changed: the scan ends here
        }
        sel.i_lo = first;
        sel.i_hi = lastr;
      }
      n_rows = 0;
      if (any) {
        DevBuf<uint32_t> keep(n_rec);
        DevBuf<uint64_t> kscan(n_rec + 1), tmp(scan_tmp_elems(n_rec));
        launch_row_flags(rk, n_rec, sel, d_terms.p, keep.p, 0, st);
        for (auto& extra : w.more) launch_row_flags(rk, n_rec, extra, d_terms.p, keep.p, 1, st);  // disjoint regions: OR
        launch_exclusive_scan_u32_to_u64(keep.p, kscan.p, n_rec, tmp.p, st);
        uint64_t tsel = 0;
        HIP_CHECK(hipMemcpyAsync(&tsel, kscan.p + n_rec, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        n_rows = tsel;
        rows_owned.alloc(std::max<uint64_t>(n_rows, 1));
        if (n_rows) launch_compact_rows(rec_off.p, keep.p, kscan.p, n_rec, rows_owned.p, 0, st);
        HIP_CHECK(hipStreamSynchronize(st));  // keep / kscan are released at the end of this scope
      }
      rows = rows_owned.p;
    }
    s.ms_select = t.stop();

    // ---- field extract -> the chunk's Arrow buffers ----
    std::shared_ptr<Result> res;
    if (n_rows) {
      t.start();
      res = extract(u, rows, n_rows, (uint32_t)(rows_emitted % batch_size), &s);
      s.ms_extract = t.stop();
      rows_emitted += n_rows;
    }
    s.n_rows = n_rows;
    s.ms_total_gpu = s.ms_inflate + s.ms_crc + s.ms_chain + s.ms_select + s.ms_extract;
    accumulate(s);
    if (res) {
      res->stats = s;
      if (to_host) start_copy_to_host(*res, st, copy_st);
    }

    // ---- advance: carry the cut record into the other buffer ----
    consumed += chunk_bytes;
    next_member = m1;
    if (stop_item) {
      item_open = false;
      carry_len = 0;
      if (la && pend[cur ^ 1].valid) {   // (a tail item ended before its range did: the look-ahead is not needed)
        HIP_CHECK(hipStreamSynchronize(k1.stream));
        pend[cur ^ 1].valid = false;
      }
    } else {
      const uint64_t c = L - end_of_records;
      const int nxt = cur ^ 1;
      const uint64_t next_bytes = p.blk_uoff[std::min<uint32_t>(w.range.b_hi, m1 + chunk_len(chunks_done + 1))] - p.blk_uoff[m1];
      if (!la) {
        if (ubuf[nxt].n < c + next_bytes + 64) ubuf[nxt].alloc(c + next_bytes + 64);
        if (c) HIP_CHECK(hipMemcpyAsync(ubuf[nxt].p, u + end_of_records, c, hipMemcpyDeviceToDevice, st));
      } else if (c <= data_off[nxt]) {
        if (c) HIP_CHECK(hipMemcpyAsync(ubuf[nxt].p + data_off[nxt] - c, u + end_of_records, c, hipMemcpyDeviceToDevice, st));
      } else {
        // the cut record is longer than the head room in front of the next chunk's bytes: wait for that inflate and move
        // both into a buffer that holds them
        HIP_CHECK(hipStreamSynchronize(k1.stream));
        DevBuf<uint8_t> g(c + next_bytes + 64);
        HIP_CHECK(hipMemcpyAsync(g.p, u + end_of_records, c, hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipMemcpyAsync(g.p + c, ubuf[nxt].p + data_off[nxt], next_bytes, hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        ubuf[nxt] = std::move(g);
        data_off[nxt] = c;
      }
      carry_len = c;
      cur = nxt;
    }
    chunks_done++;
    item_chunks++;
    // every kernel that reads this chunk's scratch (record table, keys, row list) has to be done before the scratch
    // is released at the end of this scope; the Arrow buffers live on in `res`
    HIP_CHECK(hipStreamSynchronize(st));
    return res;
  }

  void accumulate(const bioscan_scan_stats& s) {
    total.n_blocks += s.n_blocks; total.compressed_bytes += s.compressed_bytes; total.inflated_bytes += s.inflated_bytes;
    total.arrow_bytes += s.arrow_bytes; total.n_records += s.n_records; total.n_rows += s.n_rows;
    total.ms_inflate += s.ms_inflate; total.ms_crc += s.ms_crc; total.ms_chain += s.ms_chain; total.ms_keys += s.ms_keys;
    total.ms_select += s.ms_select; total.ms_extract += s.ms_extract; total.ms_total_gpu += s.ms_total_gpu;
    total.chain_iterations = std::max(total.chain_iterations, s.chain_iterations);
    total.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  }

  // Arrow column buffers (device) of `n` rows whose records start at u + rows[i]
  std::shared_ptr<Result> extract(const uint8_t* u, const uint64_t* rows, uint64_t n, uint32_t phase, bioscan_scan_stats* s) {
    auto res = std::make_shared<Result>();
    res->batch_size = batch_size;
    res->phase = phase;
    res->device = img->device;
    res->n_rows = n;
    const uint64_t nwords = (n + 63) / 64;
    const uint64_t nb = res->n_batches();
    res->cols.resize(plan.out_fields.size());
    for (size_t c = 0; c < plan.out_fields.size(); c++) {
      res->cols[c].fd = plan.out_fields[c];
      res->cols[c].n_rows = n;
    }
    // map core columns (a projection never repeats a column: bioscan_scan rejects duplicates)
    int core_col[12];
    for (int k = 0; k < 12; k++) core_col[k] = -1;
    std::vector<std::pair<int, int>> tag_cols;  // (output col, tag index)
    for (size_t c = 0; c < plan.out_fields.size(); c++) {
      int src = plan.has_projection ? plan.projection[c] : (int)c;
      if (src < 12) { if (core_col[src] < 0) core_col[src] = (int)c; }
      else tag_cols.emplace_back((int)c, src - 12);
    }
    DevBuf<uint32_t> err(1);
    HIP_CHECK(hipMemsetAsync(err.p, 0, 4, st));
    uint64_t arrow_bytes = 0;
    if (n >= 0xFFFF0000ull) throw Error("chunk holds too many rows: lower chunk_members");
    RowsCols rc{};
    auto fixed = [&](int idx) -> uint32_t* {
      if (core_col[idx] < 0) return nullptr;
      Column& col = res->cols[core_col[idx]];
      col.d_values.alloc(n * 4);
      arrow_bytes += n * 4;
      return (uint32_t*)col.d_values.p;
    };
    auto valid = [&](int idx) -> uint64_t* {
      if (core_col[idx] < 0) return nullptr;
      Column& col = res->cols[core_col[idx]];
      col.d_valid.alloc(nwords);
      arrow_bytes += nwords * 8;
      return col.d_valid.p;
    };
    rc.start = fixed(2); rc.end = fixed(3); rc.flags = fixed(4); rc.mapq = fixed(6); rc.mate_start = fixed(8);
    rc.tlen = (int32_t*)fixed(11);
    rc.v_chrom = valid(1); rc.v_start = valid(2); rc.v_end = valid(3); rc.v_mate_chrom = valid(7); rc.v_mate_start = valid(8);
    const int var_idx[6] = {0, 1, 5, 7, 9, 10};  // name, chrom, cigar, mate_chrom, sequence, quality_scores
    for (int k = 0; k < 6; k++) if (core_col[var_idx[k]] >= 0) rc.want |= 1u << k;
    // pass 1: fixed columns, validity, tile sums of the variable-length columns (+ their scan)
    const uint64_t n_tiles = (n + ROWS_TILE - 1) / ROWS_TILE;
    DevBuf<uint64_t> tile_sums(bam_rows_scratch_elems(n));
    launch_bam_rows_pass1(u, rows, n, rc, img->d_ref_name_len.p, (int32_t)p.hdr.ref_names.size(), p.zero_based ? 1 : 0,
                          p.binary_cigar ? 1 : 0, tile_sums.p, err.p, st);
    uint64_t totals[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 6; k++)
      if ((rc.want >> k) & 1u) HIP_CHECK(hipMemcpyAsync(&totals[k], tile_sums.p + (uint64_t)k * (n_tiles + 1) + n_tiles, 8, hipMemcpyDeviceToHost, st));
    throw_extract_err(read_err(err, st));  // synchronises: the totals are here as well
    for (int k = 0; k < 6; k++) {
      if (!((rc.want >> k) & 1u)) continue;
      Column& col = res->cols[core_col[var_idx[k]]];
      col.total_bytes = totals[k];
      col.d_values.alloc(std::max<uint64_t>(totals[k], 1));
      col.d_off32.alloc(nb * ((uint64_t)batch_size + 1));
      col.d_base.alloc(nb);
      rc.val[k] = col.d_values.p; rc.off32[k] = col.d_off32.p; rc.base[k] = col.d_base.p;
      arrow_bytes += totals[k] + nb * ((uint64_t)batch_size + 1) * 4;
    }
    // pass 2: per-batch offsets + every variable-length byte
    DevBuf<uint32_t> wide(1);
    HIP_CHECK(hipMemsetAsync(wide.p, 0, 4, st));
    launch_bam_rows_pass2(u, rows, n, rc, img->d_ref_names.p, img->d_ref_name_off.p, img->d_ref_name_len.p, (int32_t)p.hdr.ref_names.size(),
                          p.binary_cigar ? 1 : 0, batch_size, phase, tile_sums.p, wide.p, st);
    DevBuf<uint64_t> tmp(scan_tmp_elems(n));
    auto finish_var = [&](Column& col, uint32_t elem_bytes) {
      col.d_off64.alloc(n + 1);
      launch_exclusive_scan_u32_to_u64(col.d_len.p, col.d_off64.p, n, tmp.p, st);
      uint64_t tot = 0;
      HIP_CHECK(hipMemcpyAsync(&tot, col.d_off64.p + n, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      col.total_bytes = tot;
      col.d_values.alloc(std::max<uint64_t>(tot * elem_bytes, 1));
      col.d_off32.alloc(nb * ((uint64_t)batch_size + 1));
      col.d_base.reset();  // start_copy_to_host derives the batch bases of such a column from its offsets
      launch_batch_offsets(col.d_off64.p, n, batch_size, phase, col.d_off32.p, st);
      arrow_bytes += tot * elem_bytes + nb * ((uint64_t)batch_size + 1) * 4;
    };
    if (core_col[10] >= 0) {
      if (read_err(wide, st)) {
        // exact path for qualities >= 95 (two-byte UTF-8 chars): the column is redone with its own length pass
        Column& col = res->cols[core_col[10]];
        arrow_bytes -= col.total_bytes + nb * ((uint64_t)batch_size + 1) * 4;
        col.d_len.alloc(n);
        launch_qual_wide_len(u, rows, 0, n, col.d_len.p, st);
        finish_var(col, 1);
        launch_qual_wide_scatter(u, rows, n, col.d_off64.p, col.d_values.p, st);
      }
    }
    // ---- tags ----
    if (!tag_cols.empty()) {
      const int nt = (int)p.tag_fields.size();
      DevBuf<uint32_t> loc((uint64_t)nt * n);
      DevBuf<uint8_t> typ((uint64_t)nt * n);
      launch_tag_locate(u, rows, n, d_tags.p, nt, loc.p, typ.p, err.p, st);
      for (auto& tc : tag_cols) {
        Column& col = res->cols[tc.first];
        const uint32_t* l = loc.p + (uint64_t)tc.second * n;
        const uint8_t* ty = typ.p + (uint64_t)tc.second * n;
        col.d_valid.alloc(nwords);
        arrow_bytes += nwords * 8;
        if (col.fd.kind == AK_INT32 || col.fd.kind == AK_UINT32 || col.fd.kind == AK_FLOAT32) {
          col.d_values.alloc(n * 4);
          arrow_bytes += n * 4;
          int kind = col.fd.kind == AK_INT32 ? TAG_INT32 : col.fd.kind == AK_UINT32 ? TAG_UINT32 : TAG_FLOAT32;
          launch_tag_fixed(u, rows, 0, n, l, ty, kind, (uint32_t*)col.d_values.p, col.d_valid.p, err.p, st);
        } else if (col.fd.kind == AK_UTF8) {
          col.d_len.alloc(n);
          launch_tag_utf8_len(u, rows, 0, n, l, ty, col.d_len.p, col.d_valid.p, err.p, st);
          finish_var(col, 1);
          launch_tag_utf8_scatter(u, rows, n, l, ty, col.d_off64.p, col.d_valid.p, 0, col.d_values.p, st);
        } else if (col.is_list()) {
          col.d_len.alloc(n);
          int elem = list_elem_code(col.fd.kind);
          launch_tag_list_len(u, rows, 0, n, l, ty, elem, col.d_len.p, col.d_valid.p, err.p, st);
          finish_var(col, col.list_elem_bytes());
          launch_tag_list_scatter(u, rows, n, l, ty, elem, col.d_off64.p, col.d_values.p, err.p, st);
        } else {
          throw Error("unsupported tag column type");
        }
      }
      throw_extract_err(read_err(err, st));  // loc / typ are released at the end of this scope
    }
    throw_extract_err(read_err(err, st));
    for (auto& col : res->cols) col.d_len.reset();
    s->arrow_bytes = arrow_bytes;
    return res;
  }
};

// -------------------------------------------------------------------------------------------------
// FASTQ (bio-format-fastq/src/physical_exec.rs)
// -------------------------------------------------------------------------------------------------
static std::shared_ptr<Result> run_partition_fastq(const Plan& plan, int partition, uint32_t batch_size, bool to_host) {
  Provider& p = *plan.prov;
  std::lock_guard<std::mutex> lk(p.mu);
  const auto wall0 = std::chrono::steady_clock::now();
  p.make_resident();
  p.set_device();
  hipStream_t st = p.stream;
  auto res = std::make_shared<Result>();
  res->batch_size = batch_size;
  res->device = p.device;
  res->cols.resize(plan.out_fields.size());
  for (size_t c = 0; c < plan.out_fields.size(); c++) res->cols[c].fd = plan.out_fields[c];
  const bool bgzf = p.fq_compression == 1;
  const uint64_t total_len = bgzf ? p.ulen : p.file_len;  // length of the decoded text
  uint64_t start = 0, end = ~0ull;
  if (plan.fq_strategy != 0) { start = plan.fq_parts[partition].first; end = plan.fq_parts[partition].second; }

  // ownership threshold T (absolute decoded offset): a record belongs to the partition iff its '@'
  // lies before T (physical_exec.rs:492-494 / :530-535)
  uint64_t T = total_len;
  uint32_t b_start = 0, b_T = 0;
  if (bgzf) {
    if (plan.fq_strategy == 1) {
      auto it = std::lower_bound(p.blk_uoff.begin(), p.blk_uoff.end(), start);
      if (it == p.blk_uoff.end() || *it != start) throw Error("GZI start offset does not address a block start");
      b_start = (uint32_t)(it - p.blk_uoff.begin());
      if (end != ~0ull) {
        auto jt = std::lower_bound(p.blk_coff.begin(), p.blk_coff.end(), end);
        b_T = (uint32_t)std::min<size_t>(jt - p.blk_coff.begin(), p.n_blocks());
        T = p.blk_uoff[b_T];
      } else b_T = p.n_blocks();
    } else b_T = p.n_blocks();
  } else if (plan.fq_strategy == 2) {
    T = std::min<uint64_t>(end, total_len);
  }

  StageTimer t(st);
  DevBuf<unsigned long long> d_res(2);
  DevBuf<uint32_t> err(1);
  DevBuf<uint64_t> nl, nl_base, tmp;
  DevBuf<uint32_t> nl_cnt;
  uint64_t n_nl = 0, x0 = 0, n_rows = 0, base = 0, hi = 0;
  const uint64_t* nlp = nullptr;   // the index entries from x0 on
  const uint8_t* u = nullptr;
  uint32_t extra = 2;
  for (;;) {
    // ---- bytes of this attempt: [lo, hi) in decoded coordinates, held at u[0 .. hi-base) ----
    uint32_t b_hi = 0;
    if (bgzf) {
      b_hi = std::min<uint32_t>(p.n_blocks(), b_T + extra);
      base = p.blk_uoff[b_start];
      hi = p.blk_uoff[b_hi];
      const uint64_t bytes = hi - base;
      if (p.d_u.n < bytes + 64) p.d_u.alloc(bytes + 64);
      t.start();
      p.launch_inflate(p.d_u.p, b_hi - b_start, b_start);
      res->stats.ms_inflate += t.stop();
      // K2 counts the newlines of every 16 KiB tile of the text while it reads it for the CRC (crc32.hip): the count pass of
      // the newline index -- one more sweep over the text, 8 of the 18 ms of the index on the bench's file -- is not run
      nl_cnt.alloc(nl_chunks(0, bytes) + 1);
      HIP_CHECK(hipMemsetAsync(nl_cnt.p, 0, (nl_chunks(0, bytes) + 1) * 4, p.stream));
      t.start();
      p.launch_crc(p.d_u.p, b_hi - b_start, b_start, nl_cnt.p, 0);
      res->stats.ms_crc += t.stop();
      p.check_inflate_status(b_start, b_hi - b_start);
      u = p.d_u.p;
      res->stats.n_blocks = b_hi - b_start;
      res->stats.compressed_bytes = p.blk_coff[b_hi] - p.blk_coff[b_start];
      res->stats.inflated_bytes = bytes;
    } else {
      base = 0;
      hi = plan.fq_strategy == 2 ? std::min<uint64_t>(total_len, T + (uint64_t)extra * 65536) : total_len;
      u = p.d_comp.p;
      res->stats.inflated_bytes = hi - start;
    }
    const bool at_eof = hi == total_len;
    t.start();
    // ---- resync (only when the partition does not start at byte 0) ----
    x0 = start - base;
    bool none = false;
    if (start > 0) {
      std::vector<uint64_t> we, wc, wn;
      if (bgzf) {
        for (uint32_t b = b_start; b < b_hi; b++) {
          we.push_back(p.blk_uoff[b + 1] - base);
          wc.push_back(p.blk_coff[b]);
          wn.push_back(p.blk_coff[b + 1]);
        }
      } else {
        for (uint64_t w = start; w < hi; w += 8192) {  // std::io::BufReader default capacity
          we.push_back(std::min<uint64_t>(w + 8192, hi) - base);
          wc.push_back(0);
          wn.push_back(0);
        }
      }
      DevBuf<uint64_t> d_we(we.size() + 1), d_wc(wc.size() + 1), d_wn(wn.size() + 1);
      if (!we.empty()) {
        HIP_CHECK(hipMemcpyAsync(d_we.p, we.data(), we.size() * 8, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_wc.p, wc.data(), wc.size() * 8, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_wn.p, wn.data(), wn.size() * 8, hipMemcpyHostToDevice, st));
      }
      launch_fastq_sync(u, start - base, hi - base, d_we.p, d_wc.p, d_wn.p, (uint32_t)we.size(), end, (bgzf && end != ~0ull) ? 1 : 0,
                        d_res.p, st);
      unsigned long long r = 0;
      HIP_CHECK(hipMemcpyAsync(&r, d_res.p, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (r == ~0ull) {
        if (!at_eof) { extra *= 4; continue; }  // ran out of decoded bytes while resynchronising
        none = true;
      } else {
        x0 = r;
        if (x0 >= hi - base && !at_eof) { extra *= 4; continue; }
      }
    }
    n_rows = 0;
    n_nl = 0;
    nlp = nullptr;
    if (!none && x0 < hi - base) {
      if (bgzf) {
        // ---- newline index of the whole buffer from K2's tile counts; the entries in front of x0 are skipped ----
        const uint64_t nch = nl_chunks(0, hi - base);
        nl_base.alloc(nch + 2);
        tmp.alloc(scan_tmp_elems(nch));
        launch_exclusive_scan_u32_to_u64(nl_cnt.p, nl_base.p, nch, tmp.p, st);
        uint64_t n_all = 0;
        HIP_CHECK(hipMemcpyAsync(&n_all, nl_base.p + nch, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        nl.alloc(n_all + 1);
        launch_nl_write(u, 0, hi - base, nl_base.p, nl.p, st);
        unsigned long long skip = 0;
        if (x0) {
          launch_nl_lower_bound(nl.p, n_all, x0, d_res.p, st);
          HIP_CHECK(hipMemcpyAsync(&skip, d_res.p, 8, hipMemcpyDeviceToHost, st));
          HIP_CHECK(hipStreamSynchronize(st));
        }
        nlp = nl.p + skip;
        n_nl = n_all - skip;
      } else {
        // ---- newline index of [x0, hi) ----
        const uint64_t nch = nl_chunks(x0, hi - base);
        nl_cnt.alloc(nch + 1);
        nl_base.alloc(nch + 2);
        tmp.alloc(scan_tmp_elems(nch));
        launch_nl_count(u, x0, hi - base, nl_cnt.p, st);
        launch_exclusive_scan_u32_to_u64(nl_cnt.p, nl_base.p, nch, tmp.p, st);
        HIP_CHECK(hipMemcpyAsync(&n_nl, nl_base.p + nch, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        nl.alloc(n_nl + 1);
        launch_nl_write(u, x0, hi - base, nl_base.p, nl.p, st);
        nlp = nl.p;
      }
      // ---- how many records start before T ----
      const uint64_t T_rel = T > base ? T - base : 0;
      launch_fastq_count_owned(nlp, n_nl, x0, hi - base, T_rel, d_res.p, st);
      unsigned long long r = 0;
      HIP_CHECK(hipMemcpyAsync(&r, d_res.p, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      n_rows = r;
      // the last record that is READ must be complete: 4 newline-terminated lines, or the data really ends (the reference
      // stops reading at `limit` rows: a record cut short behind that is never looked at)
      const uint64_t n_read = plan.limit >= 0 ? std::min<uint64_t>(n_rows, (uint64_t)plan.limit) : n_rows;
      if (n_read * 4 > n_nl + (at_eof ? 1 : 0)) {
        if (!at_eof) { extra *= 4; continue; }
        throw Error("FASTQ read error: unexpected end of file inside a record");
      }
    }
    break;
  }
  res->stats.ms_chain = t.stop();
  if (plan.limit >= 0 && n_rows > (uint64_t)plan.limit) n_rows = (uint64_t)plan.limit;
  res->n_rows = n_rows;
  res->stats.n_records = n_rows;
  res->stats.n_rows = n_rows;
  const uint64_t n = n_rows, nwords = (n + 63) / 64, nb = res->n_batches();
  t.start();
  uint64_t arrow_bytes = 0;
  if (n) {
    HIP_CHECK(hipMemsetAsync(err.p, 0, 4, st));
    int col_of[4] = {-1, -1, -1, -1};
    for (size_t c = 0; c < plan.out_fields.size(); c++) {
      int src = plan.has_projection ? plan.projection[c] : (int)c;
      if (col_of[src] < 0) col_of[src] = (int)c;
    }
    // two passes, no per-row length / source arrays (fastq_kernels.hip): pass 1 -> tile sums -> column totals -> pass 2
    FqCols fc{};
    for (int k = 0; k < 4; k++) {
      if (col_of[k] < 0) continue;
      Column& col = res->cols[col_of[k]];
      col.n_rows = n;
      fc.want |= 1u << k;
      if (k == 1) { col.d_valid.alloc(nwords); fc.v_desc = col.d_valid.p; arrow_bytes += nwords * 8; }
    }
    const uint64_t n_tiles = (n + ROWS_TILE - 1) / ROWS_TILE;
    DevBuf<uint64_t> tile_sums(bam_rows_scratch_elems(n));
    launch_fastq_pass1(u, x0, hi - base, nlp, n_nl, n, fc, tile_sums.p, err.p, st);
    uint64_t totals[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++)
      if ((fc.want >> k) & 1u) HIP_CHECK(hipMemcpyAsync(&totals[k], tile_sums.p + (uint64_t)k * (n_tiles + 1) + n_tiles, 8, hipMemcpyDeviceToHost, st));
    uint32_t e = read_err(err, st);  // synchronises: the totals are here as well
    if (e == 1) throw Error("FASTQ read error: invalid name prefix");
    if (e == 2) throw Error("FASTQ read error: invalid description prefix");
    for (int k = 0; k < 4; k++) {
      if (col_of[k] < 0) continue;
      Column& col = res->cols[col_of[k]];
      col.total_bytes = totals[k];
      col.d_values.alloc(std::max<uint64_t>(totals[k], 1));
      col.d_off32.alloc(nb * ((uint64_t)batch_size + 1));
      col.d_base.alloc(nb);
      fc.val[k] = col.d_values.p; fc.off32[k] = col.d_off32.p; fc.base[k] = col.d_base.p;
      arrow_bytes += totals[k] + nb * ((uint64_t)batch_size + 1) * 4;
    }
    launch_fastq_pass2(u, x0, hi - base, nlp, n_nl, n, fc, batch_size, 0, tile_sums.p, st);
    HIP_CHECK(hipStreamSynchronize(st));
  }
  res->stats.ms_extract = t.stop();
  res->stats.arrow_bytes = arrow_bytes;
  res->stats.ms_total_gpu = res->stats.ms_inflate + res->stats.ms_crc + res->stats.ms_chain + res->stats.ms_extract;
  res->stats.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  if (to_host) copy_result_to_host(*res, st);
  return res;
}

// -------------------------------------------------------------------------------------------------
// FastqExecState: the chunk pipeline of a BGZF FASTQ partition (bio-format-fastq/src/physical_exec.rs:393-465: the
// reference reads record by record and never holds a partition).  Members are inflated `chunk_members` at a time; a
// chunk's rows are its complete records that start before the partition's ownership threshold; the record cut by the
// chunk's end is carried, as bytes, to the head of the next chunk's buffer; batches continue across chunks (`phase`).
// HBM: two inflated chunk buffers + the newline index and Arrow buffers of one chunk -- O(chunk), not O(partition).
// -------------------------------------------------------------------------------------------------
struct FastqExecState : ChunkProducer {
  const Plan& plan;
  Provider& p;
  const uint32_t batch_size;
  const bool to_host;
  const uint32_t chunk_members;
  const bool ramp;
  std::shared_ptr<DeviceImage> img;
  K1Ctx k1;
  hipStream_t st = nullptr, copy_st = nullptr;
  uint64_t start = 0, end = ~0ull, T = 0;   // decoded offset the partition starts at; GZI end (compressed); ownership threshold
  uint32_t b_start = 0;
  uint32_t next_member = 0;
  bool first = true, done = false, dead = false;
  uint64_t carry_len = 0;
  DevBuf<uint8_t> ubuf[2];
  int cur = 0;
  uint32_t chunks_done = 0, phase = 0;
  uint64_t rows_emitted = 0;
  std::chrono::steady_clock::time_point wall0 = std::chrono::steady_clock::now();

  FastqExecState(const Plan& pl, int partition, uint32_t bs, bool host)
      : plan(pl), p(*pl.prov), batch_size(bs), to_host(host),
        chunk_members(pl.prov->chunk_members ? pl.prov->chunk_members : host ? env_knobs().chunk_members : env_knobs().chunk_members_device),
        ramp(host && !pl.prov->chunk_members) {
    if (plan.fq_strategy != 0) { start = plan.fq_parts[partition].first; end = plan.fq_parts[partition].second; }
    uint32_t b_T = p.n_blocks();
    T = p.ulen;
    if (plan.fq_strategy == 1) {
      auto it = std::lower_bound(p.blk_uoff.begin(), p.blk_uoff.end(), start);
      if (it == p.blk_uoff.end() || *it != start) throw Error("GZI start offset does not address a block start");
      b_start = (uint32_t)(it - p.blk_uoff.begin());
      if (end != ~0ull) {
        auto jt = std::lower_bound(p.blk_coff.begin(), p.blk_coff.end(), end);
        b_T = (uint32_t)std::min<size_t>(jt - p.blk_coff.begin(), p.n_blocks());
        T = p.blk_uoff[b_T];
      }
    }
    next_member = b_start;
    // (the record that starts just before the threshold may end a few members behind it; a partition's image is widened
    // when that happens: BgzfSource::image_for)
    img = p.device_image(plan.device_of(partition), b_start, std::min<uint32_t>(p.n_blocks(), b_T + 4));
    HIP_CHECK(hipSetDevice(img->device));
    p.init_ctx(k1, *img, std::min<uint32_t>(chunk_members, std::max<uint32_t>(p.n_blocks() - b_start, 1)), false);
    st = k1.stream;
    if (to_host) HIP_CHECK(hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking));
  }
  ~FastqExecState() override {
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (img) (void)hipSetDevice(img->device);
    if (k1.stream) (void)hipStreamSynchronize(k1.stream);
    if (copy_st) { (void)hipStreamSynchronize(copy_st); (void)hipStreamDestroy(copy_st); }
    (void)hipSetDevice(prev);
  }
  uint32_t chunk_len(uint32_t k) const {
    uint64_t c = chunk_members;
    if (ramp) c = std::min<uint64_t>(c, 2048ull << std::min<uint32_t>(k, 16));
    return (uint32_t)std::max<uint64_t>(c, 1);
  }
  void accumulate(const bioscan_scan_stats& s) {
    total.n_blocks += s.n_blocks; total.compressed_bytes += s.compressed_bytes; total.inflated_bytes += s.inflated_bytes;
    total.arrow_bytes += s.arrow_bytes; total.n_records += s.n_records; total.n_rows += s.n_rows;
    total.ms_inflate += s.ms_inflate; total.ms_crc += s.ms_crc; total.ms_chain += s.ms_chain; total.ms_extract += s.ms_extract;
    total.ms_total_gpu += s.ms_total_gpu;
    total.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  }

  std::shared_ptr<Result> next_chunk() override {
    if (dead) throw Error("the stream ended with an error");
    HIP_CHECK(hipSetDevice(img->device));
    struct Guard { bool* d; bool ok = false; ~Guard() { if (!ok) *d = true; } } guard{&dead};
    std::shared_ptr<Result> r;
    while (!done && !(r = run_chunk())) {}
    guard.ok = true;
    return (r && r->n_rows) ? r : nullptr;
  }

  // one chunk; nullptr = it held no complete owned record (the next call takes more members behind the carried bytes)
  std::shared_ptr<Result> run_chunk() {
    StageTimer t(st);
    bioscan_scan_stats s{};
    uint32_t grow = 1;
    for (;;) {   // (repeated with more members only while the FIRST chunk cannot be synchronised)
      const uint32_t m0 = next_member;
      const uint32_t m1 = (uint32_t)std::min<uint64_t>(p.n_blocks(), (uint64_t)m0 + (uint64_t)chunk_len(chunks_done) * grow);
      if (m1 > img->m_hi || m0 < img->m_lo) { img = p.device_image(img->device, std::min(m0, img->m_lo), std::max(m1, img->m_hi)); }
      const uint64_t bytes = p.blk_uoff[m1] - p.blk_uoff[m0];
      const uint64_t L = carry_len + bytes;                 // decoded bytes in ubuf[cur]
      const uint64_t origin = p.blk_uoff[m0] - carry_len;   // decoded offset of ubuf[cur][0]
      if (ubuf[cur].n < L + 64) {
        DevBuf<uint8_t> g(L + 64);
        if (carry_len) HIP_CHECK(hipMemcpyAsync(g.p, ubuf[cur].p, carry_len, hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        ubuf[cur] = std::move(g);
      }
      uint8_t* u = ubuf[cur].p;
      const bool at_eof = m1 == p.n_blocks();
      DevBuf<uint32_t> nl_cnt;   // newlines per 16 KiB tile of u[0, L)
      if (m1 > m0) {
        if (k1.status.n < m1 - m0) p.init_ctx(k1, *img, m1 - m0, false);
        t.start();
        p.launch_inflate(k1, *img, u + carry_len, m1 - m0, m0);
        s.ms_inflate += t.stop();
        // K2 counts the newlines per 16 KiB tile of the buffer while it reads the members for their CRC; the carried bytes in
        // front of them are counted into the same tiles here (fastq_kernels.hip: k_nl_count, add mode)
        nl_cnt.alloc(nl_chunks(0, L) + 1);
        HIP_CHECK(hipMemsetAsync(nl_cnt.p, 0, (nl_chunks(0, L) + 1) * 4, st));
        if (carry_len) launch_nl_count(u, 0, carry_len, nl_cnt.p, st, true);
        t.start();
        p.launch_crc(k1, *img, u + carry_len, m1 - m0, m0, nl_cnt.p, carry_len);
        s.ms_crc += t.stop();
        p.check_inflate_status(k1, m0, m1 - m0);
      }
      s.n_blocks = m1 - m0;
      s.compressed_bytes = p.blk_coff[m1] - p.blk_coff[m0];
      s.inflated_bytes = bytes;
      t.start();
      // ---- first record of the chunk: the carried record's first byte, or (first chunk of a split) the resynchronisation ----
      uint64_t x0 = 0;
      bool none = false;
      DevBuf<unsigned long long> d_res(2);
      if (first) {
        x0 = start - origin;
        if (start > 0) {
          std::vector<uint64_t> we, wc, wn;
          for (uint32_t b = m0; b < m1; b++) {
            we.push_back(p.blk_uoff[b + 1] - origin);
            wc.push_back(p.blk_coff[b]);
            wn.push_back(p.blk_coff[b + 1]);
          }
          DevBuf<uint64_t> d_we(we.size() + 1), d_wc(wc.size() + 1), d_wn(wn.size() + 1);
          if (!we.empty()) {
            HIP_CHECK(hipMemcpyAsync(d_we.p, we.data(), we.size() * 8, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(d_wc.p, wc.data(), wc.size() * 8, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(d_wn.p, wn.data(), wn.size() * 8, hipMemcpyHostToDevice, st));
          }
          launch_fastq_sync(u, start - origin, L, d_we.p, d_wc.p, d_wn.p, (uint32_t)we.size(), end, end != ~0ull ? 1 : 0, d_res.p, st);
          unsigned long long r = 0;
          HIP_CHECK(hipMemcpyAsync(&r, d_res.p, 8, hipMemcpyDeviceToHost, st));
          HIP_CHECK(hipStreamSynchronize(st));
          if (r == ~0ull || (r >= L && !at_eof)) {
            if (!at_eof) { grow *= 4; s.ms_chain += t.stop(); continue; }  // ran out of decoded bytes while resynchronising
            none = true;
          } else x0 = r;
        }
      }
      uint64_t n_nl = 0, owned = 0;
      DevBuf<uint64_t> nl, nl_base, tmp;
      const uint64_t* nlp = nullptr;   // the index entries from x0 on
      if (!none && x0 < L) {
        const uint64_t nch = nl_chunks(0, L);
        nl_base.alloc(nch + 2);
        tmp.alloc(scan_tmp_elems(nch));
        if (m1 == m0) {   // (no member in this chunk: nothing has been counted)
          nl_cnt.alloc(nch + 1);
          launch_nl_count(u, 0, L, nl_cnt.p, st);
        }
        launch_exclusive_scan_u32_to_u64(nl_cnt.p, nl_base.p, nch, tmp.p, st);
        uint64_t n_all = 0;
        HIP_CHECK(hipMemcpyAsync(&n_all, nl_base.p + nch, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        nl.alloc(n_all + 1);
        launch_nl_write(u, 0, L, nl_base.p, nl.p, st);
        unsigned long long skip = 0;
        if (x0) {
          launch_nl_lower_bound(nl.p, n_all, x0, d_res.p, st);
          HIP_CHECK(hipMemcpyAsync(&skip, d_res.p, 8, hipMemcpyDeviceToHost, st));
          HIP_CHECK(hipStreamSynchronize(st));
        }
        nlp = nl.p + skip;
        n_nl = n_all - skip;
        const uint64_t T_rel = T > origin ? T - origin : 0;
        launch_fastq_count_owned(nlp, n_nl, x0, L, T_rel, d_res.p, st);
        unsigned long long r = 0;
        HIP_CHECK(hipMemcpyAsync(&r, d_res.p, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        owned = r;
      }
      // records whose four lines are all here (the file's last line may lack its newline)
      const uint64_t complete = (n_nl + (at_eof ? 1 : 0)) / 4;
      uint64_t n = std::min<uint64_t>(owned, complete);
      // first byte behind the last complete record = where the next record starts (the carried bytes begin there)
      uint64_t cs = none ? L : x0;
      if (complete && 4 * complete - 1 < n_nl) {
        uint64_t v = 0;
        HIP_CHECK(hipMemcpyAsync(&v, nlp + 4 * complete - 1, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        cs = (v & ((1ull << 48) - 1)) + 1;   // (fastq_kernels.hip NL_POS: the low 48 bits are the position)
      } else if (complete) cs = L;           // (the file's last record, without its final newline)
      // the partition is over when a record that starts at or behind the threshold has been seen, or the data has
      bool finished = none || owned < complete || origin + cs >= T || (at_eof && cs >= L);
      const bool limit_ends_it = plan.limit >= 0 && rows_emitted + n >= (uint64_t)plan.limit;
      if (limit_ends_it) { n = (uint64_t)plan.limit - rows_emitted; finished = true; }
      // (a record cut short by the end of the file is an error only when the reference would read it: not behind `limit` rows)
      else if (owned > complete && at_eof) throw Error("FASTQ read error: unexpected end of file inside a record");
      s.ms_chain += t.stop();
      s.n_records = n;
      s.n_rows = n;
      // ---- the chunk's rows ----
      std::shared_ptr<Result> res;
      t.start();
      if (n) res = extract(u, x0, L, nlp, n_nl, n, &s);
      s.ms_extract = t.stop();
      s.ms_total_gpu = s.ms_inflate + s.ms_crc + s.ms_chain + s.ms_extract;
      accumulate(s);
      if (res) {
        res->stats = s;
        if (to_host) start_copy_to_host(*res, st, copy_st);
      }
      rows_emitted += n;
      phase = (uint32_t)((phase + n) % batch_size);
      // ---- carry the cut record ----
      first = false;
      next_member = m1;
      chunks_done++;
      if (finished) done = true;
      else {
        const uint64_t c = L - cs;
        const int nxt = cur ^ 1;
        const uint64_t next_bytes = p.blk_uoff[std::min<uint64_t>(p.n_blocks(), (uint64_t)m1 + chunk_len(chunks_done))] - p.blk_uoff[m1];
        if (ubuf[nxt].n < c + next_bytes + 64) ubuf[nxt].alloc(c + next_bytes + 64);
        if (c) HIP_CHECK(hipMemcpyAsync(ubuf[nxt].p, u + cs, c, hipMemcpyDeviceToDevice, st));
        carry_len = c;
        cur = nxt;
      }
      HIP_CHECK(hipStreamSynchronize(st));   // the chunk's scratch (newline index) goes out of scope
      return res;
    }
  }

  std::shared_ptr<Result> extract(const uint8_t* u, uint64_t x0, uint64_t L, const uint64_t* nl, uint64_t n_nl, uint64_t n, bioscan_scan_stats* s) {
    auto res = std::make_shared<Result>();
    res->batch_size = batch_size;
    res->phase = phase;
    res->device = img->device;
    res->n_rows = n;
    res->cols.resize(plan.out_fields.size());
    for (size_t c = 0; c < plan.out_fields.size(); c++) res->cols[c].fd = plan.out_fields[c];
    const uint64_t nwords = (n + 63) / 64, nb = res->n_batches();
    DevBuf<uint32_t> err(1);
    HIP_CHECK(hipMemsetAsync(err.p, 0, 4, st));
    int col_of[4] = {-1, -1, -1, -1};
    for (size_t c = 0; c < plan.out_fields.size(); c++) {
      int src = plan.has_projection ? plan.projection[c] : (int)c;
      if (col_of[src] < 0) col_of[src] = (int)c;
    }
    uint64_t arrow_bytes = 0;
    FqCols fc{};
    for (int k = 0; k < 4; k++) {
      if (col_of[k] < 0) continue;
      Column& col = res->cols[col_of[k]];
      col.n_rows = n;
      fc.want |= 1u << k;
      if (k == 1) { col.d_valid.alloc(nwords); fc.v_desc = col.d_valid.p; arrow_bytes += nwords * 8; }
    }
    const uint64_t n_tiles = (n + ROWS_TILE - 1) / ROWS_TILE;
    DevBuf<uint64_t> tile_sums(bam_rows_scratch_elems(n));
    launch_fastq_pass1(u, x0, L, nl, n_nl, n, fc, tile_sums.p, err.p, st);
    uint64_t totals[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++)
      if ((fc.want >> k) & 1u) HIP_CHECK(hipMemcpyAsync(&totals[k], tile_sums.p + (uint64_t)k * (n_tiles + 1) + n_tiles, 8, hipMemcpyDeviceToHost, st));
    uint32_t e = read_err(err, st);  // synchronises: the totals are here as well
    if (e == 1) throw Error("FASTQ read error: invalid name prefix");
    if (e == 2) throw Error("FASTQ read error: invalid description prefix");
    for (int k = 0; k < 4; k++) {
      if (col_of[k] < 0) continue;
      Column& col = res->cols[col_of[k]];
      col.total_bytes = totals[k];
      col.d_values.alloc(std::max<uint64_t>(totals[k], 1));
      col.d_off32.alloc(nb * ((uint64_t)batch_size + 1));
      col.d_base.alloc(nb);
      fc.val[k] = col.d_values.p; fc.off32[k] = col.d_off32.p; fc.base[k] = col.d_base.p;
      arrow_bytes += totals[k] + nb * ((uint64_t)batch_size + 1) * 4;
    }
    launch_fastq_pass2(u, x0, L, nl, n_nl, n, fc, batch_size, phase, tile_sums.p, st);
    HIP_CHECK(hipStreamSynchronize(st));
    s->arrow_bytes = arrow_bytes;
    return res;
  }
};

// -------------------------------------------------------------------------------------------------
// Arrow C Data export
// -------------------------------------------------------------------------------------------------
static const char* arrow_format(ArrowKind k) {
  switch (k) {
    case AK_INT32: return "i";
    case AK_UINT32: return "I";
    case AK_FLOAT32: return "f";
    case AK_UTF8: return "u";
    case AK_BINARY: return "z";
    default: return "+l";
  }
}
static const char* list_child_format(ArrowKind k) {
  switch (k) {
    case AK_LIST_INT8: return "c";
    case AK_LIST_UINT8: return "C";
    case AK_LIST_INT16: return "s";
    case AK_LIST_UINT16: return "S";
    case AK_LIST_INT32: return "i";
    case AK_LIST_UINT32: return "I";
    default: return "f";
  }
}

struct SchemaPriv {
  std::string name, format, metadata;
  std::vector<ArrowSchema*> children;
  std::vector<std::unique_ptr<ArrowSchema>> owned;
};
static void release_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  auto* pr = (SchemaPriv*)s->private_data;
  for (auto& c : pr->owned) if (c->release) c->release(c.get());
  delete pr;
  s->release = nullptr;
}
static std::string encode_metadata(const std::vector<std::pair<std::string, std::string>>& md) {
  if (md.empty()) return std::string();
  std::string o;
  auto put32 = [&](int32_t v) { o.append((const char*)&v, 4); };
  put32((int32_t)md.size());
  for (auto& kv : md) {
    put32((int32_t)kv.first.size()); o += kv.first;
    put32((int32_t)kv.second.size()); o += kv.second;
  }
  return o;
}
static void fill_schema(ArrowSchema* s, const std::string& name, const std::string& format, bool nullable,
                        const std::vector<std::pair<std::string, std::string>>& md) {
  auto* pr = new SchemaPriv();
  pr->name = name;
  pr->format = format;
  pr->metadata = encode_metadata(md);
  memset(s, 0, sizeof(*s));
  s->format = pr->format.c_str();
  s->name = pr->name.c_str();
  s->metadata = pr->metadata.empty() ? nullptr : pr->metadata.data();
  s->flags = nullable ? ARROW_FLAG_NULLABLE : 0;
  s->release = release_schema;
  s->private_data = pr;
}
static void add_child(ArrowSchema* parent, std::unique_ptr<ArrowSchema> child) {
  auto* pr = (SchemaPriv*)parent->private_data;
  pr->children.push_back(child.get());
  pr->owned.push_back(std::move(child));
  parent->n_children = (int64_t)pr->children.size();
  parent->children = pr->children.data();
}
static void export_schema(const std::vector<FieldDef>& fields, const std::vector<std::pair<std::string, std::string>>& md, ArrowSchema* out) {
  fill_schema(out, "", "+s", false, md);
  for (auto& f : fields) {
    std::unique_ptr<ArrowSchema> c(new ArrowSchema);
    fill_schema(c.get(), f.name, arrow_format(f.kind), f.nullable, f.metadata);
    if (f.kind >= AK_LIST_INT8) {
      std::unique_ptr<ArrowSchema> item(new ArrowSchema);
      fill_schema(item.get(), "item", list_child_format(f.kind), true, {});
      add_child(c.get(), std::move(item));
    }
    add_child(out, std::move(c));
  }
}

struct ArrayPriv {
  std::shared_ptr<Result> keep;
  std::vector<const void*> buffers;
  std::vector<ArrowArray*> children;
  std::vector<std::unique_ptr<ArrowArray>> owned;
  std::vector<uint8_t> local_valid;  // repacked validity when the batch is not byte aligned
  std::vector<std::unique_ptr<uint8_t[]>> own;  // buffers of a batch stitched from several chunks
};
static void release_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* pr = (ArrayPriv*)a->private_data;
  for (auto& c : pr->owned) if (c->release) c->release(c.get());
  delete pr;
  a->release = nullptr;
}
static ArrayPriv* init_array(ArrowArray* a, std::shared_ptr<Result> keep, int64_t length) {
  auto* pr = new ArrayPriv();
  pr->keep = std::move(keep);
  memset(a, 0, sizeof(*a));
  a->length = length;
  a->release = release_array;
  a->private_data = pr;
  return pr;
}
static void finish_array(ArrowArray* a) {
  auto* pr = (ArrayPriv*)a->private_data;
  a->n_buffers = (int64_t)pr->buffers.size();
  a->buffers = pr->buffers.data();
  a->n_children = (int64_t)pr->children.size();
  a->children = pr->children.empty() ? nullptr : pr->children.data();
}
static int64_t count_nulls(const uint8_t* bits, uint64_t bit0, uint64_t nbits) {
  // set bits of [bit0, bit0 + nbits): ragged head and tail bit by bit, the aligned middle by 64-bit popcounts
  int64_t set = 0;
  uint64_t i = bit0, end = bit0 + nbits;
  for (; i < end && (i & 63); i++) set += (bits[i >> 3] >> (i & 7)) & 1;
  for (; i + 64 <= end; i += 64) {
    uint64_t w;
    memcpy(&w, bits + (i >> 3), 8);
    set += __builtin_popcountll(w);
  }
  for (; i < end; i++) set += (bits[i >> 3] >> (i & 7)) & 1;
  return (int64_t)nbits - set;
}

static void export_batch(const std::shared_ptr<Result>& res, uint64_t b, ArrowArray* out) {
  const uint64_t bs = res->batch_size;
  const uint64_t r0 = res->batch_row0(b);
  const uint64_t rows = res->batch_rows(b);
  ArrayPriv* top = init_array(out, res, (int64_t)rows);
  top->buffers.push_back(nullptr);
  for (auto& col : res->cols) {
    std::unique_ptr<ArrowArray> ca(new ArrowArray);
    ArrayPriv* pr = init_array(ca.get(), res, (int64_t)rows);
    // validity
    const void* vptr = nullptr;
    int64_t nulls = 0;
    if (col.fd.nullable && col.h_valid.p) {
      nulls = count_nulls(col.h_valid.p, r0, rows);
      if (nulls) {
        if ((r0 & 7) == 0) vptr = col.h_valid.p + (r0 >> 3);
        else {
          pr->local_valid.assign((rows + 7) / 8, 0);
          for (uint64_t i = 0; i < rows; i++) {
            uint64_t s = r0 + i;
            if ((col.h_valid.p[s >> 3] >> (s & 7)) & 1) pr->local_valid[i >> 3] |= (uint8_t)(1u << (i & 7));
          }
          vptr = pr->local_valid.data();
        }
      }
    }
    ca->null_count = nulls;
    pr->buffers.push_back(vptr);
    if (col.is_var()) {
      pr->buffers.push_back(col.h_off32.p + b * (bs + 1) * 4);
      pr->buffers.push_back(col.h_values.p + col.h_batch_base[b]);
    } else if (col.is_list()) {
      const int32_t* off = (const int32_t*)(col.h_off32.p + b * (bs + 1) * 4);
      pr->buffers.push_back(off);
      std::unique_ptr<ArrowArray> item(new ArrowArray);
      ArrayPriv* ip = init_array(item.get(), res, (int64_t)off[rows]);
      ip->buffers.push_back(nullptr);
      ip->buffers.push_back(col.h_values.p + col.h_batch_base[b] * col.list_elem_bytes());
      finish_array(item.get());
      pr->children.push_back(item.get());
      pr->owned.push_back(std::move(item));
    } else {
      pr->buffers.push_back(col.h_values.p + r0 * 4);
    }
    finish_array(ca.get());
    top->children.push_back(ca.get());
    top->owned.push_back(std::move(ca));
  }
  finish_array(out);
}

// A batch whose rows come from more than one chunk (the tail batch of one chunk + the head batch of the next, or several
// small chunks): the pieces -- each a whole device batch of its chunk -- are concatenated into buffers the exported
// array owns.  At most one batch per chunk takes this path.
struct Piece {
  std::shared_ptr<Result> res;
  uint64_t b;
};
static void export_pieces(const std::vector<Piece>& pieces, ArrowArray* out) {
  if (pieces.size() == 1) { export_batch(pieces[0].res, pieces[0].b, out); return; }
  uint64_t rows = 0;
  for (auto& pc : pieces) rows += pc.res->batch_rows(pc.b);
  const Result& first = *pieces[0].res;
  ArrayPriv* top = init_array(out, nullptr, (int64_t)rows);
  top->buffers.push_back(nullptr);
  for (size_t c = 0; c < first.cols.size(); c++) {
    const Column& c0 = first.cols[c];
    std::unique_ptr<ArrowArray> ca(new ArrowArray);
    ArrayPriv* pr = init_array(ca.get(), nullptr, (int64_t)rows);
    auto new_buf = [&](ArrayPriv* owner, size_t bytes) -> uint8_t* {
      owner->own.emplace_back(new uint8_t[bytes ? bytes : 1]);
      return owner->own.back().get();
    };
    // validity
    int64_t nulls = 0;
    const void* vptr = nullptr;
    if (c0.fd.nullable) {
      uint8_t* vb = new_buf(pr, (rows + 7) / 8);
      memset(vb, 0, (rows + 7) / 8);
      uint64_t o = 0;
      for (auto& pc : pieces) {
        const Column& col = pc.res->cols[c];
        const uint64_t r0 = pc.res->batch_row0(pc.b), nr = pc.res->batch_rows(pc.b);
        for (uint64_t i = 0; i < nr; i++, o++) {
          const uint64_t sidx = r0 + i;
          const bool v = !col.h_valid.p || ((col.h_valid.p[sidx >> 3] >> (sidx & 7)) & 1);
          if (v) vb[o >> 3] |= (uint8_t)(1u << (o & 7)); else nulls++;
        }
      }
      if (nulls) vptr = vb;
    }
    ca->null_count = nulls;
    pr->buffers.push_back(vptr);
    if (c0.is_var() || c0.is_list()) {
      const uint32_t eb = c0.is_list() ? c0.list_elem_bytes() : 1;
      uint64_t total = 0;
      for (auto& pc : pieces) {
        const Column& col = pc.res->cols[c];
        const int32_t* off = (const int32_t*)(col.h_off32.p + pc.b * ((uint64_t)pc.res->batch_size + 1) * 4);
        total += (uint64_t)off[pc.res->batch_rows(pc.b)];
      }
      int32_t* ob = (int32_t*)new_buf(pr, (rows + 1) * 4);
      uint8_t* vals = new_buf(pr, total * eb);
      uint64_t o = 0, acc = 0;
      for (auto& pc : pieces) {
        const Column& col = pc.res->cols[c];
        const uint64_t nr = pc.res->batch_rows(pc.b);
        const int32_t* off = (const int32_t*)(col.h_off32.p + pc.b * ((uint64_t)pc.res->batch_size + 1) * 4);
        for (uint64_t i = 0; i < nr; i++) ob[o++] = (int32_t)(acc + (uint64_t)off[i]);
        const uint64_t len = (uint64_t)off[nr];
        if (len) memcpy(vals + acc * eb, col.h_values.p + col.h_batch_base[pc.b] * eb, len * eb);
        acc += len;
      }
      ob[o] = (int32_t)acc;
      pr->buffers.push_back(ob);
      if (c0.is_var()) {
        pr->buffers.push_back(vals);
      } else {
        std::unique_ptr<ArrowArray> item(new ArrowArray);
        ArrayPriv* ip = init_array(item.get(), nullptr, (int64_t)acc);
        ip->buffers.push_back(nullptr);
        ip->buffers.push_back(vals);
        finish_array(item.get());
        pr->children.push_back(item.get());
        pr->owned.push_back(std::move(item));
      }
    } else {
      uint8_t* vals = new_buf(pr, rows * 4);
      uint64_t o = 0;
      for (auto& pc : pieces) {
        const Column& col = pc.res->cols[c];
        const uint64_t r0 = pc.res->batch_row0(pc.b), nr = pc.res->batch_rows(pc.b);
        memcpy(vals + o * 4, col.h_values.p + r0 * 4, nr * 4);
        o += nr;
      }
      pr->buffers.push_back(vals);
    }
    finish_array(ca.get());
    top->children.push_back(ca.get());
    top->owned.push_back(std::move(ca));
  }
  finish_array(out);
}

// SendableRecordBatchStream: batches of exactly batch_size rows (the last one short) cut out of the chunk results of
// a producer.  While the consumer works through chunk k the producer has already run chunk k + 1 and chunk k's D2H has
// been in flight since before that.
struct Stream {
  Provider* prov = nullptr;
  std::unique_ptr<ChunkProducer> exec;   // the partition's chunk pipeline (BAM, BGZF FASTQ); null for a plain-text FASTQ stream (one result) or once exhausted
  std::shared_ptr<Result> cur;
  bool on_host = true;
  uint64_t cur_batch = 0;
  std::vector<Piece> pending;           // whole device batches that do not fill a batch yet
  uint64_t pending_rows = 0;
  std::vector<std::shared_ptr<Result>> device_results;  // execute_device: the chunks stay in HBM until the stream is closed
  bioscan_scan_stats stats{};
  // The producer runs on its own thread (the reference: one OS thread per partition feeding a bounded channel,
  // bio-format-core/src/sync_stream.rs:19-29): chunk k + 2 is computed while chunk k + 1 is on the link and chunk k is
  // being handed out as batches.  Driven from the consumer's calls instead, a chunk's kernels only started once the
  // consumer came back for more -- after the previous chunk's copy had finished and its batches had been exported -- and
  // the link idled for the kernels of every chunk (41 of ~55 GB/s on a config-2 stream).
  static constexpr size_t QUEUE_DEPTH = 2;
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<Result>> ready;
  bool producer_done = false, stop = false;
  std::exception_ptr producer_error;

  void start_producer() {
    worker = std::thread([this] {
      try {
        for (;;) {
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return stop || ready.size() < QUEUE_DEPTH; });
            if (stop) break;
          }
          auto r = exec->next_chunk();   // (kernels + the start of its D2H; returns with the copy in flight)
          std::lock_guard<std::mutex> lk(mu);
          if (!r) { producer_done = true; cv.notify_all(); break; }
          ready.push_back(std::move(r));
          cv.notify_all();
        }
      } catch (...) {
        std::lock_guard<std::mutex> lk(mu);
        producer_error = std::current_exception();
        producer_done = true;
        cv.notify_all();
      }
    });
  }
  ~Stream() {
    if (worker.joinable()) {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv.notify_all();
      worker.join();
    }
    // chunks that were never handed out: their copies finish before their buffers go (finish_copy waits for the event)
    for (auto& r : ready) if (r) { try { finish_copy(*r); } catch (...) {} }
    ready.clear();
    cur.reset();
    pending.clear();
    exec.reset();
  }

  bool fetch() {
    if (!exec) return false;
    if (!worker.joinable() && !producer_done) start_producer();
    std::shared_ptr<Result> r;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return !ready.empty() || producer_done; });
      if (!ready.empty()) { r = std::move(ready.front()); ready.pop_front(); cv.notify_all(); }
      else if (producer_error) {
        // sticky: a consumer that polls again after an error gets the error again, never the leftover rows and a clean end
        // of stream that would make a truncated partition look complete
        auto e = producer_error;
        lk.unlock();
        pending.clear();
        pending_rows = 0;
        std::rethrow_exception(e);
      }
    }
    if (!r) return false;   // end of the partition
    cur = std::move(r);
    cur_batch = 0;
    HIP_CHECK(hipSetDevice(cur->device));
    finish_copy(*cur);
    return true;
  }
  // false = end of stream
  bool next(ArrowArray* out) {
    for (;;) {
      if (cur) {
        const uint64_t nb = cur->n_batches(), bs = cur->batch_size;
        while (cur_batch < nb) {
          const uint64_t b = cur_batch++;
          const uint64_t rows = cur->batch_rows(b);
          if (rows == bs && pending.empty()) { export_batch(cur, b, out); return true; }
          // the head batch completes what earlier chunks left over; a short last batch waits for the next chunk
          pending.push_back(Piece{cur, b});
          pending_rows += rows;
          if (pending_rows == bs) { flush(out); return true; }
        }
        cur.reset();
      }
      if (!fetch()) {
        if (pending_rows) { flush(out); return true; }
        return false;
      }
    }
  }
  void flush(ArrowArray* out) {
    export_pieces(pending, out);
    pending.clear();
    pending_rows = 0;
  }
};

// -------------------------------------------------------------------------------------------------
// schema determination (table_provider.rs:42-140, 447-505)
// -------------------------------------------------------------------------------------------------
struct AuxVal { char type; char subtype; };

static void infer_tags_from_prefix(Provider& p, const std::vector<std::string>& unknown, size_t sample_size,
                                   std::map<std::string, std::pair<char, ArrowKind>>* found) {
  // Records are sampled from GPU-inflated leading blocks (schema discovery only; the scan itself
  // never parses records on the host).
  uint32_t nb = 2;
  for (;;) {
    std::vector<uint8_t> u = p.inflate_prefix_to_host(nb);
    size_t o = p.hdr.first_record_offset;
    size_t count = 0;
    bool truncated = false;
    std::map<std::string, std::pair<char, ArrowKind>> f;
    while (count < sample_size && o + 4 <= u.size()) {
      int32_t bs;
      memcpy(&bs, &u[o], 4);
      if (bs < 32) break;
      if (o + 4 + (size_t)bs > u.size()) { truncated = true; break; }
      const uint8_t* r = &u[o];
      uint32_t lrn = r[12], ncig = r[16] | (r[17] << 8);
      int32_t lseq;
      memcpy(&lseq, r + 20, 4);
      size_t a = 36 + lrn + 4 * (size_t)ncig + (size_t)((lseq + 1) / 2) + (size_t)lseq, end = 4 + (size_t)bs;
      std::map<std::string, std::pair<char, ArrowKind>> first;
      while (a + 3 <= end) {
        std::string tag((const char*)r + a, 2);
        char ty = (char)r[a + 2];
        size_t vo = a + 3, sz = 0;
        std::pair<char, ArrowKind> inf{'Z', AK_UTF8};
        if (ty == 'Z' || ty == 'H') {
          size_t k = vo;
          while (k < end && r[k]) k++;
          sz = k - vo + 1;
          inf = {ty, AK_UTF8};
        } else if (ty == 'B') {
          char st = (char)r[vo];
          uint32_t cnt;
          memcpy(&cnt, r + vo + 1, 4);
          size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
          sz = 5 + es * cnt;
          ArrowKind k;
          if (!sam_array_subtype_to_arrow(st, &k)) break;
          inf = {'B', k};
        } else if (ty == 'A') { sz = 1; inf = {'A', AK_UTF8}; }
        else if (ty == 'c' || ty == 'C') { sz = 1; inf = {'i', AK_INT32}; }
        else if (ty == 's' || ty == 'S') { sz = 2; inf = {'i', AK_INT32}; }
        else if (ty == 'i') { sz = 4; inf = {'i', AK_INT32}; }
        else if (ty == 'I') { sz = 4; inf = {'I', AK_UINT32}; }
        else if (ty == 'f') { sz = 4; inf = {'f', AK_FLOAT32}; }
        else break;
        if (!first.count(tag)) first[tag] = inf;
        a = vo + sz;
      }
      for (auto& t : unknown)
        if (!f.count(t) && t.size() == 2 && first.count(t)) f[t] = first[t];
      o += 4 + (size_t)bs;
      count++;
    }
    if ((truncated || (count < sample_size && o + 4 > u.size())) && nb < p.n_blocks()) { nb *= 2; continue; }
    *found = f;
    return;
  }
}

static void determine_schema(Provider& p, const bioscan_bam_options& o) {
  p.fields.clear();
  auto add = [&](const char* n, ArrowKind k, bool nullable) { p.fields.push_back(FieldDef{n, k, nullable, {}}); };
  add("name", AK_UTF8, true);
  add("chrom", AK_UTF8, true);
  add("start", AK_UINT32, true);
  add("end", AK_UINT32, true);
  add("flags", AK_UINT32, false);
  add("cigar", p.binary_cigar ? AK_BINARY : AK_UTF8, false);
  add("mapping_quality", AK_UINT32, false);
  add("mate_chrom", AK_UTF8, true);
  add("mate_start", AK_UINT32, true);
  add("sequence", AK_UTF8, false);
  add("quality_scores", AK_UTF8, false);
  add("template_length", AK_INT32, false);
  if (p.has_tag_fields) {
    std::map<std::string, std::pair<char, ArrowKind>> hints, inferred;
    if (o.tag_type_hints && o.n_tag_type_hints > 0) {
      std::vector<std::string> h;
      for (int i = 0; i < o.n_tag_type_hints; i++) h.push_back(o.tag_type_hints[i]);
      std::string e = parse_tag_type_hints(h, &hints);
      if (!e.empty()) throw Error(e);
    }
    if (o.infer_tag_types) {
      std::vector<std::string> unknown;
      for (auto& t : p.tag_fields) if (!known_tag(t)) unknown.push_back(t);
      if (!unknown.empty()) infer_tags_from_prefix(p, unknown, (size_t)std::max(o.infer_tag_sample_size, 0), &inferred);
    }
    for (auto& tag : p.tag_fields) {
      if (tag.size() != 2) throw Error("Invalid tag name length for " + tag);
      char st;
      ArrowKind k;
      std::string desc;
      const TagDef* kt = known_tag(tag);
      if (inferred.count(tag)) {
        st = inferred[tag].first; k = inferred[tag].second;
        desc = kt ? kt->description : std::string("Tag type discovered from file (") + st + ")";
      } else if (hints.count(tag)) {
        st = hints[tag].first; k = hints[tag].second;
        desc = kt ? kt->description : std::string("Tag type from user hint (") + st + ")";
      } else if (kt) {
        st = kt->sam_type; k = kt->kind; desc = kt->description;
      } else {
        st = 'Z'; k = AK_UTF8; desc = "Unknown tag";
      }
      FieldDef fd{tag, k, true, {}};
      fd.metadata.emplace_back("bio.bam.tag.tag", tag);
      fd.metadata.emplace_back("bio.bam.tag.type", format_sam_tag_type(st, k));
      fd.metadata.emplace_back("bio.bam.tag.description", desc);
      p.fields.push_back(std::move(fd));
    }
  }
  p.metadata = extract_header_metadata(p.hdr);
  p.metadata.emplace_back("bio.coordinate_system_zero_based", p.zero_based ? "true" : "false");
  if (p.binary_cigar) p.metadata.emplace_back("bio.bam.binary_cigar", "true");
}

static bool file_exists(const std::string& s) {
  std::ifstream f(s);
  return f.good();
}

}  // namespace

// Partitions -> devices: contiguous runs in plan order, byte balanced -- partition_byte_ranges_in_order
// (bio-format-core/src/range_planning.rs:147-195) applied to the plan's per-partition estimates.  run_of[i] = run index.
static std::vector<int32_t> shard_in_order(const std::vector<uint64_t>& weights, size_t world) {
  const size_t n = weights.size();
  std::vector<int32_t> run_of(n, 0);
  if (n == 0) return run_of;
  const size_t count = std::min(std::max<size_t>(world, 1), n);
  unsigned __int128 total = 0;
  for (auto w : weights) total += w;
  size_t runs = 0, cur = 0;
  unsigned __int128 assigned = 0;
  for (size_t idx = 0; idx < n; idx++) {
    const size_t remaining_ranges = n - idx, remaining_parts = count - runs;
    const bool share_complete = count > 1 && cur > 0 && assigned * count >= total * (runs + 1);
    const bool must_close = remaining_ranges < remaining_parts;
    if (remaining_parts > 1 && (share_complete || must_close)) { runs++; cur = 0; }
    run_of[idx] = (int32_t)runs;
    cur++;
    assigned += weights[idx];
  }
  return run_of;
}

// =================================================================================================
// C ABI
// =================================================================================================
static std::atomic<int> g_live_providers{0};
struct bioscan_provider {
  Provider p;
  std::unique_ptr<VcfProviderI> vcf;
  bioscan_provider() { g_live_providers.fetch_add(1); }
  ~bioscan_provider() { g_live_providers.fetch_sub(1); }
};
struct bioscan_plan { Plan pl; std::unique_ptr<VcfPlanI> vcf; };
struct bioscan_stream { Stream s; std::unique_ptr<VcfStreamI> vcf; };

#define API_BEGIN try {
#define API_END                                     \
  }                                                 \
  catch (const std::exception& e) {                 \
    g_err = e.what();                               \
    return 1;                                       \
  }                                                 \
  catch (...) {                                     \
    g_err = "unknown error";                        \
    return 1;                                       \
  }                                                 \
  return 0;

extern "C" {

const char* bioscan_last_error(void) { return g_err.c_str(); }

void bioscan_bam_options_default(bioscan_bam_options* o) {
  memset(o, 0, sizeof(*o));
  o->coordinate_system_zero_based = 1;
  o->infer_tag_types = 1;
  o->infer_tag_sample_size = 100;
}

int bioscan_device_check(int32_t device_id, char* name_buf, int32_t cap) {
  API_BEGIN
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) throw Error("no HIP device available: the bioscan scan path requires an AMD GPU (gfx950)");
  if (device_id >= n) throw Error("device ordinal out of range");
  hipDeviceProp_t pr;
  HIP_CHECK(hipGetDeviceProperties(&pr, device_id));
  if (name_buf && cap > 0) snprintf(name_buf, cap, "%s (%s)", pr.name, pr.gcnArchName);
  API_END
}

int bioscan_bam_open(const char* path, const bioscan_bam_options* opts, bioscan_provider** out) {
  API_BEGIN
  bioscan_bam_options o;
  if (opts) o = *opts; else bioscan_bam_options_default(&o);
  std::unique_ptr<bioscan_provider> bp(new bioscan_provider);
  Provider& p = bp->p;
  p.path = path;
  p.device = o.device_id;
  p.zero_based = o.coordinate_system_zero_based != 0;
  p.binary_cigar = o.binary_cigar != 0;
  p.chunk_members = o.chunk_members > 0 ? (uint32_t)o.chunk_members : 0;
  if (o.tag_fields) {
    p.has_tag_fields = true;
    for (int i = 0; i < o.n_tag_fields; i++) p.tag_fields.push_back(o.tag_fields[i]);
  }
  {
    char nm[8];
    if (bioscan_device_check(p.device, nm, sizeof nm)) throw Error(g_err);
  }
  p.set_device();
  p.load_file();
  p.frame();
  // header: inflate leading blocks on the GPU until the header is complete
  {
    uint32_t nb = 1;
    std::string herr;
    for (;;) {
      std::vector<uint8_t> u = p.inflate_prefix_to_host(nb);
      if (parse_bam_header(u.data(), u.size(), &p.hdr, &herr)) break;
      if (!herr.empty()) throw Error("Failed to open BAM: " + herr);
      if (nb >= p.n_blocks()) throw Error("Failed to open BAM: truncated header");
      nb = std::min<uint32_t>(nb * 4, p.n_blocks());
    }
  }
  determine_schema(p, o);
  // index discovery (bio-format-core/src/index_utils.rs:68-83)
  if (o.index_path && o.index_path[0]) {
    p.index_path = o.index_path;
    p.has_index = true;
  } else if (!o.index_path) {
    std::string a = p.path + ".bai";
    std::string stem = p.path;
    size_t dot = stem.rfind('.');
    if (dot != std::string::npos && stem.find('/', dot) == std::string::npos) stem = stem.substr(0, dot);
    std::string b = stem + ".bai";
    if (file_exists(a)) { p.index_path = a; p.has_index = true; }
    else if (file_exists(b)) { p.index_path = b; p.has_index = true; }
    else if (file_exists(p.path + ".csi")) { p.index_path = p.path + ".csi"; p.has_index = true; }  // discover_bam_index: BAI first, then CSI
  }
  if (p.has_index) {
    // The reference keeps the discovered path and reads it with `bam::bai::fs::read` wherever it needs the index: a file that is
    // not a BAI (a .csi companion, a damaged index) leaves the provider open, gives `scan` unit size estimates
    // (storage.rs:344-360) and no no-coor partition (:442-449), and fails every indexed partition when it is executed
    // (`IndexedBamReader::new`, storage.rs:286; physical_exec.rs:879-881).
    std::ifstream f(p.index_path, std::ios::binary);
    std::string e;
    if (!f.good()) e = "cannot read " + p.index_path;
    else {
      std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      if (!parse_bai(d, &p.bai, &e)) p.bai = Bai();
    }
    p.index_error = e;
  }
  *out = bp.release();
  API_END
}

int bioscan_fastq_open(const char* path, int32_t device_id, bioscan_provider** out) {
  API_BEGIN
  std::unique_ptr<bioscan_provider> bp(new bioscan_provider);
  Provider& p = bp->p;
  p.kind = 1;
  p.path = path;
  p.device = device_id;
  {
    char nm[8];
    if (bioscan_device_check(p.device, nm, sizeof nm)) throw Error(g_err);
  }
  p.set_device();
  p.load_file();
  // detect_compression_sync (bio-format-fastq/src/physical_exec.rs:70-89)
  const uint8_t* d = p.file.p;
  if (p.file_len >= 18 && d[0] == 0x1f && d[1] == 0x8b && d[2] == 8 && (d[3] & 4) && d[12] == 0x42 && d[13] == 0x43) {
    p.fq_compression = 1;
    p.frame();
    std::ifstream f(p.path + ".gzi", std::ios::binary);
    if (f.good()) {
      std::vector<uint8_t> g((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      if (g.size() >= 8) {
        uint64_t n;
        memcpy(&n, g.data(), 8);
        if (g.size() >= 8 + 16 * n) {
          for (uint64_t i = 0; i < n; i++) {
            uint64_t c, u;
            memcpy(&c, g.data() + 8 + 16 * i, 8);
            memcpy(&u, g.data() + 16 + 16 * i, 8);
            p.gzi.emplace_back(c, u);
          }
          p.fq_has_gzi = true;
        }
      }
    }
  } else if (p.file_len >= 2 && d[0] == 0x1f && d[1] == 0x8b) {
    throw Error("plain gzip FASTQ (not BGZF) has no block structure to decode in parallel: not supported by the GPU scan");
  } else {
    p.blk_coff = {0, p.file_len};
    p.blk_uoff = {0, 0};
    p.ulen = 0;
  }
  p.fields.clear();
  p.fields.push_back(FieldDef{"name", AK_UTF8, false, {}});
  p.fields.push_back(FieldDef{"description", AK_UTF8, true, {}});
  p.fields.push_back(FieldDef{"sequence", AK_UTF8, false, {}});
  p.fields.push_back(FieldDef{"quality_scores", AK_UTF8, false, {}});
  *out = bp.release();
  API_END
}

void bioscan_vcf_options_default(bioscan_vcf_options* o) {
  memset(o, 0, sizeof(*o));
  o->coordinate_system_zero_based = 1;
}

int bioscan_vcf_open(const char* path, const bioscan_vcf_options* opts, bioscan_provider** out) {
  API_BEGIN
  bioscan_vcf_options o;
  if (opts) o = *opts; else bioscan_vcf_options_default(&o);
  {
    char nm[8];
    if (bioscan_device_check(o.device_id, nm, sizeof nm)) throw Error(g_err);
  }
  std::unique_ptr<bioscan_provider> bp(new bioscan_provider);
  bp->vcf.reset(vcf_open(path, &o));
  *out = bp.release();
  API_END
}

int bioscan_udf_list_avg(const struct ArrowArray* in, const struct ArrowSchema* in_schema, int32_t device_id,
                         struct ArrowArray* out, struct ArrowSchema* out_schema) {
  API_BEGIN
  {
    char nm[8];
    if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(g_err);
  }
  udf_list_avg_host(in, in_schema, device_id, out, out_schema);
  API_END
}
int bioscan_udf_list_cmp(const struct ArrowArray* in, const struct ArrowSchema* in_schema, int32_t op, double threshold,
                         int32_t device_id, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  API_BEGIN
  {
    char nm[8];
    if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(g_err);
  }
  if (op != 0 && op != 1) throw Error("list comparison op must be 0 (list_gte) or 1 (list_lte)");
  udf_list_cmp_host(in, in_schema, op, threshold, device_id, out, out_schema);
  API_END
}
int bioscan_udf_list_and(const struct ArrowArray* a, const struct ArrowSchema* a_schema, const struct ArrowArray* b,
                         const struct ArrowSchema* b_schema, int32_t device_id, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  API_BEGIN
  {
    char nm[8];
    if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(g_err);
  }
  udf_list_and_host(a, a_schema, b, b_schema, device_id, out, out_schema);
  API_END
}
int bioscan_udf_vcf_set_gts(const struct ArrowArray* gt, const struct ArrowSchema* gt_schema, const struct ArrowArray* mask,
                            const struct ArrowSchema* mask_schema, const char* replacement, int32_t device_id,
                            struct ArrowArray* out, struct ArrowSchema* out_schema) {
  API_BEGIN
  {
    char nm[8];
    if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(g_err);
  }
  udf_set_gts_host(gt, gt_schema, mask, mask_schema, replacement, device_id, out, out_schema);
  API_END
}
int bioscan_udf_vcf_allele_stats(const struct ArrowArray* gt, const struct ArrowSchema* gt_schema, const struct ArrowArray* alt,
                                 const struct ArrowSchema* alt_schema, int32_t which, int32_t device_id, struct ArrowArray* out,
                                 struct ArrowSchema* out_schema) {
  API_BEGIN
  char nm[8];
  if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(bioscan_last_error());
  udf_allele_stats_host(gt, gt_schema, alt, alt_schema, which, device_id, out, out_schema);
  API_END
}

int bioscan_stream_list_udf(bioscan_stream* s, const char* field, int32_t udf, double threshold, bioscan_udf_stats* out) {
  API_BEGIN
  if (!s->vcf) throw Error("list UDFs apply to VCF streams");
  if (udf < 0 || udf > 2) throw Error("udf must be 0 (list_avg), 1 (list_gte) or 2 (list_lte)");
  s->vcf->list_udf(field, udf, threshold, out);
  API_END
}

int bioscan_schema(const bioscan_provider* p, struct ArrowSchema* out) {
  API_BEGIN
  if (p->vcf) { p->vcf->schema(out); return 0; }
  export_schema(p->p.fields, p->p.metadata, out);
  API_END
}

int bioscan_supports_filters_pushdown(const bioscan_provider* p, const bioscan_filter* filters, int32_t n, int32_t* out) {
  API_BEGIN
  if (p->vcf) { p->vcf->supports_filters_pushdown(filters, n, out); return 0; }
  auto fs = copy_filters(filters, n);
  for (int32_t i = 0; i < n; i++) {
    if (p->p.has_index && is_genomic_coordinate_filter(fs[i])) out[i] = 1;
    else if (can_push_down_record_filter(fs[i], p->p.fields)) out[i] = 1;
    else out[i] = 0;
  }
  API_END
}

int bioscan_scan(const bioscan_provider* cp, const int32_t* projection, int32_t n_projection, const bioscan_filter* filters,
                 int32_t n_filters, int64_t limit, int32_t target_partitions, bioscan_plan** out) {
  API_BEGIN
  if (cp->vcf) {
    std::unique_ptr<bioscan_plan> vp(new bioscan_plan);
    vp->vcf.reset(const_cast<bioscan_provider*>(cp)->vcf->scan(projection, n_projection, filters, n_filters, limit, target_partitions));
    *out = vp.release();
    return 0;
  }
  Provider& p = const_cast<Provider&>(cp->p);
  std::unique_ptr<bioscan_plan> bp(new bioscan_plan);
  Plan& pl = bp->pl;
  pl.prov = &p;
  pl.limit = limit;
  if (projection) {
    pl.has_projection = true;
    for (int i = 0; i < n_projection; i++) {
      if (projection[i] < 0 || (size_t)projection[i] >= p.fields.size()) throw Error("projection index out of range");
      pl.projection.push_back(projection[i]);
      pl.out_fields.push_back(p.fields[projection[i]]);
    }
    for (size_t a = 0; a < pl.projection.size(); a++)
      for (size_t b = a + 1; b < pl.projection.size(); b++)
        if (pl.projection[a] == pl.projection[b]) throw Error("duplicate column in projection");
  } else {
    pl.out_fields = p.fields;
  }
  if (p.kind == 1) {
    // detect_local_strategy (bio-format-fastq/src/physical_exec.rs:94-138); filters are ignored (scan's `_filters`)
    const size_t target = (size_t)std::max(target_partitions, 0);
    if (p.fq_compression == 1) {
      if (p.fq_has_gzi) {
        // get_bgzf_partition_bounds (:140-175)
        std::vector<std::pair<uint64_t, uint64_t>> blocks{{0, 0}};
        blocks.insert(blocks.end(), p.gzi.begin(), p.gzi.end());
        const size_t nbk = blocks.size(), nparts = std::min(target, nbk);
        pl.fq_strategy = 1;
        if (nparts == 0) pl.fq_parts.push_back({0, ~0ull});
        size_t cur = 0;
        for (size_t i = 0; i < nparts && cur < nbk; i++) {
          const size_t cnt = nbk / nparts + (i < nbk % nparts ? 1 : 0);
          const size_t nxt = cur + cnt;
          pl.fq_parts.push_back({blocks[cur].second, nxt >= nbk ? ~0ull : blocks[nxt].first});
          cur = nxt;
        }
      }
    } else {
      const uint64_t fsz = p.file_len;
      if (fsz != 0 && target > 1 && fsz / target != 0) {
        const uint64_t chunk = fsz / target;
        pl.fq_strategy = 2;
        for (size_t i = 0; i < target; i++) pl.fq_parts.push_back({i * chunk, i + 1 == target ? fsz : (i + 1) * chunk});
      }
    }
    *out = bp.release();
    return 0;
  }
  auto fs = copy_filters(filters, n_filters);
  if (p.has_index) {
    std::vector<GenomicRegion> regions;
    bool unsat = false;
    extract_genomic_regions(fs, p.zero_based, &regions, &unsat);
    if (unsat) {
      pl.empty = true;
      *out = bp.release();
      return 0;
    }
    const bool full = regions.empty();
    if (regions.empty())
      for (auto& n : p.hdr.ref_names) { GenomicRegion r; r.chrom = n; regions.push_back(r); }
    if (!regions.empty()) {
      auto est = estimate_sizes_from_bai(p.index_error.empty() ? &p.bai : nullptr, regions, p.hdr.ref_names, p.hdr.ref_lengths);
      pl.assignments = balance_partitions(est, (size_t)std::max(target_partitions, 0));
      if (full && p.index_error.empty() && p.bai.has_no_coor && p.bai.n_no_coor > 0) {
        PartitionAssignment a;
        GenomicRegion r;
        r.chrom = "*";
        r.unmapped_tail = true;
        a.regions.push_back(r);
        a.total_estimated_bytes = std::max<uint64_t>(p.bai.n_no_coor, 1);
        pl.assignments.push_back(a);
      }
      for (auto& f : fs) if (can_push_down_record_filter(f, p.fields)) pl.residual.push_back(f);
      pl.indexed = true;
    }
  }
  *out = bp.release();
  API_END
}

int32_t bioscan_plan_num_partitions(const bioscan_plan* plan) { return plan->vcf ? plan->vcf->n_partitions() : plan->pl.n_partitions(); }

int bioscan_plan_schema(const bioscan_plan* plan, struct ArrowSchema* out) {
  API_BEGIN
  if (plan->vcf) { plan->vcf->schema(out); return 0; }
  export_schema(plan->pl.out_fields, plan->pl.prov->metadata, out);
  API_END
}

int32_t bioscan_plan_display(const bioscan_plan* plan, char* buf, int32_t cap) {
  if (plan->vcf) {
    std::string v = plan->vcf->display();
    if (buf && cap > 0) snprintf(buf, cap, "%s", v.c_str());
    return (int32_t)v.size();
  }
  std::string s = plan->pl.prov->kind == 1 ? "FastqExec: projection=[" : "BamExec: projection=[";
  if (plan->pl.has_projection) {
    for (size_t i = 0; i < plan->pl.out_fields.size(); i++) {
      if (i) s += ", ";
      s += plan->pl.out_fields[i].name;
    }
  } else {
    s += "*";
  }
  s += "]";
  if (buf && cap > 0) snprintf(buf, cap, "%s", s.c_str());
  return (int32_t)s.size();
}

int32_t bioscan_plan_partition_desc(const bioscan_plan* plan, int32_t partition, char* buf, int32_t cap) {
  std::string s;
  if (plan->vcf) {
    try { s = plan->vcf->partition_desc(partition); } catch (...) { s = "sequential"; }
  } else if (plan->pl.indexed && partition >= 0 && (size_t)partition < plan->pl.assignments.size())
    s = describe_partition(plan->pl.assignments[partition]);
  else s = "sequential";
  if (buf && cap > 0) snprintf(buf, cap, "%s", s.c_str());
  return (int32_t)s.size();
}

static int execute_impl(const bioscan_plan* plan, int32_t partition, int32_t batch_size, bool device_only, bioscan_scan_stats* stats,
                        bioscan_stream** out) {
  API_BEGIN
  if (plan->vcf) {
    std::unique_ptr<bioscan_stream> vs(new bioscan_stream);
    vs->vcf.reset(plan->vcf->execute(partition, batch_size, device_only, stats));
    *out = vs.release();
    return 0;
  }
  if (partition < 0 || partition >= plan->pl.n_partitions()) throw Error("partition index out of range");
  if (batch_size <= 0) throw Error("batch_size must be positive");
  std::unique_ptr<bioscan_stream> bs(new bioscan_stream);
  Stream& sm = bs->s;
  sm.prov = plan->pl.prov;
  sm.on_host = !device_only;
  if (plan->pl.indexed && !plan->pl.prov->index_error.empty())
    throw Error("Failed to open indexed BAM: " + plan->pl.prov->index_error);
  if (plan->pl.prov->kind == 1 && (plan->pl.prov->fq_compression != 1 || device_only)) {
    // plain text (the whole file is the resident image) and device-resident executions: one result
    sm.cur = run_partition_fastq(plan->pl, partition, (uint32_t)batch_size, !device_only);
    sm.stats = sm.cur->stats;
    if (device_only) { sm.device_results.push_back(sm.cur); sm.cur.reset(); }
  } else {
    if (plan->pl.prov->kind == 1) sm.exec.reset(new FastqExecState(plan->pl, partition, (uint32_t)batch_size, true));
    else sm.exec.reset(new BamExecState(plan->pl, partition, (uint32_t)batch_size, !device_only));
    if (device_only) {
      // the whole partition now, every chunk's Arrow buffers left in HBM
      while (auto r = sm.exec->next_chunk()) sm.device_results.push_back(std::move(r));
      sm.stats = sm.exec->total;
      sm.exec.reset();
    }
  }
  if (stats) *stats = sm.stats;
  *out = bs.release();
  API_END
}

int bioscan_execute(const bioscan_plan* plan, int32_t partition, int32_t batch_size, bioscan_stream** out) {
  return execute_impl(plan, partition, batch_size, false, nullptr, out);
}
int bioscan_execute_device(const bioscan_plan* plan, int32_t partition, int32_t batch_size, bioscan_scan_stats* stats, bioscan_stream** out) {
  return execute_impl(plan, partition, batch_size, true, stats, out);
}

int bioscan_next(bioscan_stream* s, struct ArrowArray* out, int32_t* has_batch) {
  API_BEGIN
  if (s->vcf) {
    *has_batch = s->vcf->next(out) ? 1 : 0;
    return 0;
  }
  Stream& st = s->s;
  if (!st.on_host) throw Error("stream was executed device-only; no host batches to export");
  *has_batch = st.next(out) ? 1 : 0;
  API_END
}

void bioscan_stream_close(bioscan_stream* s) { delete s; }
void bioscan_plan_close(bioscan_plan* p) { delete p; }
void bioscan_provider_close(bioscan_provider* p) {
  if (!p) return;
  delete p;
  // the caches are shared by every provider of the process: they are given back when the LAST one closes, never under a
  // provider that is still executing
  if (g_live_providers.load() == 0) {
    dev_pool_trim();
    host_pool_trim();
  }
}

int bioscan_provider_set_chunk_members(bioscan_provider* p, int32_t chunk_members) {
  API_BEGIN
  if (chunk_members < 0) throw Error("chunk_members must not be negative");
  if (p->vcf) p->vcf->set_chunk_members((uint32_t)chunk_members);
  else p->p.chunk_members = (uint32_t)chunk_members;
  API_END
}

int bioscan_provider_make_resident(bioscan_provider* p) {
  API_BEGIN
  if (p->vcf) p->vcf->make_resident();
  else if (p->p.kind == 1) p->p.make_resident();
  else p->p.device_image(p->p.device, 0, p->p.n_blocks());
  API_END
}

int bioscan_scan_devices(const bioscan_provider* cp, const int32_t* projection, int32_t n_projection, const bioscan_filter* filters,
                         int32_t n_filters, int64_t limit, int32_t target_partitions, const int32_t* device_ids, int32_t n_devices,
                         bioscan_plan** out) {
  API_BEGIN
  if (n_devices < 1 || !device_ids) throw Error("bioscan_scan_devices needs at least one device");
  if (cp->vcf || cp->p.kind != 0) {
    if (n_devices != 1) throw Error("multi-device plans are implemented for BAM providers");
    return bioscan_scan(cp, projection, n_projection, filters, n_filters, limit, target_partitions, out);
  }
  for (int32_t d = 0; d < n_devices; d++) {
    char nm[8];
    if (bioscan_device_check(device_ids[d], nm, sizeof nm)) throw Error(g_err);
  }
  bioscan_plan* pl = nullptr;
  if (bioscan_scan(cp, projection, n_projection, filters, n_filters, limit, target_partitions, &pl)) throw Error(g_err);
  std::unique_ptr<bioscan_plan> hold(pl);
  const int np = pl->pl.n_partitions();
  std::vector<uint64_t> w((size_t)np, 1);
  if (pl->pl.indexed)
    for (int i = 0; i < np; i++) w[(size_t)i] = pl->pl.assignments[(size_t)i].total_estimated_bytes;
  const auto run_of = shard_in_order(w, (size_t)n_devices);
  pl->pl.part_device.resize((size_t)np);
  for (int i = 0; i < np; i++) pl->pl.part_device[(size_t)i] = device_ids[run_of[(size_t)i]];
  *out = hold.release();
  API_END
}

int32_t bioscan_plan_partition_device(const bioscan_plan* plan, int32_t partition) {
  if (plan->vcf) return 0;
  if (partition < 0 || partition >= plan->pl.n_partitions()) return -1;
  return plan->pl.device_of(partition);
}

int bioscan_plan_make_resident(const bioscan_plan* plan, const int32_t* partitions, int32_t n) {
  API_BEGIN
  if (plan->vcf) throw Error("bioscan_plan_make_resident: use bioscan_provider_make_resident for VCF providers");
  const Plan& pl = plan->pl;
  Provider& p = *pl.prov;
  if (p.kind == 1) { p.make_resident(); return 0; }
  std::vector<FilterTerm> terms;
  (void)build_terms(pl, &terms);
  std::map<int, std::pair<uint32_t, uint32_t>> span;  // device -> members
  const int np = pl.n_partitions();
  for (int k = 0; k < (partitions ? n : np); k++) {
    const int part = partitions ? partitions[k] : k;
    if (part < 0 || part >= np) throw Error("partition index out of range");
    uint32_t lo = 0, hi = 0;
    work_span(build_work(pl, part, terms.size()), &lo, &hi);
    if (hi <= lo) continue;
    auto it = span.find(pl.device_of(part));
    if (it == span.end()) span[pl.device_of(part)] = {lo, hi};
    else { it->second.first = std::min(it->second.first, lo); it->second.second = std::max(it->second.second, hi); }
  }
  for (auto& kv : span) p.device_image(kv.first, kv.second.first, kv.second.second);
  API_END
}

int bioscan_provider_resident_range(const bioscan_provider* cp, int32_t device_id, uint64_t* lo, uint64_t* hi) {
  API_BEGIN
  *lo = 0; *hi = 0;
  if (cp->vcf) throw Error("bioscan_provider_resident_range: BAM / FASTQ providers");
  Provider& p = const_cast<Provider&>(cp->p);
  if (p.resident && device_id == p.device) { *hi = p.file_len; return 0; }
  auto img = p.image_of(device_id);
  if (img) { *lo = p.blk_coff[img->m_lo]; *hi = p.blk_coff[img->m_hi]; }
  API_END
}

int32_t bioscan_debug_extract_regions(const bioscan_filter* filters, int32_t n_filters, int32_t zero_based, char* buf, int32_t cap) {
  std::string o;
  try {
    auto fs = copy_filters(filters, n_filters);
    std::vector<GenomicRegion> regions;
    bool unsat = false;
    extract_genomic_regions(fs, zero_based != 0, &regions, &unsat);
    for (auto& r : regions) {
      if (!o.empty() && o.back() != '|') o += ";";
      o += r.chrom + ":" + (r.has_start ? std::to_string(r.start) : "") + "-" + (r.has_end ? std::to_string(r.end) : "");
    }
    int residual = 0, genomic = 0;
    for (auto& f : fs) { if (is_genomic_coordinate_filter(f)) genomic++; else residual++; }
    o += "|unsat=" + std::to_string(unsat ? 1 : 0) + "|genomic=" + std::to_string(genomic) + "|residual=" + std::to_string(residual);
  } catch (const std::exception& e) {
    o = std::string("error: ") + e.what();
  }
  if (buf && cap > 0) snprintf(buf, cap, "%s", o.c_str());
  return (int32_t)o.size();
}

int32_t bioscan_debug_shard_partitions(const uint64_t* weights, int32_t n, int32_t world, int32_t* run_of) {
  std::vector<uint64_t> w(weights, weights + std::max(n, 0));
  const auto r = shard_in_order(w, (size_t)std::max(world, 1));
  int32_t runs = 0;
  for (int32_t i = 0; i < n; i++) { run_of[i] = r[(size_t)i]; runs = std::max(runs, r[(size_t)i] + 1); }
  return runs;
}

void bioscan_free(void* p) { free(p); }

int bioscan_bgzf_inflate(const uint8_t* data, size_t len, int32_t device_id, int32_t check_crc, uint8_t** out, size_t* out_len,
                         double* kernel_ms) {
  API_BEGIN
  {
    char nm[8];
    if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(g_err);
  }
  Provider p;
  p.device = device_id;
  p.set_device();
  p.file_len = len;
  p.file.alloc(len + 4096);
  memcpy(p.file.p, data, len);
  memset(p.file.p + len, 0, 4096);
  p.frame();
  p.make_resident();
  DevBuf<uint8_t> u(p.ulen + 64);
  StageTimer t(p.stream);
  t.start();
  p.launch_inflate(u.p, p.n_blocks());
  double ms = t.stop();
  if (check_crc) p.launch_crc(u.p, p.n_blocks());
  HIP_CHECK(hipStreamSynchronize(p.stream));
  p.check_inflate_status(0, p.n_blocks());
  uint8_t* h = (uint8_t*)malloc(p.ulen ? p.ulen : 1);
  if (!h) throw Error("out of memory");
  if (p.ulen) HIP_CHECK(hipMemcpy(h, u.p, p.ulen, hipMemcpyDeviceToHost));
  *out = h;
  *out_len = p.ulen;
  if (kernel_ms) *kernel_ms = ms;
  API_END
}

static int32_t write_plan(const std::vector<PartitionAssignment>& parts, char* buf, int32_t cap) {
  std::string o;
  for (auto& p : parts) { o += describe_partition(p); o += "\n"; }
  if (buf && cap > 0) snprintf(buf, cap, "%s", o.c_str());
  return (int32_t)o.size();
}

int32_t bioscan_debug_balance_partitions(int32_t n, const char* const* chroms, const uint64_t* region_start,
                                         const uint64_t* region_end, const uint64_t* est_bytes, const uint64_t* contig_len,
                                         const uint64_t* unmapped, const uint64_t* const* bins, const int32_t* n_bins,
                                         uint64_t leaf_span, int32_t target_partitions, char* buf, int32_t cap) {
  std::vector<RegionSizeEstimate> est;
  for (int32_t i = 0; i < n; i++) {
    RegionSizeEstimate e;
    e.region.chrom = chroms[i];
    if (region_start && region_start[i]) { e.region.has_start = true; e.region.start = region_start[i]; }
    if (region_end && region_end[i]) { e.region.has_end = true; e.region.end = region_end[i]; }
    e.estimated_bytes = est_bytes[i];
    if (contig_len && contig_len[i]) { e.has_contig_length = true; e.contig_length = contig_len[i]; }
    e.unmapped_count = unmapped ? unmapped[i] : 0;
    if (bins && n_bins) e.nonempty_bin_positions.assign(bins[i], bins[i] + n_bins[i]);
    e.leaf_bin_span = leaf_span;
    est.push_back(std::move(e));
  }
  return write_plan(balance_partitions(est, (size_t)std::max(target_partitions, 0)), buf, cap);
}

int32_t bioscan_debug_plan_full_scan(const char* bai_path, int32_t n_ref, const char* const* ref_names, const int64_t* ref_lengths,
                                     int32_t target_partitions, char* buf, int32_t cap) {
  std::ifstream f(bai_path, std::ios::binary);
  if (!f.good()) return -1;
  std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  Bai bai;
  std::string e;
  if (!parse_bai(d, &bai, &e)) return -1;
  std::vector<std::string> names;
  std::vector<int64_t> lens;
  std::vector<GenomicRegion> regions;
  for (int32_t i = 0; i < n_ref; i++) {
    names.push_back(ref_names[i]);
    lens.push_back(ref_lengths[i]);
    GenomicRegion r;
    r.chrom = ref_names[i];
    regions.push_back(r);
  }
  auto parts = balance_partitions(estimate_sizes_from_bai(&bai, regions, names, lens), (size_t)std::max(target_partitions, 0));
  if (bai.has_no_coor && bai.n_no_coor > 0) {
    PartitionAssignment a;
    GenomicRegion r;
    r.chrom = "*";
    r.unmapped_tail = true;
    a.regions.push_back(r);
    a.total_estimated_bytes = std::max<uint64_t>(bai.n_no_coor, 1);
    parts.push_back(a);
  }
  return write_plan(parts, buf, cap);
}

}  // extern "C"
