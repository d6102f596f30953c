// engine.cpp -- BamTableProvider / BamExec / stream mirror over the HIP kernels + the C ABI.
//
// Mirrors (names and argument meaning) bio-format-bam/src/table_provider.rs:381-529 (new),
// :941-962 (supports_filters_pushdown), :964-1115 (scan) and bio-format-bam/src/physical_exec.rs
// :108-172 (execute), :371-598 (sequential scan), :864-1372 (indexed scan).  There is NO CPU
// decode path in this library: every inflate / record walk / field extract runs on the GPU and
// the library refuses to open a file when no HIP device is usable.
#include <algorithm>
#include <cerrno>
#include <chrono>
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
const EnvKnobs& env_knobs() {
  static const EnvKnobs k = [] {
    EnvKnobs v;
    if (const char* e = getenv("BIOSCAN_DEBUG")) v.debug = e[0] && e[0] != '0';
    if (const char* e = getenv("BIOSCAN_LAPS")) v.laps = e[0] && e[0] != '0';
    if (const char* e = getenv("BIOSCAN_K1_WAVES_PER_CU")) v.k1_waves_per_cu = atoi(e);
    if (const char* e = getenv("BIOSCAN_HOST_POOL_GB")) v.host_pool_gb = atof(e);
    if (const char* e = getenv("BIOSCAN_DEV_POOL_GB")) v.dev_pool_gb = atof(e);
    if (const char* e = getenv("BIOSCAN_CHUNK_MEMBERS")) v.chunk_members = (uint32_t)std::max(64, atoi(e));
    return v;
  }();
  return k;
}
static std::mutex g_pool_mu;
static std::multimap<std::pair<int, size_t>, void*> g_pool;  // (device, bytes) -> cached block
static size_t g_pool_bytes = 0;
static size_t dev_pool_limit() { return (size_t)(env_knobs().dev_pool_gb * (double)(1ull << 30)); }
constexpr size_t POOL_MIN = 1;  // every block is cached: hipFree of even a tiny block synchronises the device
static int cur_device() { int d = 0; (void)hipGetDevice(&d); return d; }
void* dev_pool_alloc(size_t bytes) {
  if (bytes >= POOL_MIN) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool.find({cur_device(), bytes});
    if (it != g_pool.end()) {
      void* p = it->second;
      g_pool.erase(it);
      g_pool_bytes -= bytes;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess && bytes >= POOL_MIN) {  // out of memory: give cached blocks back and retry once
    dev_pool_trim();
    e = hipMalloc(&p, bytes);
  }
  HIP_CHECK(e);
  return p;
}
void dev_pool_free(void* p, size_t bytes, int device) {
  // `device` is the one the block was allocated on (recorded by DevBuf): a block is freed from Arrow release
  // callbacks and destructors on threads whose current device may be anything.
  // device < 0: released while an exception unwinds -- kernels that read the block may still be in flight, so it goes
  // back to the driver (hipFree waits for the device) instead of into the cache where another stream could pick it up
  if (bytes >= POOL_MIN && device >= 0) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool_bytes + bytes <= dev_pool_limit()) {
      g_pool.emplace(std::make_pair(device, bytes), p);
      g_pool_bytes += bytes;
      return;
    }
  }
  (void)hipFree(p);  // over the cap (or unwinding): back to the driver
}
int dev_pool_device() { return cur_device(); }
void dev_pool_trim() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto& kv : g_pool) (void)hipFree(kv.second);  // hipFree takes a pointer of any device
  g_pool.clear();
  g_pool_bytes = 0;
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
static size_t host_pool_limit() { return (size_t)(env_knobs().host_pool_gb * (double)(1ull << 30)); }
void* host_pool_alloc(size_t bytes, size_t* cap, bool* pinned) {
  const size_t c = host_class(bytes);
  {
    std::lock_guard<std::mutex> lk(g_hpool_mu);
    auto it = g_hpool.find(c);
    if (it != g_hpool.end()) {
      void* p = it->second;
      g_hpool.erase(it);
      g_hpool_bytes -= c;
      *cap = c; *pinned = false;
      return p;
    }
  }
  // Plain (pageable) memory on purpose -- measured on this platform for 4 GiB (tools/experiments/d2h_paths.cpp): a
  // fresh pinned block costs 0.60 s to allocate and 0.39 s to free around a 0.08 s copy; a fresh malloc'd block takes
  // the copy at 17 GB/s (0.25 s, page faults included) and, once its pages have been touched, at the same 52 GB/s as
  // pinned memory.  Cached blocks keep their pages, so the steady state is the full link rate either way.
  void* p = malloc(c);
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
    if (!pinned && g_hpool_bytes + cap <= host_pool_limit()) {
      g_hpool.emplace(cap, p);
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
  bool operator==(const DecodeRange& o) const {
    return b_lo == o.b_lo && b_hi == o.b_hi && first_rel == o.first_rel && stop_rel == o.stop_rel;
  }
};

// -------------------------------------------------------------------------------------------------
// Provider
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

  BamHeader hdr;
  std::vector<FieldDef> fields;  // full schema
  std::vector<std::pair<std::string, std::string>> metadata;

  bool has_index = false;
  std::string index_path;
  Bai bai;

  bool decoded = false;
  bool have_keys = false;
  DecodeRange dec_range;
  DevBuf<uint64_t> d_rec_off;
  uint64_t n_rec = 0;
  DevBuf<int32_t> k_refid, k_pos, k_end1;
  DevBuf<uint32_t> k_fm;
  DevBuf<uint8_t> d_ref_names;
  DevBuf<uint32_t> d_ref_name_off, d_ref_name_len;
  bioscan_scan_stats decode_stats{};

  DecodeRange whole_file() const {
    DecodeRange r;
    r.b_lo = 0; r.b_hi = n_blocks();
    r.first_rel = hdr.first_record_offset; r.stop_rel = ulen;
    return r;
  }

  void upload_ref_names() {
    std::vector<uint32_t> off{0}, len;
    std::string blob;
    for (auto& n : hdr.ref_names) {
      blob += n;
      off.push_back((uint32_t)blob.size());
      len.push_back((uint32_t)n.size());
    }
    d_ref_names.alloc(std::max<size_t>(blob.size(), 1));
    d_ref_name_off.alloc(off.size());
    d_ref_name_len.alloc(std::max<size_t>(len.size(), 1));
    if (!blob.empty()) HIP_CHECK(hipMemcpy(d_ref_names.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_ref_name_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    if (!len.empty()) HIP_CHECK(hipMemcpy(d_ref_name_len.p, len.data(), len.size() * 4, hipMemcpyHostToDevice));
  }

  // Full decode: inflate every block, find every record, build the key table.  Cached.
  // Decode members [r.b_lo, r.b_hi): inflate, CRC-check, find every record that starts in
  // [r.first_rel, r.stop_rel) of the range's inflated bytes, build the key table.  One range is cached.
  // caller holds `mu` and keeps holding it until the last kernel that reads d_u / the record table has finished
  void decode(bool force, const DecodeRange& r, bool need_keys) {
    if (decoded && !force && r == dec_range && (have_keys || !need_keys)) return;
    decode_locked(r, need_keys);
  }
  void decode_locked(const DecodeRange& r, bool need_keys) {
    make_resident();
    set_device();
    decoded = false;
    const uint32_t nb_r = r.b_hi - r.b_lo;
    const uint64_t ulen = blk_uoff[r.b_hi] - blk_uoff[r.b_lo];  // shadows the file total on purpose
    const uint64_t stop = r.stop_rel;
    bioscan_scan_stats s{};
    s.n_blocks = nb_r;
    s.compressed_bytes = blk_coff[r.b_hi] - blk_coff[r.b_lo];
    s.inflated_bytes = ulen;
    StageTimer t(stream), tt(stream);
    tt.start();
    if (d_u.n < ulen + 64) d_u.alloc(ulen + 64);
    t.start();
    launch_inflate(d_u.p, nb_r, r.b_lo);
    s.ms_inflate = t.stop();
    report_v2_debug(nb_r);
    // CRC32 validation (noodles-bgzf checks every block).  Measured: running it on a second, low-priority
    // stream beside the chain / extract kernels does not shorten the step on MI355X (the extract kernels are
    // bandwidth-bound and the CRC kernel just time-slices with them), so it stays in line.
    t.start();
    launch_crc(d_u.p, nb_r, r.b_lo);
    s.ms_crc = t.stop();
    check_inflate_status(r.b_lo, nb_r);

    // ---- record chain: records starting in [first_rec, stop) ----
    t.start();
    const uint64_t first_rec = r.first_rel;
    if (first_rec > stop || stop > ulen) throw Error("BAM read error: record range extends past end of data");
    const uint64_t nseg = std::max<uint64_t>((stop + SEG_BYTES - 1) / SEG_BYTES, 1);
    DevBuf<uint64_t> entry(nseg), exit_(nseg), base(nseg + 1), tmp(scan_tmp_elems(nseg));
    DevBuf<uint32_t> count(nseg), dirty(nseg), ctr(2);
    HIP_CHECK(hipMemsetAsync(ctr.p, 0, 8, stream));
    HIP_CHECK(hipMemsetAsync(dirty.p, 0, nseg * 4, stream));
    ChainBuffers cb{entry.p, exit_.p, count.p, dirty.p, ctr.p, ctr.p + 1};
    launch_seg_guess(d_u.p, stop, first_rec, nseg, (int32_t)hdr.ref_names.size(), cb, stream);
    launch_seg_walk(d_u.p, stop, nseg, cb, 0, stream);
    for (int iter = 0;; iter++) {
      s.chain_iterations = (uint64_t)iter + 1;
      HIP_CHECK(hipMemsetAsync(ctr.p, 0, 4, stream));
      launch_seg_verify(stop, first_rec, nseg, cb, stream);
      uint32_t nfix = 0;
      HIP_CHECK(hipMemcpyAsync(&nfix, ctr.p, 4, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      if (env_knobs().debug) fprintf(stderr, "[bioscan] record chain verify round %d: %u segment(s) corrected of %llu\n", iter, nfix, (unsigned long long)nseg);
      if (nfix == 0) break;
      if ((uint64_t)iter > nseg + 2) throw Error("record boundary scan did not converge");
      launch_seg_walk(d_u.p, stop, nseg, cb, 1, stream);
    }
    launch_exclusive_scan_u32_to_u64(count.p, base.p, nseg, tmp.p, stream);
    uint64_t total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, base.p + nseg, 8, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    n_rec = total;
    if (d_rec_off.n < n_rec + 1) d_rec_off.alloc(n_rec + 1);
    launch_seg_emit(d_u.p, stop, nseg, cb, base.p, d_rec_off.p, stream);
    // last exit must be exactly the end of the stream
    {
      DevBuf<unsigned long long> lastx(2);
      launch_last_exit(exit_.p, nseg, lastx.p, stream);
      unsigned long long lx[2] = {0, 0};
      uint32_t errf = 0;
      HIP_CHECK(hipMemcpyAsync(&errf, ctr.p + 1, 4, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipMemcpyAsync(lx, lastx.p, 16, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      if (errf == 2) throw Error("BAM read error: invalid record (variable-length fields exceed block_size)");
      if (errf) throw Error("BAM read error: truncated or corrupt record (invalid block_size)");
      const uint64_t last = lx[1];
      if (last == SEG_BAD) throw Error("BAM read error: truncated or corrupt record (invalid block_size)");
      if (first_rec < stop && last != stop) throw Error("BAM read error: unexpected end of record stream");
    }
    s.ms_chain = t.stop();
    s.n_records = n_rec;
    // ---- key table ----
    have_keys = false;
    if (need_keys) {
      if (k_refid.n < n_rec) {
        k_refid.alloc(n_rec); k_pos.alloc(n_rec); k_end1.alloc(n_rec); k_fm.alloc(n_rec);
      }
      RecKeys rk{k_refid.p, k_pos.p, k_end1.p, k_fm.p};
      t.start();
      launch_rec_keys(d_u.p, d_rec_off.p, n_rec, rk, stream);
      s.ms_keys = t.stop();
      have_keys = true;
    }
    if (!d_ref_name_off.p) upload_ref_names();
    s.ms_total_gpu = tt.stop();
    decode_stats = s;
    dec_range = r;
    decoded = true;
  }
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
  // member / byte range of each partition (BAI chunk queries + block lookups), computed on first use: re-planning it on
  // every execute cost 1-3 ms of host time per partition
  mutable std::mutex range_mu;
  mutable std::vector<std::pair<bool, DecodeRange>> range_cache;
  int n_partitions() const {
    if (prov && prov->kind == 1) return fq_strategy == 0 ? 1 : (int)fq_parts.size();
    return empty ? 0 : (indexed ? (int)assignments.size() : 1);
  }
};

// -------------------------------------------------------------------------------------------------
// Result columns
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
  uint64_t total_bytes = 0;   // var: bytes; list: elements
  // host
  HostBuf h_values, h_off32, h_valid;
  std::vector<uint64_t> h_batch_base;  // per batch: first byte / element
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
  std::vector<Column> cols;
  bool on_host = false;
  bioscan_scan_stats stats{};
  uint64_t n_batches() const { return (n_rows + batch_size - 1) / batch_size; }
};

struct Stream {
  std::shared_ptr<Result> res;
  uint64_t next = 0;
  Provider* prov = nullptr;
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
      if (f.values.size() > 8) throw Error("IN list longer than 8 literals is not supported by the device filter");
      int n = 0;
      for (auto& l : f.values) {
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

// The members a partition has to inflate: the span of the BAI chunks of its regions.
// Mapped regions -> noodles' merged chunk list (reg2bins, linear-index floor).  The no-coor
// partition starts after the last placed record (coordinate-sorted file: everything a BAI indexes).
// Per-reference unmapped tails scan "from the reference's last chunk until the reference changes",
// which is only decidable by looking at every later record: those partitions decode the whole file.
static DecodeRange partition_range_uncached(const Plan& plan, int partition);
static DecodeRange partition_range(const Plan& plan, int partition) {
  if (!plan.indexed) return plan.prov->whole_file();
  std::lock_guard<std::mutex> lk(plan.range_mu);
  if (plan.range_cache.size() != plan.assignments.size()) plan.range_cache.assign(plan.assignments.size(), {false, DecodeRange{}});
  auto& slot = plan.range_cache[(size_t)partition];
  if (!slot.first) { slot.second = partition_range_uncached(plan, partition); slot.first = true; }
  return slot.second;
}
static DecodeRange partition_range_uncached(const Plan& plan, int partition) {
  const Provider& p = *plan.prov;
  if (!plan.indexed) return p.whole_file();
  uint64_t lo = ~0ull, hi = 0;
  bool to_eof = false;
  for (auto& r : plan.assignments[partition].regions) {
    if (r.unmapped_tail) {
      if (r.chrom != "*") return p.whole_file();
      if (!(p.bai.has_no_coor && p.bai.n_no_coor > 0)) continue;
      uint64_t seek = 0;
      for (auto& rf : p.bai.refs)
        for (auto& b : rf.bins)
          for (auto& c : b.second) seek = std::max(seek, c.second);
      if (seek == 0) return p.whole_file();
      lo = std::min(lo, seek);
      to_eof = true;
      continue;
    }
    long ref = -1;
    for (size_t i = 0; i < p.hdr.ref_names.size(); i++) if (p.hdr.ref_names[i] == r.chrom) { ref = (long)i; break; }
    if (ref < 0) throw Error("BAM region query failed: region reference sequence does not exist in reference sequences: " + r.chrom);
    auto chunks = bai_query_chunks(p.bai, (size_t)ref, r.has_start, r.start, r.has_end, r.end);
    for (auto& c : chunks) { lo = std::min(lo, c.first); hi = std::max(hi, c.second); }
  }
  DecodeRange d;
  if (lo == ~0ull) return d;  // nothing to read
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

// Select the rows of one partition, in region order (file order inside a region).
static void select_rows(const Plan& plan, int partition, DevBuf<uint64_t>* rows_owned, const uint64_t** rows, uint64_t* n_rows) {
  Provider& p = *plan.prov;
  hipStream_t st = p.stream;
  if (!plan.indexed) {
    *rows = p.d_rec_off.p;
    *n_rows = p.n_rec;
    return;
  }
  const auto& regions = plan.assignments[partition].regions;
  std::vector<FilterTerm> terms;
  const bool satisfiable = build_terms(plan, &terms);
  DevBuf<FilterTerm> d_terms(std::max<size_t>(terms.size(), 1));
  if (!terms.empty()) HIP_CHECK(hipMemcpy(d_terms.p, terms.data(), terms.size() * sizeof(FilterTerm), hipMemcpyHostToDevice));
  const uint64_t n = p.n_rec;
  RecKeys rk{p.k_refid.p, p.k_pos.p, p.k_end1.p, p.k_fm.p};
  DevBuf<uint32_t> keep(std::max<uint64_t>(n, 1));
  DevBuf<uint64_t> kscan(n + 1), tmp(scan_tmp_elems(n));
  DevBuf<unsigned long long> d_idx(2);
  std::vector<RowSelect> sels;
  for (auto& r : regions) {
    RowSelect s{};
    s.zero_based = p.zero_based ? 1 : 0;
    s.n_terms = (int32_t)terms.size();
    if (r.unmapped_tail) {
      if (r.chrom == "*") {
        if (!(p.bai.has_no_coor && p.bai.n_no_coor > 0)) continue;
        s.mode = 3;
      } else {
        long ref = -1;
        for (size_t i = 0; i < p.hdr.ref_names.size(); i++) if (p.hdr.ref_names[i] == r.chrom) { ref = (long)i; break; }
        if (ref < 0) throw Error("Reference '" + r.chrom + "' not found in BAM header");
        if ((size_t)ref >= p.bai.refs.size()) throw Error("Reference index " + std::to_string(ref) + " not found in BAI index");
        uint64_t seek = 0;
        bool have = false;
        for (auto& b : p.bai.refs[ref].bins) for (auto& c : b.second) { seek = have ? std::max(seek, c.second) : c.second; have = true; }
        if (!have) {
          for (auto& rf : p.bai.refs) if (!rf.intervals.empty()) { seek = have ? std::max(seek, rf.intervals.back()) : rf.intervals.back(); have = true; }
        }
        uint64_t uo = (have && seek) ? voff_to_uoff(p, seek) : p.hdr.first_record_offset;
        unsigned long long h[2];
        launch_lower_bound_u64(p.d_rec_off.p, n, uo, d_idx.p, st);
        HIP_CHECK(hipMemcpyAsync(h, d_idx.p, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        uint64_t i0 = h[0];
        h[0] = n;
        HIP_CHECK(hipMemcpyAsync(d_idx.p, h, 8, hipMemcpyHostToDevice, st));
        launch_find_first(p.k_refid.p, n, i0, (int32_t)ref, 1, d_idx.p, st);
        HIP_CHECK(hipMemcpyAsync(h, d_idx.p, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        uint64_t first = h[0];
        uint64_t last = n;
        if (first < n) {
          h[0] = n;
          HIP_CHECK(hipMemcpyAsync(d_idx.p, h, 8, hipMemcpyHostToDevice, st));
          launch_find_first(p.k_refid.p, n, first + 1, (int32_t)ref, 0, d_idx.p, st);
          HIP_CHECK(hipMemcpyAsync(h, d_idx.p, 8, hipMemcpyDeviceToHost, st));
          HIP_CHECK(hipStreamSynchronize(st));
          last = h[0];
        }
        s.mode = 2;
        s.ref = (int32_t)ref;
        s.i_lo = first;
        s.i_hi = last;
      }
    } else {
      long ref = -1;
      for (size_t i = 0; i < p.hdr.ref_names.size(); i++) if (p.hdr.ref_names[i] == r.chrom) { ref = (long)i; break; }
      if (ref < 0) throw Error("BAM region query failed: region reference sequence does not exist in reference sequences: " + r.chrom);
      s.mode = 1;
      s.ref = (int32_t)ref;
      s.start1 = r.has_start ? (int64_t)r.start : 0;
      s.end1 = r.has_end ? (int64_t)r.end : INT64_MAX;
      s.q_start1 = r.has_start ? (int64_t)r.start : 1;
    }
    sels.push_back(s);
  }
  if (!satisfiable) sels.clear();
  // One pass per selection: flags, scan, compaction straight behind the rows of the previous selections.  A record
  // belongs to one region unless the caller's regions overlap, so n rows of capacity are enough; if they are not,
  // the totals are taken first and the selections are run again into an exact allocation.
  uint64_t total = 0;
  bool overflow = false;
  rows_owned->alloc(std::max<uint64_t>(n, 1));
  for (auto& s : sels) {
    launch_row_flags(rk, n, s, d_terms.p, keep.p, st);
    launch_exclusive_scan_u32_to_u64(keep.p, kscan.p, n, tmp.p, st);
    uint64_t t = 0;
    HIP_CHECK(hipMemcpyAsync(&t, kscan.p + n, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (total + t > n) { overflow = true; break; }
    if (t) launch_compact_rows(p.d_rec_off.p, keep.p, kscan.p, n, rows_owned->p, total, st);
    total += t;
  }
  if (overflow) {
    std::vector<uint64_t> totals;
    total = 0;
    for (auto& s : sels) {
      launch_row_flags(rk, n, s, d_terms.p, keep.p, st);
      launch_exclusive_scan_u32_to_u64(keep.p, kscan.p, n, tmp.p, st);
      uint64_t t = 0;
      HIP_CHECK(hipMemcpyAsync(&t, kscan.p + n, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      totals.push_back(t);
      total += t;
    }
    HIP_CHECK(hipStreamSynchronize(st));
    rows_owned->alloc(std::max<uint64_t>(total, 1));
    uint64_t base = 0;
    for (size_t k = 0; k < sels.size(); k++) {
      if (totals[k]) {
        launch_row_flags(rk, n, sels[k], d_terms.p, keep.p, st);
        launch_exclusive_scan_u32_to_u64(keep.p, kscan.p, n, tmp.p, st);
        launch_compact_rows(p.d_rec_off.p, keep.p, kscan.p, n, rows_owned->p, base, st);
      }
      base += totals[k];
    }
  }
  HIP_CHECK(hipStreamSynchronize(st));
  *rows = rows_owned->p;
  *n_rows = total;
}

// D2H of a partition's Arrow buffers (bioscan_next then exports zero-copy windows of them)
static void copy_result_to_host(Result& res, hipStream_t st) {
  const uint64_t n = res.n_rows, nb = res.n_batches(), nwords = (n + 63) / 64;
  const uint32_t batch_size = res.batch_size;
  for (auto& col : res.cols) {
    if (col.d_values.p && n) {
      uint64_t bytes = col.is_var() ? col.total_bytes : col.is_list() ? col.total_bytes * col.list_elem_bytes() : n * 4;
      col.h_values.alloc(std::max<uint64_t>(bytes, 1));
      if (bytes) HIP_CHECK(hipMemcpyAsync(col.h_values.p, col.d_values.p, bytes, hipMemcpyDeviceToHost, st));
    }
    if (col.d_off32.p && n) {
      uint64_t bytes = nb * ((uint64_t)batch_size + 1) * 4;
      col.h_off32.alloc(bytes);
      HIP_CHECK(hipMemcpyAsync(col.h_off32.p, col.d_off32.p, bytes, hipMemcpyDeviceToHost, st));
      col.h_batch_base.resize(nb);
      DevBuf<uint64_t> d_base(nb);
      launch_batch_bases(col.d_off64.p, nb, batch_size, d_base.p, st);
      HIP_CHECK(hipMemcpyAsync(col.h_batch_base.data(), d_base.p, nb * 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));  // d_base is released at the end of this scope
    }
    if (col.d_valid.p && n) {
      col.h_valid.alloc(nwords * 8 + 8);
      HIP_CHECK(hipMemcpyAsync(col.h_valid.p, col.d_valid.p, nwords * 8, hipMemcpyDeviceToHost, st));
    }
  }
  HIP_CHECK(hipStreamSynchronize(st));
  for (auto& col : res.cols) {
    col.d_values.reset(); col.d_off64.reset(); col.d_off32.reset(); col.d_valid.reset();
  }
  res.on_host = true;
}

static std::shared_ptr<Result> run_partition(const Plan& plan, int partition, uint32_t batch_size, bool force_decode, bool to_host) {
  Provider& p = *plan.prov;
  const auto wall0 = std::chrono::steady_clock::now();
  const bool laps = env_knobs().laps;
  auto lap = [&](const char* what) {
    if (laps) fprintf(stderr, "[bioscan] bam partition %d: %-18s at %8.3f ms\n", partition, what,
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
  };
  const DecodeRange range = partition_range(plan, partition);
  lap("range planned");
  // One critical section from the decode to the last extract kernel: the provider caches ONE decoded range, and another
  // thread executing a different partition of the same plan would otherwise replace it between the two steps.
  std::lock_guard<std::mutex> lk(p.mu);
  p.decode(force_decode, range, plan.indexed);
  lap("decoded");
  p.set_device();
  hipStream_t st = p.stream;
  auto res = std::make_shared<Result>();
  res->batch_size = batch_size;
  res->stats = p.decode_stats;
  StageTimer t(st);
  t.start();
  DevBuf<uint64_t> rows_owned;
  const uint64_t* rows = nullptr;
  uint64_t n = 0;
  select_rows(plan, partition, &rows_owned, &rows, &n);
  res->stats.ms_select = t.stop();
  lap("rows selected");
  t.start();
  res->n_rows = n;
  res->stats.n_rows = n;
  const uint64_t nwords = (n + 63) / 64;
  const uint64_t nb = res->n_batches();

  // ---- columns ----
  res->cols.resize(plan.out_fields.size());
  for (size_t c = 0; c < plan.out_fields.size(); c++) {
    res->cols[c].fd = plan.out_fields[c];
    res->cols[c].n_rows = n;
  }
  // map core columns (first occurrence wins; duplicates in a projection share by copy below)
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

  if (n) {
    CoreCols cc{};
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
    auto lens = [&](int idx) -> uint32_t* {
      if (core_col[idx] < 0) return nullptr;
      Column& col = res->cols[core_col[idx]];
      col.d_len.alloc(n);
      return col.d_len.p;
    };
    cc.start = fixed(2); cc.end = fixed(3); cc.flags = fixed(4); cc.mapq = fixed(6); cc.mate_start = fixed(8);
    cc.tlen = (int32_t*)fixed(11);
    cc.v_chrom = valid(1); cc.v_start = valid(2); cc.v_end = valid(3); cc.v_mate_chrom = valid(7); cc.v_mate_start = valid(8);
    cc.len_name = lens(0); cc.len_chrom = lens(1); cc.len_cigar = lens(5); cc.len_mate_chrom = lens(7);
    cc.len_seq = lens(9); cc.len_qual = lens(10);
    RowOverride ov{};
    launch_extract_fixed(p.d_u.p, rows, 0, n, cc, p.d_ref_name_len.p, (int32_t)p.hdr.ref_names.size(), p.zero_based ? 1 : 0,
                         p.binary_cigar ? 1 : 0, ov, err.p, st);
    throw_extract_err(read_err(err, st));

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
      launch_batch_offsets(col.d_off64.p, n, batch_size, col.d_off32.p, st);
      arrow_bytes += tot * elem_bytes + nb * ((uint64_t)batch_size + 1) * 4;
    };
    const int var_idx[6] = {0, 1, 5, 7, 9, 10};
    for (int k : var_idx) if (core_col[k] >= 0) finish_var(res->cols[core_col[k]], 1);
    auto off_of = [&](int idx) -> const uint64_t* { return core_col[idx] >= 0 ? res->cols[core_col[idx]].d_off64.p : nullptr; };
    auto dat_of = [&](int idx) -> uint8_t* { return core_col[idx] >= 0 ? res->cols[core_col[idx]].d_values.p : nullptr; };
    // name rides with sequence / quality in the row-centric kernel; chrom, cigar, mate_chrom stay one row per lane
    if (core_col[1] >= 0 || core_col[5] >= 0 || core_col[7] >= 0)
      launch_scatter_small(p.d_u.p, rows, 0, n, nullptr, nullptr, off_of(1), dat_of(1), off_of(5), dat_of(5), off_of(7), dat_of(7),
                           p.d_ref_names.p, p.d_ref_name_off.p, (int32_t)p.hdr.ref_names.size(), p.binary_cigar ? 1 : 0, ov, st);
    DevBuf<uint32_t> wide(1);
    HIP_CHECK(hipMemsetAsync(wide.p, 0, 4, st));
    launch_scatter_seqqual_rows(p.d_u.p, rows, n, off_of(9), dat_of(9), off_of(10), dat_of(10), off_of(0), dat_of(0), wide.p, st);
    if (core_col[10] >= 0) {
      if (read_err(wide, st)) {
        // exact path for qualities >= 95 (two-byte UTF-8 chars)
        Column& col = res->cols[core_col[10]];
        arrow_bytes -= col.total_bytes;
        launch_qual_wide_len(p.d_u.p, rows, 0, n, col.d_len.p, st);
        finish_var(col, 1);
        arrow_bytes -= nb * ((uint64_t)batch_size + 1) * 4;
        launch_qual_wide_scatter(p.d_u.p, rows, n, col.d_off64.p, col.d_values.p, st);
      }
    }
    // ---- tags ----
    if (!tag_cols.empty()) {
      const int nt = (int)p.tag_fields.size();
      std::vector<uint16_t> tg(nt);
      for (int k = 0; k < nt; k++) tg[k] = (uint16_t)((uint8_t)p.tag_fields[k][0] | ((uint16_t)(uint8_t)p.tag_fields[k][1] << 8));
      DevBuf<uint16_t> d_tg(nt);
      HIP_CHECK(hipMemcpyAsync(d_tg.p, tg.data(), nt * 2, hipMemcpyHostToDevice, st));
      DevBuf<uint32_t> loc((uint64_t)nt * n);
      DevBuf<uint8_t> typ((uint64_t)nt * n);
      launch_tag_locate(p.d_u.p, rows, n, d_tg.p, nt, loc.p, typ.p, err.p, st);
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
          launch_tag_fixed(p.d_u.p, rows, 0, n, l, ty, kind, (uint32_t*)col.d_values.p, col.d_valid.p, err.p, st);
        } else if (col.fd.kind == AK_UTF8) {
          col.d_len.alloc(n);
          launch_tag_utf8_len(p.d_u.p, rows, 0, n, l, ty, col.d_len.p, col.d_valid.p, err.p, st);
          finish_var(col, 1);
          launch_tag_utf8_scatter(p.d_u.p, rows, n, l, ty, col.d_off64.p, col.d_valid.p, 0, col.d_values.p, st);
        } else if (col.is_list()) {
          col.d_len.alloc(n);
          int elem = list_elem_code(col.fd.kind);
          launch_tag_list_len(p.d_u.p, rows, 0, n, l, ty, elem, col.d_len.p, col.d_valid.p, err.p, st);
          finish_var(col, col.list_elem_bytes());
          launch_tag_list_scatter(p.d_u.p, rows, n, l, ty, elem, col.d_off64.p, col.d_values.p, err.p, st);
        } else {
          throw Error("unsupported tag column type");
        }
      }
    }
    throw_extract_err(read_err(err, st));
    for (auto& col : res->cols) col.d_len.reset();
  }
  // duplicate projected columns: not supported (DataFusion never sends duplicates)
  res->stats.ms_extract = t.stop();
  res->stats.arrow_bytes = arrow_bytes;
  res->stats.ms_total_gpu = p.decode_stats.ms_total_gpu + res->stats.ms_select + res->stats.ms_extract;
  lap("end");
  res->stats.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();

  if (to_host) copy_result_to_host(*res, st);
  return res;
}

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
      t.start();
      p.launch_crc(p.d_u.p, b_hi - b_start, b_start);
      res->stats.ms_crc += t.stop();
      p.check_inflate_status(b_start, b_hi - b_start);
      p.decoded = false;  // the BAM cache (if any) no longer describes d_u
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
    if (!none && x0 < hi - base) {
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
      // ---- how many records start before T ----
      const uint64_t T_rel = T > base ? T - base : 0;
      launch_fastq_count_owned(nl.p, n_nl, x0, hi - base, T_rel, d_res.p, st);
      unsigned long long r = 0;
      HIP_CHECK(hipMemcpyAsync(&r, d_res.p, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      n_rows = r;
      // the last owned record must be complete: 4 newline-terminated lines, or the data really ends
      if (n_rows * 4 > n_nl + (at_eof ? 1 : 0)) {
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
    DevBuf<uint64_t> srcs[4];
    FastqCols fc{};
    uint64_t** sp[4] = {&fc.src_name, &fc.src_desc, &fc.src_seq, &fc.src_qual};
    uint32_t** lp[4] = {&fc.len_name, &fc.len_desc, &fc.len_seq, &fc.len_qual};
    for (int k = 0; k < 4; k++) {
      if (col_of[k] < 0) continue;
      Column& col = res->cols[col_of[k]];
      col.n_rows = n;
      srcs[k].alloc(n);
      col.d_len.alloc(n);
      *sp[k] = srcs[k].p;
      *lp[k] = col.d_len.p;
      if (k == 1) { col.d_valid.alloc(nwords); fc.v_desc = col.d_valid.p; arrow_bytes += nwords * 8; }
    }
    launch_fastq_fields(u, x0, hi - base, nl.p, n_nl, n, fc, err.p, st);
    uint32_t e = read_err(err, st);
    if (e == 1) throw Error("FASTQ read error: invalid name prefix");
    if (e == 2) throw Error("FASTQ read error: invalid description prefix");
    DevBuf<uint64_t> stmp(scan_tmp_elems(n));
    for (int k = 0; k < 4; k++) {
      if (col_of[k] < 0) continue;
      Column& col = res->cols[col_of[k]];
      col.d_off64.alloc(n + 1);
      launch_exclusive_scan_u32_to_u64(col.d_len.p, col.d_off64.p, n, stmp.p, st);
      uint64_t tot = 0;
      HIP_CHECK(hipMemcpyAsync(&tot, col.d_off64.p + n, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      col.total_bytes = tot;
      col.d_values.alloc(std::max<uint64_t>(tot, 1));
      col.d_off32.alloc(nb * ((uint64_t)batch_size + 1));
      launch_batch_offsets(col.d_off64.p, n, batch_size, col.d_off32.p, st);
      launch_scatter_ranges(u, srcs[k].p, n, col.d_off64.p, col.d_values.p, tot, st);
      arrow_bytes += tot + nb * ((uint64_t)batch_size + 1) * 4;
      col.d_len.reset();
    }
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
  int64_t set = 0;
  for (uint64_t i = 0; i < nbits; i++) {
    uint64_t b = bit0 + i;
    set += (bits[b >> 3] >> (b & 7)) & 1;
  }
  return (int64_t)nbits - set;
}

static void export_batch(const std::shared_ptr<Result>& res, uint64_t b, ArrowArray* out) {
  const uint64_t bs = res->batch_size;
  const uint64_t r0 = b * bs;
  const uint64_t rows = std::min<uint64_t>(bs, res->n_rows - r0);
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

// =================================================================================================
// C ABI
// =================================================================================================
struct bioscan_provider { Provider p; std::unique_ptr<VcfProviderI> vcf; };
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
  }
  if (p.has_index) {
    std::ifstream f(p.index_path, std::ios::binary);
    if (!f.good()) throw Error("Failed to open indexed BAM: cannot read " + p.index_path);
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::string e;
    if (!parse_bai(d, &p.bai, &e)) throw Error("Failed to open indexed BAM: " + e);
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
      auto est = estimate_sizes_from_bai(&p.bai, regions, p.hdr.ref_names, p.hdr.ref_lengths);
      pl.assignments = balance_partitions(est, (size_t)std::max(target_partitions, 0));
      if (full && p.bai.has_no_coor && p.bai.n_no_coor > 0) {
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
  bs->s.prov = plan->pl.prov;
  bs->s.res = plan->pl.prov->kind == 1 ? run_partition_fastq(plan->pl, partition, (uint32_t)batch_size, !device_only)
                                       : run_partition(plan->pl, partition, (uint32_t)batch_size, device_only, !device_only);
  if (stats) *stats = bs->s.res->stats;
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
  if (!st.res->on_host) throw Error("stream was executed device-only; no host batches to export");
  if (st.next >= st.res->n_batches()) {
    *has_batch = 0;
    return 0;
  }
  export_batch(st.res, st.next, out);
  st.next++;
  *has_batch = 1;
  API_END
}

void bioscan_stream_close(bioscan_stream* s) { delete s; }
void bioscan_plan_close(bioscan_plan* p) { delete p; }
void bioscan_provider_close(bioscan_provider* p) {
  delete p;
  dev_pool_trim();
  host_pool_trim();
}

int bioscan_provider_make_resident(bioscan_provider* p) {
  API_BEGIN
  if (p->vcf) p->vcf->make_resident(); else p->p.make_resident();
  API_END
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
