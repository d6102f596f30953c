// vcf_engine.cpp -- VCF scan path: provider, plan, partition execution on the GPU, nested Arrow export, list UDFs.
//
// Mirrors (paths relative to /root/reference/datafusion/bio-format-vcf/src):
//   VcfTableProvider::new_with_samples_and_format_and_policy   table_provider.rs:849-1097
//   TableProvider::supports_filters_pushdown / scan            table_provider.rs:1203-1462
//   VcfExec::execute -> get_indexed_vcf_stream / get_local_vcf_sync   physical_exec.rs:2612-2690, 2747-3078, 912-1198
//   list UDFs                                                  udfs.rs:67-110, 606-650
// The record-level work runs in vcf_kernels.hip; this file owns planning, buffers and the Arrow C Data export.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <deque>
#include <functional>
#include <map>
#include <memory>

#include "bgzf_source.h"
#include "vcf_api.h"
#include "vcf_host.h"
#include "vcf_kernels.h"

namespace bioscan {
namespace {

struct Timer {
  hipEvent_t a, b;
  hipStream_t st;
  explicit Timer(hipStream_t s) : st(s) { HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b)); }
  ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
  void start() { HIP_CHECK(hipEventRecord(a, st)); }
  double stop() {
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
  }
};

static bool file_exists(const std::string& s) {
  std::ifstream f(s, std::ios::binary);
  return f.good();
}

static void throw_vcf_err(uint32_t e) {
  switch (e) {
    case VERR_NONE: return;
    case VERR_BLANK_LINE: throw Error("VCF read error: blank line inside the records");
    case VERR_SHORT_RECORD: throw Error("VCF read error: record has fewer than 8 tab-separated fields");
    case VERR_BAD_POS: throw Error("VCF position error: invalid digit found in string");
    case VERR_MISSING_START: throw Error("Missing variant start");
    case VERR_BAD_END: throw Error("VCF read error: invalid INFO END value");
    case VERR_BAD_QUAL: throw Error("VCF qual error: invalid float literal");
    case VERR_FLOAT_PRECISION: throw Error("VCF read error: more than 65536 float literals of one chunk lie within 2^-52 of a rounding boundary of f32");
    case VERR_DUP_INFO_KEY: throw Error("VCF read error: duplicate INFO key in one record");
    case VERR_BAD_INT: throw Error("Error reading INFO / FORMAT field: invalid integer");
    case VERR_BAD_FLOAT: throw Error("Error reading INFO / FORMAT field: invalid float literal");
    case VERR_INVALID_FLAG: throw Error("Error reading INFO field: invalid flag");
    case VERR_PERCENT: throw Error("VCF read error: invalid UTF-8 after percent-decoding a string value");
    case VERR_BAD_GT: throw Error("Error reading FORMAT field 'GT': invalid genotype");
    case VERR_BAD_CHAR: throw Error("Error reading INFO / FORMAT field: invalid character");
    case VERR_UNSUPPORTED_INFO: throw Error("Unsupported INFO value type for a Character field");
    default: throw Error("VCF read error: device error " + std::to_string(e));
  }
}

// ---- result nodes -----------------------------------------------------------------------------------------
struct VNode {
  VField fd;
  uint64_t n = 0;
  uint64_t total = 0;          // utf8: bytes; list: child length
  bool all_valid = true;       // no validity buffer
  DevBuf<uint8_t> d_values;    // fixed width values / utf8 bytes / boolean bit words
  DevBuf<uint64_t> d_off;      // utf8 / list: n + 1
  DevBuf<uint64_t> d_valid;
  HostBuf h_values, h_off, h_valid;
  // utf8 / list nodes whose offsets fit 32 bits (every chunk of a stream; a whole partition mostly): the offsets cross the link as
  // int32 and a batch is exported as an Arrow SLICE of the chunk's arrays (ArrowArray::offset) -- no per-batch rebasing, no
  // repacking of validity bits on the host (export_node)
  DevBuf<int32_t> d_off32;
  HostBuf h_off32;
  bool off32 = false;
  mutable int64_t nulls_all = -1;   // NULLs of elements [0, nulls_hi), counted on first use (a list's child is exported whole)
  mutable uint64_t nulls_hi = 0;
  std::vector<VNode> kids;
  VNode() = default;
  VNode(VNode&&) = default;
  VNode& operator=(VNode&&) = default;
};
struct VResult {
  uint64_t n_rows = 0;
  uint64_t batch_size = 8192;
  std::vector<VNode> cols;
  bool on_host = false;
  int device = 0;
  hipStream_t stream = nullptr;
  bioscan_scan_stats stats{};
  uint64_t n_batches() const { return (n_rows + batch_size - 1) / batch_size; }
};

static uint32_t fixed_width(VKind k) { return k == VK_FLOAT64 ? 8 : 4; }

// ---- provider ---------------------------------------------------------------------------------------------
struct VcfProvider : BgzfSource, VcfProviderI {
  bool bgzf = false;
  bool zero_based = true;
  VcfHeader hdr;
  VcfSchema sch;
  bool has_index = false;
  std::string index_path;
  Tbi tbi;
  std::map<size_t, std::vector<std::pair<uint64_t, uint64_t>>> whole_contig_chunks;  // per contig: merged chunks as inflated offsets (guarded by mu)
  bool index_readable = false;   // the tabix reader accepted the index (a CSI, or an unreadable file, only fails at execute)
  std::string index_error;
  std::vector<std::string> contig_names;
  std::vector<uint64_t> contig_lengths;

  void schema(ArrowSchema* out) const override;
  void supports_filters_pushdown(const bioscan_filter* filters, int32_t n, int32_t* out) const override;
  VcfPlanI* scan(const int32_t* projection, int32_t n_projection, const bioscan_filter* filters, int32_t n_filters, int64_t limit,
                 int32_t target_partitions) override;
  void make_resident() override { BgzfSource::make_resident(); }
  uint32_t chunk_members = 0;
  void set_chunk_members(uint32_t n) override { chunk_members = n; }
  const uint8_t* text_base() const { return bgzf ? d_u.p : d_comp.p; }
  uint64_t text_len() const { return bgzf ? ulen : file_len; }
};

static bool vcf_can_push_down(const Filter& f, const std::vector<VField>& schema) {  // record_filter.rs:40-55, 285-356
  const VField* fd = nullptr;
  for (auto& x : schema) if (x.name == f.column) { fd = &x; break; }
  if (!fd) return false;
  const bool is_str = fd->kind == VK_UTF8;
  const bool is_num = fd->kind == VK_INT32 || fd->kind == VK_UINT32 || fd->kind == VK_FLOAT32 || fd->kind == VK_FLOAT64;
  if (f.op == BIOSCAN_OP_EQ || f.op == BIOSCAN_OP_NE) return is_str || is_num;
  if (f.op <= BIOSCAN_OP_GE) return is_num;
  if (f.op == BIOSCAN_OP_BETWEEN || f.op == BIOSCAN_OP_NOT_BETWEEN) return is_num;
  return is_str || is_num;
}

// ---- Arrow schema export (nested) -------------------------------------------------------------------------
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
static void export_field(const VField& f, ArrowSchema* out) {
  fill_schema(out, f.name, vkind_format(f.kind), f.nullable, f.metadata);
  for (auto& c : f.children) {
    std::unique_ptr<ArrowSchema> cs(new ArrowSchema);
    export_field(c, cs.get());
    add_child(out, std::move(cs));
  }
}
static void export_vschema(const std::vector<VField>& fields, const std::vector<std::pair<std::string, std::string>>& md, ArrowSchema* out) {
  fill_schema(out, "", "+s", false, md);
  for (auto& f : fields) {
    std::unique_ptr<ArrowSchema> c(new ArrowSchema);
    export_field(f, c.get());
    add_child(out, std::move(c));
  }
}
void VcfProvider::schema(ArrowSchema* out) const { export_vschema(sch.fields, sch.metadata, out); }

void VcfProvider::supports_filters_pushdown(const bioscan_filter* filters, int32_t n, int32_t* out) const {
  auto fs = copy_filters(filters, n);
  for (int32_t i = 0; i < n; i++) {
    if (has_index && is_genomic_coordinate_filter(fs[i])) out[i] = 1;
    else if (vcf_can_push_down(fs[i], sch.fields)) out[i] = 1;
    else out[i] = 0;
  }
}

// ---- plan -------------------------------------------------------------------------------------------------
struct VcfPlan : VcfPlanI {
  VcfProvider* prov = nullptr;
  bool has_projection = false;
  std::vector<int32_t> projection;
  std::vector<VField> out_fields;
  int64_t limit = -1;
  bool indexed = false, empty = false;
  std::vector<PartitionAssignment> assignments;
  std::vector<Filter> residual;

  int32_t n_partitions() const override { return empty ? 0 : (indexed ? (int32_t)assignments.size() : 1); }
  void schema(ArrowSchema* out) const override { export_vschema(out_fields, prov->sch.metadata, out); }
  std::string display() const override {
    std::string s = "VcfExec: projection=[";
    if (!has_projection) s += "*";
    else
      for (size_t i = 0; i < out_fields.size(); i++) { if (i) s += ", "; s += out_fields[i].name; }
    return s + "]";
  }
  std::string partition_desc(int32_t partition) const override {
    if (!indexed) return "sequential";
    return describe_partition(assignments.at((size_t)partition));
  }
  VcfStreamI* execute(int32_t partition, int32_t batch_size, bool device_only, bioscan_scan_stats* stats) const override;
};

VcfPlanI* VcfProvider::scan(const int32_t* projection, int32_t n_projection, const bioscan_filter* filters, int32_t n_filters,
                            int64_t limit, int32_t target_partitions) {
  std::unique_ptr<VcfPlan> pl(new VcfPlan);
  pl->prov = this;
  pl->limit = limit;
  if (projection) {
    pl->has_projection = true;
    for (int i = 0; i < n_projection; i++) {
      if (projection[i] < 0 || (size_t)projection[i] >= sch.fields.size()) throw Error("projection index out of range");
      pl->projection.push_back(projection[i]);
      pl->out_fields.push_back(sch.fields[projection[i]]);
    }
    for (size_t a = 0; a < pl->projection.size(); a++)
      for (size_t b = a + 1; b < pl->projection.size(); b++)
        if (pl->projection[a] == pl->projection[b]) throw Error("duplicate column in projection");
  } else pl->out_fields = sch.fields;
  if (limit == 0) { pl->empty = true; return pl.release(); }
  auto fs = copy_filters(filters, n_filters);
  if (has_index) {
    std::vector<GenomicRegion> regions;
    bool unsat = false;
    extract_genomic_regions(fs, zero_based, &regions, &unsat);
    if (unsat) { pl->empty = true; return pl.release(); }
    if (regions.empty())
      for (auto& n : contig_names) { GenomicRegion r; r.chrom = n; regions.push_back(r); }
    if (!regions.empty()) {
      auto est = estimate_sizes_from_tbi(index_readable ? &tbi : nullptr, regions, contig_names, contig_lengths);
      pl->assignments = balance_partitions(est, (size_t)std::max(target_partitions, 0));
      for (auto& f : fs) if (vcf_can_push_down(f, sch.fields)) pl->residual.push_back(f);
      pl->indexed = true;
    }
  }
  return pl.release();
}

// ---- helpers for building columns -----------------------------------------------------------------------------
struct Ctx {
  VcfProvider& p;
  hipStream_t st;
  const uint8_t* u;
  DevBuf<uint64_t> tmp;  // scan scratch
  uint64_t tmp_n = 0;
  uint32_t* err;
  uint64_t arrow_bytes = 0;
  DevBuf<uint32_t> pct;  // set by the string span kernels when a value holds a percent escape
  uint64_t* scratch(uint64_t n) {
    const uint64_t need = scan_tmp_elems(n);
    if (tmp.n < need) tmp.alloc(need);
    return tmp.p;
  }
  uint64_t scan(const uint32_t* in, uint64_t* out, uint64_t n) {  // exclusive scan + total (syncs)
    launch_exclusive_scan_u32_to_u64(in, out, n, scratch(n), st);
    uint64_t tot = 0;
    HIP_CHECK(hipMemcpyAsync(&tot, out + n, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return tot;
  }
};

static void finish_utf8(Ctx& c, VNode& nd, const uint64_t* src, const uint32_t* len, uint64_t N, bool may_have_pct = false) {
  nd.n = N;
  nd.d_off.alloc(N + 1);
  nd.total = c.scan(len, nd.d_off.p, N);
  nd.d_values.alloc(std::max<uint64_t>(nd.total, 1));
  uint32_t pct = 0;
  if (may_have_pct) {
    HIP_CHECK(hipMemcpyAsync(&pct, c.pct.p, 4, hipMemcpyDeviceToHost, c.st));
    HIP_CHECK(hipStreamSynchronize(c.st));
  }
  if (pct) launch_scatter_pct(c.u, src, N, nd.d_off.p, nd.d_values.p, c.st);   // percent-decoding copy (noodles decodes INFO / FORMAT strings)
  else launch_scatter_ranges(c.u, src, N, nd.d_off.p, nd.d_values.p, nd.total, c.st);
  c.arrow_bytes += nd.total + (N + 1) * 4;
}

// Build node `nd` (leaf or List<leaf>) of length N from value spans.
static void build_from_spans(Ctx& c, VNode& nd, const uint64_t* sp_off, const uint32_t* sp_len, const uint8_t* sp_state, uint64_t N) {
  nd.n = N;
  const uint64_t nw = (N + 63) / 64;
  switch (nd.fd.kind) {
    case VK_INT32:
    case VK_FLOAT32:
      nd.d_values.alloc(std::max<uint64_t>(N, 1) * 4);
      nd.d_valid.alloc(std::max<uint64_t>(nw, 1));
      nd.all_valid = false;
      launch_span_num(c.u, sp_off, sp_len, sp_state, N, nd.fd.kind == VK_INT32 ? 0 : 1, (uint32_t*)nd.d_values.p, nd.d_valid.p, c.err, c.st);
      c.arrow_bytes += N * 4 + nw * 8;
      break;
    case VK_BOOL:
      nd.d_values.alloc(std::max<uint64_t>(nw, 1) * 8);
      launch_span_flag(c.u, sp_off, sp_len, sp_state, N, (uint64_t*)nd.d_values.p, c.err, c.st);
      c.arrow_bytes += nw * 8;
      break;
    case VK_UTF8: {
      DevBuf<uint32_t> len(std::max<uint64_t>(N, 1));
      nd.d_valid.alloc(std::max<uint64_t>(nw, 1));
      nd.all_valid = false;
      if (!c.pct.p) c.pct.alloc(1);
      HIP_CHECK(hipMemsetAsync(c.pct.p, 0, 4, c.st));
      launch_span_str(c.u, sp_off, sp_len, sp_state, N, len.p, nd.d_valid.p, c.pct.p, c.err, c.st);
      finish_utf8(c, nd, sp_off, len.p, N, true);
      c.arrow_bytes += nw * 8;
      break;
    }
    case VK_LIST: {
      DevBuf<uint32_t> cnt(std::max<uint64_t>(N, 1));
      nd.d_valid.alloc(std::max<uint64_t>(nw, 1));
      nd.all_valid = false;
      launch_span_list_count(c.u, sp_off, sp_len, sp_state, N, cnt.p, nd.d_valid.p, c.st);
      nd.d_off.alloc(N + 1);
      const uint64_t E = nd.total = c.scan(cnt.p, nd.d_off.p, N);
      nd.kids.resize(1);
      VNode& ch = nd.kids[0];
      ch.fd = nd.fd.children.at(0);
      ch.n = E;
      const uint64_t ew = (E + 63) / 64;
      DevBuf<uint8_t> evalid(std::max<uint64_t>(E, 1));
      ch.d_valid.alloc(std::max<uint64_t>(ew, 1));
      ch.all_valid = false;
      if (ch.fd.kind == VK_INT32 || ch.fd.kind == VK_FLOAT32) {
        ch.d_values.alloc(std::max<uint64_t>(E, 1) * 4);
        launch_span_list_elems(c.u, sp_off, sp_len, sp_state, N, nd.d_off.p, ch.fd.kind == VK_INT32 ? 0 : 1, (uint32_t*)ch.d_values.p,
                               nullptr, nullptr, evalid.p, nullptr, c.err, c.st);
        c.arrow_bytes += E * 4;
      } else if (ch.fd.kind == VK_UTF8) {
        DevBuf<uint64_t> esrc(std::max<uint64_t>(E, 1));
        DevBuf<uint32_t> elen(std::max<uint64_t>(E, 1));
        if (!c.pct.p) c.pct.alloc(1);
        HIP_CHECK(hipMemsetAsync(c.pct.p, 0, 4, c.st));
        launch_span_list_elems(c.u, sp_off, sp_len, sp_state, N, nd.d_off.p, 2, nullptr, esrc.p, elen.p, evalid.p, c.pct.p, c.err, c.st);
        finish_utf8(c, ch, esrc.p, elen.p, E, true);
      } else throw Error("Unsupported list element type in a VCF column");
      launch_pack_bits(evalid.p, E, ch.d_valid.p, c.st);
      HIP_CHECK(hipStreamSynchronize(c.st));  // evalid goes out of scope
      c.arrow_bytes += (N + 1) * 4 + nw * 8 + ew * 8;
      break;
    }
    default: throw Error("Unsupported VCF column type");
  }
}

// ---- stream -------------------------------------------------------------------------------------------------
struct VcfStream : VcfStreamI {
  std::shared_ptr<VResult> res;
  uint64_t next_batch = 0;
  bool next(ArrowArray* out) override;
  void list_udf(const char* field, int32_t udf, double threshold, bioscan_udf_stats* out) override;
};

static void copy_node_to_host(VNode& nd, hipStream_t st) {
  auto d2h = [&](HostBuf& h, const void* d, uint64_t bytes) {
    h.alloc(std::max<uint64_t>(bytes, 8) + 8);
    if (bytes) HIP_CHECK(hipMemcpyAsync(h.p, d, bytes, hipMemcpyDeviceToHost, st));
  };
  const uint64_t nw = (nd.n + 63) / 64;
  switch (nd.fd.kind) {
    case VK_INT32: case VK_UINT32: case VK_FLOAT32: case VK_FLOAT64:
      if (nd.d_values.p) d2h(nd.h_values, nd.d_values.p, nd.n * fixed_width(nd.fd.kind));
      break;
    case VK_BOOL:
      if (nd.d_values.p) d2h(nd.h_values, nd.d_values.p, nw * 8);
      break;
    case VK_UTF8:
      if (nd.d_values.p) d2h(nd.h_values, nd.d_values.p, nd.total);
      [[fallthrough]];
    case VK_LIST:
      if (nd.d_off.p && nd.total < 0x7FFFFFFFull) {
        nd.d_off32.alloc(nd.n + 1);
        launch_off64_to_32(nd.d_off.p, nd.n + 1, nd.d_off32.p, st);
        d2h(nd.h_off32, nd.d_off32.p, (nd.n + 1) * 4);
        nd.off32 = true;
      } else if (nd.d_off.p) d2h(nd.h_off, nd.d_off.p, (nd.n + 1) * 8);
      break;
    default: break;
  }
  if (!nd.all_valid && nd.d_valid.p) d2h(nd.h_valid, nd.d_valid.p, nw * 8);
  for (auto& k : nd.kids) copy_node_to_host(k, st);
}
static void free_node_device(VNode& nd) {
  nd.d_values.reset(); nd.d_off.reset(); nd.d_valid.reset(); nd.d_off32.reset();
  for (auto& k : nd.kids) free_node_device(k);
}

// ---- Arrow array export ---------------------------------------------------------------------------------------
struct ArrayPriv {
  std::shared_ptr<VResult> keep;
  std::vector<const void*> buffers;
  std::vector<ArrowArray*> children;
  std::vector<std::unique_ptr<ArrowArray>> owned;
  std::vector<uint8_t> local_bits;     // repacked validity / boolean values
  std::vector<uint8_t> local_bits2;
  std::vector<int32_t> local_off;      // rebased offsets
};
static void release_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* pr = (ArrayPriv*)a->private_data;
  for (auto& c : pr->owned) if (c->release) c->release(c.get());
  delete pr;
  a->release = nullptr;
}
static ArrayPriv* init_array(ArrowArray* a, std::shared_ptr<VResult> keep, int64_t length) {
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
// copy bits [bit0, bit0+n) of src (LSB-first bytes) to dst starting at bit 0; returns the number of set bits
static uint64_t copy_bits(const uint8_t* src, uint64_t bit0, uint64_t n, std::vector<uint8_t>& dst) {
  dst.assign((n + 7) / 8 + 8, 0);
  if (!n) return 0;
  uint64_t set = 0;
  const uint32_t sh = (uint32_t)(bit0 & 7);
  const uint8_t* s = src + (bit0 >> 3);
  const uint64_t last_src = ((bit0 + n - 1) >> 3) - (bit0 >> 3);  // index of the last source byte that holds a wanted bit
  const uint64_t nbytes = (n + 7) / 8;
  for (uint64_t k = 0; k < nbytes; k++) {
    uint32_t v = s[k];
    if (sh && k + 1 <= last_src) v |= (uint32_t)s[k + 1] << 8;
    uint8_t b = (uint8_t)(v >> sh);
    if (k == nbytes - 1 && (n & 7)) b &= (uint8_t)((1u << (n & 7)) - 1);
    dst[k] = b;
    set += (uint64_t)__builtin_popcount(b);
  }
  return set;
}

static uint64_t node_off(const VNode& nd, uint64_t i) {
  return nd.off32 ? (uint64_t)((const int32_t*)nd.h_off32.p)[i] : ((const uint64_t*)nd.h_off.p)[i];
}
// every utf8 / list node of the subtree has its offsets as int32: rows [lo, hi) are a slice of the node's own buffers
static bool sliceable(const VNode& nd) {
  if ((nd.fd.kind == VK_UTF8 || nd.fd.kind == VK_LIST) && !nd.off32) return false;
  for (auto& k : nd.kids) if (!sliceable(k)) return false;
  return true;
}
static int64_t count_nulls(const VNode& nd, uint64_t lo, uint64_t hi) {
  if (nd.all_valid || !nd.h_valid.p || hi <= lo) return 0;
  auto range = [&](uint64_t a, uint64_t b) {   // set bits in [a, b)
    const uint64_t* w = (const uint64_t*)nd.h_valid.p;
    uint64_t set = 0;
    for (uint64_t i = a >> 6; i <= (b - 1) >> 6; i++) {
      uint64_t v = w[i];
      if (i == a >> 6) v &= ~0ull << (a & 63);
      if (i == (b - 1) >> 6 && (b & 63)) v &= (1ull << (b & 63)) - 1ull;
      set += (uint64_t)__builtin_popcountll(v);
    }
    return set;
  };
  if (lo == 0 && hi > 8192) {   // a whole child array, asked for again by every batch of the chunk
    if (nd.nulls_all < 0 || nd.nulls_hi != hi) { nd.nulls_all = (int64_t)(hi - range(0, hi)); nd.nulls_hi = hi; }
    return nd.nulls_all;
  }
  return (int64_t)((hi - lo) - range(lo, hi));
}
// Rows [lo, hi) of a sliceable node as an Arrow slice: offset = lo over the node's own host buffers; a list's child is the
// whole child array (the list's int32 offsets index it as they are).  O(nodes) per batch, whatever the batch holds.
static void export_slice(const std::shared_ptr<VResult>& res, const VNode& nd, uint64_t lo, uint64_t hi, ArrowArray* out) {
  ArrayPriv* pr = init_array(out, res, (int64_t)(hi - lo));
  out->offset = (int64_t)lo;
  out->null_count = count_nulls(nd, lo, hi);
  pr->buffers.push_back(out->null_count ? (const void*)nd.h_valid.p : nullptr);
  static const uint64_t zero = 0;
  switch (nd.fd.kind) {
    case VK_INT32: case VK_UINT32: case VK_FLOAT32: case VK_FLOAT64: case VK_BOOL:
      pr->buffers.push_back(nd.h_values.p ? (const void*)nd.h_values.p : (const void*)&zero);
      break;
    case VK_UTF8:
      pr->buffers.push_back(nd.h_off32.p);
      pr->buffers.push_back(nd.h_values.p ? (const void*)nd.h_values.p : (const void*)"");
      break;
    case VK_LIST: {
      pr->buffers.push_back(nd.h_off32.p);
      std::unique_ptr<ArrowArray> item(new ArrowArray);
      const VNode& ch = nd.kids.at(0);
      export_slice(res, ch, 0, node_off(nd, nd.n), item.get());   // (the elements the node's offsets address)
      pr->children.push_back(item.get());
      pr->owned.push_back(std::move(item));
      break;
    }
    case VK_STRUCT:
      // (a struct's offset applies to its children on top of their own: they are exported whole)
      for (auto& k : nd.kids) {
        std::unique_ptr<ArrowArray> ca(new ArrowArray);
        export_slice(res, k, 0, k.n, ca.get());
        pr->children.push_back(ca.get());
        pr->owned.push_back(std::move(ca));
      }
      break;
  }
  finish_array(out);
}

static void export_node(const std::shared_ptr<VResult>& res, const VNode& nd, uint64_t lo, uint64_t hi, ArrowArray* out) {
  if (sliceable(nd) && hi > lo) { export_slice(res, nd, lo, hi, out); return; }
  const uint64_t len = hi - lo;
  ArrayPriv* pr = init_array(out, res, (int64_t)len);
  // validity
  const void* vptr = nullptr;
  int64_t nulls = 0;
  if (!nd.all_valid && nd.h_valid.p && len) {
    const uint64_t set = copy_bits(nd.h_valid.p, lo, len, pr->local_bits);
    nulls = (int64_t)(len - set);
    if (nulls) vptr = pr->local_bits.data();
  }
  out->null_count = nulls;
  pr->buffers.push_back(vptr);
  switch (nd.fd.kind) {
    case VK_INT32: case VK_UINT32: case VK_FLOAT32: case VK_FLOAT64:
      pr->buffers.push_back(nd.h_values.p ? nd.h_values.p + lo * fixed_width(nd.fd.kind) : nullptr);
      break;
    case VK_BOOL:
      if (len) copy_bits(nd.h_values.p, lo, len, pr->local_bits2);
      else pr->local_bits2.assign(8, 0);
      pr->buffers.push_back(pr->local_bits2.data());
      break;
    case VK_UTF8: {
      pr->local_off.resize(len + 1);
      const uint64_t b0 = len ? node_off(nd, lo) : 0;
      for (uint64_t k = 0; k <= len; k++) pr->local_off[k] = len ? (int32_t)(node_off(nd, lo + k) - b0) : 0;
      pr->buffers.push_back(pr->local_off.data());
      pr->buffers.push_back(nd.h_values.p ? nd.h_values.p + b0 : (const uint8_t*)"");
      break;
    }
    case VK_LIST: {
      pr->local_off.resize(len + 1);
      const uint64_t b0 = len ? node_off(nd, lo) : 0, b1 = len ? node_off(nd, hi) : 0;
      for (uint64_t k = 0; k <= len; k++) pr->local_off[k] = len ? (int32_t)(node_off(nd, lo + k) - b0) : 0;
      pr->buffers.push_back(pr->local_off.data());
      std::unique_ptr<ArrowArray> item(new ArrowArray);
      export_node(res, nd.kids.at(0), b0, b1, item.get());
      pr->children.push_back(item.get());
      pr->owned.push_back(std::move(item));
      break;
    }
    case VK_STRUCT:
      for (auto& k : nd.kids) {
        std::unique_ptr<ArrowArray> ca(new ArrowArray);
        export_node(res, k, lo, hi, ca.get());
        pr->children.push_back(ca.get());
        pr->owned.push_back(std::move(ca));
      }
      break;
  }
  finish_array(out);
}

bool VcfStream::next(ArrowArray* out) {
  if (!res->on_host) throw Error("stream was executed device-only: no host batches");
  if (next_batch >= res->n_batches()) return false;
  const uint64_t lo = next_batch * res->batch_size, hi = std::min<uint64_t>(lo + res->batch_size, res->n_rows);
  next_batch++;
  ArrayPriv* top = init_array(out, res, (int64_t)(hi - lo));
  top->buffers.push_back(nullptr);
  for (auto& col : res->cols) {
    std::unique_ptr<ArrowArray> ca(new ArrowArray);
    export_node(res, col, lo, hi, ca.get());
    top->children.push_back(ca.get());
    top->owned.push_back(std::move(ca));
  }
  finish_array(out);
  return true;
}

// ---- partition execution -------------------------------------------------------------------------------------
struct RegionQuery {
  GenomicRegion region;
  std::vector<std::pair<uint64_t, uint64_t>> chunks_abs;  // merged tabix chunks as absolute inflated offsets
};
static uint64_t voff_to_abs(const VcfProvider& p, uint64_t voff) {
  const uint64_t coff = voff >> 16, uo = voff & 0xFFFF;
  if (!p.bgzf) return coff + uo;
  auto it = std::lower_bound(p.blk_coff.begin(), p.blk_coff.end(), coff);
  if (it == p.blk_coff.end() || *it != coff) throw Error("tabix virtual offset does not address a BGZF member");
  return p.blk_uoff[(size_t)(it - p.blk_coff.begin())] + uo;
}
static uint32_t block_of_abs(const VcfProvider& p, uint64_t a) {  // block holding inflated byte a
  auto it = std::upper_bound(p.blk_uoff.begin(), p.blk_uoff.end(), a);
  size_t i = (size_t)(it - p.blk_uoff.begin());
  return (uint32_t)(i ? i - 1 : 0);
}

// A stretch of decoded text on the device: u[0] is the byte at absolute decoded offset `base`, the lines of interest start at
// u[x0] and the decoded bytes end at u[hi]; at_eof: they end with the file.
struct TextSpan {
  const uint8_t* u = nullptr;
  uint64_t base = 0, x0 = 0, hi = 0;
  bool at_eof = false;
};
// delimiter index + per-line keys of a span
struct LineIndex {
  DevBuf<uint64_t> nl, nl_tabs, tab, base_nl, base_tab, scan_tmp;
  DevBuf<uint32_t> cnt_nl, cnt_tab;
  DevBuf<uint32_t> k_pos, k_vend;
  DevBuf<uint8_t> k_flags;
  VcfLines L{};
  uint64_t first_open = 0;   // span offset of the first byte behind the last newline (= x0 when the span holds none)
  bool open_tail = false;    // the span does not end with a newline
};

// What one execute() works with, whatever the size of the pieces it works in: the projection, the residual filter program, the
// regions of the partition, the limit.  run_rows() turns a stretch of decoded text into the Arrow nodes of its rows; the
// one-shot form (device-resident executions, plain-text files) calls it once for the partition's whole span, a host stream
// calls it once per chunk of BGZF members (VcfChunkStream below).
// ---- the header's declarations as the kernels look them up (VcfCheckKind, vcf_kernels.h) --------------------------------
struct DevTypeTable {
  DevBuf<uint8_t> keys;
  DevBuf<VcfTypeSlot> slots;
  VcfTypeTable T{nullptr, nullptr, 0, CK_STR, 0};
  // sel (optional): for every entry 1 + its index among the keys the scan extracts, 0 = none
  void build(const std::vector<std::pair<std::string, uint32_t>>& ents, const std::vector<uint32_t>* sel = nullptr) {
    T = VcfTypeTable{nullptr, nullptr, 0, CK_STR, 0};
    if (ents.empty()) return;
    uint32_t cap = 8;
    while (cap < 2 * ents.size()) cap *= 2;
    std::vector<VcfTypeSlot> h(cap, VcfTypeSlot{0, 0, 0});
    std::string blob;
    bool sel_ok = sel != nullptr;
    for (size_t ei = 0; ei < ents.size(); ei++) {
      auto& e = ents[ei];
      uint64_t k8 = 0;
      for (size_t i = 0; i < e.first.size() && i < 8; i++) k8 |= (uint64_t)(uint8_t)e.first[i] << (8 * i);
      uint32_t i = vcf_key_hash(k8, (uint32_t)e.first.size()) & (cap - 1);
      bool dup = false;
      for (; h[i].len_kind; i = (i + 1) & (cap - 1))
        if ((h[i].len_kind & 0xFFFFFFu) == e.first.size() && blob.compare(h[i].off & 0xFFFFu, e.first.size(), e.first) == 0) { dup = true; break; }
      if (dup) continue;   // (the first declaration of an id counts)
      if (blob.size() + e.first.size() > 0xFFFFu || (sel && (*sel)[ei] > 0xFFFFu)) sel_ok = false;
      h[i] = VcfTypeSlot{k8, (uint32_t)blob.size() | (sel ? (*sel)[ei] << 16 : 0u), (uint32_t)e.first.size() | ((e.second + 1u) << 24)};
      blob += e.first;
    }
    if (!sel_ok) for (auto& sl : h) sl.off &= 0xFFFFu;
    keys.alloc(blob.size() + 1);
    slots.alloc(cap);
    HIP_CHECK(hipMemcpy(keys.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(slots.p, h.data(), cap * sizeof(VcfTypeSlot), hipMemcpyHostToDevice));
    T = VcfTypeTable{keys.p, slots.p, cap - 1, CK_STR, sel_ok ? 1u : 0u};
  }
};
static uint32_t scalar_check_kind(const std::string& type) {
  return type == "Integer" ? CK_INT : type == "Float" ? CK_FLOAT : type == "Character" ? CK_CHAR : type == "Flag" ? CK_FLAG : CK_STR;
}
// INFO: `builders` = the keys load_infos_single_pass has a builder for (every INFO field of the table, projected or not)
static uint32_t info_check_kind(const VcfFieldDefn& d, bool has_builder) {
  if (d.type == "Flag") return CK_FLAG;
  if (d.type == "Character" && has_builder) return CK_UNSUPPORTED;
  if (d.number == "1") return scalar_check_kind(d.type);
  return has_builder ? (CK_LIST | scalar_check_kind(d.type)) : CK_NONE;
}
static uint32_t format_check_kind(const VcfFieldDefn& d, bool has_builder) {
  if (d.id == "GT") return has_builder ? CK_GT : CK_NONE;
  const uint32_t sc = d.type == "Integer" ? CK_INT : d.type == "Float" ? CK_FLOAT : d.type == "Character" ? CK_CHAR : CK_STR;
  if (d.number == "1") return sc;
  return has_builder ? (CK_LIST | (sc == CK_CHAR ? (uint32_t)CK_STR : sc)) : CK_NONE;
}

struct VcfRun {
  const VcfPlan& plan;
  VcfProvider& p;
  hipStream_t st = nullptr;
  const VcfSchema& sch;
  int n_info = 0;
  std::vector<int> col_src;
  bool any_format = false, multi = false, want_end = false;
  uint64_t NS = 0, batch_size = 8192;
  std::vector<RegionQuery> queries;
  bool nothing = false, never = false;
  uint64_t lo_abs = 0, E_abs = 0;     // span of the partition (all its regions)
  // residual filter program + the region names, on the device (indexed plans)
  std::vector<VcfFilterTerm> terms;
  std::vector<uint32_t> chrom_off;
  DevBuf<uint8_t> d_blob;
  DevBuf<VcfFilterTerm> d_terms;
  DevBuf<uint32_t> err;               // error word + the queue of float cells to be rounded exactly (vcf_kernels.h)
  DevTypeTable tt_end, tt_info, tt_format;   // INFO as `Info::get` types it / as load_infos_single_pass does / FORMAT
  int need_end = 0;                   // launch_vcf_keys' argument
  uint64_t rows_left = ~0ull;         // what the plan's limit still allows

  VcfRun(const VcfPlan& pl, int32_t partition, int32_t batch_size_in) : plan(pl), p(*pl.prov), sch(pl.prov->sch) {
    n_info = (int)sch.info_fields.size();
    // projection flags (physical_exec.rs:257-281)
    col_src.resize(plan.out_fields.size());
    for (size_t c = 0; c < plan.out_fields.size(); c++) col_src[c] = plan.has_projection ? plan.projection[c] : (int)c;
    for (int s : col_src) if (s >= 8 + n_info) any_format = true;
    multi = sch.multi;
    NS = sch.samples.size();
    // effective batch size (choose_effective_batch_size; the adaptive re-tune is not restated)
    batch_size = choose_effective_batch_size((uint64_t)batch_size_in, any_format && sch.has_format, sch.format_fields.size(), NS,
                                             p.hdr.samples.size());
    want_end = plan.indexed;
    for (int s : col_src) if (s == 2) want_end = true;
    if (plan.limit >= 0) rows_left = (uint64_t)plan.limit;
    lo_abs = p.hdr.header_bytes;
    E_abs = p.text_len();
    if (plan.indexed && !p.index_readable) throw Error("Failed to open indexed VCF: " + p.index_error);  // physical_exec.rs:2766-2768
    if (plan.indexed) plan_regions(partition);
    if (lo_abs >= p.text_len()) nothing = true;
  }

  // ---- byte range of every region of the partition (tabix query, storage.rs / noodles `Query`) ----
  void plan_regions(int32_t partition) {
    std::lock_guard<std::mutex> lk(p.mu);   // (the per-contig chunk cache)
    uint64_t mn = ~0ull, mx = 0;
    for (auto& r : plan.assignments[(size_t)partition].regions) {
      if (r.unmapped_tail) continue;
      if (r.has_start && r.has_end && r.end < r.start)
        throw Error("Invalid region '" + r.chrom + "': end (" + std::to_string(r.end) + ") is less than start (" + std::to_string(r.start) + ")");
      if ((r.has_start && r.start == 0) || (r.has_end && r.end == 0))
        throw Error("Invalid region '" + r.chrom + "': " + ((r.has_start && r.start == 0) ? "start" : "end") + " position must be >= 1 (got 0)");
      long idx = -1;
      for (size_t i = 0; i < p.tbi.names.size(); i++) if (p.tbi.names[i] == r.chrom) { idx = (long)i; break; }
      if (idx < 0) continue;  // "does not exist in reference sequences": region skipped (physical_exec.rs:2844-2850)
      RegionQuery q;
      q.region = r;
      if (!r.has_start && !r.has_end) {
        // whole-contig query: the merged chunk list only depends on the index, so it is computed once per contig
        // (for the 1000-sample file of BASELINE config 4 this was 6 ms of every 33 ms scan)
        auto it = p.whole_contig_chunks.find((size_t)idx);
        if (it == p.whole_contig_chunks.end()) {
          std::vector<std::pair<uint64_t, uint64_t>> v;
          for (auto& ch : bai_query_chunks(p.tbi.idx, (size_t)idx, false, 0, false, 0)) {
            const uint64_t a = voff_to_abs(p, ch.first), b = voff_to_abs(p, ch.second);
            if (b > a) v.emplace_back(a, b);
          }
          it = p.whole_contig_chunks.emplace((size_t)idx, std::move(v)).first;
        }
        q.chunks_abs = it->second;
        for (auto& c : q.chunks_abs) { mn = std::min(mn, c.first); mx = std::max(mx, c.second); }
      } else {
        auto chunks = bai_query_chunks(p.tbi.idx, (size_t)idx, r.has_start, r.start, r.has_end, r.end);
        if (r.has_end && r.end >= 1 && (size_t)idx < p.tbi.idx.refs.size()) {
          // as for BAM (engine.cpp: build_work_uncached): the bins of the coarser levels reach up to 64 Mb behind the region's
          // end; the first line of the next non-empty LEAF bin behind the region's last 16 kb window has POS > end, and so
          // has every line after it -- chunks from there on hold no row of the answer and would only stretch the span
          const BaiRef& br = p.tbi.idx.refs[(size_t)idx];
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
        for (auto& ch : chunks) {
          const uint64_t a = voff_to_abs(p, ch.first), b = voff_to_abs(p, ch.second);
          if (b > a) { q.chunks_abs.emplace_back(a, b); mn = std::min(mn, a); mx = std::max(mx, b); }
        }
      }
      if (q.chunks_abs.size() > 4096) throw Error("region query expands to more than 4096 index chunks");
      queries.push_back(std::move(q));
    }
    if (mn == ~0ull) nothing = true;
    else { lo_abs = mn; E_abs = mx; }
  }

  // ---- the residual filter program (record_filter.rs over VcfRecordFields) and the region names, uploaded once ----
  void upload_filters(hipStream_t stream) {
    st = stream;
    err.alloc(VCF_ERR_DWORDS);
    HIP_CHECK(hipMemsetAsync(err.p, 0, 16, st));
    build_type_tables();
    if (!plan.indexed) return;
    std::string blob;
    for (auto& f : plan.residual) {
      int field;
      if (f.column == "chrom") field = 0;
      else if (f.column == "start") field = 1;
      else if (f.column == "end") field = 2;
      else if (f.column == "id") field = 3;
      else continue;  // VcfRecordFields knows no other field: the term passes (storage.rs:1000-1024)
      const bool is_str = field == 0 || field == 3;
      VcfFilterTerm tm{};
      tm.field = field;
      tm.op = f.op;
      auto num = [&](const Literal& l, double* v) {
        if (l.kind == BIOSCAN_LIT_INT) { *v = (double)l.i; return true; }
        if (l.kind == BIOSCAN_LIT_FLOAT) { *v = l.f; return true; }
        return false;
      };
      auto put_str = [&](int k, const std::string& s) {
        tm.str_off[k] = (uint32_t)blob.size();
        tm.str_len[k] = (uint32_t)s.size();
        blob += s;
      };
      if (f.op <= BIOSCAN_OP_GE) {
        if (f.values.size() != 1) continue;
        const Literal& l = f.values[0];
        if (l.kind == BIOSCAN_LIT_NULL) { never = true; break; }
        if (is_str) {
          if (l.kind != BIOSCAN_LIT_STR) continue;
          if (f.op != BIOSCAN_OP_EQ && f.op != BIOSCAN_OP_NE) continue;
          put_str(0, l.s);
        } else if (!num(l, &tm.vals[0])) continue;
        tm.n_vals = 1;
      } else if (f.op == BIOSCAN_OP_BETWEEN || f.op == BIOSCAN_OP_NOT_BETWEEN) {
        if (f.values.size() != 2) continue;
        if (f.values[0].kind == BIOSCAN_LIT_NULL || f.values[1].kind == BIOSCAN_LIT_NULL) { never = true; break; }
        if (is_str) continue;
        if (!num(f.values[0], &tm.vals[0]) || !num(f.values[1], &tm.vals[1])) continue;
        tm.n_vals = 2;
      } else {
        // eight literals per term; a longer list continues in the following terms (`more`)
        int k = 0;
        for (auto& l : f.values) {
          if (k == 8) { tm.n_vals = 8; tm.more = 1; terms.push_back(tm); tm.has_null = 0; tm.more = 0; k = 0; }
          if (is_str) {
            if (l.kind == BIOSCAN_LIT_NULL) tm.has_null = 1;
            else if (l.kind == BIOSCAN_LIT_STR) put_str(k++, l.s);
          } else {
            double v;
            if (num(l, &v)) tm.vals[k++] = v; else tm.has_null = 1;
          }
        }
        tm.n_vals = k;
      }
      terms.push_back(tm);
    }
    for (auto& q : queries) { chrom_off.push_back((uint32_t)blob.size()); blob += q.region.chrom; }
    d_blob.alloc(blob.size() + 1);
    if (!blob.empty()) HIP_CHECK(hipMemcpyAsync(d_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
    d_terms.alloc(terms.size() + 1);
    if (!terms.empty()) HIP_CHECK(hipMemcpyAsync(d_terms.p, terms.data(), terms.size() * sizeof(VcfFilterTerm), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));  // (pageable sources)
  }

  // ---- delimiter index and keys of a span; a last line without its newline is a line only at the end of the file ----
  void index_lines(const TextSpan& t, LineIndex& li) {
    const uint64_t x0 = t.x0, hi = t.hi;
    const uint64_t nch = vcf_delim_chunks(x0, hi);
    li.cnt_nl.alloc(nch + 1); li.cnt_tab.alloc(nch + 1);
    li.base_nl.alloc(nch + 2); li.base_tab.alloc(nch + 2);
    li.scan_tmp.alloc(scan_tmp_elems(nch));
    launch_vcf_delim_count(t.u, x0, hi, li.cnt_nl.p, li.cnt_tab.p, st);
    launch_exclusive_scan_u32_to_u64(li.cnt_nl.p, li.base_nl.p, nch, li.scan_tmp.p, st);
    launch_exclusive_scan_u32_to_u64(li.cnt_tab.p, li.base_tab.p, nch, li.scan_tmp.p, st);
    uint64_t n_nl = 0, n_tab = 0;
    uint8_t last = '\n';
    HIP_CHECK(hipMemcpyAsync(&n_nl, li.base_nl.p + nch, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&n_tab, li.base_tab.p + nch, 8, hipMemcpyDeviceToHost, st));
    if (hi > x0) HIP_CHECK(hipMemcpyAsync(&last, t.u + hi - 1, 1, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    li.nl.alloc(n_nl + 1); li.nl_tabs.alloc(n_nl + 1); li.tab.alloc(n_tab + 1);
    launch_vcf_delim_write(t.u, x0, hi, li.base_nl.p, li.base_tab.p, li.nl.p, li.nl_tabs.p, li.tab.p, st);
    VcfLines& L = li.L;
    L.nl = li.nl.p; L.nl_tabs = li.nl_tabs.p; L.tab = li.tab.p;
    L.n_nl = n_nl; L.n_tab = n_tab; L.x0 = x0; L.hi = hi;
    li.open_tail = hi > x0 && last != '\n';
    L.n_lines = n_nl + (li.open_tail && t.at_eof ? 1 : 0);
    li.first_open = x0;
    if (n_nl) {
      uint64_t lastnl = 0;
      HIP_CHECK(hipMemcpyAsync(&lastnl, li.nl.p + n_nl - 1, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      li.first_open = lastnl + 1;
    }
  }
  // (called once the run's device is current)
  void build_type_tables() {
    auto in = [](const std::vector<std::string>& v, const std::string& k) { return std::find(v.begin(), v.end(), k) != v.end(); };
    std::vector<std::pair<std::string, uint32_t>> e_end, e_info, e_fmt;
    std::vector<uint32_t> info_sel;   // 1 + the index build_columns gives a projected INFO column (output order), 0 = not projected
    std::vector<std::string> projected;
    for (size_t c = 0; c < col_src.size(); c++)
      if (col_src[c] >= 8 && col_src[c] < 8 + n_info) projected.push_back(sch.info_fields[(size_t)col_src[c] - 8]);
    uint32_t end_kind = CK_STR;
    bool seen_end = false;
    for (auto& d : p.hdr.infos) {
      e_end.emplace_back(d.id, info_check_kind(d, false));
      e_info.emplace_back(d.id, info_check_kind(d, in(sch.info_fields, d.id)));
      const auto it = std::find(projected.begin(), projected.end(), d.id);
      info_sel.push_back(it == projected.end() ? 0u : (uint32_t)(it - projected.begin()) + 1u);
      if (d.id == "END" && !seen_end) { seen_end = true; end_kind = info_check_kind(d, false); }
    }
    bool seen_gt = false;
    for (auto& d : p.hdr.formats) {
      e_fmt.emplace_back(d.id, format_check_kind(d, sch.has_format && in(sch.format_fields, d.id)));
      seen_gt |= d.id == "GT";
    }
    if (!seen_gt) {   // a genotype is a genotype whatever the header says
      VcfFieldDefn gt; gt.id = "GT"; gt.number = "1"; gt.type = "String";
      e_fmt.emplace_back(gt.id, format_check_kind(gt, sch.has_format && in(sch.format_fields, gt.id)));
    }
    tt_end.build(e_end); tt_info.build(e_info, &info_sel); tt_format.build(e_fmt);
    need_end = want_end ? (int)(1u + end_kind) | (plan.indexed ? 0x100 : 0) : 0;
  }
  void line_keys(const TextSpan& t, LineIndex& li) {
    const VcfLines& L = li.L;
    li.k_pos.alloc(L.n_lines + 1); li.k_vend.alloc(L.n_lines + 1); li.k_flags.alloc(L.n_lines + 1);
    launch_vcf_keys(t.u, L, li.k_pos.p, li.k_vend.p, li.k_flags.p, need_end, tt_end.T, err.p, st);
  }
  void check_err() {
    uint32_t e = 0;
    HIP_CHECK(hipMemcpyAsync(&e, err.p, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    // sequential scan: an error in a line beyond `limit` is never reached by the reference; lines are
    // validated as a whole here, which only matters for malformed files
    throw_vcf_err(e);
  }

  // ---- row selection: the lines of the span that queries [q0, q1) return, region after region, at most `cap` ----
  uint64_t select_rows(const TextSpan& t, LineIndex& li, size_t q0, size_t q1, uint64_t cap, DevBuf<uint64_t>& rows) {
    const VcfLines& L = li.L;
    uint64_t n = 0;
    if (!plan.indexed) {
      n = std::min<uint64_t>(L.n_lines, cap);
      rows.alloc(std::max<uint64_t>(n, 1));
      launch_vcf_iota_rows(rows.p, n, st);
      check_err();
      return n;
    }
    check_err();
    const uint64_t base = t.base;
    // per region: keep flags over the lines its chunks span, scan, compact
    const size_t nq = q1 - q0;
    std::vector<DevBuf<uint32_t>> keeps(nq);
    std::vector<DevBuf<uint64_t>> scans(nq);
    std::vector<uint64_t> totals(nq, 0), i_los(nq, 0), i_ns(nq, 0);
    // two host round trips for all regions together: the line ranges of the regions' chunk spans first, then -- after every
    // region's flags and scan have been queued -- the totals.  A region's chunks are clipped to the span (a stream's chunk
    // holds a part of them): positions relative to the span's first byte.
    auto rel = [&](uint64_t a) { return a <= base ? 0ull : std::min<uint64_t>(a - base, t.hi); };
    DevBuf<unsigned long long> lb(2 * nq + 2);
    std::vector<unsigned long long> h_lb(2 * nq + 2, 0);
    for (size_t k = 0; k < nq && !never; k++) {
      auto& q = queries[q0 + k];
      if (q.chunks_abs.empty()) continue;
      launch_vcf_line_lower_bound(L, rel(q.chunks_abs.front().first), lb.p + 2 * k, st);
      launch_vcf_line_lower_bound(L, rel(q.chunks_abs.back().second), lb.p + 2 * k + 1, st);
    }
    if (nq && !never) {
      HIP_CHECK(hipMemcpyAsync(h_lb.data(), lb.p, (2 * nq) * 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
    }
    std::vector<DevBuf<uint64_t>> d_chs(nq);
    uint64_t max_cnt = 0;
    for (size_t k = 0; k < nq && !never; k++)
      if (!queries[q0 + k].chunks_abs.empty() && h_lb[2 * k + 1] > h_lb[2 * k]) max_cnt = std::max<uint64_t>(max_cnt, h_lb[2 * k + 1] - h_lb[2 * k]);
    DevBuf<uint64_t> tmp2(scan_tmp_elems(std::max<uint64_t>(max_cnt, 1)));  // the scans run one after the other on the stream
    bool queued = false;
    for (size_t k = 0; k < nq && !never; k++) {
      auto& q = queries[q0 + k];
      if (q.chunks_abs.empty()) continue;
      const unsigned long long i_lo = h_lb[2 * k], i_hi = std::min<unsigned long long>(h_lb[2 * k + 1], L.n_lines);
      if (i_hi <= i_lo) continue;
      std::vector<uint64_t> ch;
      for (auto& c : q.chunks_abs) {
        if (c.second <= base || c.first >= base + t.hi) continue;   // (not in this span)
        ch.push_back(rel(c.first)); ch.push_back(rel(c.second));
      }
      if (ch.empty()) continue;
      d_chs[k].alloc(ch.size());
      HIP_CHECK(hipMemcpyAsync(d_chs[k].p, ch.data(), ch.size() * 8, hipMemcpyHostToDevice, st));  // pageable source: copied before the call returns
      VcfRowSelect S{};
      S.mode = 1;
      S.n_chunks = (int32_t)(ch.size() / 2);
      S.i_lo = i_lo; S.i_hi = i_hi;
      S.chrom_off = chrom_off[q0 + k];
      S.chrom_len = (uint32_t)q.region.chrom.size();
      S.q_start1 = q.region.has_start ? (int64_t)q.region.start : 1;
      S.q_end1 = q.region.has_end ? (int64_t)std::min<uint64_t>(q.region.end, 1ull << 29) : (int64_t)(1ull << 29);
      S.start1 = q.region.has_start ? (int64_t)q.region.start : 0;
      S.end1 = q.region.has_end ? (int64_t)q.region.end : 0;
      S.zero_based = p.zero_based ? 1 : 0;
      S.n_terms = (int32_t)terms.size();
      const uint64_t cntl = i_hi - i_lo;
      keeps[k].alloc(cntl + 1);
      scans[k].alloc(cntl + 2);
      launch_vcf_row_flags(t.u, L, li.k_pos.p, li.k_vend.p, li.k_flags.p, S, d_chs[k].p, d_terms.p, d_blob.p, keeps[k].p, st);
      launch_exclusive_scan_u32_to_u64(keeps[k].p, scans[k].p, cntl, tmp2.p, st);
      HIP_CHECK(hipMemcpyAsync(&totals[k], scans[k].p + cntl, 8, hipMemcpyDeviceToHost, st));
      queued = true;
      i_los[k] = i_lo;
      i_ns[k] = cntl;
    }
    if (queued) HIP_CHECK(hipStreamSynchronize(st));
    uint64_t total = 0;
    for (auto v : totals) total += v;
    n = std::min<uint64_t>(total, cap);
    rows.alloc(std::max<uint64_t>(n, 1));
    uint64_t rb = 0;
    for (size_t k = 0; k < nq && rb < n; k++) {
      if (!totals[k]) continue;
      launch_vcf_compact(keeps[k].p, scans[k].p, i_ns[k], i_los[k], rows.p, rb, n - rb, st);
      rb += std::min<uint64_t>(totals[k], n - rb);
    }
    HIP_CHECK(hipStreamSynchronize(st));
    return n;
  }

  // ---- the Arrow nodes of `n` selected lines ----
  void build_columns(const TextSpan& t, LineIndex& li, const DevBuf<uint64_t>& rows, uint64_t n, VResult& R) {
    VResult* res = &R;
    const uint8_t* u = t.u;
    const VcfLines& L = li.L;
    DevBuf<uint32_t>& k_pos = li.k_pos; DevBuf<uint32_t>& k_vend = li.k_vend; DevBuf<uint8_t>& k_flags = li.k_flags;
    const std::vector<VField>& out_fields = plan.out_fields;
    res->n_rows = n;
    res->cols.resize(out_fields.size());
    for (size_t c = 0; c < out_fields.size(); c++) { res->cols[c].fd = out_fields[c]; res->cols[c].n = n; }
    Ctx cx{p, st, u, {}, 0, err.p, 0, {}};
    if (n) {
      int core_col[8];
      for (int k = 0; k < 8; k++) core_col[k] = -1;
      std::vector<std::pair<int, int>> info_cols, fmt_cols;  // (output column, field index)
      int geno_col = -1;
      for (size_t c = 0; c < out_fields.size(); c++) {
        const int s = col_src[c];
        if (s < 8) core_col[s] = (int)c;
        else if (s < 8 + n_info) info_cols.emplace_back((int)c, s - 8);
        else if (multi) geno_col = (int)c;
        else fmt_cols.emplace_back((int)c, s - 8 - n_info);
      }
      // core
      {
        VcfCoreCols C{};
        DevBuf<uint64_t> src[5];
        DevBuf<uint32_t> len[5];
        const int sidx[5] = {0, 3, 4, 5, 7};
        uint64_t** sp[5] = {&C.src_chrom, &C.src_id, &C.src_ref, &C.src_alt, &C.src_filter};
        uint32_t** lp[5] = {&C.len_chrom, &C.len_id, &C.len_ref, &C.len_alt, &C.len_filter};
        for (int k = 0; k < 5; k++)
          if (core_col[sidx[k]] >= 0) { src[k].alloc(n); len[k].alloc(n); *sp[k] = src[k].p; *lp[k] = len[k].p; }
        if (core_col[1] >= 0) { auto& nd = res->cols[core_col[1]]; nd.d_values.alloc(n * 4); C.start = (uint32_t*)nd.d_values.p; cx.arrow_bytes += n * 4; }
        if (core_col[2] >= 0) { auto& nd = res->cols[core_col[2]]; nd.d_values.alloc(n * 4); C.end = (uint32_t*)nd.d_values.p; cx.arrow_bytes += n * 4; }
        if (core_col[6] >= 0) {
          auto& nd = res->cols[core_col[6]];
          nd.d_values.alloc(n * 8);
          nd.d_valid.alloc((n + 63) / 64);
          nd.all_valid = false;
          C.qual = (double*)nd.d_values.p;
          C.v_qual = nd.d_valid.p;
          cx.arrow_bytes += n * 8 + (n + 63) / 64 * 8;
        }
        launch_vcf_core(u, L, rows.p, n, k_pos.p, k_vend.p, k_flags.p, C, p.zero_based ? 1 : 0, err.p, st);
        for (int k = 0; k < 5; k++)
          if (core_col[sidx[k]] >= 0) {
            VNode& nd = res->cols[core_col[sidx[k]]];
            finish_utf8(cx, nd, src[k].p, len[k].p, n);
            if (sidx[k] == 5) launch_replace_byte(nd.d_values.p, nd.total, ',', '|', st);  // alleles re-joined with '|'
          }
        HIP_CHECK(hipStreamSynchronize(st));
      }
      // INFO
      if (!info_cols.empty()) {
        std::string blob;
        std::vector<uint32_t> koff{0};
        std::vector<uint8_t> unsupported;   // Character keys: noodles hands a Value::Character, which has no builder (:632-636)
        for (auto& ic : info_cols) {
          const std::string& tag = sch.info_fields[ic.second];
          blob += tag;
          koff.push_back((uint32_t)blob.size());
          const VcfFieldDefn* d = p.hdr.info(tag);
          unsupported.push_back(d && d->type == "Character" ? 1 : 0);
        }
        const int K = (int)info_cols.size();
        DevBuf<uint8_t> d_keys(blob.size() + 1), d_unsup(unsupported.size());
        DevBuf<uint32_t> d_koff(koff.size());
        HIP_CHECK(hipMemcpyAsync(d_keys.p, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_koff.p, koff.data(), koff.size() * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_unsup.p, unsupported.data(), unsupported.size(), hipMemcpyHostToDevice, st));
        DevBuf<uint64_t> sp_off((uint64_t)K * n);
        DevBuf<uint32_t> sp_len((uint64_t)K * n);
        DevBuf<uint8_t> sp_state((uint64_t)K * n);
        launch_vcf_info_locate(u, L, rows.p, n, d_keys.p, d_koff.p, d_unsup.p, K, tt_info.T, sp_off.p, sp_len.p, sp_state.p, err.p, st);
        for (int k = 0; k < K; k++) {
          VNode& nd = res->cols[info_cols[k].first];
          build_from_spans(cx, nd, sp_off.p + (uint64_t)k * n, sp_len.p + (uint64_t)k * n, sp_state.p + (uint64_t)k * n, n);
        }
        HIP_CHECK(hipStreamSynchronize(st));
      }
      // FORMAT
      if (sch.has_format && (geno_col >= 0 || !fmt_cols.empty())) {
        std::vector<int> sel;   // FORMAT field indices to extract, in the order the kernels number them
        std::vector<int> dst;   // where key s goes: child of `genotypes` (multi-sample) / entry of fmt_cols (single-sample)
        if (geno_col >= 0) for (size_t k = 0; k < sch.format_fields.size(); k++) { sel.push_back((int)k); dst.push_back((int)k); }
        else for (size_t k = 0; k < fmt_cols.size(); k++) { sel.push_back(fmt_cols[k].second); dst.push_back((int)k); }
        // GT first: only the first VCF_MAX_DIRECT keys are parsed inside the cell kernel, and a genotype has to be (it is validated
        // and sized there, and rendered again when an allele has leading zeros -- as the ninth key it would come out as written)
        for (size_t k = 1; k < sel.size(); k++)
          if (sch.format_fields[sel[k]] == "GT") { std::rotate(sel.begin(), sel.begin() + k, sel.begin() + k + 1); std::rotate(dst.begin(), dst.begin() + k, dst.begin() + k + 1); break; }
        const int S = (int)sel.size();
        std::string blob;
        std::vector<uint32_t> koff{0};
        int gt_field = -1;
        uint32_t char_mask = 0;   // selected Character scalars: one character, or the record is an error
        for (int s = 0; s < S; s++) {
          const std::string& tag = sch.format_fields[sel[s]];
          if (tag == "GT") gt_field = s;
          const VcfFieldDefn* d = p.hdr.format(tag);
          if (s < 32 && tag != "GT" && d && d->type == "Character" && d->number == "1") char_mask |= 1u << s;
          blob += tag;
          koff.push_back((uint32_t)blob.size());
        }
        DevBuf<uint8_t> d_keys(blob.size() + 1);
        DevBuf<uint32_t> d_koff(koff.size());
        HIP_CHECK(hipMemcpyAsync(d_keys.p, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_koff.p, koff.data(), koff.size() * 4, hipMemcpyHostToDevice, st));
        // single-sample source: the one sample (load_formats_single_pass takes `sample_count` leading samples)
        std::vector<int32_t> scol = multi ? sch.sample_header_index : std::vector<int32_t>{0};
        const uint64_t ns = scol.size();
        DevBuf<int32_t> d_scol(ns);
        HIP_CHECK(hipMemcpyAsync(d_scol.p, scol.data(), ns * 4, hipMemcpyHostToDevice, st));
        DevBuf<int16_t> fpos(n * (uint64_t)S);
        DevBuf<uint64_t> cmap(std::max<uint64_t>(n, 1));
        launch_vcf_format_keys(u, L, rows.p, n, d_keys.p, d_koff.p, S, tt_format.T, char_mask, fpos.p, cmap.p, err.p, st);
        const uint64_t N = n * ns;
        DevBuf<uint64_t> sp_off((uint64_t)S * N);
        DevBuf<uint32_t> sp_len((uint64_t)S * N);
        DevBuf<uint8_t> sp_state((uint64_t)S * N);
        // shape of the output nodes first: scalar Int32 / Float32 keys are parsed inside the cell kernel straight into
        // their value buffers; strings and lists go through spans
        std::vector<VNode*> leaf((size_t)S);
        if (geno_col >= 0) {
          VNode& g = res->cols[geno_col];
          g.kids.resize((size_t)S);
          for (int s = 0; s < S; s++) {
            VNode& lst = g.kids[(size_t)dst[s]];
            lst.fd = g.fd.children.at((size_t)dst[s]);
            lst.n = n;
            lst.total = N;
            lst.d_off.alloc(n + 1);
            launch_stride_offsets(lst.d_off.p, n, ns, st);
            lst.kids.resize(1);
            lst.kids[0].fd = lst.fd.children.at(0);
            leaf[s] = &lst.kids[0];
            cx.arrow_bytes += (n + 1) * 4;
          }
        } else {
          for (int s = 0; s < S; s++) leaf[s] = &res->cols[fmt_cols[(size_t)dst[s]].first];
        }
        VcfCellDirect D{};
        const uint64_t nwN = (N + 63) / 64;
        DevBuf<uint64_t> gt_src;
        DevBuf<uint32_t> gt_len;
        for (int s = 0; s < S && s < VCF_MAX_DIRECT; s++) {
          VNode& nd = *leaf[s];
          if (s == gt_field && nd.fd.kind == VK_UTF8) {   // GT: length + source + validity straight from the cell kernel
            nd.n = N;
            nd.d_valid.alloc(std::max<uint64_t>(nwN, 1));
            nd.all_valid = false;
            gt_src.alloc(std::max<uint64_t>(N, 1));
            gt_len.alloc(std::max<uint64_t>(N, 1));
            D.kind[s] = 3;
            D.values[s] = gt_len.p;
            D.src[s] = gt_src.p;
            D.valid[s] = nd.d_valid.p;
            cx.arrow_bytes += nwN * 8;
            continue;
          }
          if (nd.fd.kind != VK_INT32 && nd.fd.kind != VK_FLOAT32) continue;
          nd.n = N;
          nd.d_values.alloc(std::max<uint64_t>(N, 1) * 4);
          nd.d_valid.alloc(std::max<uint64_t>(nwN, 1));
          nd.all_valid = false;
          D.kind[s] = nd.fd.kind == VK_INT32 ? 1 : 2;
          D.values[s] = (uint32_t*)nd.d_values.p;
          D.valid[s] = nd.d_valid.p;
          cx.arrow_bytes += N * 4 + nwN * 8;
        }
        launch_vcf_format_cells(u, L, rows.p, n, d_scol.p, (int)ns, fpos.p, S, gt_field, D, sp_off.p, sp_len.p, sp_state.p, err.p, st);
        // err[2]: GT cells with leading zeros in an allele (sized for the re-rendered genotype, written below); err[3]: values of
        // keys without a column in this scan, typed by the header all the same
        uint32_t after_cells[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(after_cells, err.p + 2, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (after_cells[0] || after_cells[1]) HIP_CHECK(hipMemsetAsync(err.p + 2, 0, 8, st));
        if (after_cells[1]) launch_vcf_format_check(u, L, rows.p, n, d_scol.p, (int)ns, cmap.p, tt_format.T, err.p, st);
        for (int s = 0; s < S; s++) {
          if (s < VCF_MAX_DIRECT && D.kind[s] == 3) {
            finish_utf8(cx, *leaf[s], gt_src.p, gt_len.p, N);
            if (after_cells[0]) launch_gt_render(u, gt_src.p, leaf[s]->d_off.p, leaf[s]->d_values.p, N, st);
            continue;
          }
          if (s < VCF_MAX_DIRECT && D.kind[s] != 0) continue;
          build_from_spans(cx, *leaf[s], sp_off.p + (uint64_t)s * N, sp_len.p + (uint64_t)s * N, sp_state.p + (uint64_t)s * N, N);
        }
        HIP_CHECK(hipStreamSynchronize(st));
      }
      uint32_t e2[2] = {0, 0};
      HIP_CHECK(hipMemcpyAsync(e2, err.p, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      throw_vcf_err(e2[0]);
      if (e2[1]) {  // float literals within 2^-52 of a rounding boundary: decided by exact integer comparison
        launch_f32_fix(err.p, e2[1], st);
        HIP_CHECK(hipStreamSynchronize(st));
      }
    } else {
      // zero rows: struct / list kids still need their (empty) shape for export
      std::function<void(VNode&)> shape = [&](VNode& nd) {
        nd.n = 0;
        for (auto& cf : nd.fd.children) {
          nd.kids.emplace_back();
          nd.kids.back().fd = cf;
          shape(nd.kids.back());
        }
      };
      for (auto& col : res->cols) shape(col);
    }
    res->stats.arrow_bytes = cx.arrow_bytes;
  }
};

// ---- one-shot execution: the partition's whole span at once, the result left in HBM or copied to the host in one piece
//      (device-resident executions and their list UDFs; plain-text files, whose text is the resident image) ----
static std::shared_ptr<VResult> execute_oneshot(const VcfPlan& plan, int32_t partition, int32_t batch_size_in, bool device_only) {
  VcfProvider& p = *plan.prov;
  VcfRun run(plan, partition, batch_size_in);
  std::lock_guard<std::mutex> lk(p.mu);
  const auto wall0 = std::chrono::steady_clock::now();
  const bool dbg_wall = env_knobs().laps;
  auto lap = [&](const char* what) {
    if (dbg_wall) fprintf(stderr, "[bioscan] vcf execute: %-22s at %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
  };
  p.make_resident();
  p.set_device();
  hipStream_t st = p.stream;
  auto res = std::make_shared<VResult>();
  res->device = p.device;
  res->stream = st;
  res->batch_size = run.batch_size;
  lap("planning done");
  Timer t(st);
  run.upload_filters(st);
  lap("timer+err ready");
  LineIndex li;
  DevBuf<uint64_t> rows;
  TextSpan ts;
  uint64_t n = 0;
  if (!run.nothing) {
    uint32_t extra = 1;
    for (;;) {
      uint64_t hi_abs;
      if (p.bgzf) {
        const uint32_t b_lo = block_of_abs(p, run.lo_abs);
        uint32_t b_hi = std::min<uint32_t>(p.n_blocks(), block_of_abs(p, run.E_abs ? run.E_abs - 1 : 0) + 1 + (plan.indexed ? extra : 0));
        if (!plan.indexed) b_hi = p.n_blocks();
        ts.base = p.blk_uoff[b_lo];
        hi_abs = p.blk_uoff[b_hi];
        const uint64_t bytes = hi_abs - ts.base;
        if (p.d_u.n < bytes + 64) p.d_u.alloc(bytes + 64);
        t.start();
        p.launch_inflate(p.d_u.p, b_hi - b_lo, b_lo);
        lap("inflate launched");
        res->stats.ms_inflate += t.stop();
        lap("inflate done");
        t.start();
        p.launch_crc(p.d_u.p, b_hi - b_lo, b_lo);
        res->stats.ms_crc += t.stop();
        lap("inflate+crc done");
        p.check_inflate_status(b_lo, b_hi - b_lo);
        lap("status checked");
        ts.u = p.d_u.p;
        res->stats.n_blocks = b_hi - b_lo;
        res->stats.compressed_bytes = p.blk_coff[b_hi] - p.blk_coff[b_lo];
        res->stats.inflated_bytes = bytes;
      } else {
        ts.base = 0;
        hi_abs = p.file_len;
        ts.u = p.d_comp.p;
        res->stats.inflated_bytes = hi_abs - run.lo_abs;
      }
      ts.at_eof = hi_abs == p.text_len();
      ts.x0 = run.lo_abs - ts.base;
      ts.hi = hi_abs - ts.base;
      t.start();
      run.index_lines(ts, li);
      if (plan.indexed && !ts.at_eof) {
        // every line that starts before the last chunk end must be complete in the decoded bytes
        if (li.first_open < run.E_abs - ts.base) { extra *= 4; res->stats.ms_chain += t.stop(); continue; }
      } else if (!ts.at_eof && li.open_tail) {
        throw Error("VCF read error: decoded range ends inside a record");
      }
      run.line_keys(ts, li);
      res->stats.ms_chain += t.stop();
      lap("index+keys done");
      break;
    }
    res->stats.n_records = li.L.n_lines;
    t.start();
    n = run.select_rows(ts, li, 0, run.queries.size(), run.rows_left, rows);
    res->stats.ms_select = t.stop();
    lap("select done");
  }
  res->stats.n_rows = n;
  t.start();
  run.build_columns(ts, li, rows, n, *res);
  res->stats.ms_extract = t.stop();
  lap("extract done");
  res->stats.ms_total_gpu = res->stats.ms_inflate + res->stats.ms_crc + res->stats.ms_chain + res->stats.ms_select + res->stats.ms_extract;
  res->stats.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  if (!device_only) {
    for (auto& col : res->cols) copy_node_to_host(col, st);
    HIP_CHECK(hipStreamSynchronize(st));
    for (auto& col : res->cols) free_node_device(col);
    res->on_host = true;
  }
  lap("end");
  return res;
}

// ---- the chunk pipeline of a host stream (bio-format-vcf/src/physical_exec.rs:66-232, 912-1198: the reference reads line by
//      line and hands out a batch at a time, `EmissionType::Incremental`) ----
// The partition's work is a list of items -- the whole data section of a sequential plan, or one item per region of an
// indexed one, in assignment order (rows come region after region, as the reference's queries do).  An item's members are
// inflated `chunk_members` at a time; a chunk's rows are its complete lines that the item's predicate keeps; the line cut
// by the chunk's end is carried, as bytes, to the head of the next chunk's buffer.  HBM and host memory: O(chunk).
struct VcfChunkStream : VcfStreamI {
  VcfRun run;
  VcfProvider& p;
  const uint32_t chunk_members;
  const bool ramp;
  std::shared_ptr<DeviceImage> img;
  K1Ctx k1;
  hipStream_t st = nullptr;
  struct Item { size_t q0, q1; uint64_t lo_abs, E_abs; };
  std::vector<Item> items;
  size_t item = 0;
  bool item_open = false, dead = false;
  uint32_t next_member = 0, chunks_done = 0, ext = 1;
  uint64_t carry_len = 0, skip = 0;
  DevBuf<uint8_t> ubuf[2];
  int cur = 0;
  bioscan_scan_stats total{};
  // batches are cut from the chunks' rows in order; one that straddles two chunks is concatenated on the host
  std::deque<std::shared_ptr<VResult>> ready;
  uint64_t ready_off = 0;   // rows of ready.front() already handed out
  bool exhausted = false;

  VcfChunkStream(const VcfPlan& plan, int32_t partition, int32_t batch_size_in)
      : run(plan, partition, batch_size_in), p(*plan.prov),
        chunk_members(p.chunk_members ? p.chunk_members : env_knobs().chunk_members), ramp(!p.chunk_members) {
    if (!run.nothing) {
      if (!plan.indexed) items.push_back(Item{0, 0, run.lo_abs, p.text_len()});
      else
        for (size_t q = 0; q < run.queries.size(); q++) {
          auto& c = run.queries[q].chunks_abs;
          if (c.empty()) continue;
          uint64_t mn = ~0ull, mx = 0;
          for (auto& ch : c) { mn = std::min(mn, ch.first); mx = std::max(mx, ch.second); }
          items.push_back(Item{q, q + 1, mn, mx});
        }
    }
    uint32_t m_lo = p.n_blocks(), m_hi = 0;
    for (auto& it : items) {
      m_lo = std::min(m_lo, block_of_abs(p, it.lo_abs));
      m_hi = std::max<uint32_t>(m_hi, std::min<uint32_t>(p.n_blocks(), block_of_abs(p, it.E_abs ? it.E_abs - 1 : 0) + 2));
    }
    if (m_hi < m_lo) m_lo = m_hi = 0;
    img = p.image_for(p.device, m_lo, m_hi);
    HIP_CHECK(hipSetDevice(img->device));
    p.init_ctx(k1, *img, std::min<uint32_t>(chunk_members, std::max<uint32_t>(m_hi - m_lo, 1)), false);
    st = k1.stream;
    run.upload_filters(st);
  }
  ~VcfChunkStream() override {
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (img) (void)hipSetDevice(img->device);
    if (k1.stream) (void)hipStreamSynchronize(k1.stream);
    (void)hipSetDevice(prev);
  }
  uint32_t chunk_len(uint32_t k) const {
    uint64_t c = chunk_members;
    if (ramp) c = std::min<uint64_t>(c, 2048ull << std::min<uint32_t>(k, 16));
    return (uint32_t)std::max<uint64_t>(c, 1);
  }

  // the next chunk that has rows (on the host), or nullptr at the end of the partition
  std::shared_ptr<VResult> next_chunk() {
    for (;;) {
      if (run.rows_left == 0) return nullptr;
      if (!item_open) {
        if (item >= items.size()) return nullptr;
        const Item& it = items[item];
        next_member = block_of_abs(p, it.lo_abs);
        skip = it.lo_abs - p.blk_uoff[next_member];
        carry_len = 0;
        ext = 1;
        item_open = true;
      }
      auto res = run_chunk(items[item]);
      if (!item_open) item++;
      if (res && res->n_rows) return res;
    }
  }

  std::shared_ptr<VResult> run_chunk(const Item& it) {
    Timer t(st);
    const auto wall0 = std::chrono::steady_clock::now();
    auto res = std::make_shared<VResult>();
    res->device = img->device;
    res->stream = st;
    res->batch_size = run.batch_size;
    const uint32_t m0 = next_member;
    const uint32_t m_end = std::min<uint32_t>(p.n_blocks(), block_of_abs(p, it.E_abs ? it.E_abs - 1 : 0) + 1);  // members that hold the item's span
    uint32_t m1 = std::min<uint32_t>(p.n_blocks(), m0 + chunk_len(chunks_done));
    if (m0 < m_end) m1 = std::min(m1, std::max(m_end, m0 + 1));   // a chunk ends with the item's span at the latest ...
    else { m1 = (uint32_t)std::min<uint64_t>(p.n_blocks(), (uint64_t)m0 + ext); ext *= 2; }  // ... behind it: 1, 2, 4, ... members until the span's last line is whole
    if (m1 > img->m_hi || m0 < img->m_lo) img = p.image_for(img->device, std::min(m0, img->m_lo), std::max(m1, img->m_hi));
    const uint64_t bytes = p.blk_uoff[m1] - p.blk_uoff[m0];
    const uint64_t L = carry_len + bytes;
    if (ubuf[cur].n < L + 64) {
      DevBuf<uint8_t> g(L + 64);
      if (carry_len) HIP_CHECK(hipMemcpyAsync(g.p, ubuf[cur].p, carry_len, hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipStreamSynchronize(st));
      ubuf[cur] = std::move(g);
    }
    uint8_t* u = ubuf[cur].p;
    if (m1 > m0) {
      if (k1.status.n < m1 - m0) p.init_ctx(k1, *img, m1 - m0, false);
      t.start();
      p.launch_inflate(k1, *img, u + carry_len, m1 - m0, m0);
      res->stats.ms_inflate = t.stop();
      t.start();
      p.launch_crc(k1, *img, u + carry_len, m1 - m0, m0);
      res->stats.ms_crc = t.stop();
      p.check_inflate_status(k1, m0, m1 - m0);
    }
    res->stats.n_blocks = m1 - m0;
    res->stats.compressed_bytes = p.blk_coff[m1] - p.blk_coff[m0];
    res->stats.inflated_bytes = bytes;
    TextSpan ts;
    ts.u = u;
    ts.base = p.blk_uoff[m0] - carry_len;
    ts.x0 = carry_len ? 0 : skip;    // (the carried bytes begin with a line; the item's first chunk begins at its first byte)
    ts.hi = L;
    ts.at_eof = m1 == p.n_blocks();
    skip = 0;
    t.start();
    LineIndex li;
    run.index_lines(ts, li);
    if (!ts.at_eof) li.L.n_lines = li.L.n_nl;   // the open tail is the next chunk's first line
    run.line_keys(ts, li);
    res->stats.ms_chain = t.stop();
    res->stats.n_records = li.L.n_lines;
    t.start();
    DevBuf<uint64_t> rows;
    const uint64_t n = run.select_rows(ts, li, it.q0, it.q1, run.rows_left, rows);
    res->stats.ms_select = t.stop();
    res->stats.n_rows = n;
    t.start();
    if (n) run.build_columns(ts, li, rows, n, *res);
    res->stats.ms_extract = t.stop();
    if (run.rows_left != ~0ull) run.rows_left -= n;
    // ---- is the item over?  every line that starts in front of the span's end has been seen whole ----
    const bool over = ts.at_eof || ts.base + li.first_open >= it.E_abs || run.rows_left == 0;
    if (!over && m1 == p.n_blocks() && li.open_tail) throw Error("VCF read error: decoded range ends inside a record");
    next_member = m1;
    chunks_done++;
    if (over) { item_open = false; carry_len = 0; }
    else {
      const uint64_t c = L - li.first_open;
      const int nxt = cur ^ 1;
      const uint64_t next_bytes = p.blk_uoff[std::min<uint64_t>(p.n_blocks(), (uint64_t)m1 + chunk_len(chunks_done))] - p.blk_uoff[m1];
      if (ubuf[nxt].n < c + next_bytes + 64) ubuf[nxt].alloc(c + next_bytes + 64);
      if (c) HIP_CHECK(hipMemcpyAsync(ubuf[nxt].p, u + li.first_open, c, hipMemcpyDeviceToDevice, st));
      carry_len = c;
      cur = nxt;
    }
    if (n) {
      for (auto& col : res->cols) copy_node_to_host(col, st);
      HIP_CHECK(hipStreamSynchronize(st));
      for (auto& col : res->cols) free_node_device(col);
      res->on_host = true;
    } else HIP_CHECK(hipStreamSynchronize(st));
    res->stats.ms_total_gpu = res->stats.ms_inflate + res->stats.ms_crc + res->stats.ms_chain + res->stats.ms_select + res->stats.ms_extract;
    res->stats.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return n ? res : nullptr;
  }

  bool next(ArrowArray* out) override;
  void list_udf(const char*, int32_t, double, bioscan_udf_stats*) override {
    throw Error("list UDFs on a stream need a device-resident execution (bioscan_execute_device)");
  }
};

// ---- a batch out of several chunks' rows: concatenation on the host (one batch per chunk boundary at most) ----
struct NodePiece { const VNode* nd; uint64_t lo, hi; };
static void append_bits(uint8_t* dst, uint64_t pos, const uint8_t* src, uint64_t lo, uint64_t n) {  // dst is zero-initialised
  for (uint64_t k = 0; k < n; k++) {   // (at most one batch of rows per chunk boundary goes through here)
    const uint64_t s = lo + k;
    if ((src[s >> 3] >> (s & 7)) & 1u) dst[(pos + k) >> 3] |= (uint8_t)(1u << ((pos + k) & 7));
  }
}
static void concat_nodes(VNode& dst, const std::vector<NodePiece>& ps) {
  dst.fd = ps[0].nd->fd;
  uint64_t n = 0;
  bool nullable = false;
  for (auto& q : ps) { n += q.hi - q.lo; if (!q.nd->all_valid && q.nd->h_valid.p) nullable = true; }
  dst.n = n;
  const uint64_t nw = (n + 63) / 64;
  if (nullable) {
    dst.all_valid = false;
    dst.h_valid.alloc(nw * 8 + 16);
    memset(dst.h_valid.p, 0, nw * 8 + 16);
    uint64_t pos = 0;
    for (auto& q : ps) {
      const uint64_t len = q.hi - q.lo;
      if (!q.nd->all_valid && q.nd->h_valid.p) append_bits(dst.h_valid.p, pos, q.nd->h_valid.p, q.lo, len);
      else for (uint64_t k = 0; k < len; k++) dst.h_valid.p[(pos + k) >> 3] |= (uint8_t)(1u << ((pos + k) & 7));
      pos += len;
    }
  }
  switch (dst.fd.kind) {
    case VK_INT32: case VK_UINT32: case VK_FLOAT32: case VK_FLOAT64: {
      const uint32_t w = fixed_width(dst.fd.kind);
      dst.h_values.alloc(n * w + 16);
      uint64_t pos = 0;
      for (auto& q : ps) {
        const uint64_t len = q.hi - q.lo;
        if (len && q.nd->h_values.p) memcpy(dst.h_values.p + pos * w, q.nd->h_values.p + q.lo * w, len * w);
        pos += len;
      }
      break;
    }
    case VK_BOOL: {
      dst.h_values.alloc(nw * 8 + 16);
      memset(dst.h_values.p, 0, nw * 8 + 16);
      uint64_t pos = 0;
      for (auto& q : ps) {
        const uint64_t len = q.hi - q.lo;
        if (len && q.nd->h_values.p) append_bits(dst.h_values.p, pos, q.nd->h_values.p, q.lo, len);
        pos += len;
      }
      break;
    }
    case VK_UTF8: {
      uint64_t bytes = 0;
      for (auto& q : ps) if (q.hi > q.lo) bytes += node_off(*q.nd, q.hi) - node_off(*q.nd, q.lo);
      dst.total = bytes;
      dst.h_off.alloc((n + 1) * 8 + 16);
      dst.h_values.alloc(bytes + 16);
      uint64_t* d = (uint64_t*)dst.h_off.p;
      uint64_t pos = 0, b = 0;
      d[0] = 0;
      for (auto& q : ps) {
        if (q.hi <= q.lo) continue;
        const uint64_t b0 = node_off(*q.nd, q.lo), b1 = node_off(*q.nd, q.hi);
        for (uint64_t k = q.lo; k < q.hi; k++) d[++pos] = b + (node_off(*q.nd, k + 1) - b0);
        if (b1 > b0) memcpy(dst.h_values.p + b, q.nd->h_values.p + b0, b1 - b0);
        b += b1 - b0;
      }
      break;
    }
    case VK_LIST: {
      dst.h_off.alloc((n + 1) * 8 + 16);
      uint64_t* d = (uint64_t*)dst.h_off.p;
      uint64_t pos = 0, b = 0;
      d[0] = 0;
      std::vector<NodePiece> kid;
      for (auto& q : ps) {
        if (q.hi <= q.lo) continue;
        const uint64_t b0 = node_off(*q.nd, q.lo), b1 = node_off(*q.nd, q.hi);
        for (uint64_t k = q.lo; k < q.hi; k++) d[++pos] = b + (node_off(*q.nd, k + 1) - b0);
        b += b1 - b0;
        kid.push_back(NodePiece{&q.nd->kids.at(0), b0, b1});
      }
      dst.total = b;
      dst.kids.resize(1);
      if (kid.empty()) kid.push_back(NodePiece{&ps[0].nd->kids.at(0), 0, 0});
      concat_nodes(dst.kids[0], kid);
      break;
    }
    case VK_STRUCT: {
      const size_t nk = ps[0].nd->kids.size();
      dst.kids.resize(nk);
      for (size_t c = 0; c < nk; c++) {
        std::vector<NodePiece> kid;
        for (auto& q : ps) kid.push_back(NodePiece{&q.nd->kids.at(c), q.lo, q.hi});
        concat_nodes(dst.kids[c], kid);
      }
      break;
    }
  }
}

static void export_rows(const std::shared_ptr<VResult>& res, uint64_t lo, uint64_t hi, ArrowArray* out) {
  ArrayPriv* top = init_array(out, res, (int64_t)(hi - lo));
  top->buffers.push_back(nullptr);
  for (auto& col : res->cols) {
    std::unique_ptr<ArrowArray> ca(new ArrowArray);
    export_node(res, col, lo, hi, ca.get());
    top->children.push_back(ca.get());
    top->owned.push_back(std::move(ca));
  }
  finish_array(out);
}

bool VcfChunkStream::next(ArrowArray* out) {
  if (dead) throw Error("the stream ended with an error");
  struct Guard { bool* d; bool ok = false; ~Guard() { if (!ok) *d = true; } } guard{&dead};
  HIP_CHECK(hipSetDevice(img->device));
  const uint64_t bs = run.batch_size;
  uint64_t have = 0;
  for (auto& r : ready) have += r->n_rows;
  have -= ready_off;
  while (have < bs && !exhausted) {
    auto r = next_chunk();
    if (!r) { exhausted = true; break; }
    have += r->n_rows;
    ready.push_back(std::move(r));
  }
  guard.ok = true;
  if (!have) return false;
  const uint64_t take = std::min(bs, have);
  if (ready.front()->n_rows - ready_off >= take) {
    export_rows(ready.front(), ready_off, ready_off + take, out);
    ready_off += take;
  } else {
    // the batch straddles chunks: its rows are copied into a result of their own
    auto merged = std::make_shared<VResult>();
    merged->batch_size = bs;
    merged->on_host = true;
    merged->n_rows = take;
    const size_t nc = ready.front()->cols.size();
    merged->cols.resize(nc);
    std::vector<std::vector<NodePiece>> pieces(nc);
    uint64_t left = take, off = ready_off;
    for (auto& r : ready) {
      if (!left) break;
      const uint64_t len = std::min(left, r->n_rows - off);
      for (size_t c = 0; c < nc; c++) pieces[c].push_back(NodePiece{&r->cols[c], off, off + len});
      left -= len;
      off = 0;
    }
    for (size_t c = 0; c < nc; c++) concat_nodes(merged->cols[c], pieces[c]);
    export_rows(merged, 0, take, out);
    ready_off += take;
  }
  while (!ready.empty() && ready_off >= ready.front()->n_rows) { ready_off -= ready.front()->n_rows; ready.pop_front(); }
  return true;
}

VcfStreamI* VcfPlan::execute(int32_t partition, int32_t batch_size_in, bool device_only, bioscan_scan_stats* stats_out) const {
  if (partition < 0 || partition >= n_partitions()) throw Error("partition index out of range");
  if (batch_size_in <= 0) throw Error("batch_size must be positive");
  if (!device_only && prov->bgzf) {
    if (stats_out) *stats_out = bioscan_scan_stats{};
    return new VcfChunkStream(*this, partition, batch_size_in);
  }
  auto res = execute_oneshot(*this, partition, batch_size_in, device_only);
  if (stats_out) *stats_out = res->stats;
  auto* s = new VcfStream();
  s->res = res;
  return s;
}

// ---- list UDFs --------------------------------------------------------------------------------------------------
static const VNode* find_list_node(const VResult& r, const char* field) {
  for (auto& col : r.cols) {
    if (col.fd.kind == VK_STRUCT)
      for (auto& k : col.kids) if (k.fd.name == field) return &k;
    if (col.fd.kind == VK_LIST && col.fd.name == field) return &col;
  }
  return nullptr;
}

void VcfStream::list_udf(const char* field, int32_t udf, double threshold, bioscan_udf_stats* out) {
  VResult& r = *res;
  if (r.on_host) throw Error("bioscan_stream_list_udf needs a device-resident stream (bioscan_execute_device)");
  const VNode* lst = find_list_node(r, field);
  if (!lst) throw Error(std::string("no List column named '") + field + "' in this stream");
  if (lst->kids.empty()) throw Error("list column has no values");
  const VNode& ch = lst->kids[0];
  if (ch.fd.kind != VK_INT32 && ch.fd.kind != VK_FLOAT32) throw Error("list UDFs take List<Int32> or List<Float32>");
  HIP_CHECK(hipSetDevice(r.device));
  hipStream_t st = r.stream;
  Timer t(st);
  const uint64_t n = lst->n, E = ch.n;
  memset(out, 0, sizeof(*out));
  out->n_rows = n;
  out->n_elements = E;
  const int is_float = ch.fd.kind == VK_FLOAT32;
  if (udf == 0) {
    DevBuf<double> o(std::max<uint64_t>(n, 1));
    DevBuf<uint8_t> ov(std::max<uint64_t>(n, 1));
    t.start();
    launch_list_avg(lst->d_off.p, (const uint32_t*)ch.d_values.p, ch.all_valid ? nullptr : ch.d_valid.p,
                    lst->all_valid ? nullptr : lst->d_valid.p, n, is_float, o.p, ov.p, st);
    out->ms_kernel = t.stop();
    std::vector<double> h(n);
    std::vector<uint8_t> hv(n);
    if (n) {
      HIP_CHECK(hipMemcpy(h.data(), o.p, n * 8, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(hv.data(), ov.p, n, hipMemcpyDeviceToHost));
    }
    for (uint64_t i = 0; i < n; i++) {
      if (hv[i]) { out->count_a++; out->sum += h[i]; } else out->count_b++;
    }
  } else {
    const uint64_t ew = (E + 63) / 64;
    DevBuf<uint64_t> bits(std::max<uint64_t>(ew, 1));
    uint32_t thr_bits;
    if (is_float) { float f = (float)threshold; memcpy(&thr_bits, &f, 4); }
    else { int32_t v = (int32_t)threshold; memcpy(&thr_bits, &v, 4); }
    t.start();
    launch_list_cmp((const uint32_t*)ch.d_values.p, E, is_float, udf == 1 ? 0 : 1, thr_bits, bits.p, st);
    out->ms_kernel = t.stop();
    DevBuf<unsigned long long> cnt(2);
    HIP_CHECK(hipMemsetAsync(cnt.p, 0, 16, st));
    launch_count_bits(bits.p, ch.all_valid ? nullptr : ch.d_valid.p, E, cnt.p, st);
    unsigned long long h2[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(h2, cnt.p, 16, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    out->count_a = h2[0];
    out->count_b = h2[1];
  }
}

// Host Arrow in/out forms.
struct UdfKeep {
  std::vector<double> f64;
  std::vector<uint8_t> bits, bits2, lbits, data;
  std::vector<int32_t> off, off2;
};
struct UdfPriv {
  std::shared_ptr<UdfKeep> keep;
  std::vector<const void*> buffers;
  std::vector<ArrowArray*> children;
  std::vector<std::unique_ptr<ArrowArray>> owned;
};
static void release_udf_array(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* pr = (UdfPriv*)a->private_data;
  for (auto& c : pr->owned) if (c->release) c->release(c.get());
  delete pr;
  a->release = nullptr;
}
static UdfPriv* init_udf_array(ArrowArray* a, std::shared_ptr<UdfKeep> keep, int64_t len) {
  auto* pr = new UdfPriv();
  pr->keep = std::move(keep);
  memset(a, 0, sizeof(*a));
  a->length = len;
  a->release = release_udf_array;
  a->private_data = pr;
  return pr;
}
static void finish_udf_array(ArrowArray* a) {
  auto* pr = (UdfPriv*)a->private_data;
  a->n_buffers = (int64_t)pr->buffers.size();
  a->buffers = pr->buffers.data();
  a->n_children = (int64_t)pr->children.size();
  a->children = pr->children.empty() ? nullptr : pr->children.data();
}

struct ListIn {
  uint64_t n = 0, E = 0;
  int is_float = 0;
  std::vector<uint64_t> off;       // n + 1, rebased to 0
  std::vector<uint64_t> lvalid;    // words, or empty
  std::vector<uint64_t> evalid;    // words, or empty
  const uint8_t* values = nullptr;  // first element of the window
};
static void bits_to_words(const uint8_t* bits, uint64_t bit0, uint64_t n, std::vector<uint64_t>& w) {
  std::vector<uint8_t> tmp;
  copy_bits(bits, bit0, n, tmp);
  w.assign((n + 63) / 64 + 1, 0);
  memcpy(w.data(), tmp.data(), (n + 7) / 8);
}
static ListIn read_list_input(const ArrowArray* in, const ArrowSchema* sc) {
  if (!in || !sc || !sc->format || strcmp(sc->format, "+l") != 0 || sc->n_children != 1 || in->n_children != 1)
    throw Error("list UDF expects a List array");
  const char* cf = sc->children[0]->format;
  ListIn L;
  if (strcmp(cf, "i") == 0) L.is_float = 0;
  else if (strcmp(cf, "f") == 0) L.is_float = 1;
  else throw Error("list UDFs take List<Int32> or List<Float32>");
  L.n = (uint64_t)in->length;
  const int32_t* off = (const int32_t*)in->buffers[1] + in->offset;
  const ArrowArray* ch = in->children[0];
  const uint64_t e0 = L.n ? (uint64_t)off[0] : 0, e1 = L.n ? (uint64_t)off[L.n] : 0;
  L.E = e1 - e0;
  L.off.resize(L.n + 1);
  for (uint64_t i = 0; i <= L.n; i++) L.off[i] = L.n ? (uint64_t)off[i] - e0 : 0;
  if (in->buffers[0] && in->null_count != 0) bits_to_words((const uint8_t*)in->buffers[0], (uint64_t)in->offset, L.n, L.lvalid);
  if (ch->buffers[0] && ch->null_count != 0) bits_to_words((const uint8_t*)ch->buffers[0], (uint64_t)ch->offset + e0, L.E, L.evalid);
  L.values = (const uint8_t*)ch->buffers[1] + ((uint64_t)ch->offset + e0) * 4;
  return L;
}
struct ListDev {
  DevBuf<uint64_t> off, lvalid, evalid;
  DevBuf<uint32_t> values;
};
static void upload_list(const ListIn& L, ListDev& D) {
  D.off.alloc(L.n + 1);
  HIP_CHECK(hipMemcpy(D.off.p, L.off.data(), (L.n + 1) * 8, hipMemcpyHostToDevice));
  D.values.alloc(std::max<uint64_t>(L.E, 1));
  if (L.E) HIP_CHECK(hipMemcpy(D.values.p, L.values, L.E * 4, hipMemcpyHostToDevice));
  if (!L.lvalid.empty()) { D.lvalid.alloc(L.lvalid.size()); HIP_CHECK(hipMemcpy(D.lvalid.p, L.lvalid.data(), L.lvalid.size() * 8, hipMemcpyHostToDevice)); }
  if (!L.evalid.empty()) { D.evalid.alloc(L.evalid.size()); HIP_CHECK(hipMemcpy(D.evalid.p, L.evalid.data(), L.evalid.size() * 8, hipMemcpyHostToDevice)); }
}

// List<Boolean> input window: bit-packed values / validity rebased to element 0
struct BoolListIn {
  uint64_t n = 0, E = 0;
  std::vector<uint64_t> off, lvalid, val, evalid;
};
static BoolListIn read_bool_list(const ArrowArray* in, const ArrowSchema* sc) {
  if (!in || !sc || !sc->format || strcmp(sc->format, "+l") != 0 || sc->n_children != 1 || in->n_children != 1 ||
      strcmp(sc->children[0]->format, "b") != 0)
    throw Error("expected a List<Boolean> array");
  BoolListIn L;
  L.n = (uint64_t)in->length;
  const int32_t* off = (const int32_t*)in->buffers[1] + in->offset;
  const ArrowArray* ch = in->children[0];
  const uint64_t e0 = L.n ? (uint64_t)off[0] : 0, e1 = L.n ? (uint64_t)off[L.n] : 0;
  L.E = e1 - e0;
  L.off.resize(L.n + 1);
  for (uint64_t i = 0; i <= L.n; i++) L.off[i] = L.n ? (uint64_t)off[i] - e0 : 0;
  if (in->buffers[0] && in->null_count != 0) bits_to_words((const uint8_t*)in->buffers[0], (uint64_t)in->offset, L.n, L.lvalid);
  if (ch->buffers[0] && ch->null_count != 0) bits_to_words((const uint8_t*)ch->buffers[0], (uint64_t)ch->offset + e0, L.E, L.evalid);
  if (L.E) bits_to_words((const uint8_t*)ch->buffers[1], (uint64_t)ch->offset + e0, L.E, L.val);
  else L.val.assign(1, 0);
  return L;
}
template <typename T>
static void up(DevBuf<T>& d, const std::vector<T>& h) {
  d.alloc(std::max<size_t>(h.size(), 1));
  if (!h.empty()) HIP_CHECK(hipMemcpy(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
}
static bool word_bit(const std::vector<uint64_t>& w, uint64_t i) { return w.empty() || ((w[i >> 6] >> (i & 63)) & 1ull); }

}  // namespace

void udf_list_avg_host(const ArrowArray* in, const ArrowSchema* in_schema, int32_t device_id, ArrowArray* out, ArrowSchema* out_schema) {
  ListIn L = read_list_input(in, in_schema);
  HIP_CHECK(hipSetDevice(device_id));
  ListDev D;
  upload_list(L, D);
  DevBuf<double> o(std::max<uint64_t>(L.n, 1));
  DevBuf<uint8_t> ov(std::max<uint64_t>(L.n, 1));
  launch_list_avg(D.off.p, D.values.p, D.evalid.p, D.lvalid.p, L.n, L.is_float, o.p, ov.p, nullptr);
  HIP_CHECK(hipDeviceSynchronize());
  auto keep = std::make_shared<UdfKeep>();
  keep->f64.resize(L.n + 1);
  std::vector<uint8_t> hv(L.n + 1);
  if (L.n) {
    HIP_CHECK(hipMemcpy(keep->f64.data(), o.p, L.n * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hv.data(), ov.p, L.n, hipMemcpyDeviceToHost));
  }
  keep->bits.assign((L.n + 7) / 8 + 1, 0);
  int64_t nulls = 0;
  for (uint64_t i = 0; i < L.n; i++) {
    if (hv[i]) keep->bits[i >> 3] |= (uint8_t)(1u << (i & 7)); else nulls++;
  }
  UdfPriv* pr = init_udf_array(out, keep, (int64_t)L.n);
  out->null_count = nulls;
  pr->buffers.push_back(nulls ? keep->bits.data() : nullptr);
  pr->buffers.push_back(keep->f64.data());
  finish_udf_array(out);
  fill_schema(out_schema, "list_avg", "g", true, {});
}

void udf_list_cmp_host(const ArrowArray* in, const ArrowSchema* in_schema, int32_t op, double threshold, int32_t device_id,
                       ArrowArray* out, ArrowSchema* out_schema) {
  ListIn L = read_list_input(in, in_schema);
  HIP_CHECK(hipSetDevice(device_id));
  ListDev D;
  upload_list(L, D);
  const uint64_t ew = (L.E + 63) / 64;
  DevBuf<uint64_t> bits(std::max<uint64_t>(ew, 1));
  uint32_t thr_bits;
  if (L.is_float) { float f = (float)threshold; memcpy(&thr_bits, &f, 4); }
  else { int32_t v = (int32_t)threshold; memcpy(&thr_bits, &v, 4); }
  launch_list_cmp(D.values.p, L.E, L.is_float, op, thr_bits, bits.p, nullptr);
  HIP_CHECK(hipDeviceSynchronize());
  auto keep = std::make_shared<UdfKeep>();
  keep->bits.assign(ew * 8 + 8, 0);   // element values
  if (ew) HIP_CHECK(hipMemcpy(keep->bits.data(), bits.p, ew * 8, hipMemcpyDeviceToHost));
  keep->off.resize(L.n + 1);
  for (uint64_t i = 0; i <= L.n; i++) keep->off[i] = (int32_t)L.off[i];
  // list validity / element validity are the input's
  int64_t lnulls = 0, enulls = 0;
  if (!L.lvalid.empty()) {
    keep->lbits.assign((L.n + 7) / 8 + 8, 0);
    memcpy(keep->lbits.data(), L.lvalid.data(), (L.n + 7) / 8);
    for (uint64_t i = 0; i < L.n; i++) lnulls += !((L.lvalid[i >> 6] >> (i & 63)) & 1);
  }
  if (!L.evalid.empty()) {
    keep->bits2.assign((L.E + 7) / 8 + 8, 0);
    memcpy(keep->bits2.data(), L.evalid.data(), (L.E + 7) / 8);
    for (uint64_t i = 0; i < L.E; i++) enulls += !((L.evalid[i >> 6] >> (i & 63)) & 1);
  }
  UdfPriv* pr = init_udf_array(out, keep, (int64_t)L.n);
  out->null_count = lnulls;
  pr->buffers.push_back(lnulls ? keep->lbits.data() : nullptr);
  pr->buffers.push_back(keep->off.data());
  std::unique_ptr<ArrowArray> item(new ArrowArray);
  UdfPriv* ip = init_udf_array(item.get(), keep, (int64_t)L.E);
  item->null_count = enulls;
  ip->buffers.push_back(enulls ? keep->bits2.data() : nullptr);
  ip->buffers.push_back(keep->bits.data());
  finish_udf_array(item.get());
  pr->children.push_back(item.get());
  pr->owned.push_back(std::move(item));
  finish_udf_array(out);
  fill_schema(out_schema, op == 0 ? "list_gte" : "list_lte", "+l", true, {});
  std::unique_ptr<ArrowSchema> cs(new ArrowSchema);
  fill_schema(cs.get(), "item", "b", true, {});
  add_child(out_schema, std::move(cs));
}

void udf_list_and_host(const ArrowArray* a, const ArrowSchema* as, const ArrowArray* b, const ArrowSchema* bs, int32_t device_id,
                       ArrowArray* out, ArrowSchema* out_schema) {
  BoolListIn L = read_bool_list(a, as), R = read_bool_list(b, bs);
  if (L.n != R.n) throw Error("list_and: the two arguments have different lengths");
  const uint64_t n = L.n;
  std::vector<uint64_t> off_o(n + 1, 0);
  auto keep = std::make_shared<UdfKeep>();
  keep->lbits.assign((n + 7) / 8 + 8, 0);
  int64_t lnulls = 0;
  for (uint64_t i = 0; i < n; i++) {
    const bool ok = word_bit(L.lvalid, i) && word_bit(R.lvalid, i);   // NULL list on either side -> NULL list (udfs.rs:807-810)
    const uint64_t ll = L.off[i + 1] - L.off[i], rl = R.off[i + 1] - R.off[i];
    off_o[i + 1] = off_o[i] + (ok ? std::min(ll, rl) : 0);
    if (ok) keep->lbits[i >> 3] |= (uint8_t)(1u << (i & 7)); else lnulls++;
  }
  const uint64_t E = off_o[n], ew = (E + 63) / 64 + 1;
  HIP_CHECK(hipSetDevice(device_id));
  DevBuf<uint64_t> d_ol, d_or, d_oo, d_lv, d_lva, d_rv, d_rva, d_val(ew), d_valid(ew);
  up(d_ol, L.off); up(d_or, R.off); up(d_oo, off_o); up(d_lv, L.val); up(d_rv, R.val);
  if (!L.evalid.empty()) up(d_lva, L.evalid);
  if (!R.evalid.empty()) up(d_rva, R.evalid);
  HIP_CHECK(hipMemset(d_val.p, 0, ew * 8));
  HIP_CHECK(hipMemset(d_valid.p, 0, ew * 8));
  launch_list_and(d_ol.p, d_or.p, d_oo.p, d_lv.p, L.evalid.empty() ? nullptr : d_lva.p, d_rv.p, R.evalid.empty() ? nullptr : d_rva.p, n,
                  d_val.p, d_valid.p, nullptr);
  HIP_CHECK(hipDeviceSynchronize());
  keep->bits.assign(ew * 8 + 8, 0);
  keep->bits2.assign(ew * 8 + 8, 0);
  HIP_CHECK(hipMemcpy(keep->bits.data(), d_val.p, ew * 8, hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(keep->bits2.data(), d_valid.p, ew * 8, hipMemcpyDeviceToHost));
  int64_t enulls = 0;
  for (uint64_t i = 0; i < E; i++) enulls += !((keep->bits2[i >> 3] >> (i & 7)) & 1);
  keep->off.resize(n + 1);
  for (uint64_t i = 0; i <= n; i++) keep->off[i] = (int32_t)off_o[i];
  UdfPriv* pr = init_udf_array(out, keep, (int64_t)n);
  out->null_count = lnulls;
  pr->buffers.push_back(lnulls ? keep->lbits.data() : nullptr);
  pr->buffers.push_back(keep->off.data());
  std::unique_ptr<ArrowArray> item(new ArrowArray);
  UdfPriv* ip = init_udf_array(item.get(), keep, (int64_t)E);
  item->null_count = enulls;
  ip->buffers.push_back(enulls ? keep->bits2.data() : nullptr);
  ip->buffers.push_back(keep->bits.data());
  finish_udf_array(item.get());
  pr->children.push_back(item.get());
  pr->owned.push_back(std::move(item));
  finish_udf_array(out);
  fill_schema(out_schema, "list_and", "+l", true, {});
  std::unique_ptr<ArrowSchema> cs(new ArrowSchema);
  fill_schema(cs.get(), "item", "b", true, {});
  add_child(out_schema, std::move(cs));
}

void udf_set_gts_host(const ArrowArray* gt, const ArrowSchema* gs, const ArrowArray* mask, const ArrowSchema* ms, const char* replacement,
                      int32_t device_id, ArrowArray* out, ArrowSchema* out_schema) {
  if (!gt || !gs || strcmp(gs->format, "+l") != 0 || gs->n_children != 1 || strcmp(gs->children[0]->format, "u") != 0)
    throw Error("vcf_set_gts expects List<Utf8> genotypes");
  BoolListIn M = read_bool_list(mask, ms);
  const uint64_t n = (uint64_t)gt->length;
  if (M.n != n) throw Error("vcf_set_gts: the two list arguments have different lengths");
  const std::string rep = replacement ? replacement : "./.";
  const int32_t* off = (const int32_t*)gt->buffers[1] + gt->offset;
  const ArrowArray* ch = gt->children[0];
  const uint64_t e0 = n ? (uint64_t)off[0] : 0, e1 = n ? (uint64_t)off[n] : 0, E = e1 - e0;
  std::vector<uint64_t> off_g(n + 1), goff(E + 1), gvalid, glvalid;
  for (uint64_t i = 0; i <= n; i++) off_g[i] = n ? (uint64_t)off[i] - e0 : 0;
  const int32_t* so = (const int32_t*)ch->buffers[1] + ch->offset + e0;
  const uint64_t b0 = E ? (uint64_t)so[0] : 0, b1 = E ? (uint64_t)so[E] : 0;
  for (uint64_t i = 0; i <= E; i++) goff[i] = E ? (uint64_t)so[i] - b0 : 0;
  if (ch->buffers[0] && ch->null_count != 0) bits_to_words((const uint8_t*)ch->buffers[0], (uint64_t)ch->offset + e0, E, gvalid);
  if (gt->buffers[0] && gt->null_count != 0) bits_to_words((const uint8_t*)gt->buffers[0], (uint64_t)gt->offset, n, glvalid);
  const uint64_t data_len = b1 - b0;
  HIP_CHECK(hipSetDevice(device_id));
  DevBuf<uint8_t> d_u(data_len + rep.size() + 64);
  if (data_len) HIP_CHECK(hipMemcpy(d_u.p, (const uint8_t*)ch->buffers[2] + b0, data_len, hipMemcpyHostToDevice));
  if (!rep.empty()) HIP_CHECK(hipMemcpy(d_u.p + data_len, rep.data(), rep.size(), hipMemcpyHostToDevice));
  DevBuf<uint64_t> d_og, d_goff, d_gv, d_om, d_mlv, d_mv, d_mva, d_src(std::max<uint64_t>(E, 1)), d_ooff(E + 2), d_tmp(scan_tmp_elems(E));
  DevBuf<uint32_t> d_len(std::max<uint64_t>(E, 1));
  DevBuf<uint8_t> d_ov(std::max<uint64_t>(E, 1));
  up(d_og, off_g); up(d_goff, goff); up(d_om, M.off); up(d_mv, M.val);
  if (!gvalid.empty()) up(d_gv, gvalid);
  if (!M.lvalid.empty()) up(d_mlv, M.lvalid);
  if (!M.evalid.empty()) up(d_mva, M.evalid);
  launch_set_gts_plan(d_og.p, d_goff.p, gvalid.empty() ? nullptr : d_gv.p, d_om.p, M.lvalid.empty() ? nullptr : d_mlv.p, d_mv.p,
                      M.evalid.empty() ? nullptr : d_mva.p, n, data_len, (uint32_t)rep.size(), d_src.p, d_len.p, d_ov.p, nullptr);
  launch_exclusive_scan_u32_to_u64(d_len.p, d_ooff.p, E, d_tmp.p, nullptr);
  std::vector<uint64_t> ooff(E + 1);
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpy(ooff.data(), d_ooff.p, (E + 1) * 8, hipMemcpyDeviceToHost));
  const uint64_t tot = ooff[E];
  if (tot > 0x7FFFFFFFull) throw Error("vcf_set_gts: result exceeds the int32 offsets of one Utf8 array");
  DevBuf<uint8_t> d_out(std::max<uint64_t>(tot, 1));
  launch_scatter_ranges(d_u.p, d_src.p, E, d_ooff.p, d_out.p, tot, nullptr);
  HIP_CHECK(hipDeviceSynchronize());
  auto keep = std::make_shared<UdfKeep>();
  keep->data.resize(tot + 8);
  if (tot) HIP_CHECK(hipMemcpy(keep->data.data(), d_out.p, tot, hipMemcpyDeviceToHost));
  std::vector<uint8_t> ov(E + 1);
  if (E) HIP_CHECK(hipMemcpy(ov.data(), d_ov.p, E, hipMemcpyDeviceToHost));
  keep->bits2.assign((E + 7) / 8 + 8, 0);
  int64_t enulls = 0;
  for (uint64_t i = 0; i < E; i++) { if (ov[i]) keep->bits2[i >> 3] |= (uint8_t)(1u << (i & 7)); else enulls++; }
  keep->off2.resize(E + 1);
  for (uint64_t i = 0; i <= E; i++) keep->off2[i] = (int32_t)ooff[i];
  keep->off.resize(n + 1);
  for (uint64_t i = 0; i <= n; i++) keep->off[i] = (int32_t)off_g[i];
  keep->lbits.assign((n + 7) / 8 + 8, 0xFF);
  int64_t lnulls = 0;
  if (!glvalid.empty()) {
    memcpy(keep->lbits.data(), glvalid.data(), (n + 7) / 8);
    for (uint64_t i = 0; i < n; i++) lnulls += !word_bit(glvalid, i);
  }
  UdfPriv* pr = init_udf_array(out, keep, (int64_t)n);
  out->null_count = lnulls;
  pr->buffers.push_back(lnulls ? keep->lbits.data() : nullptr);
  pr->buffers.push_back(keep->off.data());
  std::unique_ptr<ArrowArray> item(new ArrowArray);
  UdfPriv* ip = init_udf_array(item.get(), keep, (int64_t)E);
  item->null_count = enulls;
  ip->buffers.push_back(enulls ? keep->bits2.data() : nullptr);
  ip->buffers.push_back(keep->off2.data());
  ip->buffers.push_back(keep->data.data());
  finish_udf_array(item.get());
  pr->children.push_back(item.get());
  pr->owned.push_back(std::move(item));
  finish_udf_array(out);
  fill_schema(out_schema, "vcf_set_gts", "+l", true, {});
  std::unique_ptr<ArrowSchema> cs(new ArrowSchema);
  fill_schema(cs.get(), "item", "u", true, {});
  add_child(out_schema, std::move(cs));
}

// vcf_an / vcf_ac / vcf_af (bio-format-vcf/src/udfs.rs:161-552).  which: 0 = AN (Int32), 1 = AC (List<Int32>), 2 = AF (List<Float64>).
// `alt` (Utf8, the provider's pipe-separated ALT column) is the optional second argument of AC / AF: the list of a row is as long
// as the larger of its ALT count and the largest allele index its genotypes call.  The GT strings are parsed on the device
// (two passes: called alleles + largest index per row, then the per-allele counts); the per-row list lengths, their prefix
// sum and AF's division run on the host over n rows.
void udf_allele_stats_host(const ArrowArray* gt, const ArrowSchema* gs, const ArrowArray* alt, const ArrowSchema* as, int32_t which,
                           int32_t device_id, ArrowArray* out, ArrowSchema* out_schema) {
  static const char* const NAMES[3] = {"vcf_an", "vcf_ac", "vcf_af"};
  if (which < 0 || which > 2) throw Error("unknown allele statistic");
  const std::string fn = NAMES[which];
  if (!gt || !gs || strcmp(gs->format, "+l") != 0 || gs->n_children != 1 || strcmp(gs->children[0]->format, "u") != 0)
    throw Error(fn + " expects List<Utf8> input");
  if (alt && (!as || strcmp(as->format, "u") != 0)) throw Error(fn + " 2nd argument must be Utf8 (alt column)");
  const uint64_t n = (uint64_t)gt->length;
  if (alt && (uint64_t)alt->length != n) throw Error(fn + ": the two arguments have different lengths");
  const int32_t* off = (const int32_t*)gt->buffers[1] + gt->offset;
  const ArrowArray* ch = gt->children[0];
  const uint64_t e0 = n ? (uint64_t)off[0] : 0, e1 = n ? (uint64_t)off[n] : 0, E = e1 - e0;
  std::vector<uint64_t> off_g(n + 1), goff(E + 1), gvalid, glvalid;
  for (uint64_t i = 0; i <= n; i++) off_g[i] = n ? (uint64_t)off[i] - e0 : 0;
  const int32_t* so = (const int32_t*)ch->buffers[1] + ch->offset + e0;
  const uint64_t b0 = E ? (uint64_t)so[0] : 0, b1 = E ? (uint64_t)so[E] : 0;
  for (uint64_t i = 0; i <= E; i++) goff[i] = E ? (uint64_t)so[i] - b0 : 0;
  if (ch->buffers[0] && ch->null_count != 0) bits_to_words((const uint8_t*)ch->buffers[0], (uint64_t)ch->offset + e0, E, gvalid);
  if (gt->buffers[0] && gt->null_count != 0) bits_to_words((const uint8_t*)gt->buffers[0], (uint64_t)gt->offset, n, glvalid);
  const uint64_t data_len = b1 - b0;
  HIP_CHECK(hipSetDevice(device_id));
  DevBuf<uint8_t> d_u(data_len + 64);
  if (data_len) HIP_CHECK(hipMemcpy(d_u.p, (const uint8_t*)ch->buffers[2] + b0, data_len, hipMemcpyHostToDevice));
  DevBuf<uint64_t> d_og, d_goff, d_gv;
  up(d_og, off_g); up(d_goff, goff);
  if (!gvalid.empty()) up(d_gv, gvalid);
  DevBuf<int32_t> d_an(std::max<uint64_t>(n, 1));
  DevBuf<unsigned long long> d_mx(std::max<uint64_t>(n, 1));
  launch_gt_stats(d_u.p, d_goff.p, gvalid.empty() ? nullptr : d_gv.p, d_og.p, n, d_an.p, d_mx.p, nullptr);
  std::vector<int32_t> an(n);
  std::vector<unsigned long long> mx(n);
  HIP_CHECK(hipDeviceSynchronize());
  if (n) {
    HIP_CHECK(hipMemcpy(an.data(), d_an.p, n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(mx.data(), d_mx.p, n * 8, hipMemcpyDeviceToHost));
  }
  auto keep = std::make_shared<UdfKeep>();
  keep->lbits.assign((n + 7) / 8 + 8, 0xFF);
  int64_t lnulls = 0;
  if (!glvalid.empty()) {
    memcpy(keep->lbits.data(), glvalid.data(), (n + 7) / 8);
    for (uint64_t i = 0; i < n; i++) lnulls += !word_bit(glvalid, i);
  }
  if (which == 0) {
    keep->off.resize(n + 1);   // (Int32 values live in `off`)
    for (uint64_t i = 0; i < n; i++) keep->off[i] = word_bit(glvalid, i) ? an[i] : 0;
    UdfPriv* pr = init_udf_array(out, keep, (int64_t)n);
    out->null_count = lnulls;
    pr->buffers.push_back(lnulls ? keep->lbits.data() : nullptr);
    pr->buffers.push_back(keep->off.data());
    finish_udf_array(out);
    fill_schema(out_schema, "vcf_an", "i", true, {});
    return;
  }
  // list lengths: max(ALT count of the row, largest called allele index); a NULL list is a NULL (empty) list
  auto alt_valid = [&](uint64_t i) {
    if (!alt) return false;
    const uint8_t* v = (const uint8_t*)alt->buffers[0];
    const uint64_t j = i + (uint64_t)alt->offset;
    return alt->null_count == 0 || !v || ((v[j >> 3] >> (j & 7)) & 1);
  };
  auto is_space = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); };
  std::vector<uint64_t> out_off(n + 1, 0);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t len = 0;
    if (word_bit(glvalid, i)) {
      len = mx[i];
      if (alt_valid(i)) {
        // count_alt_alleles (udfs.rs:146-154): trimmed; "" and "." have none; else pieces between '|'
        const int32_t* ao = (const int32_t*)alt->buffers[1] + alt->offset;
        const uint8_t* ad = (const uint8_t*)alt->buffers[2];
        uint64_t a = (uint64_t)ao[i], b = (uint64_t)ao[i + 1];
        while (a < b && is_space(ad[a])) a++;
        while (b > a && is_space(ad[b - 1])) b--;
        uint64_t cnt = 0;
        if (b > a && !(b - a == 1 && ad[a] == '.')) { cnt = 1; for (uint64_t k = a; k < b; k++) cnt += ad[k] == '|'; }
        len = std::max<uint64_t>(len, cnt);
      }
      if (len > 0x7FFFFFFFull) throw Error(fn + ": allele index " + std::to_string(len) + " is out of range");
    }
    out_off[i + 1] = out_off[i] + len;
  }
  const uint64_t T = out_off[n];
  if (T > 0x7FFFFFFFull) throw Error(fn + ": result exceeds the int32 offsets of one list array");
  DevBuf<uint64_t> d_oo;
  up(d_oo, out_off);
  DevBuf<int32_t> d_cnt(std::max<uint64_t>(T, 1));
  HIP_CHECK(hipMemset(d_cnt.p, 0, std::max<uint64_t>(T, 1) * 4));
  launch_gt_ac(d_u.p, d_goff.p, gvalid.empty() ? nullptr : d_gv.p, d_og.p, n, d_oo.p, d_cnt.p, nullptr);
  HIP_CHECK(hipDeviceSynchronize());
  std::vector<int32_t> counts(T);
  if (T) HIP_CHECK(hipMemcpy(counts.data(), d_cnt.p, T * 4, hipMemcpyDeviceToHost));
  keep->off.resize(n + 1);
  for (uint64_t i = 0; i <= n; i++) keep->off[i] = (int32_t)out_off[i];
  int64_t enulls = 0;
  if (which == 1) {
    keep->off2.assign(counts.begin(), counts.end());   // Int32 element values
    keep->off2.push_back(0);
  } else {
    keep->f64.assign(T + 1, 0.0);
    keep->bits2.assign((T + 7) / 8 + 8, 0);
    for (uint64_t i = 0; i < n; i++)
      for (uint64_t k = out_off[i]; k < out_off[i + 1]; k++) {
        if (an[i] == 0) { enulls++; continue; }            // all missing: NULLs of the list's length (udfs.rs:505-511)
        keep->f64[k] = (double)counts[k] / (double)an[i];
        keep->bits2[k >> 3] |= (uint8_t)(1u << (k & 7));
      }
  }
  UdfPriv* pr = init_udf_array(out, keep, (int64_t)n);
  out->null_count = lnulls;
  pr->buffers.push_back(lnulls ? keep->lbits.data() : nullptr);
  pr->buffers.push_back(keep->off.data());
  std::unique_ptr<ArrowArray> item(new ArrowArray);
  UdfPriv* ip = init_udf_array(item.get(), keep, (int64_t)T);
  item->null_count = enulls;
  ip->buffers.push_back(enulls ? keep->bits2.data() : nullptr);
  if (which == 1) ip->buffers.push_back(keep->off2.data()); else ip->buffers.push_back(keep->f64.data());
  finish_udf_array(item.get());
  pr->children.push_back(item.get());
  pr->owned.push_back(std::move(item));
  finish_udf_array(out);
  fill_schema(out_schema, which == 1 ? "vcf_ac" : "vcf_af", "+l", true, {});
  std::unique_ptr<ArrowSchema> cs(new ArrowSchema);
  fill_schema(cs.get(), "item", which == 1 ? "i" : "g", true, {});
  add_child(out_schema, std::move(cs));
}

// ---- open -----------------------------------------------------------------------------------------------------
VcfProviderI* vcf_open(const char* path, const bioscan_vcf_options* o) {
  std::unique_ptr<VcfProvider> pp(new VcfProvider);
  VcfProvider& p = *pp;
  p.what = "VCF";
  p.path = path;
  p.device = o->device_id;
  p.zero_based = o->coordinate_system_zero_based != 0;
  p.set_device();
  p.load_file();
  const uint8_t* d = p.file.p;
  std::vector<uint8_t> head;
  if (p.file_len >= 18 && d[0] == 0x1f && d[1] == 0x8b && d[2] == 8 && (d[3] & 4) && d[12] == 0x42 && d[13] == 0x43) {
    p.bgzf = true;
    p.frame();
    uint32_t nb = 1;
    std::string herr;
    for (;;) {
      head = p.inflate_prefix_to_host(nb);
      const bool all = nb >= p.n_blocks();
      if (parse_vcf_header(head.data(), head.size(), all, &p.hdr, &herr)) break;
      if (!herr.empty()) throw Error("Failed to open VCF: " + herr);
      if (all) throw Error("Failed to open VCF: truncated header");
      nb = std::min<uint32_t>(nb * 4, p.n_blocks());
    }
  } else if (p.file_len >= 2 && d[0] == 0x1f && d[1] == 0x8b) {
    throw Error("plain gzip VCF (not BGZF) has no block structure to decode in parallel: not supported by the GPU scan");
  } else {
    p.blk_coff = {0, p.file_len};
    p.blk_uoff = {0, 0};
    std::string herr;
    if (!parse_vcf_header(d, p.file_len, true, &p.hdr, &herr)) throw Error("Failed to open VCF: " + (herr.empty() ? "truncated header" : herr));
  }
  // index discovery: explicit path, else `<path>.tbi` for BGZF input (index_utils.rs:85-94)
  if (o->index_path && o->index_path[0]) { p.index_path = o->index_path; p.has_index = true; }
  else if (!o->index_path && p.bgzf) {
    if (file_exists(p.path + ".tbi")) { p.index_path = p.path + ".tbi"; p.has_index = true; }
    else if (file_exists(p.path + ".csi")) { p.index_path = p.path + ".csi"; p.has_index = true; }
  }
  for (auto& c : p.hdr.contigs) { p.contig_names.push_back(c.first); p.contig_lengths.push_back(c.second > 0 ? (uint64_t)c.second : 0); }
  std::vector<std::string> index_names;
  if (p.has_index) {
    std::string lower = p.index_path;
    for (auto& ch : lower) ch = (char)tolower(ch);
    const bool csi = lower.size() >= 4 && lower.compare(lower.size() - 4, 4, ".csi") == 0;  // table_provider.rs:970-974
    // An index that cannot be read is not an error here: the reference logs it and carries on with no index names
    // (table_provider.rs:1012-1024); the scan then plans with unit estimates (storage.rs:826-840) and every partition
    // fails when it opens the index (IndexedVcfReader::new, storage.rs:766).  A CSI only contributes its names.
    try {
      // the index is BGZF itself: it is inflated by the same GPU kernel
      BgzfSource ix;
      ix.what = csi ? "CSI index" : "tabix index";
      ix.path = p.index_path;
      ix.device = p.device;
      ix.load_file();
      ix.frame();
      std::vector<uint8_t> raw = ix.inflate_prefix_to_host(ix.n_blocks());
      std::string e;
      if (csi) {
        if (parse_csi_names(raw, &index_names, &e)) e = "invalid tabix header (the file is a CSI index; the VCF text reader only accepts tabix)";
        p.index_error = e;
      } else if (parse_tbi(raw, &p.tbi, &e)) {
        p.index_readable = true;
        index_names = p.tbi.names;
      } else p.index_error = e;
    } catch (const Error& ex) {
      p.index_error = ex.what();
    }
    if (!index_names.empty()) {  // table_provider.rs:1037-1075
      std::vector<uint64_t> lens;
      for (auto& n : index_names) {
        uint64_t L = 0;
        for (auto& c : p.hdr.contigs) if (c.first == n && c.second >= 0) L = (uint64_t)c.second;
        lens.push_back(L);
      }
      p.contig_names = index_names;
      p.contig_lengths = lens;
    }
  }
  std::vector<std::string> inf, fmt, smp;
  if (o->has_info_fields) for (int i = 0; i < o->n_info_fields; i++) inf.push_back(o->info_fields[i]);
  if (o->has_format_fields) for (int i = 0; i < o->n_format_fields; i++) fmt.push_back(o->format_fields[i]);
  if (o->has_samples) for (int i = 0; i < o->n_samples; i++) smp.push_back(o->samples[i]);
  std::string e = determine_vcf_schema(p.hdr, o->has_info_fields ? &inf : nullptr, o->has_format_fields ? &fmt : nullptr,
                                       o->has_samples ? &smp : nullptr, p.zero_based, p.has_index ? &index_names : nullptr, &p.sch);
  if (!e.empty()) throw Error("Failed to open VCF: " + e);
  return pp.release();
}

}  // namespace bioscan
