// bam_writer.cpp -- host side of the BAM write path: RecordBatches (Arrow C Data) -> BAM records -> BGZF members -> file.
//
// Mirrors BamLocalWriter (bio-format-bam/src/writer.rs:57-283: write_header, write_records, finish) with the record
// serialisation of bio-format-core/src/sam_record_serializer.rs and the BGZF framing of noodles-bgzf's Writer.  The
// serialisation, CRC32 and DEFLATE run on the GPU (bam_write.hip); the host uploads the batch's column buffers, resolves
// chrom / mate_chrom names against the header (a dictionary lookup) and writes the finished members to the file.
// Tag columns (fields carrying bio.bam.tag.tag metadata) become aux fields in schema order (build_tag_data,
// bio-format-core/src/sam_tag_io.rs:109-147); type errors are the reference's, worded on the host from the batch.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/bioscan.h"
#include "common.h"
#include "kernels.h"

using namespace bioscan;

namespace {

// BGZF end-of-file marker (SAM spec 4.1.2)
const uint8_t BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct Compressor {
  int device = 0;
  hipStream_t st = nullptr;
  double kernel_ms = 0;
  explicit Compressor(int dev) : device(dev) {
    HIP_CHECK(hipSetDevice(dev));
    HIP_CHECK(hipStreamCreate(&st));
  }
  ~Compressor() {
    if (st) { (void)hipSetDevice(device); (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  }
  // compresses d_payload[0 .. len) (device, padded by >= 16 readable bytes) into members of <= BGZF_MAX_PAYLOAD bytes and
  // appends them to `out`; at most MAX_BATCH members per launch (the parse of a member lives in 256 KB of token scratch)
  static constexpr uint32_t MAX_BATCH = 2048;
  void compress(const uint8_t* d_payload, uint64_t len, std::vector<uint8_t>* out, uint64_t* n_members) {
    HIP_CHECK(hipSetDevice(device));
    const uint64_t nm_all = (len + BGZF_MAX_PAYLOAD - 1) / BGZF_MAX_PAYLOAD;
    for (uint64_t m0 = 0; m0 < nm_all; m0 += MAX_BATCH) {
      const uint32_t nm = (uint32_t)std::min<uint64_t>(MAX_BATCH, nm_all - m0);
      compress_batch(d_payload + m0 * BGZF_MAX_PAYLOAD, std::min<uint64_t>(len - m0 * BGZF_MAX_PAYLOAD, (uint64_t)nm * BGZF_MAX_PAYLOAD), nm, out);
    }
    if (n_members) *n_members += nm_all;
  }
  void compress_batch(const uint8_t* d_payload, uint64_t len, uint32_t nm, std::vector<uint8_t>* out) {
    std::vector<uint64_t> off(nm + 1);
    for (uint32_t m = 0; m <= nm; m++) off[m] = std::min<uint64_t>((uint64_t)m * BGZF_MAX_PAYLOAD, len);
    DevBuf<uint64_t> d_off(nm + 1), d_out_off(nm + 1);
    DevBuf<uint32_t> d_crc(nm), d_sizes(nm), tokens((uint64_t)nm * BGZF_TOKENS_PER_MEMBER);
    DevBuf<uint8_t> slots((uint64_t)nm * BGZF_SLOT_BYTES);
    HIP_CHECK(hipMemcpyAsync(d_off.p, off.data(), (nm + 1) * 8, hipMemcpyHostToDevice, st));
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, st));
    launch_crc32_store(d_payload, d_off.p, nm, d_crc.p, st);
    launch_bgzf_deflate(d_payload, d_off.p, nm, d_crc.p, slots.p, BGZF_SLOT_BYTES, d_sizes.p, tokens.p, st);
    HIP_CHECK(hipEventRecord(b, st));
    std::vector<uint32_t> sizes(nm);
    HIP_CHECK(hipMemcpyAsync(sizes.data(), d_sizes.p, nm * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    kernel_ms += ms;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    std::vector<uint64_t> out_off(nm + 1, 0);
    for (uint32_t m = 0; m < nm; m++) {
      if (sizes[m] < 28 || sizes[m] > 65536) throw Error("BGZF write error: member size out of range");
      out_off[m + 1] = out_off[m] + sizes[m];
    }
    DevBuf<uint8_t> packed(out_off[nm] + 16);
    HIP_CHECK(hipMemcpyAsync(d_out_off.p, out_off.data(), (nm + 1) * 8, hipMemcpyHostToDevice, st));
    launch_compact_members(slots.p, BGZF_SLOT_BYTES, d_sizes.p, d_out_off.p, nm, packed.p, st);
    const size_t base = out->size();
    out->resize(base + out_off[nm]);
    HIP_CHECK(hipMemcpyAsync(out->data() + base, packed.p, out_off[nm], hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
};

struct Writer {
  std::string path;
  FILE* f = nullptr;
  bool zero_based = true;
  std::unordered_map<std::string, int32_t> ref_map;
  std::unique_ptr<Compressor> comp;
  DevBuf<uint8_t> d_stream;   // serialized bytes not yet compressed (device)
  uint64_t stream_len = 0;
  uint64_t n_records = 0, n_members = 0, n_bytes = 0;
  bool finished = false;
  static constexpr uint64_t FLUSH_BYTES = 256ull * BGZF_MAX_PAYLOAD;  // compress once this much has accumulated

  ~Writer() { if (f) fclose(f); }

  void reserve(uint64_t extra) {
    const uint64_t need = stream_len + extra + 64;
    if (d_stream.n >= need) return;
    DevBuf<uint8_t> g(std::max<uint64_t>(need, d_stream.n * 2));
    if (stream_len) HIP_CHECK(hipMemcpyAsync(g.p, d_stream.p, stream_len, hipMemcpyDeviceToDevice, comp->st));
    HIP_CHECK(hipStreamSynchronize(comp->st));
    d_stream = std::move(g);
  }
  void append_host(const uint8_t* p, uint64_t n) {
    reserve(n);
    HIP_CHECK(hipMemcpyAsync(d_stream.p + stream_len, p, n, hipMemcpyHostToDevice, comp->st));
    HIP_CHECK(hipStreamSynchronize(comp->st));
    stream_len += n;
  }
  // compress whole members (all of the stream when `all`), write them, keep the rest at the front of the buffer
  void flush(bool all) {
    const uint64_t take = all ? stream_len : stream_len / BGZF_MAX_PAYLOAD * BGZF_MAX_PAYLOAD;
    if (!take) return;
    HIP_CHECK(hipMemsetAsync(d_stream.p + stream_len, 0, 16, comp->st));  // the match finder reads a few bytes past the end
    std::vector<uint8_t> out;
    comp->compress(d_stream.p, take, &out, &n_members);
    if (fwrite(out.data(), 1, out.size(), f) != out.size()) throw Error("Failed to write BAM records: " + std::string(strerror(errno)));
    n_bytes += out.size();
    const uint64_t rest = stream_len - take;
    if (rest) {
      DevBuf<uint8_t> tmp(rest);
      HIP_CHECK(hipMemcpyAsync(tmp.p, d_stream.p + take, rest, hipMemcpyDeviceToDevice, comp->st));
      HIP_CHECK(hipMemcpyAsync(d_stream.p, tmp.p, rest, hipMemcpyDeviceToDevice, comp->st));
      HIP_CHECK(hipStreamSynchronize(comp->st));
    }
    stream_len = rest;
  }
};

// ---- Arrow C Data helpers ------------------------------------------------------------------------------------------
struct Col {
  const ArrowArray* a = nullptr;
  std::string format;
};
static Col find_col(const ArrowArray* batch, const ArrowSchema* schema, const char* name, bool required) {
  for (int64_t i = 0; i < schema->n_children; i++)
    if (schema->children[i]->name && !strcmp(schema->children[i]->name, name)) {
      Col c;
      c.a = batch->children[i];
      c.format = schema->children[i]->format ? schema->children[i]->format : "";
      return c;
    }
  if (required) throw Error(std::string("Required column '") + name + "' not found in batch");
  return Col{};
}
static void want_format(const Col& c, const char* name, const char* fmt, const char* what) {
  if (c.format != fmt) throw Error(std::string("Column '") + name + "' must be " + what + " type");
}

struct DevCol {   // one column's buffers on the device
  DevBuf<uint8_t> valid, values;
  DevBuf<int32_t> off;
};
static const uint8_t* up_valid(const ArrowArray* a, DevCol* d, hipStream_t st) {
  if (a->null_count == 0 || !a->buffers[0]) return nullptr;
  const size_t bytes = (size_t)((a->offset + a->length + 7) / 8);
  d->valid.alloc(bytes + 1);
  HIP_CHECK(hipMemcpyAsync(d->valid.p, a->buffers[0], bytes, hipMemcpyHostToDevice, st));
  return d->valid.p;
}
static void up_var(const ArrowArray* a, DevCol* d, hipStream_t st) {
  const int32_t* off = (const int32_t*)a->buffers[1];
  const size_t n_off = (size_t)(a->offset + a->length + 1);
  d->off.alloc(n_off);
  HIP_CHECK(hipMemcpyAsync(d->off.p, off, n_off * 4, hipMemcpyHostToDevice, st));
  const size_t bytes = (size_t)off[n_off - 1];
  d->values.alloc(bytes + 16);
  if (bytes) HIP_CHECK(hipMemcpyAsync(d->values.p, a->buffers[2], bytes, hipMemcpyHostToDevice, st));
}
// The core columns are uploaded REBASED: rows [eff, eff + n) of the Arrow array (eff = the array's own offset + the offset of
// the struct array the batch came as) become rows [0, n) on the device, so columns sliced differently -- and sliced
// batches -- need nothing from the kernels.  A validity bitmap whose first bit is not at a byte boundary is repacked on the
// host; `keep` owns such staging blocks until the batch's copies have been waited for.
static const uint8_t* up_valid_at(const ArrowArray* a, int64_t eff, int64_t n, DevCol* d, std::vector<std::vector<uint8_t>>* keep, hipStream_t st) {
  if (a->null_count == 0 || !a->buffers[0]) return nullptr;
  const uint8_t* v = (const uint8_t*)a->buffers[0];
  const size_t bytes = (size_t)((n + 7) / 8);
  d->valid.alloc(bytes + 1);
  if ((eff & 7) == 0) {
    HIP_CHECK(hipMemcpyAsync(d->valid.p, v + (eff >> 3), bytes, hipMemcpyHostToDevice, st));
  } else {
    keep->emplace_back(bytes + 1, 0);
    std::vector<uint8_t>& t = keep->back();
    const int sh = (int)(eff & 7);
    const size_t b0 = (size_t)(eff >> 3), last = (size_t)((eff + n - 1) >> 3);
    for (size_t k = 0; k < bytes; k++) {
      const uint32_t lo = v[b0 + k], hi = b0 + k + 1 <= last ? v[b0 + k + 1] : 0u;
      t[k] = (uint8_t)((lo >> sh) | (hi << (8 - sh)));
    }
    HIP_CHECK(hipMemcpyAsync(d->valid.p, t.data(), bytes, hipMemcpyHostToDevice, st));
  }
  return d->valid.p;
}
static void up_var_at(const ArrowArray* a, int64_t eff, int64_t n, DevCol* d, hipStream_t st) {
  const int32_t* off = (const int32_t*)a->buffers[1] + eff;   // the offsets stay absolute: the value bytes are uploaded from byte 0
  d->off.alloc((size_t)n + 1);
  HIP_CHECK(hipMemcpyAsync(d->off.p, off, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st));
  const size_t bytes = (size_t)off[n];
  d->values.alloc(bytes + 16);
  if (bytes) HIP_CHECK(hipMemcpyAsync(d->values.p, a->buffers[2], bytes, hipMemcpyHostToDevice, st));
}
static void up_fixed_at(const ArrowArray* a, int64_t eff, int64_t n, DevCol* d, hipStream_t st) {
  const size_t bytes = (size_t)n * 4;
  d->values.alloc(bytes + 4);
  if (bytes) HIP_CHECK(hipMemcpyAsync(d->values.p, (const uint8_t*)a->buffers[1] + (size_t)eff * 4, bytes, hipMemcpyHostToDevice, st));
}

// ---- tag columns ---------------------------------------------------------------------------------------------------
// Arrow C Data metadata: int32 n, then n x (int32 key length, key, int32 value length, value)
static std::unordered_map<std::string, std::string> field_metadata(const ArrowSchema* f) {
  std::unordered_map<std::string, std::string> m;
  const char* p = f->metadata;
  if (!p) return m;
  int32_t n;
  memcpy(&n, p, 4); p += 4;
  for (int32_t i = 0; i < n; i++) {
    int32_t kl, vl;
    memcpy(&kl, p, 4); p += 4;
    std::string k(p, (size_t)kl); p += kl;
    memcpy(&vl, p, 4); p += 4;
    std::string v(p, (size_t)vl); p += vl;
    m.emplace(std::move(k), std::move(v));
  }
  return m;
}
static uint8_t kind_of(const std::string& fmt) {
  if (fmt == "c") return SK_I8; if (fmt == "s") return SK_I16; if (fmt == "i") return SK_I32; if (fmt == "l") return SK_I64;
  if (fmt == "C") return SK_U8; if (fmt == "S") return SK_U16; if (fmt == "I") return SK_U32; if (fmt == "L") return SK_U64;
  if (fmt == "f") return SK_F32; if (fmt == "g") return SK_F64; if (fmt == "u") return SK_UTF8; if (fmt == "+l") return SK_LIST;
  return 0;
}
static size_t kind_width(uint8_t k) {
  switch (k) { case SK_I8: case SK_U8: return 1; case SK_I16: case SK_U16: return 2; case SK_I32: case SK_U32: case SK_F32: return 4;
               case SK_I64: case SK_U64: case SK_F64: return 8; default: return 0; }
}
static bool kind_is_int(uint8_t k) { return k >= SK_I8 && k <= SK_U64; }
static std::string arrow_type_name(const ArrowSchema* f) {   // the DataType's Debug form for the common cases
  const std::string fmt = f->format ? f->format : "";
  static const std::unordered_map<std::string, std::string> names = {
      {"c", "Int8"}, {"C", "UInt8"}, {"s", "Int16"}, {"S", "UInt16"}, {"i", "Int32"}, {"I", "UInt32"}, {"l", "Int64"}, {"L", "UInt64"},
      {"f", "Float32"}, {"g", "Float64"}, {"u", "Utf8"}, {"U", "LargeUtf8"}, {"z", "Binary"}, {"b", "Boolean"}, {"n", "Null"}};
  auto it = names.find(fmt);
  if (it != names.end()) return it->second;
  if (fmt == "+l" && f->n_children == 1) return "List(" + arrow_type_name(f->children[0]) + ")";
  return fmt;
}
// f64 as Rust's Display prints it: shortest digits that round-trip, positional notation, no exponent
static std::string rust_f64(double v) {
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
  char buf[64];
  int prec = 0;
  for (; prec < 17; prec++) {
    snprintf(buf, sizeof buf, "%.*e", prec, v);
    if (strtod(buf, nullptr) == v) break;
  }
  std::string t = buf;                       // [-]d.ddddde[+-]xx
  const bool neg = t[0] == '-';
  if (neg) t.erase(0, 1);
  const size_t ep = t.find('e');
  const int ex = atoi(t.c_str() + ep + 1);
  std::string dig = t.substr(0, ep);
  dig.erase(std::remove(dig.begin(), dig.end(), '.'), dig.end());
  std::string out;
  if (ex >= 0) {
    if ((int)dig.size() <= ex + 1) out = dig + std::string((size_t)(ex + 1 - (int)dig.size()), '0');
    else out = dig.substr(0, (size_t)ex + 1) + "." + dig.substr((size_t)ex + 1);
  } else {
    out = "0." + std::string((size_t)(-ex - 1), '0') + dig;
  }
  while (out.find('.') != std::string::npos && out.back() == '0') out.pop_back();
  if (!out.empty() && out.back() == '.') out.pop_back();
  return (neg ? "-" : "") + out;
}
static bool host_valid(const ArrowArray* a, int64_t i) {
  const uint8_t* v = (const uint8_t*)a->buffers[0];
  const int64_t j = i + a->offset;
  return a->null_count == 0 || !v || ((v[j >> 3] >> (j & 7)) & 1);
}
static std::string host_num(const ArrowArray* a, uint8_t kind, int64_t j) {   // element j (absolute index) as Rust prints it
  const void* v = a->buffers[1];
  switch (kind) {
    case SK_I8: return std::to_string((int)((const int8_t*)v)[j]); case SK_I16: return std::to_string(((const int16_t*)v)[j]);
    case SK_I32: return std::to_string(((const int32_t*)v)[j]); case SK_I64: return std::to_string(((const int64_t*)v)[j]);
    case SK_U8: return std::to_string((unsigned)((const uint8_t*)v)[j]); case SK_U16: return std::to_string(((const uint16_t*)v)[j]);
    case SK_U32: return std::to_string(((const uint32_t*)v)[j]); case SK_U64: return std::to_string(((const uint64_t*)v)[j]);
    case SK_F32: return rust_f64((double)((const float*)v)[j]); case SK_F64: return rust_f64(((const double*)v)[j]);
    default: return "?";
  }
}
static bool host_int_fits(uint8_t t, const ArrowArray* a, uint8_t kind, int64_t j) {
  const void* v = a->buffers[1];
  bool neg = false; uint64_t u = 0; int64_t sv = 0;
  switch (kind) {
    case SK_I8: sv = ((const int8_t*)v)[j]; break; case SK_I16: sv = ((const int16_t*)v)[j]; break;
    case SK_I32: sv = ((const int32_t*)v)[j]; break; case SK_I64: sv = ((const int64_t*)v)[j]; break;
    case SK_U8: u = ((const uint8_t*)v)[j]; break; case SK_U16: u = ((const uint16_t*)v)[j]; break;
    case SK_U32: u = ((const uint32_t*)v)[j]; break; case SK_U64: u = ((const uint64_t*)v)[j]; break;
    default: return true;
  }
  if (kind <= SK_I64) { neg = sv < 0; u = (uint64_t)sv; }
  switch (t) {
    case 'c': return neg ? sv >= -128 : u <= 127u; case 's': return neg ? sv >= -32768 : u <= 32767u;
    case 'i': return neg ? sv >= -2147483648ll : u <= 2147483647u; case 'C': return !neg && u <= 255u;
    case 'S': return !neg && u <= 65535u; case 'I': return !neg && u <= 4294967295ull; default: return true;
  }
}
struct TagPlan {            // one tag column: what the device needs + what the host needs to word an error
  SerTagCol d{};
  const ArrowArray* a = nullptr;
  const ArrowSchema* f = nullptr;
  DevCol dev;
  DevBuf<uint8_t> evalid;
  int64_t row0 = 0;   // index of the batch's row 0 in the column (its own offset + the struct array's)
};
[[noreturn]] static void throw_tag_err(const std::vector<std::unique_ptr<TagPlan>>& plans, unsigned long long key) {
  const int64_t row = (int64_t)(key >> 16);
  const uint32_t ci = (uint32_t)((key >> 8) & 0xFFu), code = (uint32_t)(key & 0xFFu);
  const TagPlan& t = *plans.at(ci);
  const int64_t j = row + t.row0;
  auto str_at = [&]() {
    const int32_t* off = (const int32_t*)t.a->buffers[1];
    return std::string((const char*)t.a->buffers[2] + off[j], (size_t)(off[j + 1] - off[j]));
  };
  switch (code) {
    case 20: throw Error("Integer value " + host_num(t.a, t.d.kind, j) + " does not fit SAM type '" + std::string(1, (char)t.d.sam_type) + "'");
    case 21: throw Error("Float value " + host_num(t.a, t.d.kind, j) + " does not fit SAM type 'f'");
    case 22: { std::string v = str_at(); for (auto& ch : v) if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32); throw Error("Invalid SAM hex tag value '" + v + "'"); }
    case 23: throw Error("Character tags must be a single ASCII byte, got '" + str_at() + "'");
    case 24: throw Error("Character tag value " + host_num(t.a, t.d.kind, j) + " does not fit into a single byte");
    case 25: throw Error("SAM array tags cannot contain null elements");
    case 26: {
      const int32_t* off = (const int32_t*)t.a->buffers[1];
      const ArrowArray* ch = t.a->children[0];
      for (int32_t k = off[j]; k < off[j + 1]; k++) {
        const int64_t q = (int64_t)k + ch->offset;
        bool bad;
        if (t.d.subtype == 'f') {
          const double v = t.d.ekind == SK_F64 ? ((const double*)ch->buffers[1])[q] : (double)((const float*)ch->buffers[1])[q];
          bad = !(std::isfinite(v) && v >= -3.4028234663852886e38 && v <= 3.4028234663852886e38);
        } else bad = !host_int_fits(t.d.subtype, ch, t.d.ekind, q);
        if (bad) throw Error("Array element " + host_num(ch, t.d.ekind, q) + " does not fit SAM subtype '" + std::string(1, (char)t.d.subtype) + "'");
      }
      throw Error("Array element does not fit its SAM subtype");
    }
    default: throw Error("Failed to write BAM records: tag error " + std::to_string(code));
  }
}

static void throw_ser_err(uint32_t e) {
  switch (e) {
    case 0: return;
    case 1: throw Error("Flag value does not fit into 16-bit SAM flags");
    case 2: throw Error("Failed to parse CIGAR: invalid operation or length");
    case 3: throw Error("Failed to write BAM records: sequence and quality scores differ in length");
    case 4: throw Error("Failed to write BAM records: read name longer than 254 bytes");
    case 5: throw Error("Failed to write BAM records: more than 65535 CIGAR operations");
    default: throw Error("Failed to write BAM records: device error " + std::to_string(e));
  }
}

}  // namespace

// ---- SAM header from Arrow schema metadata (bio-format-bam/src/header_builder.rs:42-195) -------------------------------
// A reader for the JSON the provider writes into schema metadata (arrays of flat objects: strings, numbers, null, and one
// nested string map "other_fields"; serde's output for ReferenceSequenceMetadata / ReadGroupMetadata / ProgramMetadata,
// bio-format-core/src/metadata.rs:249-300).  Anything that does not parse makes the key count as absent, like
// from_json_string returning None.
struct JVal {
  enum Kind { NUL, STR, NUM, OBJ, ARR, BOOL } kind = NUL;
  std::string s;                                       // STR: the text; NUM: the literal
  std::vector<std::pair<std::string, JVal>> obj;       // OBJ, in document order
  std::vector<JVal> arr;
  const JVal* get(const char* k) const {
    for (auto& kv : obj) if (kv.first == k) return &kv.second;
    return nullptr;
  }
};
struct JParser {
  const char* p; const char* e; bool ok = true;
  void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  static void put_utf8(std::string& o, uint32_t c) {
    if (c < 0x80) o.push_back((char)c);
    else if (c < 0x800) { o.push_back((char)(0xC0 | (c >> 6))); o.push_back((char)(0x80 | (c & 63))); }
    else if (c < 0x10000) { o.push_back((char)(0xE0 | (c >> 12))); o.push_back((char)(0x80 | ((c >> 6) & 63))); o.push_back((char)(0x80 | (c & 63))); }
    else { o.push_back((char)(0xF0 | (c >> 18))); o.push_back((char)(0x80 | ((c >> 12) & 63))); o.push_back((char)(0x80 | ((c >> 6) & 63))); o.push_back((char)(0x80 | (c & 63))); }
  }
  bool hex4(uint32_t* v) {
    if (e - p < 4) return false;
    uint32_t x = 0;
    for (int i = 0; i < 4; i++) {
      const char c = p[i];
      x <<= 4;
      if (c >= '0' && c <= '9') x |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') x |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') x |= (uint32_t)(c - 'A' + 10);
      else return false;
    }
    p += 4; *v = x;
    return true;
  }
  std::string str() {
    std::string o;
    if (p >= e || *p != '"') { ok = false; return o; }
    p++;
    while (p < e && *p != '"') {
      if (*p == '\\') {
        if (++p >= e) { ok = false; return o; }
        const char c = *p++;
        switch (c) {
          case 'n': o.push_back('\n'); break; case 't': o.push_back('\t'); break; case 'r': o.push_back('\r'); break;
          case 'b': o.push_back('\b'); break; case 'f': o.push_back('\f'); break;
          case 'u': {
            uint32_t c1 = 0;
            if (!hex4(&c1)) { ok = false; return o; }
            if (c1 >= 0xD800 && c1 < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {
              p += 2;
              uint32_t c2 = 0;
              if (!hex4(&c2)) { ok = false; return o; }
              c1 = 0x10000 + ((c1 - 0xD800) << 10) + (c2 - 0xDC00);
            }
            put_utf8(o, c1);
            break;
          }
          default: o.push_back(c);  // \" \\ \/
        }
      } else {
        o.push_back(*p++);
      }
    }
    if (p >= e) { ok = false; return o; }
    p++;
    return o;
  }
  JVal val(int depth = 0) {
    JVal v;
    ws();
    if (p >= e || depth > 8) { ok = false; return v; }
    if (*p == '"') { v.kind = JVal::STR; v.s = str(); }
    else if (*p == '{') {
      v.kind = JVal::OBJ; p++; ws();
      if (p < e && *p == '}') { p++; return v; }
      while (ok) {
        ws();
        std::string k = str();
        ws();
        if (!ok || p >= e || *p != ':') { ok = false; break; }
        p++;
        JVal c = val(depth + 1);
        v.obj.emplace_back(std::move(k), std::move(c));
        ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == '}') { p++; break; }
        ok = false;
      }
    } else if (*p == '[') {
      v.kind = JVal::ARR; p++; ws();
      if (p < e && *p == ']') { p++; return v; }
      while (ok) {
        v.arr.push_back(val(depth + 1));
        ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == ']') { p++; break; }
        ok = false;
      }
    } else if (e - p >= 4 && !strncmp(p, "null", 4)) { p += 4; }
    else if (e - p >= 4 && !strncmp(p, "true", 4)) { p += 4; v.kind = JVal::BOOL; v.s = "true"; }
    else if (e - p >= 5 && !strncmp(p, "false", 5)) { p += 5; v.kind = JVal::BOOL; v.s = "false"; }
    else {
      v.kind = JVal::NUM;
      while (p < e && (isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) v.s.push_back(*p++);
      if (v.s.empty()) ok = false;
    }
    return v;
  }
};
static bool parse_json(const std::string& text, JVal* out) {
  JParser jp{text.data(), text.data() + text.size()};
  *out = jp.val();
  jp.ws();
  return jp.ok && jp.p == jp.e;
}

struct BuiltHeader {
  std::string text;
  std::vector<std::string> ref_names;
  std::vector<int64_t> ref_lengths;
};
// Optional fields of a header line that noodles keeps ("other_fields" whose key is one of the standard tags of that record
// kind; header_builder.rs:192-250).  serde writes other_fields from a HashMap, so their order in the reference is that
// map's iteration order -- unspecified; here they follow in key order.
static void append_other(std::string& line, const JVal* other, const char* const* allowed) {
  if (!other || other->kind != JVal::OBJ) return;
  std::vector<std::pair<std::string, std::string>> kv;
  for (auto& f : other->obj) {
    bool ok = false;
    for (const char* const* a = allowed; *a; a++) ok = ok || f.first == *a;
    if (ok && f.second.kind == JVal::STR) kv.emplace_back(f.first, f.second.s);
  }
  std::sort(kv.begin(), kv.end());
  for (auto& f : kv) line += "\t" + f.first + ":" + f.second;
}
static BuiltHeader build_bam_header(const std::unordered_map<std::string, std::string>& md) {
  BuiltHeader h;
  auto get = [&](const char* k) -> const std::string* { auto it = md.find(k); return it == md.end() ? nullptr : &it->second; };
  // @HD: VN from bio.bam.file_format_version when it parses as major.minor, else 1.6; SO / GO / SS when present
  std::string vn = "1.6";
  if (const std::string* v = get("bio.bam.file_format_version")) {
    const size_t dot = v->find('.');
    auto digits = [](const std::string& t) { return !t.empty() && t.find_first_not_of("0123456789") == std::string::npos; };
    if (dot != std::string::npos && digits(v->substr(0, dot)) && digits(v->substr(dot + 1)))
      vn = std::to_string(strtoul(v->substr(0, dot).c_str(), nullptr, 10)) + "." + std::to_string(strtoul(v->substr(dot + 1).c_str(), nullptr, 10));
  }
  h.text = "@HD\tVN:" + vn;
  if (const std::string* v = get("bio.bam.sort_order")) h.text += "\tSO:" + *v;
  if (const std::string* v = get("bio.bam.group_order")) h.text += "\tGO:" + *v;
  if (const std::string* v = get("bio.bam.subsort_order")) h.text += "\tSS:" + *v;
  h.text += "\n";
  JVal j;
  if (const std::string* v = get("bio.bam.reference_sequences"); v && parse_json(*v, &j) && j.kind == JVal::ARR) {
    bool shape = true;
    for (auto& r : j.arr) {
      const JVal* nm = r.get("name"); const JVal* ln = r.get("length");
      shape = shape && r.kind == JVal::OBJ && nm && nm->kind == JVal::STR && ln && ln->kind == JVal::NUM && ln->s.find_first_not_of("0123456789") == std::string::npos;
    }
    if (shape) {
      static const char* const SQ[] = {"AH", "AN", "AS", "DS", "M5", "SP", "TP", "UR", nullptr};
      for (auto& r : j.arr) {
        const unsigned long long len = strtoull(r.get("length")->s.c_str(), nullptr, 10);
        if (len == 0) throw Error("Reference sequence length cannot be zero");
        std::string line = "@SQ\tSN:" + r.get("name")->s + "\tLN:" + std::to_string(len);
        append_other(line, r.get("other_fields"), SQ);
        h.text += line + "\n";
        h.ref_names.push_back(r.get("name")->s);
        h.ref_lengths.push_back((int64_t)len);
      }
    }
  }
  auto opt = [](const JVal& r, const char* k) -> const std::string* {
    const JVal* v = r.get(k);
    return v && v->kind == JVal::STR ? &v->s : nullptr;
  };
  if (const std::string* v = get("bio.bam.read_groups"); v && parse_json(*v, &j) && j.kind == JVal::ARR) {
    bool shape = true;
    for (auto& r : j.arr) shape = shape && r.kind == JVal::OBJ && opt(r, "id");
    if (shape) {
      static const char* const RG[] = {"BC", "CN", "DT", "FO", "KS", "PG", "PI", "PM", "PU", nullptr};
      for (auto& r : j.arr) {
        std::string line = "@RG\tID:" + *opt(r, "id");
        if (auto x = opt(r, "sample")) line += "\tSM:" + *x;
        if (auto x = opt(r, "platform")) line += "\tPL:" + *x;
        if (auto x = opt(r, "library")) line += "\tLB:" + *x;
        if (auto x = opt(r, "description")) line += "\tDS:" + *x;
        append_other(line, r.get("other_fields"), RG);
        h.text += line + "\n";
      }
    }
  }
  if (const std::string* v = get("bio.bam.program_info"); v && parse_json(*v, &j) && j.kind == JVal::ARR) {
    bool shape = true;
    for (auto& r : j.arr) shape = shape && r.kind == JVal::OBJ && opt(r, "id");
    if (shape) {
      static const char* const PG[] = {"PP", "DS", nullptr};
      for (auto& r : j.arr) {
        std::string line = "@PG\tID:" + *opt(r, "id");
        if (auto x = opt(r, "name")) line += "\tPN:" + *x;
        if (auto x = opt(r, "version")) line += "\tVN:" + *x;
        if (auto x = opt(r, "command_line")) line += "\tCL:" + *x;
        append_other(line, r.get("other_fields"), PG);
        h.text += line + "\n";
      }
    }
  }
  if (const std::string* v = get("bio.bam.comments"); v && parse_json(*v, &j) && j.kind == JVal::ARR) {
    bool shape = true;
    for (auto& c : j.arr) shape = shape && c.kind == JVal::STR;
    if (shape) for (auto& c : j.arr) h.text += "@CO\t" + c.s + "\n";
  }
  return h;
}

struct bioscan_bam_writer { Writer w; };

#define W_BEGIN try {
#define W_END                                        \
  }                                                  \
  catch (const std::exception& e) {                  \
    ::bioscan::set_last_error(e.what());             \
    return 1;                                        \
  }                                                  \
  return 0;

extern "C" {

int bioscan_bam_writer_open(const char* path, const char* header_text, const char* const* ref_names, const int64_t* ref_lengths,
                            int32_t n_ref, int32_t coordinate_system_zero_based, int32_t device_id, bioscan_bam_writer** out) {
  W_BEGIN
  {
    // BamCompressionType::from_path (bio-format-bam/src/writer.rs:27-43): a path ending in .sam selects the plain SAM text
    // writer in the reference.  This library writes BGZF BAM only (SAM text is not on the scan path) and
    // says so instead of writing BAM bytes into a .sam file.
    std::string low = path ? path : "";
    for (auto& ch : low) ch = (char)tolower((unsigned char)ch);
    if (low.size() >= 4 && low.compare(low.size() - 4, 4, ".sam") == 0)
      throw Error("Failed to create output file: '.sam' selects the plain SAM text writer, which this library does not provide (BGZF BAM only): " + std::string(path));
  }
  char nm[8];
  if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(bioscan_last_error());
  std::unique_ptr<bioscan_bam_writer> bw(new bioscan_bam_writer);
  Writer& w = bw->w;
  w.path = path;
  w.zero_based = coordinate_system_zero_based != 0;
  w.f = fopen(path, "wb");
  if (!w.f) throw Error(std::string("Failed to create output file: ") + strerror(errno));
  w.comp.reset(new Compressor(device_id));
  // BAM header (SAM spec 4.2): magic, l_text, text, n_ref, { l_name, name\0, l_ref }
  std::string text = header_text ? header_text : "";
  std::vector<uint8_t> h;
  auto put32 = [&](int32_t v) { const uint8_t* p = (const uint8_t*)&v; h.insert(h.end(), p, p + 4); };
  h.insert(h.end(), {'B', 'A', 'M', 1});
  put32((int32_t)text.size());
  h.insert(h.end(), text.begin(), text.end());
  put32(n_ref);
  for (int32_t i = 0; i < n_ref; i++) {
    const std::string name = ref_names[i];
    put32((int32_t)name.size() + 1);
    h.insert(h.end(), name.begin(), name.end());
    h.push_back(0);
    put32((int32_t)ref_lengths[i]);
    w.ref_map.emplace(name, i);
  }
  w.append_host(h.data(), h.size());
  *out = bw.release();
  W_END
}

// schema-level metadata of the struct schema a RecordBatch is exported as, with the overrides of insert_into applied
static std::unordered_map<std::string, std::string> effective_metadata(const struct ArrowSchema* schema, int32_t sort_on_write) {
  auto md = field_metadata(schema);
  // table_provider.rs:1156-1164: the provider's sort_on_write flag decides @HD SO, whatever the schema says
  if (sort_on_write >= 0) md["bio.bam.sort_order"] = sort_on_write ? "coordinate" : "unsorted";
  return md;
}

int bioscan_bam_header_from_schema(const struct ArrowSchema* schema, int32_t sort_on_write, char** header_text) {
  W_BEGIN
  const BuiltHeader h = build_bam_header(effective_metadata(schema, sort_on_write));
  char* o = (char*)malloc(h.text.size() + 1);
  if (!o) throw Error("out of host memory");
  memcpy(o, h.text.c_str(), h.text.size() + 1);
  *header_text = o;
  W_END
}

int bioscan_bam_writer_open_schema(const char* path, const struct ArrowSchema* schema, int32_t sort_on_write, int32_t device_id,
                                   bioscan_bam_writer** out) {
  W_BEGIN
  const auto md = effective_metadata(schema, sort_on_write);
  const BuiltHeader h = build_bam_header(md);
  // table_provider.rs:1131-1135: the coordinate system of the rows is the schema's, 0-based when it does not say
  bool zero_based = true;
  if (auto it = md.find("bio.coordinate_system_zero_based"); it != md.end()) {
    if (it->second == "true") zero_based = true;
    else if (it->second == "false") zero_based = false;
  }
  std::vector<const char*> names;
  for (auto& n : h.ref_names) names.push_back(n.c_str());
  if (names.empty()) names.push_back("");
  if (bioscan_bam_writer_open(path, h.text.c_str(), names.data(), h.ref_lengths.empty() ? nullptr : h.ref_lengths.data(),
                              (int32_t)h.ref_names.size(), zero_based ? 1 : 0, device_id, out))
    throw Error(bioscan_last_error());
  W_END
}

int bioscan_bam_writer_write(bioscan_bam_writer* bw, const struct ArrowArray* batch, const struct ArrowSchema* schema) {
  W_BEGIN
  Writer& w = bw->w;
  if (w.finished) throw Error("BAM writer is finished");
  const int64_t n = batch->length;
  if (n == 0) return 0;
  const int64_t boff = batch->offset;   // a sliced struct array: its offset adds to every child's own
  hipStream_t st = w.comp->st;
  HIP_CHECK(hipSetDevice(w.comp->device));
  const Col name = find_col(batch, schema, "name", true), chrom = find_col(batch, schema, "chrom", true),
            start = find_col(batch, schema, "start", true), flags = find_col(batch, schema, "flags", true),
            cigar = find_col(batch, schema, "cigar", true), mapq = find_col(batch, schema, "mapping_quality", true),
            mchrom = find_col(batch, schema, "mate_chrom", true), mstart = find_col(batch, schema, "mate_start", true),
            seq = find_col(batch, schema, "sequence", true), qual = find_col(batch, schema, "quality_scores", true),
            tlen = find_col(batch, schema, "template_length", true);
  want_format(name, "name", "u", "String"); want_format(chrom, "chrom", "u", "String");
  want_format(mchrom, "mate_chrom", "u", "String"); want_format(seq, "sequence", "u", "String");
  want_format(qual, "quality_scores", "u", "String");
  want_format(start, "start", "I", "UInt32"); want_format(flags, "flags", "I", "UInt32");
  want_format(mapq, "mapping_quality", "I", "UInt32"); want_format(mstart, "mate_start", "I", "UInt32");
  want_format(tlen, "template_length", "i", "Int32");
  const bool cigar_binary = cigar.format == "z";
  if (!cigar_binary && cigar.format != "u") throw Error("Column 'cigar' must be String or Binary type");
  const ArrowArray* all[] = {name.a, chrom.a, start.a, flags.a, cigar.a, mapq.a, mchrom.a, mstart.a, seq.a, qual.a, tlen.a};
  for (auto* a : all)
    if (a->length < boff + n) throw Error("Failed to write BAM records: columns differ in length");
  // chrom / mate_chrom -> reference ids (sam_record_serializer.rs:145-151, 176-186)
  std::vector<int32_t> refid((size_t)n), mrefid((size_t)n);
  {
    auto is_valid = [boff](const ArrowArray* a, int64_t i) {
      const uint8_t* v = (const uint8_t*)a->buffers[0];
      const int64_t j = i + a->offset + boff;
      return a->null_count == 0 || !v || ((v[j >> 3] >> (j & 7)) & 1);
    };
    auto str_at = [boff](const ArrowArray* a, int64_t i) {
      const int32_t* off = (const int32_t*)a->buffers[1];
      const char* d = (const char*)a->buffers[2];
      const int64_t j = i + a->offset + boff;
      return std::string(d + off[j], (size_t)(off[j + 1] - off[j]));
    };
    std::string last_c, last_m;
    int32_t last_ci = -1, last_mi = -1;
    bool have_c = false, have_m = false;
    for (int64_t i = 0; i < n; i++) {
      int32_t r = -1;
      if (is_valid(chrom.a, i)) {
        std::string s = str_at(chrom.a, i);
        if (have_c && s == last_c) r = last_ci;
        else { auto it = w.ref_map.find(s); r = it == w.ref_map.end() ? -1 : it->second; last_c = std::move(s); last_ci = r; have_c = true; }
      }
      refid[(size_t)i] = r;
      int32_t m = -1;
      if (is_valid(mchrom.a, i)) {
        std::string s = str_at(mchrom.a, i);
        if (s == "=") m = r;
        else if (have_m && s == last_m) m = last_mi;
        else { auto it = w.ref_map.find(s); m = it == w.ref_map.end() ? -1 : it->second; last_m = std::move(s); last_mi = m; have_m = true; }
      }
      mrefid[(size_t)i] = m;
    }
  }
  DevCol d_name, d_cigar, d_seq, d_qual, d_start, d_flags, d_mapq, d_mstart, d_tlen;
  DevBuf<int32_t> d_refid((size_t)n), d_mrefid((size_t)n);
  SerCols c{};
  c.offset = 0;   // (rebased uploads, see up_valid_at)
  c.zero_based = w.zero_based ? 1 : 0;
  c.cigar_binary = cigar_binary ? 1 : 0;
  std::vector<std::vector<uint8_t>> keep;
  auto eff = [boff](const ArrowArray* a) { return a->offset + boff; };
  c.name_valid = up_valid_at(name.a, eff(name.a), n, &d_name, &keep, st); up_var_at(name.a, eff(name.a), n, &d_name, st); c.name_off = d_name.off.p; c.name = d_name.values.p;
  up_var_at(cigar.a, eff(cigar.a), n, &d_cigar, st); c.cigar_off = d_cigar.off.p; c.cigar = d_cigar.values.p;
  up_var_at(seq.a, eff(seq.a), n, &d_seq, st); c.seq_off = d_seq.off.p; c.seq = d_seq.values.p;
  up_var_at(qual.a, eff(qual.a), n, &d_qual, st); c.qual_off = d_qual.off.p; c.qual = d_qual.values.p;
  c.start_valid = up_valid_at(start.a, eff(start.a), n, &d_start, &keep, st); up_fixed_at(start.a, eff(start.a), n, &d_start, st); c.start = (const uint32_t*)d_start.values.p;
  up_fixed_at(flags.a, eff(flags.a), n, &d_flags, st); c.flags = (const uint32_t*)d_flags.values.p;
  up_fixed_at(mapq.a, eff(mapq.a), n, &d_mapq, st); c.mapq = (const uint32_t*)d_mapq.values.p;
  c.mate_start_valid = up_valid_at(mstart.a, eff(mstart.a), n, &d_mstart, &keep, st); up_fixed_at(mstart.a, eff(mstart.a), n, &d_mstart, st); c.mate_start = (const uint32_t*)d_mstart.values.p;
  up_fixed_at(tlen.a, eff(tlen.a), n, &d_tlen, st); c.tlen = (const int32_t*)d_tlen.values.p;
  HIP_CHECK(hipMemcpyAsync(d_refid.p, refid.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(d_mrefid.p, mrefid.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  c.refid = d_refid.p; c.mate_refid = d_mrefid.p;
  // tag columns: every field that carries bio.bam.tag.tag metadata, in schema order (table_provider.rs:1137-1143,
  // sam_record_serializer.rs:281-299); the SAM type comes from bio.bam.tag.type, "Z" when absent (sam_tag_io.rs:129-133)
  std::vector<std::unique_ptr<TagPlan>> plans;
  for (int64_t ci = 0; ci < schema->n_children; ci++) {
    const ArrowSchema* f = schema->children[ci];
    const auto md = field_metadata(f);
    if (!md.count("bio.bam.tag.tag")) continue;
    const std::string tname = f->name ? f->name : "";
    const ArrowArray* a = batch->children[ci];
    if (a->length < boff + n) throw Error("Failed to write BAM records: columns differ in length");
    if (a->null_count == a->length && a->length > 0) continue;   // nothing to write; the reference never looks at the type
    if (tname.size() != 2) continue;                             // sam_tag_io.rs:135-138
    std::string spec = "Z";
    if (auto it = md.find("bio.bam.tag.type"); it != md.end()) spec = it->second;
    // parse_sam_tag_type (tag_registry.rs:78-106)
    char sam_type = 0, subtype = 0;
    {
      const size_t colon = spec.find(':');
      if (colon == std::string::npos) {
        if (spec.size() != 1) throw Error("Invalid SAM tag type metadata: Invalid SAM tag type '" + spec + "': type must be a single character");
        sam_type = spec[0];
      } else if (spec.compare(0, colon, "B") == 0 && spec.find(':', colon + 1) == std::string::npos) {
        const std::string sub = spec.substr(colon + 1);
        if (sub.size() != 1) throw Error("Invalid SAM tag type metadata: Invalid SAM tag array type '" + spec + "': subtype must be a single character");
        if (!strchr("cCsSiIf", sub[0])) throw Error("Invalid SAM tag type metadata: Unsupported SAM array subtype '" + sub + "'");
        sam_type = 'B'; subtype = sub[0];
      } else {
        throw Error("Invalid SAM tag type metadata: Invalid SAM tag type '" + spec + "': expected 'TYPE' or 'B:SUBTYPE'");
      }
    }
    const std::string fmt = f->format ? f->format : "";
    const uint8_t kind = kind_of(fmt);
    const std::string tn = arrow_type_name(f);
    std::unique_ptr<TagPlan> t(new TagPlan);
    t->a = a; t->f = f;
    t->d.tag[0] = (uint8_t)tname[0]; t->d.tag[1] = (uint8_t)tname[1];
    t->d.kind = kind; t->d.offset = a->offset + boff; t->row0 = a->offset + boff;
    if (strchr("icsCSI", sam_type)) {
      if (!kind_is_int(kind)) throw Error("Tag value type mismatch for integer: " + tn);
    } else if (sam_type == 'f') {
      if (kind != SK_F32 && kind != SK_F64) throw Error("Tag value type mismatch for float: " + tn);
    } else if (sam_type == 'Z') {
      if (kind != SK_UTF8) throw Error("Tag value type mismatch for string: " + tn);
    } else if (sam_type == 'H') {
      if (kind != SK_UTF8) throw Error("Tag value type mismatch for hex string: " + tn);
    } else if (sam_type == 'A') {
      if (kind != SK_UTF8 && !kind_is_int(kind)) throw Error("Tag value type mismatch for character: " + tn);
    } else if (sam_type == 'B') {
      if (kind != SK_LIST) throw Error("Tag value type mismatch for array: " + tn);
      const ArrowSchema* cf = f->children[0];
      const uint8_t ek = kind_of(cf->format ? cf->format : "");
      if (!subtype) {  // sam_array_subtype_from_arrow_type (tag_registry.rs:48-59)
        switch (ek) { case SK_I8: subtype = 'c'; break; case SK_U8: subtype = 'C'; break; case SK_I16: subtype = 's'; break;
                      case SK_U16: subtype = 'S'; break; case SK_I32: subtype = 'i'; break; case SK_U32: subtype = 'I'; break;
                      case SK_F32: subtype = 'f'; break; default: break; }
        if (!subtype) throw Error("Unable to determine SAM array subtype for Arrow type " + arrow_type_name(cf));
      }
      const bool ok = subtype == 'f' ? (ek == SK_F32 || ek == SK_F64) : kind_is_int(ek);
      if (!ok) throw Error("Unsupported array element type for SAM subtype '" + std::string(1, subtype) + "': " + arrow_type_name(cf));
      t->d.ekind = ek;
    } else {
      if (kind != SK_UTF8) continue;   // any other type character: a string column is written as Z, anything else skipped (:227-233)
      sam_type = 'Z';
    }
    t->d.sam_type = (uint8_t)sam_type; t->d.subtype = (uint8_t)subtype;
    // upload
    t->d.valid = up_valid(a, &t->dev, st);
    if (kind == SK_UTF8) { up_var(a, &t->dev, st); t->d.off = t->dev.off.p; t->d.values = t->dev.values.p; }
    else if (kind == SK_LIST) {
      const ArrowArray* ch = a->children[0];
      const size_t n_off = (size_t)(a->offset + a->length + 1);
      t->dev.off.alloc(n_off);
      HIP_CHECK(hipMemcpyAsync(t->dev.off.p, a->buffers[1], n_off * 4, hipMemcpyHostToDevice, st));
      const size_t bytes = (size_t)(ch->offset + ch->length) * kind_width(t->d.ekind);
      t->dev.values.alloc(bytes + 8);
      if (bytes) HIP_CHECK(hipMemcpyAsync(t->dev.values.p, ch->buffers[1], bytes, hipMemcpyHostToDevice, st));
      t->d.off = t->dev.off.p; t->d.values = t->dev.values.p; t->d.eoffset = ch->offset;
      if (ch->null_count != 0 && ch->buffers[0]) {
        const size_t vb = (size_t)((ch->offset + ch->length + 7) / 8);
        t->evalid.alloc(vb + 1);
        HIP_CHECK(hipMemcpyAsync(t->evalid.p, ch->buffers[0], vb, hipMemcpyHostToDevice, st));
        t->d.evalid = t->evalid.p;
      }
    } else {
      const size_t bytes = (size_t)(a->offset + a->length) * kind_width(kind);
      t->dev.values.alloc(bytes + 8);
      if (bytes) HIP_CHECK(hipMemcpyAsync(t->dev.values.p, a->buffers[1], bytes, hipMemcpyHostToDevice, st));
      t->d.values = t->dev.values.p;
    }
    if (plans.size() >= 255) throw Error("Failed to write BAM records: more than 255 tag columns");
    plans.push_back(std::move(t));
  }
  std::vector<SerTagCol> tag_cols;
  for (auto& t : plans) tag_cols.push_back(t->d);
  DevBuf<SerTagCol> d_tag_cols(std::max<size_t>(tag_cols.size(), 1));
  if (!tag_cols.empty()) HIP_CHECK(hipMemcpyAsync(d_tag_cols.p, tag_cols.data(), tag_cols.size() * sizeof(SerTagCol), hipMemcpyHostToDevice, st));
  SerTags tg{(int32_t)tag_cols.size(), d_tag_cols.p};
  DevBuf<unsigned long long> tag_err(1);
  HIP_CHECK(hipMemsetAsync(tag_err.p, 0xFF, 8, st));
  // sizes -> offsets -> bytes
  DevBuf<uint32_t> rec_bytes((size_t)n), err(1);
  DevBuf<uint64_t> rec_off((size_t)n + 1), tmp(scan_tmp_elems((uint64_t)n));
  HIP_CHECK(hipMemsetAsync(err.p, 0, 4, st));
  launch_ser_sizes(c, tg, (uint64_t)n, rec_bytes.p, err.p, tag_err.p, st);
  launch_exclusive_scan_u32_to_u64(rec_bytes.p, rec_off.p, (uint64_t)n, tmp.p, st);
  uint64_t total = 0;
  uint32_t e = 0;
  unsigned long long te = ~0ull;
  HIP_CHECK(hipMemcpyAsync(&total, rec_off.p + n, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&e, err.p, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&te, tag_err.p, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  throw_ser_err(e);
  if (te != ~0ull) throw_tag_err(plans, te);
  w.reserve(total);
  launch_ser_write(c, tg, (uint64_t)n, rec_off.p, w.d_stream.p + w.stream_len, err.p, st);
  HIP_CHECK(hipMemcpyAsync(&e, err.p, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  throw_ser_err(e);
  w.stream_len += total;
  w.n_records += (uint64_t)n;
  if (w.stream_len >= Writer::FLUSH_BYTES) w.flush(false);
  W_END
}

int bioscan_bam_writer_finish(bioscan_bam_writer* bw, uint64_t* n_records, uint64_t* n_members, uint64_t* n_bytes) {
  W_BEGIN
  Writer& w = bw->w;
  if (!w.finished) {
    w.flush(true);
    if (fwrite(BGZF_EOF, 1, sizeof BGZF_EOF, w.f) != sizeof BGZF_EOF) throw Error("Failed to finish BAM file: " + std::string(strerror(errno)));
    w.n_bytes += sizeof BGZF_EOF;
    if (fclose(w.f) != 0) { w.f = nullptr; throw Error("Failed to finish BAM file: " + std::string(strerror(errno))); }
    w.f = nullptr;
    w.finished = true;
  }
  if (n_records) *n_records = w.n_records;
  if (n_members) *n_members = w.n_members;
  if (n_bytes) *n_bytes = w.n_bytes;
  W_END
}

void bioscan_bam_writer_close(bioscan_bam_writer* w) { delete w; }

int bioscan_bgzf_deflate(const uint8_t* data, size_t len, int32_t device_id, int32_t add_eof, uint8_t** out, size_t* out_len, double* kernel_ms) {
  W_BEGIN
  char nm[8];
  if (bioscan_device_check(device_id, nm, sizeof nm)) throw Error(bioscan_last_error());
  Compressor comp(device_id);
  DevBuf<uint8_t> d(len + 64);
  if (len) HIP_CHECK(hipMemcpyAsync(d.p, data, len, hipMemcpyHostToDevice, comp.st));
  HIP_CHECK(hipMemsetAsync(d.p + len, 0, 64, comp.st));
  std::vector<uint8_t> o;
  uint64_t nmem = 0;
  comp.compress(d.p, len, &o, &nmem);
  if (add_eof) o.insert(o.end(), BGZF_EOF, BGZF_EOF + sizeof BGZF_EOF);
  uint8_t* h = (uint8_t*)malloc(o.size() ? o.size() : 1);
  if (!h) throw Error("out of memory");
  if (!o.empty()) memcpy(h, o.data(), o.size());
  *out = h;
  *out_len = o.size();
  if (kernel_ms) *kernel_ms = comp.kernel_ms;
  W_END
}

}  // extern "C"
