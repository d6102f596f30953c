// bam_writer.cpp -- host side of the BAM write path: RecordBatches (Arrow C Data) -> BAM records -> BGZF members -> file.
//
// Mirrors BamLocalWriter (bio-format-bam/src/writer.rs:57-283: write_header, write_records, finish) with the record
// serialisation of bio-format-core/src/sam_record_serializer.rs and the BGZF framing of noodles-bgzf's Writer.  The
// serialisation, CRC32 and DEFLATE run on the GPU (bam_write.hip); the host uploads the batch's column buffers, resolves
// chrom / mate_chrom names against the header (a dictionary lookup) and writes the finished members to the file.
// Core columns only: tag columns of a batch are not written (noted in DESIGN.md).
#include <cerrno>
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
  // appends them to `out`
  void compress(const uint8_t* d_payload, uint64_t len, std::vector<uint8_t>* out, uint64_t* n_members) {
    if (!len) return;
    HIP_CHECK(hipSetDevice(device));
    const uint32_t nm = (uint32_t)((len + BGZF_MAX_PAYLOAD - 1) / BGZF_MAX_PAYLOAD);
    std::vector<uint64_t> off(nm + 1);
    for (uint32_t m = 0; m <= nm; m++) off[m] = std::min<uint64_t>((uint64_t)m * BGZF_MAX_PAYLOAD, len);
    DevBuf<uint64_t> d_off(nm + 1), d_out_off(nm + 1);
    DevBuf<uint32_t> d_crc(nm), d_sizes(nm);
    DevBuf<uint8_t> slots((uint64_t)nm * BGZF_SLOT_BYTES);
    HIP_CHECK(hipMemcpyAsync(d_off.p, off.data(), (nm + 1) * 8, hipMemcpyHostToDevice, st));
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, st));
    launch_crc32_store(d_payload, d_off.p, nm, d_crc.p, st);
    launch_bgzf_deflate(d_payload, d_off.p, nm, d_crc.p, slots.p, BGZF_SLOT_BYTES, d_sizes.p, st);
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
    if (n_members) *n_members += nm;
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
static void up_fixed(const ArrowArray* a, DevCol* d, hipStream_t st) {
  const size_t bytes = (size_t)(a->offset + a->length) * 4;
  d->values.alloc(bytes + 4);
  if (bytes) HIP_CHECK(hipMemcpyAsync(d->values.p, a->buffers[1], bytes, hipMemcpyHostToDevice, st));
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

int bioscan_bam_writer_write(bioscan_bam_writer* bw, const struct ArrowArray* batch, const struct ArrowSchema* schema) {
  W_BEGIN
  Writer& w = bw->w;
  if (w.finished) throw Error("BAM writer is finished");
  const int64_t n = batch->length;
  if (n == 0) return 0;
  if (batch->offset != 0) throw Error("Failed to write BAM records: sliced struct arrays are not supported");
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
    if (a->length != n || a->offset != name.a->offset) throw Error("Failed to write BAM records: columns differ in length or offset");
  // chrom / mate_chrom -> reference ids (sam_record_serializer.rs:145-151, 176-186)
  std::vector<int32_t> refid((size_t)n), mrefid((size_t)n);
  {
    auto is_valid = [](const ArrowArray* a, int64_t i) {
      const uint8_t* v = (const uint8_t*)a->buffers[0];
      const int64_t j = i + a->offset;
      return a->null_count == 0 || !v || ((v[j >> 3] >> (j & 7)) & 1);
    };
    auto str_at = [](const ArrowArray* a, int64_t i) {
      const int32_t* off = (const int32_t*)a->buffers[1];
      const char* d = (const char*)a->buffers[2];
      const int64_t j = i + a->offset;
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
  c.offset = name.a->offset;
  c.zero_based = w.zero_based ? 1 : 0;
  c.cigar_binary = cigar_binary ? 1 : 0;
  c.name_valid = up_valid(name.a, &d_name, st); up_var(name.a, &d_name, st); c.name_off = d_name.off.p; c.name = d_name.values.p;
  up_var(cigar.a, &d_cigar, st); c.cigar_off = d_cigar.off.p; c.cigar = d_cigar.values.p;
  up_var(seq.a, &d_seq, st); c.seq_off = d_seq.off.p; c.seq = d_seq.values.p;
  up_var(qual.a, &d_qual, st); c.qual_off = d_qual.off.p; c.qual = d_qual.values.p;
  c.start_valid = up_valid(start.a, &d_start, st); up_fixed(start.a, &d_start, st); c.start = (const uint32_t*)d_start.values.p;
  up_fixed(flags.a, &d_flags, st); c.flags = (const uint32_t*)d_flags.values.p;
  up_fixed(mapq.a, &d_mapq, st); c.mapq = (const uint32_t*)d_mapq.values.p;
  c.mate_start_valid = up_valid(mstart.a, &d_mstart, st); up_fixed(mstart.a, &d_mstart, st); c.mate_start = (const uint32_t*)d_mstart.values.p;
  up_fixed(tlen.a, &d_tlen, st); c.tlen = (const int32_t*)d_tlen.values.p;
  HIP_CHECK(hipMemcpyAsync(d_refid.p, refid.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(d_mrefid.p, mrefid.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  c.refid = d_refid.p; c.mate_refid = d_mrefid.p;
  // sizes -> offsets -> bytes
  DevBuf<uint32_t> rec_bytes((size_t)n), err(1);
  DevBuf<uint64_t> rec_off((size_t)n + 1), tmp(scan_tmp_elems((uint64_t)n));
  HIP_CHECK(hipMemsetAsync(err.p, 0, 4, st));
  launch_ser_sizes(c, (uint64_t)n, rec_bytes.p, err.p, st);
  launch_exclusive_scan_u32_to_u64(rec_bytes.p, rec_off.p, (uint64_t)n, tmp.p, st);
  uint64_t total = 0;
  uint32_t e = 0;
  HIP_CHECK(hipMemcpyAsync(&total, rec_off.p + n, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&e, err.p, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  throw_ser_err(e);
  w.reserve(total);
  launch_ser_write(c, (uint64_t)n, rec_off.p, w.d_stream.p + w.stream_len, err.p, st);
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
