// vcf_host.cpp -- see vcf_host.h
#include "vcf_host.h"

#include <string.h>

#include <algorithm>
#include <map>
#include <set>

namespace bioscan {

static void jstr(std::string& o, const std::string& s) {
  o += '"';
  for (unsigned char c : s) {
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      default:
        if (c < 0x20) {
          char b[8];
          snprintf(b, sizeof b, "\\u%04x", c);
          o += b;
        } else o += (char)c;
    }
  }
  o += '"';
}
std::string json_string_array(const std::vector<std::string>& v) {
  std::string o = "[";
  for (size_t i = 0; i < v.size(); i++) {
    if (i) o += ',';
    jstr(o, v[i]);
  }
  return o + "]";
}

const VcfFieldDefn* VcfHeader::info(const std::string& id) const {
  for (auto& d : infos) if (d.id == id) return &d;
  return nullptr;
}
const VcfFieldDefn* VcfHeader::format(const std::string& id) const {
  for (auto& d : formats) if (d.id == id) return &d;
  return nullptr;
}

// `ID=x,Number=1,Description="a, b"` -> ordered (key, value) list; quoted values may hold commas and \-escapes
static std::vector<std::pair<std::string, std::string>> parse_struct_fields(const std::string& body) {
  std::vector<std::pair<std::string, std::string>> out;
  size_t i = 0, n = body.size();
  while (i < n) {
    size_t j = body.find('=', i);
    if (j == std::string::npos) break;
    std::string key = body.substr(i, j - i);
    while (!key.empty() && key.front() == ' ') key.erase(key.begin());
    while (!key.empty() && key.back() == ' ') key.pop_back();
    i = j + 1;
    std::string val;
    if (i < n && body[i] == '"') {
      i++;
      while (i < n && body[i] != '"') {
        if (body[i] == '\\' && i + 1 < n) i++;
        val += body[i++];
      }
      i++;
    } else {
      size_t k = body.find(',', i);
      if (k == std::string::npos) k = n;
      val = body.substr(i, k - i);
      i = k;
    }
    out.emplace_back(key, val);
    if (i < n && body[i] == ',') i++;
  }
  return out;
}
static std::string get_kv(const std::vector<std::pair<std::string, std::string>>& kv, const char* k, const char* dflt, bool* found = nullptr) {
  for (auto& p : kv) if (p.first == k) { if (found) *found = true; return p.second; }
  if (found) *found = false;
  return dflt;
}

bool parse_vcf_header(const uint8_t* u, size_t n, bool at_eof, VcfHeader* out, std::string* err) {
  *out = VcfHeader();
  err->clear();
  size_t pos = 0;
  bool saw_chrom = false;
  while (pos < n && u[pos] == '#') {
    const uint8_t* e = (const uint8_t*)memchr(u + pos, '\n', n - pos);
    if (!e && !at_eof) return false;  // incomplete line
    size_t le = e ? (size_t)(e - u) : n;
    size_t nxt = e ? le + 1 : n;
    size_t l2 = le;
    if (l2 > pos && u[l2 - 1] == '\r') l2--;
    std::string line((const char*)u + pos, l2 - pos);
    pos = nxt;
    if (line.size() >= 2 && line[1] == '#') {
      size_t eq = line.find('=');
      if (eq == std::string::npos) continue;
      std::string k = line.substr(2, eq - 2), v = line.substr(eq + 1);
      if (k == "fileformat") out->file_format = v;
      else if (v.size() >= 2 && v.front() == '<' && v.back() == '>') {
        auto kv = parse_struct_fields(v.substr(1, v.size() - 2));
        std::string id = get_kv(kv, "ID", "");
        if (k == "INFO" || k == "FORMAT") {
          VcfFieldDefn d{id, get_kv(kv, "Number", "."), get_kv(kv, "Type", "String"), get_kv(kv, "Description", "")};
          auto& vec = k == "INFO" ? out->infos : out->formats;
          bool dup = false;
          for (auto& x : vec) if (x.id == id) { x = d; dup = true; }
          if (!dup) vec.push_back(d);
        } else if (k == "FILTER") out->filters.emplace_back(id, get_kv(kv, "Description", ""));
        else if (k == "contig") {
          bool f = false;
          std::string ln = get_kv(kv, "length", "", &f);
          int64_t L = -1;
          if (f && !ln.empty() && ln.find_first_not_of("0123456789") == std::string::npos) L = (int64_t)strtoull(ln.c_str(), nullptr, 10);
          out->contigs.emplace_back(id, L);
        } else if (k == "ALT") out->alts.emplace_back(id, get_kv(kv, "Description", ""));
      }
    } else {
      // #CHROM line
      size_t p = 0;
      int col = 0;
      while (p <= line.size()) {
        size_t t = line.find('\t', p);
        if (t == std::string::npos) t = line.size();
        if (col >= 9) out->samples.push_back(line.substr(p, t - p));
        col++;
        p = t + 1;
      }
      saw_chrom = true;
      break;
    }
  }
  if (!saw_chrom) {
    if (pos >= n && !at_eof) return false;  // ran out of bytes inside the header
    if (pos < n && u[pos] != '#' && pos == 0) { *err = "missing VCF header"; return false; }
    if (!at_eof && pos >= n) return false;
    *err = "VCF header has no #CHROM line";
    return false;
  }
  out->header_bytes = pos;
  return true;
}

const char* vkind_format(VKind k) {
  switch (k) {
    case VK_INT32: return "i";
    case VK_UINT32: return "I";
    case VK_FLOAT32: return "f";
    case VK_FLOAT64: return "g";
    case VK_BOOL: return "b";
    case VK_UTF8: return "u";
    case VK_LIST: return "+l";
    default: return "+s";
  }
}

static VKind scalar_kind(const std::string& ty, bool format) {
  if (ty == "Integer") return VK_INT32;
  if (ty == "Float") return VK_FLOAT32;
  if (ty == "Flag" && !format) return VK_BOOL;
  return VK_UTF8;
}
VcfValueType info_value_type(const VcfHeader& h, const std::string& tag) {
  VcfValueType t;
  const VcfFieldDefn* d = h.info(tag);
  if (!d) return t;
  t.scalar = scalar_kind(d->type, false);
  t.is_list = !(d->number == "0" || d->number == "1");
  return t;
}
VcfValueType format_value_type(const VcfHeader& h, const std::string& tag) {
  VcfValueType t;
  if (tag == "GT") return t;
  const VcfFieldDefn* d = h.format(tag);
  if (!d) return t;
  t.scalar = scalar_kind(d->type, true);
  t.is_list = !(d->number == "0" || d->number == "1");
  return t;
}

static VField leaf_or_list(const std::string& name, VcfValueType t, bool nullable) {
  VField f;
  f.name = name;
  f.nullable = nullable;
  if (t.is_list) {
    f.kind = VK_LIST;
    VField item;
    item.name = "item";
    item.kind = t.scalar;
    item.nullable = true;
    f.children.push_back(item);
  } else f.kind = t.scalar;
  return f;
}

// storage.rs:643-661
static std::string resolve_single_sample_format_column_name(const std::set<std::string>& used, const std::string& id) {
  if (!used.count(id)) return id;
  std::string cand = "fmt_" + id;
  if (used.count(cand)) cand = "format_" + id;
  int k = 2;
  while (used.count(cand)) cand = "format_" + id + "_" + std::to_string(k++);
  return cand;
}

std::string determine_vcf_schema(const VcfHeader& h, const std::vector<std::string>* info_fields,
                                 const std::vector<std::string>* format_fields, const std::vector<std::string>* samples,
                                 bool zero_based, const std::vector<std::string>* index_names, VcfSchema* out) {
  *out = VcfSchema();
  auto core = [&](const char* n, VKind k, bool nullable) {
    VField f;
    f.name = n; f.kind = k; f.nullable = nullable;
    out->fields.push_back(f);
  };
  core("chrom", VK_UTF8, false); core("start", VK_UINT32, false); core("end", VK_UINT32, false); core("id", VK_UTF8, true);
  core("ref", VK_UTF8, false); core("alt", VK_UTF8, false); core("qual", VK_FLOAT64, true); core("filter", VK_UTF8, true);
  if (info_fields) out->info_fields = *info_fields;
  else for (auto& d : h.infos) out->info_fields.push_back(d.id);
  if (format_fields) out->format_fields = *format_fields;
  else for (auto& d : h.formats) out->format_fields.push_back(d.id);
  // sample selection (genotype.rs resolve_samples, MissingSamplePolicy::Ignore for text VCF)
  {
    std::map<std::string, int> src;
    for (size_t i = 0; i < h.samples.size(); i++) {
      if (src.count(h.samples[i])) return "source sample name is ambiguous: " + h.samples[i];
      src[h.samples[i]] = (int)i;
    }
    if (!samples) {
      out->samples = h.samples;
      for (size_t i = 0; i < h.samples.size(); i++) out->sample_header_index.push_back((int32_t)i);
    } else {
      std::set<std::string> seen;
      for (auto& s : *samples) {
        if (!seen.insert(s).second) continue;
        auto it = src.find(s);
        if (it == src.end()) continue;
        out->samples.push_back(s);
        out->sample_header_index.push_back(it->second);
      }
    }
  }
  for (auto& tag : out->info_fields) {
    const VcfFieldDefn* d = h.info(tag);
    if (!d) return "INFO field '" + tag + "' is not defined in the VCF header";  // the reference unwraps (panics)
    VField f = leaf_or_list(tag, info_value_type(h, tag), d->type != "Flag");
    f.metadata = {{"bio.vcf.field.description", d->description}, {"bio.vcf.field.type", d->type},
                  {"bio.vcf.field.number", d->number}, {"bio.vcf.field.field_type", "INFO"}};
    out->fields.push_back(f);
  }
  auto fmt_meta = [&](const std::string& tag) {
    std::vector<std::pair<std::string, std::string>> md;
    if (const VcfFieldDefn* d = h.format(tag)) {
      md.emplace_back("bio.vcf.field.description", d->description);
      md.emplace_back("bio.vcf.field.type", d->type);
      md.emplace_back("bio.vcf.field.number", d->number);
    }
    md.emplace_back("bio.vcf.field.field_type", "FORMAT");
    md.emplace_back("bio.vcf.field.format_id", tag);
    return md;
  };
  out->multi = h.samples.size() > 1;
  if (!out->format_fields.empty() && !out->samples.empty()) {
    out->has_format = true;
    if (h.samples.size() == 1) {
      std::set<std::string> used;
      for (auto& f : out->fields) used.insert(f.name);
      for (auto& tag : out->format_fields) {
        std::string name = resolve_single_sample_format_column_name(used, tag);
        used.insert(name);
        VField f = leaf_or_list(name, format_value_type(h, tag), true);
        f.metadata = fmt_meta(tag);
        out->fields.push_back(f);
      }
    } else {
      VField g;
      g.name = "genotypes";
      g.kind = VK_STRUCT;
      g.nullable = true;
      for (auto& tag : out->format_fields) {
        VField l;
        l.name = tag;
        l.kind = VK_LIST;
        l.nullable = true;
        l.metadata = fmt_meta(tag);
        VField item = leaf_or_list("item", format_value_type(h, tag), true);
        l.children.push_back(item);
        g.children.push_back(l);
      }
      std::string names = json_string_array(out->samples);
      g.metadata = {{"bio.genotype.sample_names", names}, {"bio.vcf.genotypes.sample_names", names}};
      out->fields.push_back(g);
    }
  }
  // schema metadata
  auto& md = out->metadata;
  md.emplace_back("bio.coordinate_system_zero_based", zero_based ? "true" : "false");
  md.emplace_back("bio.vcf.file_format", h.file_format);
  {
    std::string o = "[";
    for (size_t i = 0; i < h.filters.size(); i++) {
      if (i) o += ',';
      o += "{\"id\":"; jstr(o, h.filters[i].first); o += ",\"description\":"; jstr(o, h.filters[i].second); o += '}';
    }
    md.emplace_back("bio.vcf.filters", o + "]");
  }
  {
    std::string o = "[";
    for (size_t i = 0; i < h.contigs.size(); i++) {
      if (i) o += ',';
      o += "{\"id\":"; jstr(o, h.contigs[i].first);
      if (h.contigs[i].second >= 0) o += ",\"length\":" + std::to_string(h.contigs[i].second);
      o += '}';
    }
    md.emplace_back("bio.vcf.contigs", o + "]");
  }
  {
    std::string o = "[";
    for (size_t i = 0; i < h.alts.size(); i++) {
      if (i) o += ',';
      o += "{\"id\":"; jstr(o, h.alts[i].first); o += ",\"description\":"; jstr(o, h.alts[i].second); o += '}';
    }
    md.emplace_back("bio.vcf.alternative_alleles", o + "]");
  }
  md.emplace_back("bio.vcf.samples", json_string_array(out->samples));
  {
    std::map<std::string, const VcfFieldDefn*> m;  // BTreeMap: sorted by tag
    for (auto& tag : out->format_fields) if (const VcfFieldDefn* d = h.format(tag)) m[tag] = d;
    std::string o = "{";
    bool first = true;
    for (auto& kv : m) {
      if (!first) o += ',';
      first = false;
      jstr(o, kv.first);
      o += ":{\"number\":"; jstr(o, kv.second->number);
      o += ",\"type\":"; jstr(o, kv.second->type);
      o += ",\"description\":"; jstr(o, kv.second->description);
      o += '}';
    }
    md.emplace_back("bio.vcf.format_fields", o + "}");
  }
  if (index_names && !index_names->empty()) md.emplace_back("bio.vcf.contigs.indexed", json_string_array(*index_names));
  return "";
}

// ---- tabix ---------------------------------------------------------------------------------------
bool parse_csi_names(const std::vector<uint8_t>& d, std::vector<std::string>* names, std::string* err) {
  names->clear();
  if (d.size() < 16 || memcmp(d.data(), "CSI\1", 4) != 0) { *err = "invalid CSI header"; return false; }
  int32_t l_aux;
  memcpy(&l_aux, d.data() + 12, 4);
  if (l_aux < 0 || 16 + (size_t)l_aux > d.size()) { *err = "truncated CSI index"; return false; }
  if (l_aux < 28) return true;  // no tabix-style aux block: no names
  int32_t l_nm;
  memcpy(&l_nm, d.data() + 16 + 24, 4);
  if (l_nm < 0 || 28 + (size_t)l_nm > (size_t)l_aux) { *err = "truncated CSI index"; return false; }
  size_t p = 16 + 28, e = p + (size_t)l_nm;
  while (p < e) {
    const uint8_t* z = (const uint8_t*)memchr(d.data() + p, 0, e - p);
    size_t q = z ? (size_t)(z - d.data()) : e;
    names->emplace_back((const char*)d.data() + p, q - p);
    p = q + 1;
  }
  return true;
}

bool parse_tbi(const std::vector<uint8_t>& d, Tbi* out, std::string* err) {
  *out = Tbi();
  if (d.size() < 36 || memcmp(d.data(), "TBI\1", 4) != 0) { *err = "invalid tabix header"; return false; }
  int32_t h[8];
  memcpy(h, d.data() + 4, 32);
  const int32_t n_ref = h[0], l_nm = h[7];
  out->format = h[1]; out->col_seq = h[2]; out->col_beg = h[3]; out->col_end = h[4]; out->meta = h[5]; out->skip = h[6];
  size_t o = 36;
  if (n_ref < 0 || l_nm < 0 || o + (size_t)l_nm > d.size()) { *err = "truncated tabix index"; return false; }
  {
    size_t p = o, e = o + (size_t)l_nm;
    while (p < e) {
      const uint8_t* z = (const uint8_t*)memchr(d.data() + p, 0, e - p);
      size_t q = z ? (size_t)(z - d.data()) : e;
      out->names.emplace_back((const char*)d.data() + p, q - p);
      p = q + 1;
    }
  }
  o += (size_t)l_nm;
  auto need = [&](size_t k) { return o + k <= d.size(); };
  for (int32_t r = 0; r < n_ref; r++) {
    BaiRef ref;
    if (!need(4)) { *err = "truncated tabix index"; return false; }
    int32_t n_bin;
    memcpy(&n_bin, d.data() + o, 4);
    o += 4;
    for (int32_t b = 0; b < n_bin; b++) {
      if (!need(8)) { *err = "truncated tabix index"; return false; }
      uint32_t bin;
      int32_t n_chunk;
      memcpy(&bin, d.data() + o, 4);
      memcpy(&n_chunk, d.data() + o + 4, 4);
      o += 8;
      if (n_chunk < 0 || !need((size_t)n_chunk * 16)) { *err = "truncated tabix index"; return false; }
      std::vector<std::pair<uint64_t, uint64_t>> chunks((size_t)n_chunk);
      for (int32_t c = 0; c < n_chunk; c++) {
        memcpy(&chunks[c].first, d.data() + o, 8);
        memcpy(&chunks[c].second, d.data() + o + 8, 8);
        o += 16;
      }
      if (bin == 37450) {
        if (n_chunk == 2) {
          ref.has_meta = true;
          ref.ref_beg = chunks[0].first; ref.ref_end = chunks[0].second;
          ref.n_mapped = chunks[1].first; ref.n_unmapped = chunks[1].second;
        }
      } else ref.bins[bin] = std::move(chunks);
    }
    if (!need(4)) { *err = "truncated tabix index"; return false; }
    int32_t n_intv;
    memcpy(&n_intv, d.data() + o, 4);
    o += 4;
    if (n_intv < 0 || !need((size_t)n_intv * 8)) { *err = "truncated tabix index"; return false; }
    ref.intervals.resize((size_t)n_intv);
    if (n_intv) memcpy(ref.intervals.data(), d.data() + o, (size_t)n_intv * 8);
    o += (size_t)n_intv * 8;
    out->idx.refs.push_back(std::move(ref));
  }
  if (need(8)) {
    out->idx.has_no_coor = true;
    memcpy(&out->idx.n_no_coor, d.data() + o, 8);
  }
  return true;
}

std::vector<RegionSizeEstimate> estimate_sizes_from_tbi(const Tbi* tbi, const std::vector<GenomicRegion>& regions,
                                                        const std::vector<std::string>& contig_names,
                                                        const std::vector<uint64_t>& contig_lengths) {
  std::vector<RegionSizeEstimate> out;
  static const uint64_t LEVEL_OFF[6] = {0, 1, 9, 73, 585, 4681};
  static const uint64_t LEVEL_SPAN[6] = {1ull << 29, 1ull << 26, 1ull << 23, 1ull << 20, 1ull << 17, 1ull << 14};
  for (auto& r : regions) {
    RegionSizeEstimate e;
    e.region = r;
    if (!tbi) { e.estimated_bytes = 1; out.push_back(e); continue; }
    long idx = -1;
    for (size_t i = 0; i < tbi->names.size(); i++) if (tbi->names[i] == r.chrom) idx = (long)i;
    if (idx < 0) {
      for (size_t i = 0; i < contig_names.size(); i++) if (contig_names[i] == r.chrom) idx = (long)i;
      if (idx >= 0 && (size_t)idx >= tbi->idx.refs.size()) idx = -1;
    }
    const BaiRef* ref = (idx >= 0 && (size_t)idx < tbi->idx.refs.size()) ? &tbi->idx.refs[idx] : nullptr;
    if (ref) {
      uint64_t mn = ~0ull, mx = 0;
      for (auto& b : ref->bins)
        for (auto& c : b.second) {
          mn = std::min(mn, c.first >> 16);
          mx = std::max(mx, c.second >> 16);
        }
      e.estimated_bytes = mx > mn ? mx - mn : 0;
      for (auto& b : ref->bins)
        if (b.first >= 4681 && b.first <= 37448) e.nonempty_bin_positions.push_back((uint64_t)(b.first - 4681) * 16384 + 1);
      std::sort(e.nonempty_bin_positions.begin(), e.nonempty_bin_positions.end());
    } else e.estimated_bytes = 1;
    // contig length: header length by name, else from the leaf bins, else from any bin level
    bool have = false;
    uint64_t L = 0;
    for (size_t i = 0; i < contig_names.size(); i++)
      if (contig_names[i] == r.chrom && i < contig_lengths.size() && contig_lengths[i] > 0) { have = true; L = contig_lengths[i]; }
    if (!have && !e.nonempty_bin_positions.empty()) { have = true; L = e.nonempty_bin_positions.back() + 16384 - 1; }
    if (!have && ref) {
      for (auto& b : ref->bins) {
        for (int li = 5; li >= 0; li--) {
          const uint64_t nxt = li < 5 ? LEVEL_OFF[li + 1] : 37449;
          if (b.first >= LEVEL_OFF[li] && b.first < nxt) {
            const uint64_t v = (b.first - LEVEL_OFF[li] + 1) * LEVEL_SPAN[li];
            L = have ? std::max(L, v) : v;
            have = true;
            break;
          }
        }
      }
    }
    e.has_contig_length = have;
    e.contig_length = L;
    e.unmapped_count = 0;
    e.leaf_bin_span = 16384;
    out.push_back(std::move(e));
  }
  return out;
}

uint64_t choose_effective_batch_size(uint64_t requested, bool any_format, uint64_t n_format_fields, uint64_t n_selected,
                                     uint64_t n_source) {
  if (!any_format || n_source <= 1 || n_selected == 0) return std::max<uint64_t>(requested, 1);
  const uint64_t ffc = std::max<uint64_t>(n_format_fields, 1);
  const uint64_t cells = n_selected * ffc;
  if (cells == 0) return std::max<uint64_t>(requested, 1);
  const uint64_t bytes_per_sample = 16 + ffc * 8;
  const uint64_t bytes_per_row = std::max<uint64_t>(n_selected * bytes_per_sample, 1);
  const uint64_t by_cells = std::max<uint64_t>(100000 / cells, 1);
  const uint64_t by_bytes = std::max<uint64_t>(8000000 / bytes_per_row, 1);
  uint64_t eff = std::min(requested, std::min(by_cells, by_bytes));
  if (by_cells >= 8 && by_bytes >= 8 && requested > 8) eff = std::max<uint64_t>(eff, 8);
  return std::max<uint64_t>(eff, 1);
}

}  // namespace bioscan
