// bam_host.cpp -- see bam_host.h.  Host-side planning only; no device code.
#include "bam_host.h"

#include <algorithm>
#include <cstring>
#include <sstream>

namespace bioscan {

static inline uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline int32_t rd_i32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// =================================================================================================
// BAM header (SAM spec 4.2)
// =================================================================================================
bool parse_bam_header(const uint8_t* u, size_t n, BamHeader* out, std::string* err) {
  if (n < 12) return false;
  if (memcmp(u, "BAM\1", 4) != 0) { *err = "not a BAM file (bad magic)"; return false; }
  int32_t l_text = rd_i32(u + 4);
  if (l_text < 0) { *err = "invalid BAM header text length"; return false; }
  size_t o = 8 + (size_t)l_text;
  if (n < o + 4) return false;
  int32_t n_ref = rd_i32(u + o);
  if (n_ref < 0) { *err = "invalid BAM reference count"; return false; }
  o += 4;
  std::vector<std::string> names;
  std::vector<int64_t> lens;
  for (int32_t i = 0; i < n_ref; i++) {
    if (n < o + 4) return false;
    int32_t l_name = rd_i32(u + o);
    if (l_name < 1) { *err = "invalid BAM reference name length"; return false; }
    if (n < o + 4 + (size_t)l_name + 4) return false;
    names.emplace_back((const char*)u + o + 4, (size_t)l_name - 1);
    lens.push_back(rd_i32(u + o + 4 + l_name));
    o += 8 + (size_t)l_name;
  }
  size_t tl = strnlen((const char*)u + 8, (size_t)l_text);
  out->text.assign((const char*)u + 8, tl);
  out->ref_names = std::move(names);
  out->ref_lengths = std::move(lens);
  out->first_record_offset = o;
  err->clear();
  return true;
}

// ---- tiny JSON writer (serde_json compact form) -------------------------------------------------
static void json_str(std::string& o, const std::string& s) {
  o.push_back('"');
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
        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back((char)c);
    }
  }
  o.push_back('"');
}
typedef std::vector<std::pair<std::string, std::string>> KV;
static void json_obj_fields(std::string& o, const KV& kv, bool& first) {
  for (auto& p : kv) {
    if (!first) o.push_back(',');
    first = false;
    json_str(o, p.first);
    o.push_back(':');
    json_str(o, p.second);
  }
}
static const std::string* kv_get(const KV& kv, const char* k) {
  for (auto& p : kv) if (p.first == k) return &p.second;
  return nullptr;
}

std::vector<std::pair<std::string, std::string>> extract_header_metadata(const BamHeader& h) {
  KV md;
  std::vector<KV> sq, rg, pg;
  std::vector<std::string> co;
  KV hd;
  bool have_hd = false;
  size_t p = 0;
  const std::string& t = h.text;
  while (p < t.size()) {
    size_t e = t.find('\n', p);
    if (e == std::string::npos) e = t.size();
    std::string line = t.substr(p, e - p);
    p = e + 1;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.size() < 3 || line[0] != '@') continue;
    std::string kind = line.substr(1, 2);
    if (kind == "CO") { co.push_back(line.size() > 4 ? line.substr(4) : std::string()); continue; }
    KV kv;
    size_t q = line.find('\t');
    while (q != std::string::npos) {
      size_t r = line.find('\t', q + 1);
      std::string f = line.substr(q + 1, r == std::string::npos ? std::string::npos : r - q - 1);
      if (f.size() >= 3 && f[2] == ':') kv.emplace_back(f.substr(0, 2), f.substr(3));
      q = r;
    }
    if (kind == "HD") { hd = kv; have_hd = true; }
    else if (kind == "SQ") sq.push_back(kv);
    else if (kind == "RG") rg.push_back(kv);
    else if (kind == "PG") pg.push_back(kv);
  }
  if (have_hd) {
    if (auto v = kv_get(hd, "VN")) md.emplace_back("bio.bam.file_format_version", *v);
    if (auto v = kv_get(hd, "SO")) md.emplace_back("bio.bam.sort_order", *v);
    if (auto v = kv_get(hd, "GO")) md.emplace_back("bio.bam.group_order", *v);
    if (auto v = kv_get(hd, "SS")) md.emplace_back("bio.bam.subsort_order", *v);
  }
  auto emit_list = [&](const char* key, const std::vector<KV>& recs, const char* id_tag, const char* id_name,
                       std::vector<std::pair<const char*, const char*>> named, bool length_field) {
    if (recs.empty()) return;
    std::string o = "[";
    bool firstrec = true;
    for (auto& kv : recs) {
      if (!firstrec) o.push_back(',');
      firstrec = false;
      o.push_back('{');
      bool first = true;
      const std::string* idv = kv_get(kv, id_tag);
      KV idf{{id_name, idv ? *idv : std::string()}};
      json_obj_fields(o, idf, first);
      if (length_field) {
        const std::string* ln = kv_get(kv, "LN");
        o += ",\"length\":";
        o += ln ? *ln : "0";
      }
      for (auto& nm : named) {
        if (auto v = kv_get(kv, nm.first)) {
          KV one{{nm.second, *v}};
          json_obj_fields(o, one, first);
        }
      }
      KV other;
      for (auto& f : kv) {
        bool skip = f.first == id_tag || (length_field && f.first == "LN");
        for (auto& nm : named) if (f.first == nm.first) skip = true;
        if (!skip) other.push_back(f);
      }
      if (!other.empty()) {
        o += ",\"other_fields\":{";
        bool f2 = true;
        json_obj_fields(o, other, f2);
        o.push_back('}');
      }
      o.push_back('}');
    }
    o.push_back(']');
    md.emplace_back(key, o);
  };
  if (sq.empty() && !h.ref_names.empty()) {
    // no @SQ lines: noodles adopts the binary reference list
    for (size_t i = 0; i < h.ref_names.size(); i++)
      sq.push_back(KV{{"SN", h.ref_names[i]}, {"LN", std::to_string(h.ref_lengths[i])}});
  }
  emit_list("bio.bam.reference_sequences", sq, "SN", "name", {}, true);
  emit_list("bio.bam.read_groups", rg, "ID", "id", {{"SM", "sample"}, {"PL", "platform"}, {"LB", "library"}, {"DS", "description"}}, false);
  emit_list("bio.bam.program_info", pg, "ID", "id", {{"PN", "name"}, {"VN", "version"}, {"CL", "command_line"}}, false);
  if (!co.empty()) {
    std::string o = "[";
    for (size_t i = 0; i < co.size(); i++) {
      if (i) o.push_back(',');
      json_str(o, co[i]);
    }
    o.push_back(']');
    md.emplace_back("bio.bam.comments", o);
  }
  return md;
}

// =================================================================================================
// tag registry
// =================================================================================================
static const std::map<std::string, TagDef>& registry() {
  static const std::map<std::string, TagDef> r = {
#include "tag_registry_table.inc"
  };
  return r;
}
const TagDef* known_tag(const std::string& tag) {
  auto& r = registry();
  auto it = r.find(tag);
  return it == r.end() ? nullptr : &it->second;
}
ArrowKind sam_tag_type_to_arrow(char c) {
  switch (c) {
    case 'c': case 's': case 'i': return AK_INT32;
    case 'C': case 'S': case 'I': return AK_UINT32;
    case 'f': return AK_FLOAT32;
    case 'B': return AK_LIST_INT32;
    default: return AK_UTF8;
  }
}
bool sam_array_subtype_to_arrow(char c, ArrowKind* out) {
  switch (c) {
    case 'c': *out = AK_LIST_INT8; return true;
    case 'C': *out = AK_LIST_UINT8; return true;
    case 's': *out = AK_LIST_INT16; return true;
    case 'S': *out = AK_LIST_UINT16; return true;
    case 'i': *out = AK_LIST_INT32; return true;
    case 'I': *out = AK_LIST_UINT32; return true;
    case 'f': *out = AK_LIST_FLOAT32; return true;
    default: return false;
  }
}
std::string parse_tag_type_hints(const std::vector<std::string>& hints, std::map<std::string, std::pair<char, ArrowKind>>* out) {
  for (auto& h : hints) {
    std::vector<std::string> parts;
    size_t p = 0;
    for (;;) {
      size_t e = h.find(':', p);
      parts.push_back(h.substr(p, e == std::string::npos ? std::string::npos : e - p));
      if (e == std::string::npos) break;
      p = e + 1;
    }
    if (parts.size() == 2) {
      if (parts[1].size() != 1) return "Invalid tag type hint '" + h + "': TYPE must be a single character";
      char t = parts[1][0];
      if (t == 'B') return "Invalid tag type hint '" + h + "': array type 'B' requires a subtype. Use 'TAG:B:c|C|s|S|i|I|f'";
      if (!strchr("AcCsSiIfZH", t)) return "Invalid tag type hint '" + h + "': unsupported SAM type '" + std::string(1, t) + "'. Supported types: A, c, C, s, S, i, I, f, Z, H";
      (*out)[parts[0]] = {t, sam_tag_type_to_arrow(t)};
    } else if (parts.size() == 3 && parts[1] == "B") {
      if (parts[2].size() != 1) return "Invalid tag type hint '" + h + "': array subtype must be a single character";
      ArrowKind k;
      if (!sam_array_subtype_to_arrow(parts[2][0], &k)) return "Invalid tag type hint '" + h + "': unsupported array subtype";
      (*out)[parts[0]] = {'B', k};
    } else {
      return "Invalid tag type hint '" + h + "': expected 'TAG:TYPE' or 'TAG:B:SUBTYPE' format";
    }
  }
  return "";
}
std::string format_sam_tag_type(char sam_type, ArrowKind k) {
  if (sam_type == 'B') {
    const char* sub = "";
    switch (k) {
      case AK_LIST_INT8: sub = "c"; break;
      case AK_LIST_UINT8: sub = "C"; break;
      case AK_LIST_INT16: sub = "s"; break;
      case AK_LIST_UINT16: sub = "S"; break;
      case AK_LIST_INT32: sub = "i"; break;
      case AK_LIST_UINT32: sub = "I"; break;
      case AK_LIST_FLOAT32: sub = "f"; break;
      default: return "B";
    }
    return std::string("B:") + sub;
  }
  return std::string(1, sam_type);
}

// =================================================================================================
// BAI (SAM spec 5.2)
// =================================================================================================
bool parse_bai(const std::vector<uint8_t>& d, Bai* out, std::string* err) {
  const size_t n = d.size();
  auto need = [&](size_t o, size_t k) { return o + k <= n; };
  if (n < 8 || memcmp(d.data(), "BAI\1", 4) != 0) { *err = "invalid BAI header"; return false; }
  int32_t n_ref = rd_i32(&d[4]);
  size_t o = 8;
  out->refs.clear();
  for (int32_t r = 0; r < n_ref; r++) {
    BaiRef ref;
    if (!need(o, 4)) { *err = "truncated BAI"; return false; }
    int32_t n_bin = rd_i32(&d[o]); o += 4;
    for (int32_t b = 0; b < n_bin; b++) {
      if (!need(o, 8)) { *err = "truncated BAI"; return false; }
      uint32_t bin = rd_u32(&d[o]);
      int32_t n_chunk = rd_i32(&d[o + 4]);
      o += 8;
      if (n_chunk < 0 || !need(o, 16 * (size_t)n_chunk)) { *err = "truncated BAI"; return false; }
      std::vector<std::pair<uint64_t, uint64_t>> ch;
      for (int32_t c = 0; c < n_chunk; c++) ch.emplace_back(rd_u64(&d[o + 16 * c]), rd_u64(&d[o + 16 * c + 8]));
      o += 16 * (size_t)n_chunk;
      if (bin == 37450) {
        if (n_chunk == 2) {
          ref.has_meta = true;
          ref.ref_beg = ch[0].first; ref.ref_end = ch[0].second;
          ref.n_mapped = ch[1].first; ref.n_unmapped = ch[1].second;
        }
      } else {
        ref.bins[bin] = std::move(ch);
      }
    }
    if (!need(o, 4)) { *err = "truncated BAI"; return false; }
    int32_t n_intv = rd_i32(&d[o]); o += 4;
    if (n_intv < 0 || !need(o, 8 * (size_t)n_intv)) { *err = "truncated BAI"; return false; }
    for (int32_t i = 0; i < n_intv; i++) ref.intervals.push_back(rd_u64(&d[o + 8 * i]));
    o += 8 * (size_t)n_intv;
    out->refs.push_back(std::move(ref));
  }
  out->has_no_coor = need(o, 8);
  out->n_no_coor = out->has_no_coor ? rd_u64(&d[o]) : 0;
  return true;
}

std::vector<std::pair<uint64_t, uint64_t>> bai_query_chunks(const Bai& bai, size_t ref_idx, bool has_start, uint64_t start1,
                                                            bool has_end, uint64_t end1) {
  const uint64_t MAXP = 1ull << 29;
  uint64_t s = has_start ? start1 : 1, e = has_end ? end1 : MAXP;
  if (e > MAXP) e = MAXP;
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  if (ref_idx >= bai.refs.size() || s == 0 || s > MAXP) return chunks;
  const BaiRef& ref = bai.refs[ref_idx];
  uint64_t beg0 = s - 1, end0 = e - 1;  // inclusive 0-based
  auto add = [&](uint32_t b) {
    auto it = ref.bins.find(b);
    if (it != ref.bins.end()) chunks.insert(chunks.end(), it->second.begin(), it->second.end());
  };
  const int shifts[5] = {26, 23, 20, 17, 14};
  const uint32_t bases[5] = {1, 9, 73, 585, 4681};
  uint64_t n_candidates = 1;
  for (int l = 0; l < 5; l++) n_candidates += (end0 >> shifts[l]) - (beg0 >> shifts[l]) + 1;
  if (n_candidates > 4 * ref.bins.size() + 16) {
    // wide interval: walk the bins that exist (same set, same order: std::map iterates bin ids ascending,
    // reg2bins enumerates level by level in ascending id) instead of probing tens of thousands of ids
    for (auto& kv : ref.bins) {
      const uint32_t b = kv.first;
      bool hit = b == 0;
      for (int l = 0; l < 5 && !hit; l++) {
        const uint64_t lo = bases[l] + (beg0 >> shifts[l]), hi = bases[l] + (end0 >> shifts[l]);
        const uint64_t next_base = l < 4 ? bases[l + 1] : 37449;
        if (b >= bases[l] && b < next_base) { hit = b >= lo && b <= hi; break; }
      }
      if (hit) chunks.insert(chunks.end(), kv.second.begin(), kv.second.end());
    }
  } else {
    add(0);
    for (int l = 0; l < 5; l++)
      for (uint64_t b = bases[l] + (beg0 >> shifts[l]); b <= bases[l] + (end0 >> shifts[l]); b++) add((uint32_t)b);
  }
  size_t li = (size_t)(beg0 >> 14);
  uint64_t min_off = li < ref.intervals.size() ? ref.intervals[li] : 0;
  std::vector<std::pair<uint64_t, uint64_t>> f;
  for (auto& c : chunks) if (c.second > min_off) f.push_back(c);
  std::sort(f.begin(), f.end());
  std::vector<std::pair<uint64_t, uint64_t>> m;
  for (auto& c : f) {
    if (!m.empty() && c.first <= m.back().second) { if (c.second > m.back().second) m.back().second = c.second; }
    else m.push_back(c);
  }
  return m;
}

// =================================================================================================
// estimates + balancer
// =================================================================================================
std::vector<RegionSizeEstimate> estimate_sizes_from_bai(const Bai* bai, const std::vector<GenomicRegion>& regions,
                                                        const std::vector<std::string>& ref_names,
                                                        const std::vector<int64_t>& ref_lengths) {
  std::vector<RegionSizeEstimate> out;
  for (auto& r : regions) {
    RegionSizeEstimate e;
    e.region = r;
    if (!bai) { e.estimated_bytes = 1; out.push_back(e); continue; }
    long idx = -1;
    // HashMap<&str, usize> built by enumerate(): a later duplicate name overwrites an earlier one
    for (size_t i = 0; i < ref_names.size(); i++) if (ref_names[i] == r.chrom) idx = (long)i;
    const BaiRef* ref = (idx >= 0 && (size_t)idx < bai->refs.size()) ? &bai->refs[idx] : nullptr;
    if (ref) {
      uint64_t mn = ~0ull, mx = 0;
      for (auto& b : ref->bins)
        for (auto& c : b.second) {
          mn = std::min(mn, c.first >> 16);
          mx = std::max(mx, c.second >> 16);
        }
      e.estimated_bytes = mx > mn ? mx - mn : 0;  // saturating_sub
    } else {
      e.estimated_bytes = 1;
    }
    if (idx >= 0 && (size_t)idx < ref_lengths.size() && ref_lengths[idx] > 0) {
      e.has_contig_length = true;
      e.contig_length = (uint64_t)ref_lengths[idx];
    }
    e.unmapped_count = (ref && ref->has_meta) ? ref->n_unmapped : 0;
    if (ref)
      for (auto& b : ref->bins)
        if (b.first >= 4681 && b.first <= 37448) e.nonempty_bin_positions.push_back((uint64_t)(b.first - 4681) * 16384 + 1);
    std::sort(e.nonempty_bin_positions.begin(), e.nonempty_bin_positions.end());
    e.leaf_bin_span = 16384;
    out.push_back(std::move(e));
  }
  return out;
}

std::vector<PartitionAssignment> balance_partitions(const std::vector<RegionSizeEstimate>& estimates, size_t target_partitions) {
  typedef unsigned __int128 u128;
  std::vector<PartitionAssignment> parts;
  if (estimates.empty()) return parts;
  const size_t target = std::max<size_t>(target_partitions, 1);
  uint64_t total = 0;
  for (auto& e : estimates) total += e.estimated_bytes;
  if (target == 1) {
    PartitionAssignment p;
    for (auto& e : estimates) p.regions.push_back(e.region);
    p.total_estimated_bytes = total;
    parts.push_back(std::move(p));
    return parts;
  }
  if (total == 0) {
    size_t nb = std::min(target, estimates.size());
    parts.resize(nb);
    for (size_t i = 0; i < estimates.size(); i++) parts[i % nb].regions.push_back(estimates[i].region);
    return parts;
  }
  const uint64_t eff_target = std::min<uint64_t>(target, total);
  const uint64_t base = total / eff_target;
  const uint64_t extra = total % eff_target;
  auto budget_for = [&](size_t i) { return i < extra ? base + 1 : base; };
  parts.emplace_back();
  uint64_t budget = budget_for(0);
  for (auto& est : estimates) {
    uint64_t remaining = est.estimated_bytes;
    uint64_t eff_start = 0, eff_end = 0;
    if (est.region.has_start && est.region.has_end && est.region.end >= est.region.start) {
      eff_start = est.region.start; eff_end = est.region.end;
    } else if (est.has_contig_length && est.contig_length > 0) {
      eff_start = 1; eff_end = est.contig_length;
    }
    const bool can_split = eff_end > 0 && eff_end >= eff_start;
    uint64_t pos = eff_start;
    bool was_split = false;
    if (remaining == 0) {
      size_t mi = 0;
      for (size_t i = 1; i < parts.size(); i++) if (parts[i].regions.size() < parts[mi].regions.size()) mi = i;
      parts[mi].regions.push_back(est.region);
      continue;
    }
    while (remaining > 0) {
      if (budget == 0 && parts.size() < eff_target) {
        parts.emplace_back();
        budget = budget_for(parts.size() - 1);
      }
      const bool is_last = parts.size() >= eff_target;
      const uint64_t remaining_bp = (can_split && pos <= eff_end) ? eff_end - pos + 1 : 0;
      const bool splittable = remaining_bp > 1;
      if (remaining <= budget || is_last || !splittable) {
        GenomicRegion region = est.region;
        if (can_split && pos <= eff_end && was_split) {
          region = GenomicRegion();
          region.chrom = est.region.chrom;
          region.has_start = true; region.start = pos;
          region.has_end = false;
        }
        auto& p = parts.back();
        p.regions.push_back(region);
        p.total_estimated_bytes += remaining;
        budget = budget > remaining ? budget - remaining : 0;
        remaining = 0;
      } else {
        was_split = true;
        uint64_t sub_end;
        const auto& nbp = est.nonempty_bin_positions;
        auto bp_split = [&]() {
          uint64_t bp = (uint64_t)((u128)remaining_bp * budget / remaining);
          bp = std::max<uint64_t>(1, std::min(bp, remaining_bp - 1));
          return pos + bp - 1;
        };
        if (!nbp.empty() && est.leaf_bin_span > 0) {
          size_t r0 = std::lower_bound(nbp.begin(), nbp.end(), pos) - nbp.begin();         // p < pos
          size_t r1 = std::upper_bound(nbp.begin(), nbp.end(), eff_end) - nbp.begin();     // p <= eff_end
          size_t nbins = r1 - r0;
          if (nbins > 1) {
            size_t take = (size_t)((u128)nbins * budget / remaining);
            take = std::max<size_t>(1, std::min(take, nbins - 1));
            uint64_t bin_start = nbp[r0 + take - 1];
            sub_end = std::min(bin_start + est.leaf_bin_span - 1, eff_end - 1);
          } else {
            sub_end = bp_split();
          }
        } else {
          sub_end = bp_split();
        }
        GenomicRegion region;
        region.chrom = est.region.chrom;
        region.has_start = true; region.start = pos;
        region.has_end = true; region.end = sub_end;
        auto& p = parts.back();
        p.regions.push_back(region);
        p.total_estimated_bytes += budget;
        remaining -= budget;
        pos = sub_end + 1;
        if (parts.size() < eff_target) {
          parts.emplace_back();
          budget = budget_for(parts.size() - 1);
        } else {
          budget = 0;
        }
      }
    }
    if (est.unmapped_count > 0) {
      GenomicRegion tail;
      tail.chrom = est.region.chrom;
      tail.unmapped_tail = true;
      auto& p = parts.back();
      p.regions.push_back(tail);
      p.total_estimated_bytes += 1;
      budget = budget > 1 ? budget - 1 : 0;
    }
  }
  std::vector<PartitionAssignment> out;
  for (auto& p : parts) if (!p.regions.empty()) out.push_back(std::move(p));
  return out;
}

std::string describe_partition(const PartitionAssignment& p) {
  std::ostringstream o;
  o << p.total_estimated_bytes << "|";
  for (size_t i = 0; i < p.regions.size(); i++) {
    auto& r = p.regions[i];
    if (i) o << ";";
    o << r.chrom << ":";
    if (r.has_start) o << r.start;
    o << "-";
    if (r.has_end) o << r.end;
    if (r.unmapped_tail) o << "*";
  }
  return o.str();
}

// =================================================================================================
// filters
// =================================================================================================
std::vector<Filter> copy_filters(const bioscan_filter* f, int32_t n) {
  std::vector<Filter> out;
  for (int32_t i = 0; i < n; i++) {
    Filter x;
    x.column = f[i].column ? f[i].column : "";
    x.op = f[i].op;
    for (int32_t k = 0; k < f[i].n_values; k++) {
      Literal l;
      l.kind = f[i].values[k].kind;
      l.i = f[i].values[k].i;
      l.f = f[i].values[k].f;
      if (l.kind == BIOSCAN_LIT_STR && f[i].values[k].s) l.s = f[i].values[k].s;
      x.values.push_back(std::move(l));
    }
    out.push_back(std::move(x));
  }
  return out;
}

static bool lit_u64(const Literal& l, uint64_t* v) {
  if (l.kind == BIOSCAN_LIT_INT && l.i >= 0) { *v = (uint64_t)l.i; return true; }
  return false;
}

void extract_genomic_regions(const std::vector<Filter>& filters, bool zero_based, std::vector<GenomicRegion>* regions, bool* unsat) {
  std::vector<std::string> chroms;
  bool has_lo = false, has_hi = false;
  uint64_t lo = 0, hi = 0;
  auto set_lo = [&](uint64_t v) { lo = has_lo ? std::max(lo, v) : v; has_lo = true; };
  auto set_hi = [&](uint64_t v) { hi = has_hi ? std::min(hi, v) : v; has_hi = true; };
  for (auto& f : filters) {
    const bool cmp = f.op <= BIOSCAN_OP_GE && f.values.size() == 1;
    uint64_t v;
    if (f.column == "chrom" && f.op == BIOSCAN_OP_EQ && cmp && f.values[0].kind == BIOSCAN_LIT_STR) {
      chroms.push_back(f.values[0].s);
    } else if (f.column == "chrom" && f.op == BIOSCAN_OP_IN) {
      for (auto& l : f.values) if (l.kind == BIOSCAN_LIT_STR) chroms.push_back(l.s);
    } else if (f.column == "start" && cmp && lit_u64(f.values[0], &v)) {
      uint64_t v1 = zero_based ? v + 1 : v;
      switch (f.op) {
        case BIOSCAN_OP_EQ: set_lo(v1); set_hi(v1); break;
        case BIOSCAN_OP_GT: set_lo(v1 + 1); break;
        case BIOSCAN_OP_GE: set_lo(v1); break;
        case BIOSCAN_OP_LT: set_hi(v1 ? v1 - 1 : 0); break;
        case BIOSCAN_OP_LE: set_hi(v1); break;
        default: break;
      }
    } else if (f.column == "end" && cmp && lit_u64(f.values[0], &v)) {
      switch (f.op) {
        case BIOSCAN_OP_EQ: set_hi(v); break;
        case BIOSCAN_OP_LT: set_hi(v ? v - 1 : 0); break;
        case BIOSCAN_OP_LE: set_hi(v); break;
        default: break;
      }
    } else if (f.column == "start" && f.op == BIOSCAN_OP_BETWEEN && f.values.size() == 2) {
      uint64_t a, b;
      if (lit_u64(f.values[0], &a) && lit_u64(f.values[1], &b)) {
        set_lo(zero_based ? a + 1 : a);
        set_hi(zero_based ? b + 1 : b);
      }
    }
  }
  std::sort(chroms.begin(), chroms.end());
  chroms.erase(std::unique(chroms.begin(), chroms.end()), chroms.end());
  *unsat = has_lo && has_hi && lo > hi;
  regions->clear();
  if (chroms.empty() || *unsat) return;
  for (auto& c : chroms) {
    GenomicRegion r;
    r.chrom = c;
    r.has_start = has_lo; r.start = lo;
    r.has_end = has_hi; r.end = hi;
    regions->push_back(r);
  }
}

bool is_genomic_coordinate_filter(const Filter& f) {
  if (f.op <= BIOSCAN_OP_GE) return f.column == "chrom" || f.column == "start" || f.column == "end";
  if (f.op == BIOSCAN_OP_BETWEEN || f.op == BIOSCAN_OP_NOT_BETWEEN) return f.column == "start" || f.column == "end";
  return f.column == "chrom";  // InList
}

bool can_push_down_record_filter(const Filter& f, const std::vector<FieldDef>& schema) {
  const FieldDef* fd = nullptr;
  for (auto& x : schema) if (x.name == f.column) { fd = &x; break; }
  if (!fd) return false;
  const bool is_str = fd->kind == AK_UTF8;
  const bool is_num = fd->kind == AK_INT32 || fd->kind == AK_UINT32 || fd->kind == AK_FLOAT32;
  if (f.op == BIOSCAN_OP_EQ || f.op == BIOSCAN_OP_NE) return is_str || is_num;
  if (f.op <= BIOSCAN_OP_GE) return is_num;
  if (f.op == BIOSCAN_OP_BETWEEN || f.op == BIOSCAN_OP_NOT_BETWEEN) return is_num;
  return is_str || is_num;
}

}  // namespace bioscan
