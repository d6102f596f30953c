// vcf_host.h -- host-side planning for the VCF scan: header, schema, tabix index, size estimates.
// Pure C++ (no HIP).  Mirrors bio-format-vcf/src/table_provider.rs:91-338 (schema), :995-1076 (index
// contig names), storage.rs:815-986 (estimate_sizes_from_tbi), physical_exec.rs:81-137
// (choose_effective_batch_size); the partition balancer and filter analysis are shared with bam_host.h.
#pragma once
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

#include "bam_host.h"

namespace bioscan {

struct VcfFieldDefn {
  std::string id, number, type, description;  // number: "0","1",...,"A","R","G","."; type: Integer/Float/Flag/Character/String
};
struct VcfHeader {
  std::string file_format = "VCFv4.3";
  std::vector<VcfFieldDefn> infos, formats;                      // header order
  std::vector<std::pair<std::string, std::string>> filters;     // (id, description)
  std::vector<std::pair<std::string, int64_t>> contigs;         // (id, length or -1)
  std::vector<std::pair<std::string, std::string>> alts;
  std::vector<std::string> samples;
  uint64_t header_bytes = 0;  // bytes up to and including the #CHROM line
  const VcfFieldDefn* info(const std::string& id) const;
  const VcfFieldDefn* format(const std::string& id) const;
};
// Returns false when `u` does not yet hold the whole header (caller decodes more blocks), unless at_eof.
bool parse_vcf_header(const uint8_t* u, size_t n, bool at_eof, VcfHeader* out, std::string* err);

// ---- nested schema -------------------------------------------------------------------------------
enum VKind : int32_t { VK_INT32 = 0, VK_UINT32, VK_FLOAT32, VK_FLOAT64, VK_BOOL, VK_UTF8, VK_LIST, VK_STRUCT };
struct VField {
  std::string name;
  VKind kind = VK_UTF8;
  bool nullable = true;
  std::vector<std::pair<std::string, std::string>> metadata;
  std::vector<VField> children;  // LIST: one "item" child; STRUCT: members
};
const char* vkind_format(VKind k);  // Arrow C format string

// value type of one INFO / FORMAT tag: scalar kind + whether it is a list (Number not 0 / 1)
struct VcfValueType {
  VKind scalar = VK_UTF8;
  bool is_list = false;
};
VcfValueType info_value_type(const VcfHeader& h, const std::string& tag);     // table_provider.rs:1602-1626
VcfValueType format_value_type(const VcfHeader& h, const std::string& tag);   // table_provider.rs:370-396

struct VcfSchema {
  std::vector<VField> fields;
  std::vector<std::pair<std::string, std::string>> metadata;
  std::vector<std::string> info_fields, format_fields;
  std::vector<std::string> samples;          // selected, output order
  std::vector<int32_t> sample_header_index;  // header column of each selected sample
  bool multi = false;                        // source has more than one sample -> nested `genotypes`
  bool has_format = false;                   // FORMAT columns present in the schema
};
// determine_schema_from_header.  info_fields / format_fields / samples: nullptr = all.  Returns error text or "".
std::string determine_vcf_schema(const VcfHeader& h, const std::vector<std::string>* info_fields,
                                 const std::vector<std::string>* format_fields, const std::vector<std::string>* samples,
                                 bool zero_based, const std::vector<std::string>* index_names, VcfSchema* out);

// ---- tabix ---------------------------------------------------------------------------------------
struct Tbi {
  std::vector<std::string> names;
  Bai idx;  // same bin / chunk / linear-index layout as BAI
  int32_t format = 0, col_seq = 0, col_beg = 0, col_end = 0, meta = 0, skip = 0;
};
bool parse_tbi(const std::vector<uint8_t>& inflated, Tbi* out, std::string* err);
// CSI header (`CSI\1`, min_shift, depth, l_aux, aux = the tabix header with the reference names): only the names are
// used — the reference reads them for `bio.vcf.contigs.indexed` (table_provider.rs:1019-1025) and then hands the file
// to the tabix reader, which rejects it (storage.rs:766).
bool parse_csi_names(const std::vector<uint8_t>& inflated, std::vector<std::string>* names, std::string* err);
// bio-format-vcf/src/storage.rs:815-986
std::vector<RegionSizeEstimate> estimate_sizes_from_tbi(const Tbi* tbi, const std::vector<GenomicRegion>& regions,
                                                        const std::vector<std::string>& contig_names,
                                                        const std::vector<uint64_t>& contig_lengths);
// physical_exec.rs:81-137
uint64_t choose_effective_batch_size(uint64_t requested, bool any_format, uint64_t n_format_fields, uint64_t n_selected,
                                     uint64_t n_source);

std::string json_string_array(const std::vector<std::string>& v);

}  // namespace bioscan
