// bam_host.h -- host-side planning for the BAM scan: header, BAI, regions, partition balancer,
// filter analysis, tag registry.  Pure C++ (no HIP): mirrors the reference's planning layer
// (bio-format-core/src/{genomic_filter,partition_balancer,record_filter,tag_registry,metadata}.rs,
//  bio-format-bam/src/{storage,table_provider}.rs).
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bioscan.h"

namespace bioscan {

// ---- BAM header ---------------------------------------------------------------------------------
struct BamHeader {
  std::string text;
  std::vector<std::string> ref_names;
  std::vector<int64_t> ref_lengths;
  uint64_t first_record_offset = 0;
};
// Returns false if `u` does not yet contain the complete header (caller inflates more blocks).
bool parse_bam_header(const uint8_t* u, size_t n, BamHeader* out, std::string* err);

// bio-format-core/src/metadata.rs:321-485 -> ordered key/value list for the schema metadata
std::vector<std::pair<std::string, std::string>> extract_header_metadata(const BamHeader& h);

// ---- tag registry / schema ----------------------------------------------------------------------
enum ArrowKind : int32_t {
  AK_INT32 = 0, AK_UINT32, AK_FLOAT32, AK_UTF8, AK_BINARY,
  AK_LIST_INT8, AK_LIST_UINT8, AK_LIST_INT16, AK_LIST_UINT16, AK_LIST_INT32, AK_LIST_UINT32, AK_LIST_FLOAT32
};
struct TagDef {
  char sam_type;
  ArrowKind kind;
  const char* description;
};
const TagDef* known_tag(const std::string& tag);
ArrowKind sam_tag_type_to_arrow(char c);                          // tag_registry.rs:757-768
bool sam_array_subtype_to_arrow(char c, ArrowKind* out);
// "TAG:TYPE" / "TAG:B:SUBTYPE" (tag_registry.rs:698-755); returns error text or "".
std::string parse_tag_type_hints(const std::vector<std::string>& hints, std::map<std::string, std::pair<char, ArrowKind>>* out);
std::string format_sam_tag_type(char sam_type, ArrowKind k);

struct FieldDef {
  std::string name;
  ArrowKind kind;
  bool nullable;
  std::vector<std::pair<std::string, std::string>> metadata;
};

// ---- BAI ----------------------------------------------------------------------------------------
struct BaiRef {
  std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;  // without pseudo-bin 37450
  std::vector<uint64_t> intervals;
  bool has_meta = false;
  uint64_t ref_beg = 0, ref_end = 0, n_mapped = 0, n_unmapped = 0;
};
struct Bai {
  std::vector<BaiRef> refs;
  bool has_no_coor = false;
  uint64_t n_no_coor = 0;
};
bool parse_bai(const std::vector<uint8_t>& data, Bai* out, std::string* err);

// ---- regions / partitions -----------------------------------------------------------------------
struct GenomicRegion {
  std::string chrom;
  bool has_start = false, has_end = false;
  uint64_t start = 0, end = 0;  // 1-based inclusive
  bool unmapped_tail = false;
};
struct RegionSizeEstimate {
  GenomicRegion region;
  uint64_t estimated_bytes = 0;
  bool has_contig_length = false;
  uint64_t contig_length = 0;
  uint64_t unmapped_count = 0;
  std::vector<uint64_t> nonempty_bin_positions;
  uint64_t leaf_bin_span = 0;
};
struct PartitionAssignment {
  std::vector<GenomicRegion> regions;
  uint64_t total_estimated_bytes = 0;
};
// bio-format-bam/src/storage.rs:336-436
std::vector<RegionSizeEstimate> estimate_sizes_from_bai(const Bai* bai, const std::vector<GenomicRegion>& regions,
                                                        const std::vector<std::string>& ref_names,
                                                        const std::vector<int64_t>& ref_lengths);
// bio-format-core/src/partition_balancer.rs:61-295
std::vector<PartitionAssignment> balance_partitions(const std::vector<RegionSizeEstimate>& estimates, size_t target_partitions);

// ---- filters ------------------------------------------------------------------------------------
struct Literal {
  int32_t kind = BIOSCAN_LIT_NULL;
  int64_t i = 0;
  double f = 0;
  std::string s;
};
struct Filter {
  std::string column;
  int32_t op = 0;
  std::vector<Literal> values;
};
std::vector<Filter> copy_filters(const bioscan_filter* f, int32_t n);
// bio-format-core/src/genomic_filter.rs:51-105
void extract_genomic_regions(const std::vector<Filter>& filters, bool zero_based, std::vector<GenomicRegion>* regions, bool* unsatisfiable);
bool is_genomic_coordinate_filter(const Filter& f);                                  // genomic_filter.rs:120-147
bool can_push_down_record_filter(const Filter& f, const std::vector<FieldDef>& schema);  // record_filter.rs:40-55,285-356

// noodles-csi BinningIndex::query: merged chunk list for (ref, [start1,end1]) (0 = unbounded)
std::vector<std::pair<uint64_t, uint64_t>> bai_query_chunks(const Bai& bai, size_t ref_idx, bool has_start, uint64_t start1,
                                                            bool has_end, uint64_t end1);

std::string describe_partition(const PartitionAssignment& p);

}  // namespace bioscan
