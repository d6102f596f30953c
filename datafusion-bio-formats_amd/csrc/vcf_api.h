// vcf_api.h -- interface between the C ABI (engine.cpp) and the VCF scan path (vcf_engine.cpp).
#pragma once
#include <string>

#include "../../include/bioscan.h"
#include "common.h"

namespace bioscan {

struct VcfStreamI {
  virtual ~VcfStreamI() {}
  virtual bool next(ArrowArray* out) = 0;  // false = end of stream
  virtual void list_udf(const char* field, int32_t udf, double threshold, bioscan_udf_stats* out) = 0;
};
struct VcfPlanI {
  virtual ~VcfPlanI() {}
  virtual int32_t n_partitions() const = 0;
  virtual void schema(ArrowSchema* out) const = 0;
  virtual std::string display() const = 0;
  virtual std::string partition_desc(int32_t partition) const = 0;
  virtual VcfStreamI* execute(int32_t partition, int32_t batch_size, bool device_only, bioscan_scan_stats* stats) const = 0;
};
struct VcfProviderI {
  virtual ~VcfProviderI() {}
  virtual void schema(ArrowSchema* out) const = 0;
  virtual void supports_filters_pushdown(const bioscan_filter* filters, int32_t n, int32_t* out) const = 0;
  virtual VcfPlanI* scan(const int32_t* projection, int32_t n_projection, const bioscan_filter* filters, int32_t n_filters,
                         int64_t limit, int32_t target_partitions) = 0;
  virtual void make_resident() = 0;
  virtual void set_chunk_members(uint32_t n) = 0;  // BGZF members per pipeline chunk of a stream (0 = default)
};
VcfProviderI* vcf_open(const char* path, const bioscan_vcf_options* opts);

void udf_list_avg_host(const ArrowArray* in, const ArrowSchema* in_schema, int32_t device_id, ArrowArray* out, ArrowSchema* out_schema);
void udf_list_cmp_host(const ArrowArray* in, const ArrowSchema* in_schema, int32_t op, double threshold, int32_t device_id,
                       ArrowArray* out, ArrowSchema* out_schema);

void udf_list_and_host(const ArrowArray* a, const ArrowSchema* as, const ArrowArray* b, const ArrowSchema* bs, int32_t device_id,
                       ArrowArray* out, ArrowSchema* out_schema);
void udf_set_gts_host(const ArrowArray* gt, const ArrowSchema* gs, const ArrowArray* mask, const ArrowSchema* ms, const char* replacement,
                      int32_t device_id, ArrowArray* out, ArrowSchema* out_schema);
void udf_allele_stats_host(const ArrowArray* gt, const ArrowSchema* gs, const ArrowArray* alt, const ArrowSchema* as, int32_t which,
                           int32_t device_id, ArrowArray* out, ArrowSchema* out_schema);

}  // namespace bioscan
