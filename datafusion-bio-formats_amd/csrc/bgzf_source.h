// bgzf_source.h -- a compressed (BGZF) or plain input file resident in HBM, shared by the BAM, FASTQ and
// VCF scan paths: host framing of the member chain, upload, and launches of the inflate / CRC kernels.
// Replaces noodles-bgzf's `io::Reader` (call sites bam/src/storage.rs:161-169, fastq/src/physical_exec.rs:482-491,
// vcf/src/storage.rs:117-122, 766-776) at block granularity.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace bioscan {

const char* inflate_status_str(uint32_t s);

// What one execution needs besides the resident file to run K1 / K2 on its own: a stream, the member counter, the
// per-wave scratch and a status slot per member of a launch.  Streams of one provider each own one, so partitions
// execute concurrently without sharing any mutable device state (bio-format-bam/src/physical_exec.rs:878-881: every
// execute opens its own reader).
struct K1Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuf<uint32_t> ctr;
  DevBuf<unsigned long long> scratch;
  DevBuf<uint32_t> status;  // status[i] = member b0 + i of the last launch
  uint32_t grid = 0;
  // one-shot launches (a wave per member, see inflate_v3.hip): scratch strides are borrowed through these flags, so
  // several launches of one context may be in flight at once
  DevBuf<uint32_t> slots;
  uint32_t n_slots = 0;
  DevBuf<uint32_t> pre;     // K0's header records of the members of one launch (V3_PRE_DWORDS each); empty = K0 off
  K1Ctx() = default;
  K1Ctx(const K1Ctx&) = delete;
  K1Ctx& operator=(const K1Ctx&) = delete;
  ~K1Ctx();
};

// The part of a file one device holds: the compressed bytes of members [m_lo, m_hi) and the member tables, resident in
// that device's HBM (each GPU receives only the compressed byte ranges its partitions cover).  Immutable once
// built: a wider span is a new image, executions in flight keep the one they started with.
struct DeviceImage {
  int device = 0;
  uint32_t m_lo = 0, m_hi = 0;
  DevBuf<uint8_t> d_comp;              // bytes [blk_coff[m_lo], blk_coff[m_hi]) + 4 KiB of zero padding
  const uint8_t* comp_base = nullptr;  // d_comp.p - blk_coff[m_lo]: comp_base + blk_coff[b] is member b
  DevBuf<uint64_t> d_coff, d_uoff;     // the whole tables (8 B per member each)
  uint32_t grid_max = 0;               // persistent grid of K1 on this device (waves)
  size_t scratch_stride = 0;
  DevBuf<uint8_t> d_ref_names;         // BAM: reference-name table for chrom / mate_chrom
  DevBuf<uint32_t> d_ref_name_off, d_ref_name_len;
};

struct BgzfSource {
  std::string path;
  const char* what = "BAM";  // format name used in error messages
  int device = 0;

  HostBuf file;  // pinned copy of the file (+ slack)
  size_t file_len = 0;
  std::vector<uint64_t> blk_coff, blk_uoff;  // n_blocks + 1 entries each
  uint64_t ulen = 0;

  std::mutex mu;
  hipStream_t stream = nullptr;
  bool resident = false;
  DevBuf<uint8_t> d_comp;
  DevBuf<uint64_t> d_coff, d_uoff;
  DevBuf<uint32_t> d_status;
  DevBuf<uint32_t> d_k1_ctr;               // [0] member counter, [1..] debug counters
  DevBuf<unsigned long long> d_k1_scratch;  // per-wave scratch of K1 (match list + checkpoint rows)
  uint32_t k1_grid = 0;  // persistent grid of K1 (waves) of the whole-file image below
  DevBuf<uint8_t> d_u;  // inflated bytes of the range decoded last

  ~BgzfSource();
  uint32_t n_blocks() const { return (uint32_t)(blk_coff.size() - 1); }
  void set_device() { HIP_CHECK(hipSetDevice(device)); }
  void load_file();
  void frame();  // BGZF framing (SAM spec 4.1): walk the member chain; inflated offsets from the ISIZE trailers
  void make_resident();
  // Inflate members [b0, b0+nb) so that member b0's payload lands at dst[0].
  void launch_inflate(uint8_t* dst, uint32_t nb, uint32_t b0 = 0);
  // nl_cnt / nl_head (FASTQ): per-tile newline counts of the text buffer in which `dst` lies nl_head bytes behind the start
  void launch_crc(const uint8_t* dst, uint32_t nb, uint32_t b0 = 0, uint32_t* nl_cnt = nullptr, uint64_t nl_head = 0);
  void report_k1_debug(uint32_t nb);
  void report_k1_debug(const uint32_t* ctr_dev, uint32_t nb);  // BIOSCAN_DEBUG=1: pass / phase counters of the last K1 launch
  void check_inflate_status(uint32_t b0, uint32_t nb);
  // Per-device, per-range residency (BAM): the image of `device` that covers members [m_lo, m_hi), built or widened on
  // demand from the mapped file.  The three K1 / K2 steps on a caller-owned context (its stream, its scratch, status
  // relative to b0) against an image.
  std::map<int, std::shared_ptr<DeviceImage>> images;  // guarded by mu
  std::shared_ptr<DeviceImage> image_for(int dev, uint32_t m_lo, uint32_t m_hi);
  std::shared_ptr<DeviceImage> image_of(int dev);      // the current image of a device, or null
  std::shared_ptr<DeviceImage> build_image(int dev, uint32_t m_lo, uint32_t m_hi);
  void init_ctx(K1Ctx& c, const DeviceImage& img, uint32_t max_members, bool oneshot = false);
  void launch_inflate(K1Ctx& c, const DeviceImage& img, uint8_t* dst, uint32_t nb, uint32_t b0);
  void launch_crc(K1Ctx& c, const DeviceImage& img, const uint8_t* dst, uint32_t nb, uint32_t b0, uint32_t* nl_cnt = nullptr, uint64_t nl_head = 0);
  void check_inflate_status(K1Ctx& c, uint32_t b0, uint32_t nb);
  // the same three steps with the status slots and the stream named by the caller (look-ahead inflate: K1 of the next
  // chunk runs on the context's stream while CRC and the status check of this chunk run on the execute's own stream)
  void launch_inflate_to(K1Ctx& c, const DeviceImage& img, uint8_t* dst, uint32_t nb, uint32_t b0, uint32_t* status);
  void launch_crc_on(const DeviceImage& img, const uint8_t* dst, uint32_t nb, uint32_t b0, uint32_t* status, hipStream_t st,
                     uint32_t* nl_cnt = nullptr, uint64_t nl_head = 0);
  void check_inflate_status_on(uint32_t* status, hipStream_t st, uint32_t b0, uint32_t nb);
  // Inflate blocks [0, b1) into a temporary device buffer and copy to the host (header / sampling).
  std::vector<uint8_t> inflate_prefix_to_host(uint32_t b1);
};

}  // namespace bioscan
