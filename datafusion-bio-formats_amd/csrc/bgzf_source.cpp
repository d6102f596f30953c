// bgzf_source.cpp -- see bgzf_source.h
#include "bgzf_source.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace bioscan {

const char* inflate_status_str(uint32_t s) {
  switch (s) {
    case INF_BAD_HEADER: return "invalid BGZF header";
    case INF_BAD_BTYPE: return "invalid DEFLATE block type";
    case INF_BAD_CODE: return "invalid Huffman code";
    case INF_BAD_DIST: return "invalid match distance";
    case INF_OVERRUN: return "output overrun";
    case INF_SIZE_MISMATCH: return "ISIZE mismatch";
    case INF_BAD_STORED: return "invalid stored block";
    case INF_CRC_MISMATCH: return "CRC32 mismatch";
    case 0xFF: return "no wave took the member (a bounded inflate launch ran out of scratch strides)";
    default: return "unknown";
  }
}


BgzfSource::~BgzfSource() {
  if (stream) (void)hipStreamDestroy(stream);
}

void BgzfSource::load_file() {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) throw Error(std::string("Failed to open ") + what + ": " + path + ": " + strerror(errno));
  struct stat sb;
  if (fstat(fd, &sb) != 0) { const int e = errno; close(fd); throw Error(std::string("Failed to open ") + what + ": " + path + ": " + strerror(e)); }
  file_len = (size_t)sb.st_size;
  // the file is mapped, not read: framing touches only the member headers and the upload streams the pages straight
  // out of the page cache (16.5 GB: 3.2 s with a pinned staging copy); a file that cannot be mapped is read
  if (file_len && S_ISREG(sb.st_mode) && file.map_file(fd, file_len)) { close(fd); return; }
  file.alloc(file_len + 4096, /*use_pool=*/false);
  size_t got = 0;
  while (got < file_len) {
    const ssize_t r = read(fd, file.p + got, std::min<size_t>(file_len - got, 1u << 30));
    if (r <= 0) break;
    got += (size_t)r;
  }
  close(fd);
  if (got != file_len) throw Error("short read on " + path);
  memset(file.p + file_len, 0, 4096);
}

void BgzfSource::frame() {
  blk_coff.clear();
  blk_uoff.clear();
  uint64_t o = 0, uo = 0;
  const uint8_t* d = file.p;
  while (o < file_len) {
    if (file_len - o < 18) throw Error("BGZF: truncated block header at offset " + std::to_string(o));
    if (d[o] != 0x1f || d[o + 1] != 0x8b || d[o + 2] != 8 || !(d[o + 3] & 4))
      throw Error("BGZF: invalid block header at offset " + std::to_string(o));
    uint32_t xlen = d[o + 10] | (d[o + 11] << 8);
    if (o + 12 + (uint64_t)xlen + 8 > file_len) throw Error("BGZF: truncated block header at offset " + std::to_string(o));
    uint64_t p = o + 12, pe = o + 12 + xlen;
    int64_t bsize = -1;
    while (p + 4 <= pe) {
      uint32_t slen = d[p + 2] | (d[p + 3] << 8);
      if (d[p] == 66 && d[p + 1] == 67 && slen == 2 && p + 6 <= pe) bsize = (int64_t)(d[p + 4] | (d[p + 5] << 8)) + 1;
      p += 4 + slen;
    }
    if (bsize < 0 || o + (uint64_t)bsize > file_len || (uint64_t)bsize < 12 + xlen + 8)
      throw Error("BGZF: invalid block size at offset " + std::to_string(o));
    uint32_t isize;
    memcpy(&isize, d + o + bsize - 4, 4);
    if (isize > 65536) throw Error("BGZF: ISIZE > 64 KiB at offset " + std::to_string(o));
    blk_coff.push_back(o);
    blk_uoff.push_back(uo);
    o += (uint64_t)bsize;
    uo += isize;
  }
  blk_coff.push_back(o);
  blk_uoff.push_back(uo);
  ulen = uo;
}

static void k1_launch_params(int device, uint32_t* grid, size_t* stride) {
  hipDeviceProp_t pr;
  HIP_CHECK(hipGetDeviceProperties(&pr, device));
  const int occ = env_knobs().k1_version == 4 ? v4_resident_wg_per_cu() : v3_resident_wg_per_cu();
  *stride = V3_SCRATCH_STRIDE;
  const int per_cu = env_knobs().k1_waves_per_cu > 0 ? env_knobs().k1_waves_per_cu : occ;
  *grid = (uint32_t)pr.multiProcessorCount * (uint32_t)per_cu;
  if (env_knobs().debug) fprintf(stderr, "[bioscan] K1 residency: %d waves per CU x %d CUs\n", per_cu, pr.multiProcessorCount);
}

void BgzfSource::make_resident() {
  if (resident) return;
  set_device();
  if (!stream) HIP_CHECK(hipStreamCreate(&stream));
  d_comp.alloc(file_len + 4096);
  if (file_len) HIP_CHECK(hipMemcpyAsync(d_comp.p, file.p, file_len, hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipMemsetAsync(d_comp.p + file_len, 0, 4096, stream));  // the kernels over-read a little past the last member
  d_coff.alloc(blk_coff.size());
  d_uoff.alloc(blk_uoff.size());
  HIP_CHECK(hipMemcpyAsync(d_coff.p, blk_coff.data(), blk_coff.size() * 8, hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipMemcpyAsync(d_uoff.p, blk_uoff.data(), blk_uoff.size() * 8, hipMemcpyHostToDevice, stream));
  d_status.alloc(std::max<size_t>(n_blocks(), 1));
  {
    size_t stride = 0;
    k1_launch_params(device, &k1_grid, &stride);
    k1_grid = std::min<uint32_t>(k1_grid, std::max<uint32_t>(n_blocks(), 1));
    d_k1_ctr.alloc(64);
    d_k1_scratch.alloc(((size_t)k1_grid + 8) * stride);
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  // the compressed bytes now live in HBM.  A mapped file stays mapped (it costs address space, not memory): a later execute
  // that streams a partition in chunks uploads its own range from it (image_for); a read copy is given back.
  if (!file.mapped) file.reset();
  images.clear();  // (the header image of the open call)
  resident = true;
}

// members K1 v4 left for the wide-table kernel (it counts them in ctr[1]); waits for the launch
static uint32_t k1_retry_count(const uint32_t* ctr_dev, hipStream_t st) {
  uint32_t n = 0;
  HIP_CHECK(hipMemcpyAsync(&n, ctr_dev + 1, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return n;
}

// waves of the retry launch: a handful of members (config 2: a few per 65 536) do not need the whole device -- and dispatching
// its 5 120 one-wave workgroups costs 0.7 ms; enough waves that finding the members in the status array stays a few loads each
static uint32_t k1_retry_grid(uint32_t full, uint32_t n_retry) { return std::min<uint32_t>(full, std::max<uint32_t>(256u, 8u * n_retry)); }

void BgzfSource::launch_inflate(uint8_t* dst, uint32_t nb, uint32_t b0) {
  uint8_t* base = dst - blk_uoff[b0];
  HIP_CHECK(hipMemsetAsync(d_k1_ctr.p, 0, 256, stream));
  if (env_knobs().k1_version == 4) {
    launch_bgzf_inflate_v4(d_comp.p, d_coff.p + b0, d_uoff.p + b0, base, nb, d_status.p + b0, d_k1_ctr.p, d_k1_scratch.p, k1_grid,
                           env_knobs().debug ? d_k1_ctr.p + 2 : nullptr, stream, nullptr, 0, 0, 0, nullptr);
    // members whose Huffman codes do not fit v4's table pool (it counts them in ctr[1]): the wide-table kernel, same stream --
    // launched only when there is one (an empty persistent grid still costs 0.7 ms of dispatch; the host waits for K1 here
    // instead, which it does a stage later anyway to look at the members' status)
    if (const uint32_t nr = k1_retry_count(d_k1_ctr.p, stream))
      launch_bgzf_inflate_v3(d_comp.p, d_coff.p + b0, d_uoff.p + b0, base, nb, d_status.p + b0, d_k1_ctr.p, d_k1_scratch.p,
                             k1_retry_grid(k1_grid, nr), nullptr, stream, nullptr, 0, 0, 0, nullptr, true);
  } else {
    launch_bgzf_inflate_v3(d_comp.p, d_coff.p + b0, d_uoff.p + b0, base, nb, d_status.p + b0, d_k1_ctr.p, d_k1_scratch.p, k1_grid,
                           env_knobs().debug ? d_k1_ctr.p + 2 : nullptr, stream, nullptr, 0, 0, 0, nullptr);
  }
}

K1Ctx::~K1Ctx() {
  if (stream) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    (void)hipStreamSynchronize(stream);  // nothing of this context may still be running when its scratch goes back to the pool
    (void)hipStreamDestroy(stream);
    (void)hipSetDevice(prev);
  }
}

std::shared_ptr<DeviceImage> BgzfSource::image_of(int dev) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = images.find(dev);
  return it == images.end() ? nullptr : it->second;
}

std::shared_ptr<DeviceImage> BgzfSource::image_for(int dev, uint32_t m_lo, uint32_t m_hi) {
  std::lock_guard<std::mutex> lk(mu);
  m_hi = std::min(m_hi, n_blocks());
  if (m_lo > m_hi) m_lo = m_hi;
  auto it = images.find(dev);
  if (it != images.end()) {
    const DeviceImage& cur = *it->second;
    if (m_lo == m_hi || (cur.m_lo <= m_lo && m_hi <= cur.m_hi)) return it->second;  // (an empty request needs no bytes)
    if (cur.m_lo == cur.m_hi) { images.erase(it); it = images.end(); }
  }
  if (it != images.end()) {
    m_lo = std::min(m_lo, it->second->m_lo);  // widen: one contiguous span per device
    m_hi = std::max(m_hi, it->second->m_hi);
  }
  auto img = build_image(dev, m_lo, m_hi);
  images[dev] = img;
  return img;
}

// uploads members [m_lo, m_hi) to `dev` (caller holds mu or owns the source exclusively)
std::shared_ptr<DeviceImage> BgzfSource::build_image(int dev, uint32_t m_lo, uint32_t m_hi) {
  if (!file.p && file_len) throw Error("the host image of " + path + " has been released (make_resident): cannot upload another range");
  HIP_CHECK(hipSetDevice(dev));
  auto img = std::make_shared<DeviceImage>();
  img->device = dev;
  img->m_lo = m_lo; img->m_hi = m_hi;
  const uint64_t c0 = blk_coff[m_lo], c1 = blk_coff[m_hi];
  hipStream_t up;
  HIP_CHECK(hipStreamCreate(&up));
  img->d_comp.alloc(c1 - c0 + 4096);
  if (c1 > c0) HIP_CHECK(hipMemcpyAsync(img->d_comp.p, file.p + c0, c1 - c0, hipMemcpyHostToDevice, up));
  HIP_CHECK(hipMemsetAsync(img->d_comp.p + (c1 - c0), 0, 4096, up));  // the kernels over-read a little past the last member
  img->comp_base = img->d_comp.p - c0;
  img->d_coff.alloc(blk_coff.size());
  img->d_uoff.alloc(blk_uoff.size());
  HIP_CHECK(hipMemcpyAsync(img->d_coff.p, blk_coff.data(), blk_coff.size() * 8, hipMemcpyHostToDevice, up));
  HIP_CHECK(hipMemcpyAsync(img->d_uoff.p, blk_uoff.data(), blk_uoff.size() * 8, hipMemcpyHostToDevice, up));
  k1_launch_params(dev, &img->grid_max, &img->scratch_stride);
  HIP_CHECK(hipStreamSynchronize(up));
  (void)hipStreamDestroy(up);
  return img;
}

void BgzfSource::init_ctx(K1Ctx& c, const DeviceImage& img, uint32_t max_members, bool oneshot) {
  HIP_CHECK(hipSetDevice(img.device));
  c.device = img.device;
  if (!c.stream) HIP_CHECK(hipStreamCreate(&c.stream));
  c.grid = std::min<uint32_t>(img.grid_max, std::max<uint32_t>(max_members, 1));
  if (!c.ctr.p) c.ctr.alloc(64);
  size_t need = ((size_t)c.grid + 8) * img.scratch_stride;
  if (oneshot) {
    // twice what the device holds at once: a wave finds a free stride within a few probes
    c.n_slots = (uint32_t)((uint64_t)img.grid_max * (uint64_t)std::max(105, env_knobs().k1_slots_pct) / 100u);
    need = (size_t)c.n_slots * img.scratch_stride;
    if (c.slots.n < c.n_slots) c.slots.alloc(c.n_slots);
    HIP_CHECK(hipMemsetAsync(c.slots.p, 0, (size_t)c.n_slots * 4, c.stream));
  }
  if (c.scratch.n < need) c.scratch.alloc(need);
  if (c.status.n < max_members) c.status.alloc(std::max<uint32_t>(max_members, 1));
  if (env_knobs().k1_preheaders && c.pre.n < (size_t)max_members * V3_PRE_DWORDS) c.pre.alloc((size_t)std::max<uint32_t>(max_members, 1) * V3_PRE_DWORDS);
}

void BgzfSource::launch_inflate_to(K1Ctx& c, const DeviceImage& img, uint8_t* dst, uint32_t nb, uint32_t b0, uint32_t* status) {
  if (b0 < img.m_lo || b0 + nb > img.m_hi) throw Error("internal: members outside the resident range of the device image");
  uint8_t* base = dst - blk_uoff[b0];
  HIP_CHECK(hipMemsetAsync(c.ctr.p, 0, 256, c.stream));
  if (c.n_slots) HIP_CHECK(hipMemsetAsync(status, 0xFF, (size_t)nb * 4, c.stream));  // a member no bounded wave took reads as an error
  // K0: the first block header of every member, one member per lane, ahead of K1 on the same stream
  const bool k0 = c.pre.p && (size_t)nb * V3_PRE_DWORDS <= c.pre.n;
  if (k0) launch_bgzf_headers(img.comp_base, img.d_coff.p + b0, nb, c.pre.p, c.stream);
  if (env_knobs().k1_version == 4 && !c.n_slots) {
    launch_bgzf_inflate_v4(img.comp_base, img.d_coff.p + b0, img.d_uoff.p + b0, base, nb, status, c.ctr.p, c.scratch.p, c.grid,
                           env_knobs().debug ? c.ctr.p + 2 : nullptr, c.stream, nullptr, 0, 0, 0, k0 ? c.pre.p : nullptr);
    if (const uint32_t nr = k1_retry_count(c.ctr.p, c.stream))
      launch_bgzf_inflate_v3(img.comp_base, img.d_coff.p + b0, img.d_uoff.p + b0, base, nb, status, c.ctr.p, c.scratch.p,
                             k1_retry_grid(c.grid, nr), nullptr, c.stream, nullptr, 0, 0, 0, nullptr, true);
  } else {
    launch_bgzf_inflate_v3(img.comp_base, img.d_coff.p + b0, img.d_uoff.p + b0, base, nb, status, c.ctr.p, c.scratch.p, c.grid,
                           env_knobs().debug ? c.ctr.p + 2 : nullptr, c.stream, c.n_slots ? c.slots.p : nullptr, c.n_slots,
                           (uint32_t)std::max(1, env_knobs().k1_per_wave), (uint32_t)env_knobs().k1_bounded_wpw, k0 ? c.pre.p : nullptr);
  }
}
void BgzfSource::launch_inflate(K1Ctx& c, const DeviceImage& img, uint8_t* dst, uint32_t nb, uint32_t b0) {
  launch_inflate_to(c, img, dst, nb, b0, c.status.p);
}

void BgzfSource::launch_crc_on(const DeviceImage& img, const uint8_t* dst, uint32_t nb, uint32_t b0, uint32_t* status, hipStream_t st,
                               uint32_t* nl_cnt, uint64_t nl_head) {
  if (nl_cnt && (((uintptr_t)dst - nl_head) & 15)) throw Error("internal: text buffer not 16-byte aligned");
  launch_bgzf_crc32(img.comp_base, img.d_coff.p + b0, img.d_uoff.p + b0, dst - blk_uoff[b0], nb, status, st, nl_cnt, nl_head - blk_uoff[b0]);
}
void BgzfSource::launch_crc(K1Ctx& c, const DeviceImage& img, const uint8_t* dst, uint32_t nb, uint32_t b0, uint32_t* nl_cnt, uint64_t nl_head) {
  launch_crc_on(img, dst, nb, b0, c.status.p, c.stream, nl_cnt, nl_head);
}

void BgzfSource::check_inflate_status_on(uint32_t* status, hipStream_t st, uint32_t b0, uint32_t nb) {
  DevBuf<uint32_t> res(1);
  launch_first_bad_status(status, nb, res.p, st);
  uint32_t i = 0xFFFFFFFFu;
  HIP_CHECK(hipMemcpyAsync(&i, res.p, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (i != 0xFFFFFFFFu) {
    uint32_t stv = 0;
    HIP_CHECK(hipMemcpyAsync(&stv, status + i, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    throw Error(std::string(what) + " read error: BGZF block " + std::to_string(b0 + i) + " at offset " +
                std::to_string(blk_coff[b0 + i]) + ": " + inflate_status_str(stv & 0xFF) + " (code " + std::to_string(stv) + ")");
  }
}
void BgzfSource::check_inflate_status(K1Ctx& c, uint32_t b0, uint32_t nb) { check_inflate_status_on(c.status.p, c.stream, b0, nb); }

void BgzfSource::launch_crc(const uint8_t* dst, uint32_t nb, uint32_t b0, uint32_t* nl_cnt, uint64_t nl_head) {
  if (nl_cnt && (((uintptr_t)dst - nl_head) & 15)) throw Error("internal: text buffer not 16-byte aligned");
  launch_bgzf_crc32(d_comp.p, d_coff.p + b0, d_uoff.p + b0, dst - blk_uoff[b0], nb, d_status.p + b0, stream, nl_cnt, nl_head - blk_uoff[b0]);
}

void BgzfSource::report_k1_debug(uint32_t nb) { report_k1_debug(d_k1_ctr.p, nb); }
void BgzfSource::report_k1_debug(const uint32_t* ctr_dev, uint32_t nb) {
  if (!env_knobs().debug || !ctr_dev) return;
  uint32_t h[64];
  HIP_CHECK(hipMemcpy(h, ctr_dev, 256, hipMemcpyDeviceToHost));
  unsigned long long tc[5];
  memcpy(tc, h + 4, sizeof tc);  // dbg = ctr+1; cycle sums start at dbg+2 (8-byte aligned: ctr+3 -> see kernel) 
  fprintf(stderr, "[bioscan] inflate: %u members, %u rounds, %u decode passes (%.2f per round)\n", nb, h[2], h[3],
          h[2] ? (double)h[3] / h[2] : 0.0);
  unsigned long long tx[2];
  memcpy(tx, h + 28, sizeof tx);   // dbg + 26, dbg + 28: header parse alone, sync pass alone
  double tot = (double)tx[0] + (double)tx[1];
  for (int i = 0; i < 5; i++) tot += (double)tc[i];
  const char* nm[5] = {"table builds", "stage", "count + fix passes", "scan+write pass", "resolve"};
  fprintf(stderr, "[bioscan]   %-18s %6.2f %% of wave cycles\n", "header parse", tot ? 100.0 * (double)tx[0] / tot : 0.0);
  for (int i = 0; i < 5; i++) {
    fprintf(stderr, "[bioscan]   %-18s %6.2f %% of wave cycles\n", nm[i], tot ? 100.0 * (double)tc[i] / tot : 0.0);
    if (i == 1) fprintf(stderr, "[bioscan]   %-18s %6.2f %% of wave cycles\n", "sync pass", tot ? 100.0 * (double)tx[1] / tot : 0.0);
  }
  {
    // K1 v4 only: the write phase and the resolve split further (dbg + 32 ...)
    unsigned long long ty[8];
    memcpy(ty, h + 34, sizeof ty);
    const char* ny[8] = {"mini-round setup", "write loop", "resolve: entries", "resolve: far copy", "resolve: near walk", "resolve: flush", "-", "-"};
    if (ty[0] | ty[1] | ty[2])
      for (int i = 0; i < 6; i++) fprintf(stderr, "[bioscan]     %-18s %6.2f %% of wave cycles\n", ny[i], tot ? 100.0 * (double)ty[i] / tot : 0.0);
  }
  fprintf(stderr, "[bioscan]   write mini-rounds %u (%.2f per round), lanes idle behind END-OF-BLOCK %.1f per round, %u mini-rounds through HBM\n", h[24], h[2] ? (double)h[24] / h[2] : 0.0,
          h[2] ? (double)h[25] / h[2] : 0.0, h[26]);
  fprintf(stderr, "[bioscan]   LZ77 matches %u (%.1f per round), %.1f %% with a source inside the round's window\n", h[14], h[2] ? (double)h[14] / h[2] : 0.0,
          h[14] ? 100.0 * h[15] / h[14] : 0.0);
  if (h[20])  // -DV3_FIXSTAT builds only: how much of the wave each fix pass of the cascade re-decodes
    for (int k = 0; k < 4; k++)
      fprintf(stderr, "[bioscan]   fix pass %d%s: run in %.1f %% of rounds, %.2f lanes re-decoded per run\n", k + 1, k == 3 ? "+" : "",
              h[2] ? 100.0 * h[20 + k] / h[2] : 0.0, h[20 + k] ? (double)h[16 + k] / h[20 + k] : 0.0);
}

void BgzfSource::check_inflate_status(uint32_t b0, uint32_t nb) {
  // the first failing member is found on the device: 8 bytes come back instead of the whole status array
  DevBuf<uint32_t> res(1);
  launch_first_bad_status(d_status.p + b0, nb, res.p, stream);
  uint32_t i = 0xFFFFFFFFu;
  HIP_CHECK(hipMemcpyAsync(&i, res.p, 4, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  if (i != 0xFFFFFFFFu) {
    uint32_t st = 0;
    HIP_CHECK(hipMemcpy(&st, d_status.p + b0 + i, 4, hipMemcpyDeviceToHost));
    throw Error(std::string(what) + " read error: BGZF block " + std::to_string(b0 + i) + " at offset " +
                std::to_string(blk_coff[b0 + i]) + ": " + inflate_status_str(st & 0xFF) + " (code " + std::to_string(st) + ")");
  }
}

std::vector<uint8_t> BgzfSource::inflate_prefix_to_host(uint32_t b1) {
  // header / sampling: only the leading members are uploaded (a provider that will execute a far range of the file, or on
  // another device, never sends the whole file to this one)
  b1 = std::min(b1, n_blocks());
  auto img = build_image(device, 0, b1);  // private to this call: it is not what an execute will want resident
  K1Ctx c;
  init_ctx(c, *img, std::max<uint32_t>(b1, 1));
  uint64_t bytes = blk_uoff[b1];
  DevBuf<uint8_t> tmp(bytes + 64);
  launch_inflate(c, *img, tmp.p, b1, 0);
  launch_crc(c, *img, tmp.p, b1, 0);
  HIP_CHECK(hipStreamSynchronize(c.stream));
  check_inflate_status(c, 0, b1);
  std::vector<uint8_t> out(bytes);
  if (bytes) HIP_CHECK(hipMemcpy(out.data(), tmp.p, bytes, hipMemcpyDeviceToHost));
  return out;
}

}  // namespace bioscan
