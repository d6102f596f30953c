// kernels.hip -- record-boundary scan, row selection, field extract -> Arrow scatter, tags (gfx950).
//
// These kernels replace the per-record loop of the reference's BamExec
// (bio-format-bam/src/physical_exec.rs:408-573 sequential, :1269-1356 indexed) and the Arrow
// builders it drives (bio-format-core/src/alignment_utils.rs:383-644, sam_tag_io.rs:154-204,
// 658-1036).  All integer / byte work: no MFMA.  Layout rule: row i of a launch is lane
// (i & 63) of wave (i >> 6), so fixed-width columns and validity words are written fully
// coalesced and one wave owns exactly one 64-bit validity word per nullable column.
#include "kernels.h"
#include "f32_display.h"
#include "../../include/bioscan.h"

namespace bioscan {

#define WAVE 64

// BAM fields sit at arbitrary byte alignment: one unaligned dword / word load each (gfx950 global loads need no
// alignment) instead of four byte loads
struct __attribute__((packed, aligned(1))) ld_u32_t { uint32_t v; };
struct __attribute__((packed, aligned(1))) ld_u16_t { uint16_t v; };
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return ((const ld_u32_t*)p)->v; }
__device__ __forceinline__ int32_t ld_i32(const uint8_t* p) { return (int32_t)ld_u32(p); }
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return (uint32_t)((const ld_u16_t*)p)->v; }

// =================================================================================================
// exclusive scan u32 -> u64 (three-kernel, 2048 elements per block)
// =================================================================================================
constexpr int SCAN_T = 1024;  // 2 elements per thread: a lane stores 16 contiguous bytes, a wave 1 KiB
constexpr int SCAN_PER = 2;
constexpr int SCAN_BLOCK = SCAN_T * SCAN_PER;

size_t scan_tmp_elems(uint64_t n) { return (size_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK) + 2; }

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t lo = __shfl_up((uint32_t)v, d, WAVE);
    uint32_t hi = __shfl_up((uint32_t)(v >> 32), d, WAVE);
    uint64_t o = ((uint64_t)hi << 32) | lo;
    if (lane >= d) v += o;
  }
  return v;
}

// block-wide exclusive scan of per-thread totals; returns exclusive prefix, *total = block sum
__device__ uint64_t block_excl_scan(uint64_t v, uint64_t* total) {
  __shared__ uint64_t s_w[SCAN_T / WAVE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t inc = wave_incl_scan(v, lane);
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
  for (int i = 0; i < SCAN_T / WAVE; i++) {
    if (i < w) base += s_w[i];
    tot += s_w[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// Reduce-then-scan: pass 1 only sums each tile (4 B read per element), pass 2 scans the tile sums, pass 3 re-reads the
// input, scans the tile in registers and writes the final offsets (4 B read + 8 B written): 16 B of traffic per element
// instead of 28 for "local scan, then add the tile base to every element".
__global__ __launch_bounds__(SCAN_T) void k_scan_local(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ block_sums) {
  const uint64_t b0 = (uint64_t)blockIdx.x * SCAN_BLOCK;
  const uint64_t t0 = b0 + (uint64_t)threadIdx.x * SCAN_PER;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++) {
    uint64_t i = t0 + k;
    s += i < n ? in[i] : 0u;
  }
  uint64_t tot;
  (void)block_excl_scan(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block sums in place; block_sums[nb] = grand total
__global__ __launch_bounds__(SCAN_T) void k_scan_sums(uint64_t* block_sums, uint64_t nb) {
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t base = 0; base < nb; base += SCAN_T) {
    uint64_t i = base + threadIdx.x;
    uint64_t v = i < nb ? block_sums[i] : 0;
    uint64_t tot;
    uint64_t ex = block_excl_scan(v, &tot);
    uint64_t c = carry_s;
    if (i < nb) block_sums[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[nb] = carry_s;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_add(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n,
                                                      const uint64_t* __restrict__ block_sums, uint64_t nb) {
  const uint64_t b0 = (uint64_t)blockIdx.x * SCAN_BLOCK;
  const uint64_t t0 = b0 + (uint64_t)threadIdx.x * SCAN_PER;
  uint32_t v[SCAN_PER];
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++) {
    uint64_t i = t0 + k;
    v[k] = i < n ? in[i] : 0u;
    s += v[k];
  }
  uint64_t tot;
  uint64_t ex = block_excl_scan(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++) {
    uint64_t i = t0 + k;
    if (i < n) out[i] = ex;
    ex += v[k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_sums[nb];
}

void launch_exclusive_scan_u32_to_u64(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t st) {
  if (n == 0) {
    hipMemsetAsync(out, 0, sizeof(uint64_t), st);
    return;
  }
  uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  hipLaunchKernelGGL(k_scan_local, dim3((uint32_t)nb), dim3(SCAN_T), 0, st, in, n, tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, st, tmp, nb);
  hipLaunchKernelGGL(k_scan_add, dim3((uint32_t)nb), dim3(SCAN_T), 0, st, in, out, n, tmp, nb);
}

// =================================================================================================
// K3: record boundary scan.  The record chain (block_size prefixes) is a linked list through the
// whole inflated stream.  It is cut into SEG_BYTES segments; each segment's entry (first record
// start >= segment begin) is GUESSED with a plausibility test, every segment is walked in
// parallel from its guess, and the guesses are VERIFIED exactly: entry[s] must equal the exit of
// the nearest preceding non-empty segment.  Segment 0's entry is exact (end of the BAM header),
// so by induction a verified chain is the true chain.  Mismatches are corrected and re-walked
// until none remain (normally zero iterations).
// =================================================================================================
__device__ __forceinline__ bool plausible_rec(const uint8_t* u, uint64_t ulen, uint64_t p, int32_t n_ref, uint64_t* next) {
  if (p + 36 > ulen) return false;
  const uint8_t* r = u + p;
  int32_t bs = ld_i32(r);
  if (bs < 32) return false;
  if (p + 4 + (uint64_t)bs > ulen) return false;
  int32_t refid = ld_i32(r + 4), pos = ld_i32(r + 8);
  uint32_t lrn = r[12];
  uint32_t ncig = ld_u16(r + 16);
  int32_t lseq = ld_i32(r + 20);
  int32_t nref = ld_i32(r + 24), npos = ld_i32(r + 28);
  if (refid < -1 || refid >= n_ref || nref < -1 || nref >= n_ref) return false;
  if (pos < -1 || npos < -1 || lseq < 0 || lrn == 0) return false;
  uint64_t need = 32ull + lrn + 4ull * ncig + (uint64_t)((lseq + 1) / 2) + (uint64_t)lseq;
  if (need > (uint64_t)bs) return false;
  if (r[36 + lrn - 1] != 0) return false;
  // read names are [!-?A-~]{1,254} (SAM spec 1.4): a cheap, very selective test.  It only steers the
  // GUESS; a file with exotic names still decodes exactly through the verify/fix loop.
  for (uint32_t k = 0; k + 1 < lrn; k++) {
    uint8_t c = r[36 + k];
    if (c < 0x21 || c > 0x7E) return false;
  }
  *next = p + 4 + (uint64_t)bs;
  return true;
}

__global__ __launch_bounds__(WAVE) void k_seg_guess(const uint8_t* __restrict__ u, uint64_t ulen, uint64_t first_rec,
                                                     uint64_t nseg, int32_t n_ref, uint64_t* __restrict__ entry) {
  const uint64_t s = blockIdx.x;
  if (s >= nseg) return;
  const int lane = threadIdx.x;
  const uint64_t B = s * SEG_BYTES;
  uint64_t E = B + SEG_BYTES;
  if (E > ulen) E = ulen;
  uint64_t res = SEG_NONE;
  if (first_rec >= E) {
    res = SEG_NONE;  // still inside the BAM header
  } else if (first_rec >= B) {
    res = first_rec;  // exact anchor
  } else {
    for (uint64_t p0 = B; p0 < E; p0 += WAVE) {
      uint64_t p = p0 + lane;
      uint64_t nx = 0, nx2 = 0;
      bool ok = p < E && plausible_rec(u, ulen, p, n_ref, &nx);
      if (ok) ok = (nx == ulen) || plausible_rec(u, ulen, nx, n_ref, &nx2);
      unsigned long long m = __ballot(ok);
      if (m) {
        res = p0 + (uint64_t)__builtin_ctzll(m);
        break;
      }
    }
  }
  if (lane == 0) entry[s] = res;
}

__global__ void k_seg_walk(const uint8_t* __restrict__ u, uint64_t ulen, uint64_t nseg, ChainBuffers cb, int only_dirty, int allow_partial) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  if (only_dirty && !cb.dirty[s]) return;
  const uint64_t B = s * (uint64_t)SEG_BYTES;
  uint64_t E = B + SEG_BYTES;
  if (E > ulen) E = ulen;
  uint64_t p = cb.entry[s];
  uint32_t cnt = 0;
  uint64_t ex = SEG_NONE;
  // the record starts found on the way go into the segment's slot (segment-relative, SEG_SLOT entries: a record is at
  // least 36 bytes); k_seg_gather copies the verified ones into the dense table, so the chain is walked once, not twice
  uint32_t* slot = cb.starts + s * (uint64_t)SEG_SLOT;
  if (p != SEG_NONE) {
    ex = p;
    while (ex < E) {
      // allow_partial: the buffer is one chunk of a longer stream; a record that runs past its end belongs to the next
      // chunk (the caller carries its bytes over).  SEG_PARTIAL | start is larger than any segment end, so every later
      // segment is expected to hold no record start, exactly as behind a record that covers them.
      if (ex + 4 > ulen) { ex = allow_partial ? (SEG_PARTIAL | ex) : SEG_BAD; break; }
      int32_t bs = ld_i32(u + ex);
      if (bs < 32) { ex = SEG_BAD; break; }
      if (ex + 4 + (uint64_t)bs > ulen) { ex = allow_partial ? (SEG_PARTIAL | ex) : SEG_BAD; break; }
      slot[cnt] = (uint32_t)(ex - B);
      ex += 4 + (uint64_t)bs;
      cnt++;
    }
  }
  cb.exit_[s] = ex;
  cb.count[s] = cnt;
  cb.dirty[s] = 0;
}

__global__ void k_seg_verify(uint64_t ulen, uint64_t first_rec, uint64_t nseg, ChainBuffers cb) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const uint64_t first_seg = first_rec / SEG_BYTES;
  if (s <= first_seg) return;  // exact anchor / header segments
  const uint64_t B = s * (uint64_t)SEG_BYTES;
  uint64_t E = B + SEG_BYTES;
  if (E > ulen) E = ulen;
  uint64_t t = s - 1;
  while (t > first_seg && cb.exit_[t] == SEG_NONE) t--;
  const uint64_t e = cb.exit_[t];
  // A BAD predecessor exit means its entry is a false positive that will be corrected this round
  // (or the data is corrupt, which is reported after convergence): do not judge this segment yet.
  if (e == SEG_BAD) return;
  uint64_t expect;
  if (e == SEG_NONE) expect = SEG_NONE;          // nothing upstream has a record yet (only inside the header)
  else if (e < B) return;                        // a segment between t and s must start at e but is NONE right now:
                                                 // it is corrected this round; judge s once its exit is known
  else if (e < E) expect = e;
  else expect = SEG_NONE;                        // a long record covers this whole segment (or the stream ended)
  if (cb.entry[s] != expect) {
    cb.entry[s] = expect;
    cb.dirty[s] = 1;
    atomicAdd(cb.nfix, 1u);
  }
}

// dense record table from the segments' slots: one wave per segment, coalesced on both sides
__global__ __launch_bounds__(256) void k_seg_gather(uint64_t nseg, ChainBuffers cb, const uint64_t* __restrict__ base,
                                                    uint64_t* __restrict__ rec_off) {
  const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= nseg) return;
  const int lane = threadIdx.x & 63;
  if (cb.entry[s] == SEG_NONE) return;
  if (cb.exit_[s] == SEG_BAD) {
    if (lane == 0) atomicExch(cb.err, 1u);
    return;
  }
  const uint32_t n = cb.count[s];
  const uint64_t B = s * (uint64_t)SEG_BYTES, o = base[s];
  const uint32_t* slot = cb.starts + s * (uint64_t)SEG_SLOT;
  for (uint32_t k = lane; k < n; k += 64) rec_off[o + k] = B + slot[k];
}

void launch_seg_guess(const uint8_t* u, uint64_t ulen, uint64_t first_rec, uint64_t nseg, int32_t n_ref, ChainBuffers cb, hipStream_t st) {
  hipLaunchKernelGGL(k_seg_guess, dim3((uint32_t)nseg), dim3(WAVE), 0, st, u, ulen, first_rec, nseg, n_ref, cb.entry);
}
void launch_seg_walk(const uint8_t* u, uint64_t ulen, uint64_t nseg, ChainBuffers cb, int only_dirty, int allow_partial, hipStream_t st) {
  hipLaunchKernelGGL(k_seg_walk, dim3((uint32_t)((nseg + 63) / 64)), dim3(64), 0, st, u, ulen, nseg, cb, only_dirty, allow_partial);
}
void launch_seg_verify(uint64_t ulen, uint64_t first_rec, uint64_t nseg, ChainBuffers cb, hipStream_t st) {
  hipLaunchKernelGGL(k_seg_verify, dim3((uint32_t)((nseg + 63) / 64)), dim3(64), 0, st, ulen, first_rec, nseg, cb);
}
void launch_seg_gather(uint64_t nseg, ChainBuffers cb, const uint64_t* base, uint64_t* rec_off, hipStream_t st) {
  hipLaunchKernelGGL(k_seg_gather, dim3((uint32_t)((nseg + 3) / 4)), dim3(256), 0, st, nseg, cb, base, rec_off);
}

// exit of the last segment that holds a record start: res[0] = index + 1 of that segment (0: none), res[1] = its exit.
// (The host used to copy the whole exit array back for this: 5 MB of pageable D2H per scan of config 2.)
__global__ void k_last_exit_idx(const uint64_t* __restrict__ exit_, uint64_t nseg, unsigned long long* res) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool hit = k < nseg && exit_[k] != SEG_NONE;
  const unsigned long long m = __ballot(hit);
  if (m && (threadIdx.x & 63) == 0) atomicMax(res, (unsigned long long)(k + 64 - __builtin_clzll(m)));
}
__global__ void k_last_exit_val(const uint64_t* __restrict__ exit_, unsigned long long* res) { res[1] = res[0] ? exit_[res[0] - 1] : SEG_NONE; }
void launch_last_exit(const uint64_t* exit_, uint64_t nseg, unsigned long long* res, hipStream_t st) {
  hipMemsetAsync(res, 0, 16, st);
  hipLaunchKernelGGL(k_last_exit_idx, dim3((uint32_t)((nseg + 255) / 256)), dim3(256), 0, st, exit_, nseg, res);
  hipLaunchKernelGGL(k_last_exit_val, dim3(1), dim3(1), 0, st, exit_, res);
}
// first BGZF member whose inflate status is not INF_OK: res[0] = index + 1 (0: all fine)
__global__ void k_first_bad_status(const uint32_t* __restrict__ status, uint32_t n, uint32_t* res) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n && status[k] != 0u) atomicMin(res, k);
}
void launch_first_bad_status(const uint32_t* status, uint32_t n, uint32_t* res, hipStream_t st) {
  hipMemsetAsync(res, 0xFF, 4, st);
  if (n) hipLaunchKernelGGL(k_first_bad_status, dim3((n + 255) / 256), dim3(256), 0, st, status, n, res);
}

// =================================================================================================
// record key table + row selection
// =================================================================================================
__device__ __forceinline__ uint32_t cigar_ref_span(const uint8_t* cig, uint32_t n) {
  uint32_t span = 0;
  for (uint32_t k = 0; k < n; k++) {
    uint32_t v = ld_u32(cig + 4 * k);
    uint32_t op = v & 15u;
    // M(0) D(2) N(3) =(7) X(8) consume the reference
    if ((0x18Du >> op) & 1u) span += v >> 4;
  }
  return span;
}

// 1-based inclusive end; 0 encodes None.  noodles: end = start + span - 1, Position::new(0) = None.
__device__ __forceinline__ uint32_t rec_end1(const uint8_t* r) {
  int32_t pos = ld_i32(r + 8);
  if (pos < 0) return 0;
  uint32_t lrn = r[12];
  uint32_t ncig = ld_u16(r + 16);
  uint32_t span = cigar_ref_span(r + 36 + lrn, ncig);
  return (uint32_t)pos + span;  // (pos+1) + span - 1
}

__global__ void k_rec_keys(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rec_off, uint64_t n, RecKeys k, uint32_t* err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = u + rec_off[i];
  {
    // the first kernel of an indexed scan that interprets a record: its variable-length fields must fit in block_size
    // (bam_rows.hip makes the same check for scans that build no key table)
    const uint32_t bs = ld_u32(r), lrn = r[12], ncig = ld_u16(r + 16);
    const int32_t lseq = ld_i32(r + 20);
    if (lrn == 0 || lseq < 0 || 32ull + lrn + 4ull * ncig + (((uint64_t)(uint32_t)lseq + 1) >> 1) + (uint64_t)(uint32_t)lseq > (uint64_t)bs) {
      atomicExch(err, 8u);
      k.refid[i] = -1; k.pos[i] = -1; k.end1[i] = 0; k.flag_mapq[i] = 0;
      return;
    }
  }
  k.refid[i] = ld_i32(r + 4);
  k.pos[i] = ld_i32(r + 8);
  k.end1[i] = (int32_t)rec_end1(r);
  k.flag_mapq[i] = ld_u16(r + 18) | ((uint32_t)r[13] << 16);
}
void launch_rec_keys(const uint8_t* u, const uint64_t* rec_off, uint64_t n, RecKeys k, uint32_t* err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_rec_keys, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rec_off, n, k, err);
}

// residual filter evaluation (bio-format-core/src/record_filter.rs:57-283 with BamRecordFields,
// bio-format-bam/src/storage.rs:456-494): numeric compares in f64, NULL fields pass through.
__device__ bool eval_terms(const FilterTerm* t, int n, bool has_chrom, int32_t chrom_ref, bool has_start, uint32_t start,
                           bool has_end, uint32_t end, uint32_t mapq, uint32_t flags) {
  for (int k = 0; k < n; k++) {
    const FilterTerm& f = t[k];
    bool have;
    double v;
    switch (f.field) {
      case 0: have = has_chrom; v = (double)chrom_ref; break;
      case 1: have = has_start; v = (double)start; break;
      case 2: have = has_end; v = (double)end; break;
      case 3: have = true; v = (double)mapq; break;
      default: have = true; v = (double)flags; break;
    }
    if (!have) {          // field not found on record -> pass through (all terms of a long list)
      while (t[k].more && k + 1 < n) k++;
      continue;
    }
    bool ok = true;
    switch (f.op) {
      case BIOSCAN_OP_EQ: ok = v == f.vals[0]; break;
      case BIOSCAN_OP_NE: ok = v != f.vals[0]; break;
      case BIOSCAN_OP_LT: ok = v < f.vals[0]; break;
      case BIOSCAN_OP_LE: ok = v <= f.vals[0]; break;
      case BIOSCAN_OP_GT: ok = v > f.vals[0]; break;
      case BIOSCAN_OP_GE: ok = v >= f.vals[0]; break;
      case BIOSCAN_OP_BETWEEN: ok = v >= f.vals[0] && v <= f.vals[1]; break;
      case BIOSCAN_OP_NOT_BETWEEN: ok = !(v >= f.vals[0] && v <= f.vals[1]); break;
      case BIOSCAN_OP_IN:
      case BIOSCAN_OP_NOT_IN: {
        // a list longer than eight literals spans consecutive terms (`more`): the verdict is taken over the whole list
        bool hit = false, has_null = false;
        for (;;) {
          const FilterTerm& g = t[k];
          for (int j = 0; j < g.n_vals; j++) hit = hit || (v == g.vals[j]);
          has_null = has_null || g.has_null;
          if (!g.more || k + 1 >= n) break;
          k++;
        }
        const bool neg = f.op == BIOSCAN_OP_NOT_IN;
        ok = hit ? !neg : (!has_null && neg);
      } break;
    }
    if (!ok) return false;
  }
  return true;
}

__global__ void k_row_flags(RecKeys k, uint64_t n, RowSelect sel, const FilterTerm* __restrict__ terms, uint32_t* __restrict__ keep, int accumulate) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t refid = k.refid[i], pos = k.pos[i];
  const uint32_t end1 = (uint32_t)k.end1[i];
  const uint32_t fm = k.flag_mapq[i];
  bool kp = false;
  bool has_chrom = false, has_start = false, has_end = false;
  int32_t chrom_ref = -1;
  uint32_t start_out = 0;
  if (sel.mode == 0) {
    kp = true;
  } else if (sel.mode == 1) {
    // noodles query intersects() then the reference's sub-region dedup (physical_exec.rs:1280-1314)
    if (refid == sel.ref && pos >= 0 && end1 != 0) {
      const int64_t s1 = (int64_t)pos + 1;
      const bool inter = sel.q_start1 <= (int64_t)end1 && s1 <= sel.end1;
      const bool dedup = s1 >= sel.start1 && s1 <= sel.end1;
      kp = inter && dedup;
      has_chrom = true; chrom_ref = refid;
      has_start = true; start_out = sel.zero_based ? (uint32_t)pos : (uint32_t)pos + 1u;
      has_end = true;
    } else if (refid == sel.ref && pos >= 0 && end1 == 0) {
      kp = false;  // alignment_end() == None -> intersects() false
    }
  } else if (sel.mode == 2) {
    kp = i >= sel.i_lo && i < sel.i_hi && refid == sel.ref && pos < 0;
    has_chrom = true; chrom_ref = sel.ref;
  } else {
    kp = refid < 0 && pos < 0;
  }
  if (kp && sel.n_terms)
    kp = eval_terms(terms, sel.n_terms, has_chrom, chrom_ref, has_start, start_out, has_end, end1, fm >> 16, fm & 0xFFFFu);
  if (accumulate) { if (kp) keep[i] = 1u; }   // a further region of the same decode (regions are disjoint)
  else keep[i] = kp ? 1u : 0u;
}
// One record's verdict for one selection (shared by the key-table form above and the fused form below)
// is the absolute inflated offset `apos` inside one of the region's chunks?  (sorted, disjoint [begin, end) pairs)
__device__ __forceinline__ bool in_region_chunks(const RowSelect& sel, const uint64_t* __restrict__ tab, uint64_t apos) {
  if (!tab || sel.ch_n == 0) return true;
  const uint64_t* t = tab + 2ull * sel.ch_lo;
  uint32_t lo = 0, hi = sel.ch_n;   // first chunk whose begin is > apos
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (t[2 * mid] <= apos) lo = mid + 1; else hi = mid;
  }
  return lo > 0 && apos < t[2 * (lo - 1) + 1];
}

__device__ __forceinline__ bool row_verdict(const RowSelect& sel, uint64_t i, int32_t refid, int32_t pos, uint32_t end1, uint32_t fm,
                                            const FilterTerm* __restrict__ terms, const uint64_t* __restrict__ chunk_tab = nullptr,
                                            uint64_t apos = 0) {
  bool kp = false;
  bool has_chrom = false, has_start = false, has_end = false;
  int32_t chrom_ref = -1;
  uint32_t start_out = 0;
  if (sel.mode == 0) {
    kp = true;
  } else if (sel.mode == 1) {
    if (refid == sel.ref && pos >= 0 && end1 != 0) {
      const int64_t s1 = (int64_t)pos + 1;
      const bool inter = sel.q_start1 <= (int64_t)end1 && s1 <= sel.end1;
      const bool dedup = s1 >= sel.start1 && s1 <= sel.end1;
      kp = inter && dedup && in_region_chunks(sel, chunk_tab, apos);
      has_chrom = true; chrom_ref = refid;
      has_start = true; start_out = sel.zero_based ? (uint32_t)pos : (uint32_t)pos + 1u;
      has_end = true;
    }
  } else if (sel.mode == 2) {
    kp = i >= sel.i_lo && i < sel.i_hi && refid == sel.ref && pos < 0;
    has_chrom = true; chrom_ref = sel.ref;
  } else {
    kp = refid < 0 && pos < 0;
  }
  if (kp && sel.n_terms)
    kp = eval_terms(terms, sel.n_terms, has_chrom, chrom_ref, has_start, start_out, has_end, end1, fm >> 16, fm & 0xFFFFu);
  return kp;
}

// Fused form for region and no-coor items: the record's keys (reference, position, end from the CIGAR, flag, mapq) stay in
// registers and every region of the decode is tried at once (regions are disjoint: OR).  It replaces k_rec_keys + one
// k_row_flags per region + the 16-byte-per-record key table between them (config 2 as a BAI plan: 9.4 + 13.2 ms).
__global__ void k_row_flags_rec(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rec_off, uint64_t n, const RowSelect* __restrict__ sels,
                                int n_sel, const FilterTerm* __restrict__ terms, uint32_t* __restrict__ keep, uint32_t* err,
                                const uint64_t* __restrict__ chunk_tab, uint64_t buf_base_abs) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = u + rec_off[i];
  // the 36-byte header as two 16-byte loads (the eight fields read one by one were eight dependent-free but separate requests
  // to lines 300 bytes apart)
  struct __attribute__((packed, aligned(1))) hdr16 { uint32_t x, y, z, w; };
  const hdr16 h0 = *(const hdr16*)r, h1 = *(const hdr16*)(r + 16);
  const uint32_t bs = h0.x, lrn = h0.w & 0xFFu, ncig = h1.x & 0xFFFFu;
  const int32_t lseq = (int32_t)h1.y;
  if (lrn == 0 || lseq < 0 || 32ull + lrn + 4ull * ncig + (((uint64_t)(uint32_t)lseq + 1) >> 1) + (uint64_t)(uint32_t)lseq > (uint64_t)bs) {
    atomicExch(err, 8u);
    keep[i] = 0u;
    return;
  }
  const int32_t refid = (int32_t)h0.y, pos = (int32_t)h0.z;
  const uint32_t fm = (h1.x >> 16) | (((h0.w >> 8) & 0xFFu) << 16);
  // The end needs the CIGAR (a dependent load on another line).  A region's overlap test asks `q_start1 <= end`: a record that
  // starts behind the query's start passes it whatever its CIGAR says (end >= start - 1 >= q_start1), so only a record that starts
  // at or before some region's query start on that region's reference pays for the CIGAR -- or one whose residual terms look at
  // `end`.  (In a coordinate-sorted file those are the few records in front of each region.)
  bool want_end = false;
  for (int k = 0; k < n_sel; k++)
    want_end = want_end || (sels[k].mode == 1 && sels[k].ref == refid && (sels[k].n_terms != 0 || (int64_t)pos + 1 <= sels[k].q_start1));
  const uint32_t end1 = want_end ? rec_end1(r) : (pos >= 0 ? (uint32_t)pos + 1u : 0u);
  bool kp = false;
  const uint64_t apos = buf_base_abs + rec_off[i];
  for (int k = 0; k < n_sel && !kp; k++) kp = row_verdict(sels[k], i, refid, pos, end1, fm, terms, chunk_tab, apos);
  keep[i] = kp ? 1u : 0u;
}
void launch_row_flags_rec(const uint8_t* u, const uint64_t* rec_off, uint64_t n, const RowSelect* sels_dev, int n_sel, const FilterTerm* terms_dev,
                          uint32_t* keep, uint32_t* err, hipStream_t st, const uint64_t* chunk_tab, uint64_t buf_base_abs) {
  if (!n) return;
  hipLaunchKernelGGL(k_row_flags_rec, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rec_off, n, sels_dev, n_sel, terms_dev, keep, err,
                     chunk_tab, buf_base_abs);
}

void launch_row_flags(RecKeys k, uint64_t n, RowSelect sel, const FilterTerm* terms_dev, uint32_t* keep, int accumulate, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_row_flags, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, k, n, sel, terms_dev, keep, accumulate);
}

__global__ void k_compact_rows(const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ keep,
                               const uint64_t* __restrict__ keep_scan, uint64_t n, uint64_t* __restrict__ rows, uint64_t row_base) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep[i]) rows[row_base + keep_scan[i]] = rec_off[i];
}
void launch_compact_rows(const uint64_t* rec_off, const uint32_t* keep, const uint64_t* keep_scan, uint64_t n, uint64_t* rows,
                         uint64_t row_base, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_compact_rows, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, rec_off, keep, keep_scan, n, rows, row_base);
}

__global__ void k_find_first(const int32_t* __restrict__ refid, uint64_t n, uint64_t from, int32_t ref, int want_equal,
                             unsigned long long* result) {
  const uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool eq = refid[i] == ref;
  if (eq == (want_equal != 0)) atomicMin(result, (unsigned long long)i);
}
void launch_find_first(const int32_t* refid, uint64_t n, uint64_t from, int32_t ref, int want_equal, unsigned long long* result, hipStream_t st) {
  if (from >= n) return;
  uint64_t m = n - from;
  hipLaunchKernelGGL(k_find_first, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, st, refid, n, from, ref, want_equal, result);
}
__global__ void k_lower_bound(const uint64_t* __restrict__ arr, uint64_t n, uint64_t key, unsigned long long* result) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (arr[mid] < key) lo = mid + 1; else hi = mid;
  }
  *result = lo;
}
void launch_lower_bound_u64(const uint64_t* arr, uint64_t n, uint64_t key, unsigned long long* result, hipStream_t st) {
  hipLaunchKernelGGL(k_lower_bound, dim3(1), dim3(1), 0, st, arr, n, key, result);
}

// =================================================================================================
// per-batch offsets of columns that keep an offset array (tags, the wide-quality path); the twelve core columns are
// written by bam_rows.hip
// =================================================================================================
// per-batch int32 offsets of a chunk of rows whose first `phase` batch slots were filled by the previous chunk: batch b
// covers rows [max(0, b*bs - phase), (b+1)*bs - phase) of the chunk, off32[b*(bs+1) + j] = off64[start_b + j] - off64[start_b]
__device__ __forceinline__ uint64_t batch_start_row(uint64_t b, uint32_t bs, uint32_t phase) { return b ? b * bs - phase : 0; }
__global__ void k_batch_offsets(const uint64_t* __restrict__ off64, uint64_t n_rows, uint32_t bs, uint32_t phase, int32_t* __restrict__ off32) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t nb = (n_rows + phase + bs - 1) / bs;
  const uint64_t per = (uint64_t)bs + 1;
  if (t >= nb * per) return;
  const uint64_t b = t / per, r = t % per;
  const uint64_t start = batch_start_row(b, bs, phase);
  uint64_t row = start + r;
  if (row > n_rows) row = n_rows;
  off32[t] = (int32_t)(off64[row] - off64[start]);
}
void launch_batch_offsets(const uint64_t* off64, uint64_t n_rows, uint32_t batch_size, uint32_t phase, int32_t* off32, hipStream_t st) {
  if (!n_rows) return;
  uint64_t nb = (n_rows + phase + batch_size - 1) / batch_size;
  uint64_t tot = nb * ((uint64_t)batch_size + 1);
  hipLaunchKernelGGL(k_batch_offsets, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, st, off64, n_rows, batch_size, phase, off32);
}

// base[b] = off64[first row of batch b]: first byte / element of every batch (host export builds zero-copy windows from it)
__global__ void k_batch_bases(const uint64_t* __restrict__ off64, uint64_t nb, uint32_t bs, uint32_t phase, uint64_t* __restrict__ base) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) base[b] = off64[batch_start_row(b, bs, phase)];
}
void launch_batch_bases(const uint64_t* off64, uint64_t nb, uint32_t bs, uint32_t phase, uint64_t* base, hipStream_t st) {
  if (!nb) return;
  hipLaunchKernelGGL(k_batch_bases, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, st, off64, nb, bs, phase, base);
}

// exact wide-quality path: `char::from(q + 33)` pushed into a String -> chars >= U+0080 take two UTF-8 bytes
__global__ void k_qual_wide_len(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t row0, uint64_t n,
                                uint32_t* __restrict__ len_qual) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n) return;
  const uint8_t* r = u + rows[li];
  const uint32_t lrn = r[12], ncig = ld_u16(r + 16);
  const int32_t lseq = ld_i32(r + 20);
  const uint8_t* q = r + 36 + lrn + 4 * ncig + (lseq + 1) / 2;
  uint32_t l = 0;
  for (int32_t k = 0; k < lseq; k++) l += ((((uint32_t)q[k] + 33u) & 0xFFu) >= 128u) ? 2u : 1u;
  len_qual[row0 + li] = l;
}
void launch_qual_wide_len(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, uint32_t* len_qual, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_qual_wide_len, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, row0, n, len_qual);
}
__global__ void k_qual_wide_scatter(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t n,
                                    const uint64_t* __restrict__ off64, uint8_t* __restrict__ dst) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = u + rows[i];
  const uint32_t lrn = r[12], ncig = ld_u16(r + 16);
  const int32_t lseq = ld_i32(r + 20);
  const uint8_t* q = r + 36 + lrn + 4 * ncig + (lseq + 1) / 2;
  uint8_t* d = dst + off64[i];
  for (int32_t k = 0; k < lseq; k++) {
    uint32_t c = ((uint32_t)q[k] + 33u) & 0xFFu;
    if (c < 128u) *d++ = (uint8_t)c;
    else { *d++ = (uint8_t)(0xC0u | (c >> 6)); *d++ = (uint8_t)(0x80u | (c & 0x3Fu)); }
  }
}
void launch_qual_wide_scatter(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint64_t* off64, uint8_t* dst, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_qual_wide_scatter, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, n, off64, dst);
}

// =================================================================================================
// K8: tags.  One record per lane walks the aux fields once and records where each requested tag's
// value lives; typed column kernels then read straight from those locations.
// =================================================================================================
__device__ __forceinline__ uint32_t aux_elem_size(uint8_t t) {
  switch (t) {
    case 'c': case 'C': case 'A': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    default: return 0;
  }
}

__global__ void k_tag_locate(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t n,
                             const uint16_t* __restrict__ tags, int32_t n_tags, uint32_t* __restrict__ loc,
                             uint8_t* __restrict__ typ, uint32_t* err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = u + rows[i];
  const uint32_t bs = ld_u32(r);
  const uint32_t lrn = r[12], ncig = ld_u16(r + 16);
  const int32_t lseq = ld_i32(r + 20);
  uint32_t o = 36 + lrn + 4 * ncig + (uint32_t)((lseq + 1) / 2) + (uint32_t)lseq;
  const uint32_t end = 4 + bs;
  for (int t = 0; t < n_tags; t++) typ[(uint64_t)t * n + i] = 0;
  while (o + 3 <= end) {
    const uint32_t tag = ld_u16(r + o);
    const uint8_t ty = r[o + 2];
    const uint32_t vo = o + 3;
    uint32_t sz;
    if (ty == 'Z' || ty == 'H') {
      uint32_t k = vo;
      while (k < end && r[k] != 0) k++;
      if (k >= end) { atomicExch(err, 4u); return; }
      sz = k - vo + 1;
    } else if (ty == 'B') {
      if (vo + 5 > end) { atomicExch(err, 4u); return; }
      uint32_t es = aux_elem_size(r[vo]);
      uint32_t cnt = ld_u32(r + vo + 1);
      if (es == 0 || r[vo] == 'A') { atomicExch(err, 4u); return; }
      sz = 5 + es * cnt;
    } else {
      sz = aux_elem_size(ty);
      if (sz == 0) { atomicExch(err, 4u); return; }
    }
    if (vo + sz > end) { atomicExch(err, 4u); return; }
    for (int t = 0; t < n_tags; t++) {
      if (tags[t] == tag) {
        const uint64_t x = (uint64_t)t * n + i;
        if (typ[x] != 0) atomicExch(err, 5u);  // duplicate tag: the reference appends twice (row misalignment)
        loc[x] = vo;
        typ[x] = ty;
      }
    }
    o = vo + sz;
  }
}
void launch_tag_locate(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint16_t* tags_dev, int32_t n_tags, uint32_t* loc,
                       uint8_t* typ, uint32_t* err, hipStream_t st) {
  if (!n || !n_tags) return;
  hipLaunchKernelGGL(k_tag_locate, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, n, tags_dev, n_tags, loc, typ, err);
}

// integer value of an aux scalar as i64 (c,C,s,S,i,I,A)
__device__ __forceinline__ int64_t aux_int(const uint8_t* p, uint8_t ty) {
  switch (ty) {
    case 'c': return (int8_t)p[0];
    case 'C': case 'A': return p[0];
    case 's': return (int16_t)ld_u16(p);
    case 'S': return ld_u16(p);
    case 'i': return ld_i32(p);
    default: return (int64_t)ld_u32(p);  // 'I'
  }
}

// sam_tag_io.rs:658-742: Int32 / UInt32 / Float32 builders
__global__ void k_tag_fixed(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t row0, uint64_t n,
                            const uint32_t* __restrict__ loc, const uint8_t* __restrict__ typ, int32_t kind,
                            uint32_t* __restrict__ values, uint64_t* __restrict__ valid, uint32_t* err) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = li < n;
  const uint64_t i = row0 + li;
  bool v = false;
  uint32_t out = 0;
  if (act) {
    const uint8_t ty = typ[li];
    if (ty) {
      const uint8_t* p = u + rows[li] + loc[li];
      if (ty == 'f') {
        if (kind == TAG_FLOAT32) { out = ld_u32(p); v = true; }
        else atomicExch(err, 6u);  // append_float on an integer builder: type mismatch
      } else if (ty == 'Z' || ty == 'H' || ty == 'B') {
        atomicExch(err, 6u);
      } else {
        const int64_t x = aux_int(p, ty);
        if (kind == TAG_UINT32) {
          if (x < 0 || x > 0xFFFFFFFFll) atomicExch(err, 7u); else { out = (uint32_t)x; v = true; }
        } else if (kind == TAG_INT32) {
          if (x < -2147483648ll || x > 2147483647ll) atomicExch(err, 7u); else { out = (uint32_t)(int32_t)x; v = true; }
        } else {
          atomicExch(err, 6u);  // integer into a Float32 builder
        }
      }
    }
    values[i] = out;
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && act) valid[i >> 6] = m;
}
void launch_tag_fixed(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                      int32_t kind, uint32_t* values, uint64_t* valid, uint32_t* err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_tag_fixed, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, row0, n, loc, typ, kind, values, valid, err);
}

// UTF-8 helpers
__device__ __forceinline__ uint32_t utf8_len_cp(uint32_t cp) { return cp < 0x80 ? 1 : cp < 0x800 ? 2 : cp < 0x10000 ? 3 : 4; }
__device__ __forceinline__ uint32_t utf8_write_cp(uint8_t* d, uint32_t cp) {
  if (cp < 0x80) { d[0] = (uint8_t)cp; return 1; }
  if (cp < 0x800) { d[0] = (uint8_t)(0xC0 | (cp >> 6)); d[1] = (uint8_t)(0x80 | (cp & 0x3F)); return 2; }
  if (cp < 0x10000) { d[0] = (uint8_t)(0xE0 | (cp >> 12)); d[1] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); d[2] = (uint8_t)(0x80 | (cp & 0x3F)); return 3; }
  d[0] = (uint8_t)(0xF0 | (cp >> 18)); d[1] = (uint8_t)(0x80 | ((cp >> 12) & 0x3F)); d[2] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); d[3] = (uint8_t)(0x80 | (cp & 0x3F));
  return 4;
}
__device__ bool utf8_valid(const uint8_t* s, uint32_t n) {
  uint32_t i = 0;
  while (i < n) {
    uint8_t c = s[i];
    if (c < 0x80) { i++; continue; }
    uint32_t need; uint32_t cp;
    if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1F; if (cp < 2) return false; }
    else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0F; }
    else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07; if (cp > 4) return false; }
    else return false;
    if (i + need >= n) return false;
    for (uint32_t k = 1; k <= need; k++) {
      uint8_t d = s[i + k];
      if ((d & 0xC0) != 0x80) return false;
      cp = (cp << 6) | (d & 0x3F);
    }
    if (need == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
    if (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
    i += need + 1;
  }
  return true;
}
__device__ __forceinline__ bool valid_scalar_cp(int64_t x) { return x >= 0 && x <= 0x10FFFF && !(x >= 0xD800 && x <= 0xDFFF); }
__device__ __forceinline__ uint32_t dec_len_i64(int64_t x) {
  uint32_t l = 0;
  uint64_t a;
  if (x < 0) { l = 1; a = (uint64_t)(-(x + 1)) + 1; } else a = (uint64_t)x;
  do { l++; a /= 10; } while (a);
  return l;
}
__device__ __forceinline__ uint32_t dec_write_i64(uint8_t* d, int64_t x) {
  uint32_t l = dec_len_i64(x);
  uint64_t a;
  if (x < 0) { d[0] = '-'; a = (uint64_t)(-(x + 1)) + 1; } else a = (uint64_t)x;
  for (int k = (int)l - 1; k >= (x < 0 ? 1 : 0); k--) { d[k] = (uint8_t)('0' + a % 10); a /= 10; }
  return l;
}

// Utf8 tag builder (sam_tag_io.rs:658-732): Z/H -> string (invalid UTF-8 -> NULL), A -> 1 char,
// ints -> the Unicode char of that code point if valid else decimal; f -> Rust's f32 Display (f32_display.h).
__global__ void k_tag_utf8_len(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t row0, uint64_t n,
                               const uint32_t* __restrict__ loc, const uint8_t* __restrict__ typ, uint32_t* __restrict__ len,
                               uint64_t* __restrict__ valid, uint32_t* err) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = li < n;
  const uint64_t i = row0 + li;
  bool v = false;
  uint32_t l = 0;
  if (act) {
    const uint8_t ty = typ[li];
    if (ty) {
      const uint8_t* p = u + rows[li] + loc[li];
      if (ty == 'Z' || ty == 'H') {
        uint32_t k = 0;
        while (p[k]) k++;
        if (utf8_valid(p, k)) { v = true; l = k; }
      } else if (ty == 'A') {
        v = true; l = utf8_len_cp(p[0]);
      } else if (ty == 'f') {
        uint8_t tmp[56];
        v = true; l = f32disp::f32_display(ld_u32(p), tmp);  // Rust f32::to_string
      } else if (ty == 'B') {
        atomicExch(err, 6u);
      } else {
        const int64_t x = aux_int(p, ty);
        v = true;
        l = valid_scalar_cp(x) ? utf8_len_cp((uint32_t)x) : dec_len_i64(x);
      }
    }
    len[i] = l;
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && act) valid[i >> 6] = m;
}
void launch_tag_utf8_len(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                         uint32_t* len, uint64_t* valid, uint32_t* err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_tag_utf8_len, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, row0, n, loc, typ, len, valid, err);
}
__global__ void k_tag_utf8_scatter(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t n,
                                   const uint32_t* __restrict__ loc, const uint8_t* __restrict__ typ,
                                   const uint64_t* __restrict__ off64, const uint64_t* __restrict__ valid, uint64_t row0,
                                   uint8_t* __restrict__ dst) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n) return;
  const uint64_t i = row0 + li;
  if (!((valid[i >> 6] >> (i & 63)) & 1ull)) return;
  const uint8_t ty = typ[li];
  const uint8_t* p = u + rows[li] + loc[li];
  uint8_t* d = dst + off64[i];
  if (ty == 'Z' || ty == 'H') {
    for (uint32_t k = 0; p[k]; k++) d[k] = p[k];
  } else if (ty == 'A') {
    utf8_write_cp(d, p[0]);
  } else if (ty == 'f') {
    uint8_t tmp[56];
    const uint32_t l = f32disp::f32_display(ld_u32(p), tmp);
    for (uint32_t k = 0; k < l; k++) d[k] = tmp[k];
  } else {
    const int64_t x = aux_int(p, ty);
    if (valid_scalar_cp(x)) utf8_write_cp(d, (uint32_t)x); else dec_write_i64(d, x);
  }
}
void launch_tag_utf8_scatter(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                             const uint64_t* off64, const uint64_t* valid, uint64_t row0, uint8_t* dst, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_tag_utf8_scatter, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, n, loc, typ, off64, valid, row0, dst);
}

// List<T> tag builder (sam_tag_io.rs:761-1036): range-checked element casts, float <-> int mismatch is an error
__device__ __forceinline__ uint32_t list_elem_bytes(int32_t elem) { return elem < 2 ? 1u : elem < 4 ? 2u : 4u; }
__global__ void k_tag_list_len(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t row0, uint64_t n,
                               const uint32_t* __restrict__ loc, const uint8_t* __restrict__ typ, int32_t elem,
                               uint32_t* __restrict__ len, uint64_t* __restrict__ valid, uint32_t* err) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = li < n;
  const uint64_t i = row0 + li;
  bool v = false;
  uint32_t l = 0;
  if (act) {
    const uint8_t ty = typ[li];
    if (ty) {
      if (ty != 'B') atomicExch(err, 6u);
      else {
        const uint8_t* p = u + rows[li] + loc[li];
        const uint8_t st = p[0];
        if ((st == 'f') != (elem == 6)) atomicExch(err, 6u);
        else { v = true; l = ld_u32(p + 1); }
      }
    }
    len[i] = l;
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && act) valid[i >> 6] = m;
}
void launch_tag_list_len(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                         int32_t elem, uint32_t* len, uint64_t* valid, uint32_t* err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_tag_list_len, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, row0, n, loc, typ, elem, len, valid, err);
}
__global__ void k_tag_list_scatter(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t n,
                                   const uint32_t* __restrict__ loc, const uint8_t* __restrict__ typ, int32_t elem,
                                   const uint64_t* __restrict__ off64, uint8_t* __restrict__ dst, uint32_t* err) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n) return;
  if (typ[li] != 'B') return;
  const uint8_t* p = u + rows[li] + loc[li];
  const uint8_t st = p[0];
  const uint32_t cnt = ld_u32(p + 1);
  const uint8_t* src = p + 5;
  const uint32_t es = aux_elem_size(st), ob = list_elem_bytes(elem);
  uint8_t* d = dst + off64[li] * ob;  // off64 here is indexed by local row (caller passes the partition-local scan)
  static const int64_t lo[6] = {-128, 0, -32768, 0, -2147483648ll, 0};
  static const int64_t hi[6] = {127, 255, 32767, 65535, 2147483647ll, 4294967295ll};
  for (uint32_t k = 0; k < cnt; k++) {
    if (elem == 6) {
      uint32_t w = ld_u32(src + 4 * k);
      d[4 * k] = (uint8_t)w; d[4 * k + 1] = (uint8_t)(w >> 8); d[4 * k + 2] = (uint8_t)(w >> 16); d[4 * k + 3] = (uint8_t)(w >> 24);
    } else {
      int64_t x = aux_int(src + es * k, st);
      if (x < lo[elem] || x > hi[elem]) { atomicExch(err, 7u); x = 0; }
      uint64_t w = (uint64_t)x;
      for (uint32_t q = 0; q < ob; q++) d[ob * k + q] = (uint8_t)(w >> (8 * q));
    }
  }
}
void launch_tag_list_scatter(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint32_t* loc, const uint8_t* typ, int32_t elem,
                             const uint64_t* off64, uint8_t* dst, uint32_t* err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_tag_list_scatter, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, rows, n, loc, typ, elem, off64, dst, err);
}

}  // namespace bioscan
