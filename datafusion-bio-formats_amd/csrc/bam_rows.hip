// bam_rows.hip -- BAM record -> Arrow columns in two passes over the selected rows (gfx950).
//
// Replaces the per-record loop of the reference's BamExec (bio-format-bam/src/physical_exec.rs:408-573 sequential,
// :1269-1356 indexed) and the Arrow builders it drives (bio-format-core/src/alignment_utils.rs:383-644) for the twelve
// core columns.  A record's header is read once per pass and every Arrow byte is written exactly once:
//   pass 1  one row per lane: fixed-width columns + validity words, and the SUM of every variable-length column's row
//           lengths per tile of 256 rows (one u64 per column and tile -- no per-row length arrays, no offset arrays);
//   scan    two-level exclusive scan of the tile sums -> tile bases and column totals;
//   bases   first byte of every RecordBatch of the chunk (tile base + the rows of that tile in front of the batch);
//   pass 2  one workgroup per tile: lengths are recomputed from the records, scanned inside the tile, the per-batch int32
//           offsets are written straight from that scan, chrom / cigar / mate_chrom are written one row per lane, and
//           name / sequence / quality by 16-lane groups from metadata staged in LDS (16-byte chunks, the loads of four
//           steps in flight before the first store).
// A lane's dependent loads are kept few (header = three loads, the first four CIGAR operations one, a reference name one):
// the row-per-lane phases are bound by that chain of latencies, the 16-lane phase by HBM bandwidth (5.8 TB/s each way).
// All integer / byte work: no MFMA.  Bound: HBM (the inflated records are read twice -- headers only in pass 1 -- and the
// Arrow buffers written once).
#include "kernels.h"

namespace bioscan {

#define WAVE 64

struct __attribute__((packed, aligned(1))) br_u32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) br_u16 { uint16_t v; };
struct __attribute__((packed, aligned(1))) br_u64 { uint64_t v; };
struct __attribute__((packed, aligned(1))) br_u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint32_t br_ld32(const uint8_t* p) { return ((const br_u32*)p)->v; }
__device__ __forceinline__ uint32_t br_ld16(const uint8_t* p) { return (uint32_t)((const br_u16*)p)->v; }

__device__ __forceinline__ uint32_t br_dec_digits(uint32_t v) {
  uint32_t d = 1;
  while (v >= 10) { v /= 10; d++; }
  return d;
}

// what one record contributes to each variable-length column (k: 0 name, 1 chrom, 2 cigar, 3 mate_chrom, 4 sequence, 5 quality)
struct RowInfo {
  uint32_t len[6];
  int32_t refid, nref, pos, npos, tlen;
  uint32_t lrn, ncig, lseq, flag, mapq, end1;
  br_u32x4 ops4;  // the first four CIGAR operations (when the CIGAR was read)
  bool bad_ref, bad_op, bad_rec;
};

// reads the 36-byte header (and, when the CIGAR string or `end` is wanted, the CIGAR ops) of the record at r
__device__ __forceinline__ RowInfo row_info(const uint8_t* r, const uint32_t* __restrict__ ref_name_len, int32_t n_ref,
                                            int32_t binary_cigar, bool want_cigar, bool want_end) {
  RowInfo ri;
  // three loads, not eleven: records are ~300 bytes apart, so every load instruction of a wave touches 64 different
  // lines, and the lines of 24 resident waves do not survive in the 32 KB L1 until the next field is read
  const br_u32x4 h0 = *(const br_u32x4*)r, h1 = *(const br_u32x4*)(r + 16);
  ri.refid = (int32_t)h0.y; ri.pos = (int32_t)h0.z;
  ri.lrn = h0.w & 0xFFu; ri.mapq = (h0.w >> 8) & 0xFFu; ri.ncig = h1.x & 0xFFFFu; ri.flag = h1.x >> 16;
  ri.lseq = h1.y; ri.nref = (int32_t)h1.z; ri.npos = (int32_t)h1.w; ri.tlen = (int32_t)br_ld32(r + 32);
  // Every later step trusts l_read_name / n_cigar_op / l_seq: they must fit inside block_size (noodles fails such a record
  // with an I/O error; a CRC-valid member can still carry one).  A bad record is treated as empty and reported.
  ri.bad_rec = ri.lrn == 0 || (int32_t)ri.lseq < 0 ||
               32ull + ri.lrn + 4ull * ri.ncig + (((uint64_t)ri.lseq + 1) >> 1) + (uint64_t)ri.lseq > (uint64_t)h0.x;
  if (ri.bad_rec) { ri.lrn = 1; ri.ncig = 0; ri.lseq = 0; }
  ri.bad_ref = ri.refid >= n_ref || ri.nref >= n_ref;
  if (ri.bad_ref) { ri.refid = -1; ri.nref = -1; }
  ri.bad_op = false;
  ri.ops4 = br_u32x4{0, 0, 0, 0};
  uint32_t clen = 0, span = 0;
  if ((want_cigar || want_end) && ri.ncig) {
    // the first four operations in one load (short reads rarely have more; the load may run past the CIGAR, never past the
    // padded buffer): a loop of dependent 4-byte loads costs one memory latency per operation
    const uint8_t* cg = r + 36 + ri.lrn;
    ri.ops4 = *(const br_u32x4*)cg;
    auto one = [&](uint32_t v) {
      const uint32_t op = v & 15u;
      if (op > 8u) ri.bad_op = true;
      if ((0x18Du >> op) & 1u) span += v >> 4;  // M(0) D(2) N(3) =(7) X(8) consume the reference
      clen += br_dec_digits(v >> 4) + 1;
    };
    one(ri.ops4.x);
    if (ri.ncig > 1) one(ri.ops4.y);
    if (ri.ncig > 2) one(ri.ops4.z);
    if (ri.ncig > 3) one(ri.ops4.w);
    for (uint32_t k = 4; k < ri.ncig; k++) one(br_ld32(cg + 4 * k));
  }
  ri.end1 = ri.pos < 0 ? 0u : (uint32_t)ri.pos + span;  // 1-based inclusive end; 0 = None (noodles: start + span - 1)
  ri.len[0] = ri.lrn ? ri.lrn - 1 : 0;  // noodles strips the trailing NUL; a missing name ("*\0") is rendered "*"
  ri.len[1] = ri.refid >= 0 ? ref_name_len[ri.refid] : 0u;
  ri.len[2] = binary_cigar ? 4 * ri.ncig : clen;
  ri.len[3] = ri.nref >= 0 ? ref_name_len[ri.nref] : 0u;
  ri.len[4] = ri.lseq;
  ri.len[5] = ri.lseq;
  return ri;
}

__device__ __forceinline__ uint64_t br_wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t lo = __shfl_up((uint32_t)v, d, WAVE);
    const uint32_t hi = __shfl_up((uint32_t)(v >> 32), d, WAVE);
    if (lane >= d) v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

// ---- pass 1 -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROWS_TILE) void k_bam_rows_pass1(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t n,
                                                              RowsCols c, const uint32_t* __restrict__ ref_name_len, int32_t n_ref,
                                                              int32_t zero_based, int32_t binary_cigar, uint64_t n_tiles,
                                                              uint64_t* __restrict__ tile_sums, uint32_t* err) {
  __shared__ uint64_t s_w[6][ROWS_TILE / WAVE];
  const uint64_t i = (uint64_t)blockIdx.x * ROWS_TILE + threadIdx.x;
  const bool act = i < n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool want_cigar = (c.want >> 2) & 1u, want_end = c.end != nullptr || c.v_end != nullptr;
  RowInfo ri{};
  ri.refid = -1; ri.nref = -1; ri.pos = -1; ri.npos = -1;
  if (act) {
    ri = row_info(u + rows[i], ref_name_len, n_ref, binary_cigar, want_cigar, want_end);
    if (ri.bad_rec) atomicExch(err, 8u);
    else if (ri.bad_ref) atomicExch(err, 2u);
    if (ri.bad_op && want_cigar && !binary_cigar) atomicExch(err, 3u);
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) ri.len[k] = 0;
  }
  const bool v_start = act && ri.pos >= 0;
  const bool v_end = act && want_end && ri.end1 != 0;
  const bool v_chrom = act && ri.refid >= 0;
  const bool v_mchrom = act && ri.nref >= 0;
  const bool v_mstart = act && ri.npos >= 0;
  if (act) {
    if (c.start) c.start[i] = v_start ? (zero_based ? (uint32_t)ri.pos : (uint32_t)ri.pos + 1u) : 0u;
    if (c.end) c.end[i] = v_end ? ri.end1 : 0u;
    if (c.flags) c.flags[i] = ri.flag;
    if (c.mapq) c.mapq[i] = ri.mapq;
    if (c.mate_start) c.mate_start[i] = v_mstart ? (zero_based ? (uint32_t)ri.npos : (uint32_t)ri.npos + 1u) : 0u;
    if (c.tlen) c.tlen[i] = ri.tlen;
  }
  // validity words: one 64-bit word per wave (rows are wave-aligned)
  const uint64_t word = i >> 6;
  unsigned long long m;
  if (c.v_chrom) { m = __ballot(v_chrom); if (lane == 0 && act) c.v_chrom[word] = m; }
  if (c.v_start) { m = __ballot(v_start); if (lane == 0 && act) c.v_start[word] = m; }
  if (c.v_end) { m = __ballot(v_end); if (lane == 0 && act) c.v_end[word] = m; }
  if (c.v_mate_chrom) { m = __ballot(v_mchrom); if (lane == 0 && act) c.v_mate_chrom[word] = m; }
  if (c.v_mate_start) { m = __ballot(v_mstart); if (lane == 0 && act) c.v_mate_start[word] = m; }
  // tile sums of the projected variable-length columns
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (!((c.want >> k) & 1u)) continue;
    uint64_t v = ri.len[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t lo = __shfl_down((uint32_t)v, d, WAVE);
      const uint32_t hi = __shfl_down((uint32_t)(v >> 32), d, WAVE);
      v += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) s_w[k][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6 && ((c.want >> threadIdx.x) & 1u)) {
    uint64_t t = 0;
    for (int q = 0; q < ROWS_TILE / WAVE; q++) t += s_w[threadIdx.x][q];
    tile_sums[(uint64_t)threadIdx.x * (n_tiles + 1) + blockIdx.x] = t;
  }
}

// Exclusive scan of every column's tile sums, two levels: k_bam_tile_scan_blocks scans groups of 1024 tiles in place and
// leaves each group's total in `aux`; k_bam_tile_scan_top scans the group totals (one workgroup per column) and writes the
// column's grand total to entry n_tiles.  The first byte of tile t is tile_sums[k][t] + aux[k][t / 1024] (tile_base()).
__global__ __launch_bounds__(TS_GROUP) void k_bam_tile_scan_blocks(uint64_t* __restrict__ tile_sums, uint64_t n_tiles, uint64_t n_groups,
                                                                   uint64_t* __restrict__ aux, uint32_t want) {
  const int k = blockIdx.y;
  if (!((want >> k) & 1u)) return;
  uint64_t* a = tile_sums + (uint64_t)k * (n_tiles + 1);
  __shared__ uint64_t s_w[TS_GROUP / WAVE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * TS_GROUP + threadIdx.x;
  const uint64_t v = i < n_tiles ? a[i] : 0;
  const uint64_t inc = br_wave_incl_scan(v, lane);
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
  for (int q = 0; q < TS_GROUP / WAVE; q++) { if (q < w) base += s_w[q]; tot += s_w[q]; }
  if (i < n_tiles) a[i] = base + inc - v;
  if (threadIdx.x == 0) aux[(uint64_t)k * (n_groups + 1) + blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_bam_tile_scan_top(uint64_t* __restrict__ tile_sums, uint64_t n_tiles, uint64_t n_groups,
                                                            uint64_t* __restrict__ aux, uint32_t want) {
  if (!((want >> blockIdx.x) & 1u)) return;
  uint64_t* a = aux + (uint64_t)blockIdx.x * (n_groups + 1);
  __shared__ uint64_t s_w[1024 / WAVE];
  __shared__ uint64_t carry_s;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t b0 = 0; b0 < n_groups; b0 += 1024) {
    const uint64_t i = b0 + threadIdx.x;
    const uint64_t v = i < n_groups ? a[i] : 0;
    const uint64_t inc = br_wave_incl_scan(v, lane);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int q = 0; q < 1024 / WAVE; q++) { if (q < w) base += s_w[q]; tot += s_w[q]; }
    const uint64_t cy = carry_s;
    if (i < n_groups) a[i] = cy + base + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = cy + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sums[(uint64_t)blockIdx.x * (n_tiles + 1) + n_tiles] = carry_s;
}
__device__ __forceinline__ uint64_t tile_base(const uint64_t* __restrict__ tile_sums, const uint64_t* __restrict__ aux, int k, uint64_t tile,
                                              uint64_t n_tiles) {
  const uint64_t n_groups = (n_tiles + TS_GROUP - 1) / TS_GROUP;
  return tile_sums[(uint64_t)k * (n_tiles + 1) + tile] + aux[(uint64_t)k * (n_groups + 1) + tile / TS_GROUP];
}

__device__ __forceinline__ uint64_t br_batch_start_row(uint64_t b, uint32_t bs, uint32_t phase) { return b ? b * bs - phase : 0; }

// first byte of every batch: base of the tile that holds the batch's first row + the rows of that tile in front of it
__global__ __launch_bounds__(ROWS_TILE) void k_bam_batch_bases(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t n,
                                                               RowsCols c, const uint32_t* __restrict__ ref_name_len, int32_t n_ref,
                                                               int32_t binary_cigar, uint32_t bs, uint32_t phase,
                                                               uint64_t n_tiles, const uint64_t* __restrict__ tile_sums,
                                                               const uint64_t* __restrict__ aux) {
  const uint32_t want = c.want;
  __shared__ uint64_t s_w[6][ROWS_TILE / WAVE];
  const uint64_t b = blockIdx.x;
  const uint64_t s = br_batch_start_row(b, bs, phase);
  const uint64_t tile = s / ROWS_TILE, part = s % ROWS_TILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t i = tile * ROWS_TILE + threadIdx.x;
  RowInfo ri;
#pragma unroll
  for (int k = 0; k < 6; k++) ri.len[k] = 0;
  if (threadIdx.x < part && i < n) ri = row_info(u + rows[i], ref_name_len, n_ref, binary_cigar, (want >> 2) & 1u, false);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (!((want >> k) & 1u)) continue;
    uint64_t v = ri.len[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t lo = __shfl_down((uint32_t)v, d, WAVE);
      const uint32_t hi = __shfl_down((uint32_t)(v >> 32), d, WAVE);
      v += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) s_w[k][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6 && ((want >> threadIdx.x) & 1u)) {
    uint64_t t = tile < n_tiles ? tile_base(tile_sums, aux, (int)threadIdx.x, tile, n_tiles) : tile_sums[(uint64_t)threadIdx.x * (n_tiles + 1) + n_tiles];
    for (int q = 0; q < ROWS_TILE / WAVE; q++) t += s_w[threadIdx.x][q];
    c.base[threadIdx.x][b] = t;
  }
}

#ifndef BR_UNROLL
#define BR_UNROLL 4
#endif
// ---- pass 2 -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t br_write_dec(uint8_t* d, uint32_t v) {
  const uint32_t nd = br_dec_digits(v);
  for (int k = (int)nd - 1; k >= 0; k--) { d[k] = (uint8_t)('0' + v % 10); v /= 10; }
  return nd;
}
// a reference name of `len` bytes at names + off, whose first 8 bytes are v (the table is padded by 8 bytes)
__device__ __forceinline__ void br_put_name(uint8_t* d, const uint8_t* __restrict__ names, uint32_t off, uint32_t len, uint64_t v) {
  if (len > 8) { for (uint32_t k = 0; k < len; k++) d[k] = names[off + k]; return; }
  if (len & 8) { ((br_u64*)d)->v = v; return; }
  if (len & 4) { ((br_u32*)d)->v = (uint32_t)v; v >>= 32; d += 4; }
  if (len & 2) { ((br_u16*)d)->v = (uint16_t)v; v >>= 16; d += 2; }
  if (len & 1) *d = (uint8_t)v;
}
__device__ __forceinline__ uint32_t br_qual_swar(uint32_t w) { return ((w & 0x7F7F7F7Fu) + 0x21212121u) ^ (w & 0x80808080u); }  // (q + 33) mod 256 per byte

__global__ __launch_bounds__(ROWS_TILE) void k_bam_rows_pass2(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rows, uint64_t n,
                                                              RowsCols c, const uint8_t* __restrict__ ref_names,
                                                              const uint32_t* __restrict__ ref_name_off, const uint32_t* __restrict__ ref_name_len,
                                                              int32_t n_ref, int32_t binary_cigar, uint32_t bs, uint32_t phase,
                                                              uint64_t n_tiles, const uint64_t* __restrict__ tile_sums,
                                                              const uint64_t* __restrict__ aux, uint32_t* qual_wide) {
  __shared__ uint64_t s_w[6][ROWS_TILE / WAVE];
  __shared__ uint64_t s_rec[ROWS_TILE], s_on[ROWS_TILE + 1], s_os[ROWS_TILE + 1], s_oq[ROWS_TILE + 1];
  __shared__ uint32_t s_meta[ROWS_TILE], s_lseq[ROWS_TILE];
  __shared__ uint16_t s_pair[256];  // packed byte -> two ASCII bases (high nibble first), little-endian u16
  {
    const char* Lt = "=ACMGRSVTWYHKDBN";
    s_pair[threadIdx.x] = (uint16_t)((uint8_t)Lt[threadIdx.x >> 4] | ((uint16_t)(uint8_t)Lt[threadIdx.x & 15] << 8));
  }
  const uint64_t tile = blockIdx.x;
  const uint64_t i = tile * ROWS_TILE + threadIdx.x;
  const bool act = i < n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool want_cigar = (c.want >> 2) & 1u;
  RowInfo ri{};
  uint64_t ro = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) ri.len[k] = 0;
  if (act) {
    ro = rows[i];
    ri = row_info(u + ro, ref_name_len, n_ref, binary_cigar, want_cigar, false);
  }
  // reference names: offset and first 8 bytes now, so that the two dependent loads overlap the scans below
  uint32_t nm1_off = 0, nm3_off = 0;
  uint64_t nm1 = 0, nm3 = 0;
  if (act && c.val[1] && ri.refid >= 0) { nm1_off = ref_name_off[ri.refid]; nm1 = ((const br_u64*)(ref_names + nm1_off))->v; }
  if (act && c.val[3] && ri.nref >= 0) { nm3_off = ref_name_off[ri.nref]; nm3 = ((const br_u64*)(ref_names + nm3_off))->v; }
  // exclusive scan of every projected column inside the tile
  uint64_t off[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    off[k] = 0;
    if (!((c.want >> k) & 1u)) continue;
    const uint64_t inc = br_wave_incl_scan((uint64_t)ri.len[k], lane);
    if (lane == 63) s_w[k][w] = inc;
    off[k] = inc - ri.len[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (!((c.want >> k) & 1u)) continue;
    uint64_t base = tile_base(tile_sums, aux, k, tile, n_tiles);
    for (int q = 0; q < w; q++) base += s_w[k][q];
    off[k] += base;
  }
  if (act) {
    // per-batch int32 offsets: entry j of batch b, and the closing entry when this is the batch's (or the chunk's) last row
    const uint32_t v = (uint32_t)i + phase;
    const uint32_t b = v / bs;
    const uint64_t start = br_batch_start_row(b, bs, phase);
    const uint64_t j = i - start;
    const bool closes = (v + 1) % bs == 0 || i + 1 == n;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      if (!((c.want >> k) & 1u)) continue;
      const uint64_t bb = c.base[k][b];
      int32_t* o32 = c.off32[k] + (uint64_t)b * ((uint64_t)bs + 1);
      o32[j] = (int32_t)(off[k] - bb);
      if (closes) o32[j + 1] = (int32_t)(off[k] + ri.len[k] - bb);
    }
    // chrom / mate_chrom / cigar: one row per lane.  Names of up to 8 bytes (chr1 ... chrUn) are one load and at most three
    // stores; the CIGAR's first four operations are already in registers.
    if (c.val[1] && ri.refid >= 0) br_put_name(c.val[1] + off[1], ref_names, nm1_off, ri.len[1], nm1);
    if (c.val[3] && ri.nref >= 0) br_put_name(c.val[3] + off[3], ref_names, nm3_off, ri.len[3], nm3);
    if (c.val[2]) {
      const uint8_t* cg = u + ro + 36 + ri.lrn;
      uint8_t* d = c.val[2] + off[2];
      if (binary_cigar) {
        if (ri.ncig > 0) ((br_u32*)d)[0].v = ri.ops4.x;
        if (ri.ncig > 1) ((br_u32*)d)[1].v = ri.ops4.y;
        if (ri.ncig > 2) ((br_u32*)d)[2].v = ri.ops4.z;
        if (ri.ncig > 3) ((br_u32*)d)[3].v = ri.ops4.w;
        for (uint32_t k = 4; k < ri.ncig; k++) ((br_u32*)d)[k].v = br_ld32(cg + 4 * k);
      } else {
        const char ops[] = "MIDNSHP=X???????";
        auto put = [&](uint32_t v2) { d += br_write_dec(d, v2 >> 4); *d++ = (uint8_t)ops[v2 & 15u]; };
        if (ri.ncig > 0) put(ri.ops4.x);
        if (ri.ncig > 1) put(ri.ops4.y);
        if (ri.ncig > 2) put(ri.ops4.z);
        if (ri.ncig > 3) put(ri.ops4.w);
        for (uint32_t k = 4; k < ri.ncig; k++) put(br_ld32(cg + 4 * k));
      }
    }
  }
  s_rec[threadIdx.x] = ro;
  s_meta[threadIdx.x] = ri.lrn | (ri.ncig << 8);
  s_lseq[threadIdx.x] = act ? ri.lseq : 0u;
  s_on[threadIdx.x] = off[0]; s_os[threadIdx.x] = off[4]; s_oq[threadIdx.x] = off[5];
  const uint32_t nrow = (uint32_t)((n - tile * ROWS_TILE) < ROWS_TILE ? (n - tile * ROWS_TILE) : ROWS_TILE);
  if (threadIdx.x + 1 == nrow) {  // closing entries: where the tile's name / sequence / quality bytes end
    s_on[nrow] = off[0] + ri.len[0]; s_os[nrow] = off[4] + ri.len[4]; s_oq[nrow] = off[5] + ri.len[5];
  }
  __syncthreads();
  uint8_t* const d_name = c.val[0];
  uint8_t* const d_seq = c.val[4];
  uint8_t* const d_qual = c.val[5];
  if (!d_name && !d_seq && !d_qual) return;
  // ---- name / sequence / quality ------------------------------------------------------------------------------------
  // Four 16-lane groups per wave, each on its own row of the wave's 64; a lane moves 16-byte chunks at per-row addresses
  // (a partial last chunk is served by the overlapping 16 bytes that END at the segment's end); all loads of BR_UNROLL
  // steps are issued before the first store.  Measured on 13.6 M rows of 150-base reads: the loads alone and the stores
  // alone each run at 5.8 TB/s -- this phase is at the HBM roofline; an LDS-staged variant with aligned, fully coalesced
  // global accesses was built and is 40 % slower (four times the instructions per row, no traffic saved), aligning the
  // accesses changes nothing (-3 %), and 8- or 32-lane groups are equal or worse.
  constexpr uint32_t G = 16, GS = G * 16;  // lanes per row group, bytes one group moves per step
  const int g = lane / (int)G, sl = lane % (int)G;
  bool wide = false;
  auto seq16 = [&](uint64_t pkd) {
    br_u32x4 v;
    v.x = (uint32_t)s_pair[pkd & 0xFF] | ((uint32_t)s_pair[(pkd >> 8) & 0xFF] << 16);
    v.y = (uint32_t)s_pair[(pkd >> 16) & 0xFF] | ((uint32_t)s_pair[(pkd >> 24) & 0xFF] << 16);
    v.z = (uint32_t)s_pair[(pkd >> 32) & 0xFF] | ((uint32_t)s_pair[(pkd >> 40) & 0xFF] << 16);
    v.w = (uint32_t)s_pair[(pkd >> 48) & 0xFF] | ((uint32_t)s_pair[(pkd >> 56) & 0xFF] << 16);
    return v;
  };
  auto qual16 = [&](br_u32x4 v) {
    v.x = br_qual_swar(v.x); v.y = br_qual_swar(v.y); v.z = br_qual_swar(v.z); v.w = br_qual_swar(v.w);
    wide = wide || ((v.x | v.y | v.z | v.w) & 0x80808080u);  // a byte >= 128 is a two-byte UTF-8 char: exact wide path
    return v;
  };
  // chunk of a segment of `len` bytes that lane-chunk c0 serves: a partial last chunk is served by the (overlapping) 16
  // bytes that END at the segment's end, so no lane runs a byte loop unless the whole segment is shorter than 16 bytes
  auto tail_chunk = [](uint32_t c0, uint32_t len) { return (c0 + 16 <= len || len < 16) ? c0 : len - 16; };
  struct Geom {
    uint64_t rec, ono, oso, oqo;     // record start in u; first byte of the row in the three value buffers
    uint32_t no, so, qo;             // name / packed bases / qualities, relative to rec
    uint32_t ln, lseq;
  };
  auto geom = [&](uint32_t k) {
    Geom e;
    const uint32_t meta = s_meta[k];
    const uint32_t lrn = meta & 0xFFu, ncig = meta >> 8;
    e.lseq = s_lseq[k];
    e.ono = s_on[k]; e.oso = s_os[k]; e.oqo = s_oq[k];
    e.rec = s_rec[k];
    e.no = 36;  // read_name starts 36 bytes into the record
    e.so = 36 + lrn + 4 * ncig;
    e.qo = e.so + ((e.lseq + 1) >> 1);
    e.ln = lrn ? lrn - 1 : 0;
    return e;
  };

  // direct path for the U x 4 rows of this wave from kb on
  constexpr uint32_t U = BR_UNROLL;
  const uint32_t kend_w = nrow < ((uint32_t)w + 1) * WAVE ? nrow : ((uint32_t)w + 1) * WAVE;  // this wave's rows end here
  auto direct_rows = [&](uint32_t kb) {
    br_u32x4 vn[U], vq[U];
    uint64_t pk[U];
    const uint32_t c0 = (uint32_t)sl * 16;
#pragma unroll
    for (uint32_t q = 0; q < U; q++) {
      const uint32_t k = kb + q * (WAVE / G) + (uint32_t)g;
      vn[q] = br_u32x4{0, 0, 0, 0}; vq[q] = br_u32x4{0, 0, 0, 0}; pk[q] = 0;
      if (k >= kend_w) continue;
      const Geom e = geom(k);
      const uint8_t* r = u + e.rec;
      if (d_name && c0 < e.ln && e.ln >= 16) vn[q] = *(const br_u32x4*)(r + e.no + tail_chunk(c0, e.ln));
      if (d_seq && c0 < e.lseq && e.lseq >= 16) pk[q] = ((const br_u64*)(r + e.so + ((tail_chunk(c0, e.lseq) & ~1u) >> 1)))->v;
      if (d_qual && c0 < e.lseq && e.lseq >= 16) vq[q] = *(const br_u32x4*)(r + e.qo + tail_chunk(c0, e.lseq));
    }
#pragma unroll
    for (uint32_t q = 0; q < U; q++) {
      const uint32_t k = kb + q * (WAVE / G) + (uint32_t)g;
      if (k >= kend_w) continue;
      const Geom e = geom(k);
      const uint32_t ln = e.ln, lseq = e.lseq;
      const uint8_t *np = u + e.rec + e.no, *sp = u + e.rec + e.so, *qp = u + e.rec + e.qo;
      if (d_name && c0 < ln) {
        if (ln >= 16) *(br_u32x4*)(d_name + e.ono + tail_chunk(c0, ln)) = vn[q];
        else for (uint32_t j = 0; j < ln; j++) d_name[e.ono + j] = np[j];
      }
      if (d_seq && c0 < lseq) {
        if (lseq >= 16) {
          const uint32_t cs = tail_chunk(c0, lseq) & ~1u;  // packed bytes start on even bases
          *(br_u32x4*)(d_seq + e.oso + cs) = seq16(pk[q]);
          if (cs != c0 && (lseq & 1u)) d_seq[e.oso + lseq - 1] = (uint8_t)s_pair[sp[(lseq - 1) >> 1]];  // the vector ended one base early
        } else {
          for (uint32_t j = 0; j < lseq; j += 2) {
            const uint16_t pr = s_pair[sp[j >> 1]];
            d_seq[e.oso + j] = (uint8_t)pr;
            if (j + 1 < lseq) d_seq[e.oso + j + 1] = (uint8_t)(pr >> 8);
          }
        }
      }
      if (d_qual && c0 < lseq) {
        if (lseq >= 16) *(br_u32x4*)(d_qual + e.oqo + tail_chunk(c0, lseq)) = qual16(vq[q]);
        else for (uint32_t j = 0; j < lseq; j++) {
          const uint32_t qv = ((uint32_t)qp[j] + 33u) & 0xFFu;
          wide = wide || qv >= 128u;
          d_qual[e.oqo + j] = (uint8_t)qv;
        }
      }
      // rows longer than 256 bytes per segment (long reads): remaining chunks, same scheme
      if (d_name) for (uint32_t cc0 = c0 + GS; cc0 < ln; cc0 += GS) {
        const uint32_t cc = tail_chunk(cc0, ln);
        *(br_u32x4*)(d_name + e.ono + cc) = *(const br_u32x4*)(np + cc);
      }
      if (d_seq) for (uint32_t cc0 = c0 + GS; cc0 < lseq; cc0 += GS) {
        const uint32_t cc = tail_chunk(cc0, lseq) & ~1u;
        *(br_u32x4*)(d_seq + e.oso + cc) = seq16(((const br_u64*)(sp + (cc >> 1)))->v);
        if (cc != cc0 && (lseq & 1u)) d_seq[e.oso + lseq - 1] = (uint8_t)s_pair[sp[(lseq - 1) >> 1]];
      }
      if (d_qual) for (uint32_t cc0 = c0 + GS; cc0 < lseq; cc0 += GS) {
        const uint32_t cc = tail_chunk(cc0, lseq);
        *(br_u32x4*)(d_qual + e.oqo + cc) = qual16(*(const br_u32x4*)(qp + cc));
      }
    }
  };

  for (uint32_t k0 = (uint32_t)w * WAVE; k0 < kend_w; k0 += U * (WAVE / G)) direct_rows(k0);
  if (d_qual && __any(wide) && lane == 0) atomicExch(qual_wide, 1u);
}

// ---- host wrappers ----------------------------------------------------------------------------------------------
void launch_bam_rows_pass1(const uint8_t* u, const uint64_t* rows, uint64_t n, RowsCols c, const uint32_t* ref_name_len, int32_t n_ref,
                           int32_t zero_based, int32_t binary_cigar, uint64_t* tile_sums, uint32_t* err, hipStream_t st) {
  if (!n) return;
  const uint64_t n_tiles = (n + ROWS_TILE - 1) / ROWS_TILE;
  hipLaunchKernelGGL(k_bam_rows_pass1, dim3((uint32_t)n_tiles), dim3(ROWS_TILE), 0, st, u, rows, n, c, ref_name_len, n_ref, zero_based,
                     binary_cigar, n_tiles, tile_sums, err);
  if (c.want) launch_tile_scan(tile_sums, n_tiles, c.want, st);
}
void launch_tile_scan(uint64_t* tile_sums, uint64_t n_tiles, uint32_t want, hipStream_t st) {
  const uint64_t n_groups = (n_tiles + TS_GROUP - 1) / TS_GROUP;
  uint64_t* aux = tile_sums + 6 * (n_tiles + 1);  // 6 x (n_groups + 1) group totals behind the tile sums (bam_rows_scratch_elems)
  hipLaunchKernelGGL(k_bam_tile_scan_blocks, dim3((uint32_t)n_groups, 6), dim3(TS_GROUP), 0, st, tile_sums, n_tiles, n_groups, aux, want);
  hipLaunchKernelGGL(k_bam_tile_scan_top, dim3(6), dim3(1024), 0, st, tile_sums, n_tiles, n_groups, aux, want);
}
size_t bam_rows_scratch_elems(uint64_t n) {
  const uint64_t n_tiles = (n + ROWS_TILE - 1) / ROWS_TILE, n_groups = (n_tiles + TS_GROUP - 1) / TS_GROUP;
  return (size_t)(6 * (n_tiles + 1) + 6 * (n_groups + 1));
}
void launch_bam_rows_pass2(const uint8_t* u, const uint64_t* rows, uint64_t n, RowsCols c, const uint8_t* ref_names,
                           const uint32_t* ref_name_off, const uint32_t* ref_name_len, int32_t n_ref, int32_t binary_cigar,
                           uint32_t batch_size, uint32_t phase, const uint64_t* tile_sums, uint32_t* qual_wide, hipStream_t st) {
  if (!n || !c.want) return;
  const uint64_t n_tiles = (n + ROWS_TILE - 1) / ROWS_TILE;
  const uint64_t nb = (n + phase + batch_size - 1) / batch_size;
  const uint64_t* aux = tile_sums + 6 * (n_tiles + 1);
  hipLaunchKernelGGL(k_bam_batch_bases, dim3((uint32_t)nb), dim3(ROWS_TILE), 0, st, u, rows, n, c, ref_name_len, n_ref, binary_cigar,
                     batch_size, phase, n_tiles, tile_sums, aux);
  hipLaunchKernelGGL(k_bam_rows_pass2, dim3((uint32_t)n_tiles), dim3(ROWS_TILE), 0, st, u, rows, n, c, ref_names, ref_name_off, ref_name_len,
                     n_ref, binary_cigar, batch_size, phase, n_tiles, tile_sums, aux, qual_wide);
}

}  // namespace bioscan
