/* bioscan_oracle.c -- plain-C CPU restatement of the BGZF -> BAM -> Arrow column scan.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (checker), __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Never linked into libbioscan.so and never called by the product path.
 *
 * Restates (paths relative to /root/reference/datafusion):
 *   - BGZF member framing + inflate + CRC32 check: noodles-bgzf 0.49.0 Reader (un-vendored;
 *     call site bio-format-bam/src/storage.rs:161-169).  Inflate = libdeflate when present
 *     (the library the reference links, Cargo.toml:28,41), else zlib.
 *   - sequential record loop + field rules: bio-format-bam/src/physical_exec.rs:408-573.
 *   - builder semantics: bio-format-core/src/alignment_utils.rs:383-644, 695-701;
 *     tags (Int32 / Utf8 only): bio-format-core/src/sam_tag_io.rs:658-742.
 * Threading model mirrors the reference's "one OS thread per partition"
 * (bio-format-core/src/sync_stream.rs:7-33): phase A inflates block ranges on T threads, the
 * record chain is walked once, phase B builds columns for row ranges on T threads.
 *
 * PARITY PINNING: validated against oracle/bam_oracle.py (tests/test_oracle_c.py), which is
 * pinned by the reference's fixtures; per-value column parity is otherwise unpinned (see the
 * header of bam_oracle.py).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

typedef struct {
  uint8_t* data;      /* utf8 bytes or 4-byte values */
  int64_t* offsets;   /* n_rows+1 (var-len only) */
  uint8_t* valid;     /* bitmap, NULL when the column has no validity */
  uint64_t data_len;
} oracle_col;

typedef struct {
  uint64_t n_rows, n_blocks, compressed_bytes, inflated_bytes;
  oracle_col cols[12 + 8];
  int n_cols;
  double seconds_inflate, seconds_chain, seconds_columns, seconds_total;
  int threads;
  int used_libdeflate;
  char error[256];
} oracle_result;

typedef void* (*ld_allocd_t)(void);
typedef int (*ld_decomp_t)(void*, const void*, size_t, void*, size_t, size_t*);
typedef void (*ld_freed_t)(void*);
typedef uint32_t (*ld_crc32_t)(uint32_t, const void*, size_t);
static ld_crc32_t ld_crc32;   /* libdeflate's CRC32 (what noodles-bgzf's libdeflate backend uses), zlib's otherwise */
static ld_allocd_t ld_allocd;
static ld_decomp_t ld_decomp;
static ld_freed_t ld_freed;
static int ld_tried;
static void load_libdeflate(void) {
  if (ld_tried) return;
  ld_tried = 1;
  const char* names[] = {"libdeflate.so.0", "/opt/conda/lib/libdeflate.so.0", "libdeflate.so", NULL};
  for (int i = 0; names[i]; i++) {
    void* h = dlopen(names[i], RTLD_NOW);
    if (!h) continue;
    ld_allocd = (ld_allocd_t)dlsym(h, "libdeflate_alloc_decompressor");
    ld_decomp = (ld_decomp_t)dlsym(h, "libdeflate_deflate_decompress");
    ld_freed = (ld_freed_t)dlsym(h, "libdeflate_free_decompressor");
    if (ld_allocd && ld_decomp && ld_freed) { ld_crc32 = (ld_crc32_t)dlsym(h, "libdeflate_crc32"); return; }
    ld_allocd = NULL;
  }
}

static inline uint32_t crc_of(const uint8_t* p, size_t n) {
  return ld_crc32 ? ld_crc32(0, p, n) : (uint32_t)crc32(crc32(0, NULL, 0), p, (uInt)n);
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline int32_t rdi32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

typedef struct {
  const uint8_t* file;
  const uint64_t* coff;
  const uint64_t* uoff;
  uint8_t* u;
  uint64_t b0, b1;
  int err;
} inflate_job;

static void* inflate_worker(void* arg) {
  inflate_job* j = (inflate_job*)arg;
  void* d = ld_allocd ? ld_allocd() : NULL;
  for (uint64_t b = j->b0; b < j->b1; b++) {
    const uint8_t* m = j->file + j->coff[b];
    uint64_t msz = j->coff[b + 1] - j->coff[b];
    uint32_t xlen = rd16(m + 10);
    const uint8_t* payload = m + 12 + xlen;
    size_t plen = (size_t)(msz - 12 - xlen - 8);
    uint32_t isize = rd32(m + msz - 4), crc = rd32(m + msz - 8);
    uint8_t* dst = j->u + j->uoff[b];
    if (d) {
      size_t got = 0;
      if (ld_decomp(d, payload, plen, dst, isize, &got) != 0 || got != isize) { j->err = 1; break; }
    } else {
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      inflateInit2(&zs, -15);
      zs.next_in = (Bytef*)payload; zs.avail_in = (uInt)plen;
      zs.next_out = dst; zs.avail_out = isize;
      int rc = inflate(&zs, Z_FINISH);
      uint32_t got = (uint32_t)zs.total_out;
      inflateEnd(&zs);
      if ((rc != Z_STREAM_END && !(rc == Z_BUF_ERROR && isize == 0)) || got != isize) { j->err = 1; break; }
    }
    if (crc_of(dst, isize) != crc) { j->err = 2; break; }
  }
  if (d) ld_freed(d);
  return NULL;
}

static inline uint32_t ndigits(uint32_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; d++; } return d; }

typedef struct {
  const uint8_t* u;
  const uint64_t* rec;
  uint64_t r0, r1;
  int zero_based;
  const char** ref_names;
  const uint32_t* ref_name_len;
  int n_ref;
  int n_tags;
  const char* tags;      /* 2 bytes each */
  const int* tag_kinds;  /* 0 = Int32, 3 = Utf8 */
  int pass;              /* 0 = lengths, 1 = write */
  oracle_col* cols;
  int64_t* len[12 + 8];  /* per var column per-row lengths during pass 0 (aliases offsets+1) */
  int err;
} col_job;

/* core column kinds: 1 = utf8, 0 = u32/i32 */
static const int CORE_VAR[12] = {1, 1, 0, 0, 0, 1, 0, 1, 0, 1, 1, 0};

static const uint8_t* find_tag(const uint8_t* r, uint32_t bs, const char* tag, uint8_t* ty) {
  uint32_t lrn = r[12], ncig = rd16(r + 16);
  int32_t lseq = rdi32(r + 20);
  uint32_t o = 36 + lrn + 4 * ncig + (uint32_t)((lseq + 1) / 2) + (uint32_t)lseq, end = 4 + bs;
  while (o + 3 <= end) {
    uint8_t t = r[o + 2];
    uint32_t vo = o + 3, sz;
    if (t == 'Z' || t == 'H') { uint32_t k = vo; while (k < end && r[k]) k++; sz = k - vo + 1; }
    else if (t == 'B') {
      uint8_t st = r[vo];
      uint32_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
      sz = 5 + es * rd32(r + vo + 1);
    } else if (t == 'A' || t == 'c' || t == 'C') sz = 1;
    else if (t == 's' || t == 'S') sz = 2;
    else sz = 4;
    if (r[o] == (uint8_t)tag[0] && r[o + 1] == (uint8_t)tag[1]) { *ty = t; return r + vo; }
    o = vo + sz;
  }
  return NULL;
}

static void* col_worker(void* arg) {
  col_job* j = (col_job*)arg;
  static const char SEQ[] = "=ACMGRSVTWYHKDBN";
  static const char OPS[] = "MIDNSHP=X???????";
  for (uint64_t i = j->r0; i < j->r1; i++) {
    const uint8_t* r = j->u + j->rec[i];
    uint32_t bs = rd32(r);
    int32_t refid = rdi32(r + 4), pos = rdi32(r + 8);
    uint32_t lrn = r[12], mapq = r[13], ncig = rd16(r + 16), flag = rd16(r + 18);
    int32_t lseq = rdi32(r + 20), nref = rdi32(r + 24), npos = rdi32(r + 28), tlen = rdi32(r + 32);
    const uint8_t* cg = r + 36 + lrn;
    const uint8_t* sq = cg + 4 * ncig;
    const uint8_t* ql = sq + (lseq + 1) / 2;
    if (refid >= j->n_ref || nref >= j->n_ref) { j->err = 1; return NULL; }
    if (j->pass == 0) {
      oracle_col* c = j->cols;
      c[0].offsets[i + 1] = lrn ? lrn - 1 : 0;
      c[1].offsets[i + 1] = refid >= 0 ? j->ref_name_len[refid] : 0;
      uint32_t cl = 0, span = 0;
      for (uint32_t k = 0; k < ncig; k++) {
        uint32_t v = rd32(cg + 4 * k), op = v & 15;
        cl += ndigits(v >> 4) + 1;
        if ((0x18Du >> op) & 1u) span += v >> 4;
      }
      c[5].offsets[i + 1] = cl;
      c[7].offsets[i + 1] = nref >= 0 ? j->ref_name_len[nref] : 0;
      c[9].offsets[i + 1] = lseq;
      uint32_t qlw = 0;
      for (int32_t k = 0; k < lseq; k++) qlw += (((uint32_t)ql[k] + 33u) & 0xFFu) >= 128u ? 2u : 1u;
      c[10].offsets[i + 1] = qlw;
      /* fixed columns are written in pass 0 */
      uint32_t* v;
      v = (uint32_t*)c[2].data; v[i] = pos >= 0 ? (uint32_t)(j->zero_based ? pos : pos + 1) : 0;
      uint32_t end1 = pos >= 0 ? (uint32_t)pos + span : 0;
      v = (uint32_t*)c[3].data; v[i] = end1;
      v = (uint32_t*)c[4].data; v[i] = flag;
      v = (uint32_t*)c[6].data; v[i] = mapq;
      v = (uint32_t*)c[8].data; v[i] = npos >= 0 ? (uint32_t)(j->zero_based ? npos : npos + 1) : 0;
      v = (uint32_t*)c[11].data; v[i] = (uint32_t)tlen;
      /* validity bits: set with atomic or (rows of one byte may belong to two threads) */
      if (refid >= 0) __atomic_fetch_or(&c[1].valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
      if (pos >= 0) __atomic_fetch_or(&c[2].valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
      if (end1 != 0) __atomic_fetch_or(&c[3].valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
      if (nref >= 0) __atomic_fetch_or(&c[7].valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
      if (npos >= 0) __atomic_fetch_or(&c[8].valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
      for (int t = 0; t < j->n_tags; t++) {
        oracle_col* tc = &c[12 + t];
        uint8_t ty = 0;
        const uint8_t* p = find_tag(r, bs, j->tags + 2 * t, &ty);
        if (j->tag_kinds[t] == 0) {
          if (p) {
            int64_t x;
            switch (ty) {
              case 'c': x = (int8_t)p[0]; break;
              case 'C': case 'A': x = p[0]; break;
              case 's': x = (int16_t)rd16(p); break;
              case 'S': x = rd16(p); break;
              case 'i': x = rdi32(p); break;
              case 'I': x = (int64_t)rd32(p); break;
              default: j->err = 2; return NULL;
            }
            if (x < INT32_MIN || x > INT32_MAX) { j->err = 2; return NULL; }
            ((int32_t*)tc->data)[i] = (int32_t)x;
            __atomic_fetch_or(&tc->valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
          } else ((int32_t*)tc->data)[i] = 0;
        } else {
          if (p) {
            if (ty != 'Z' && ty != 'H') { j->err = 2; return NULL; }
            tc->offsets[i + 1] = (int64_t)strlen((const char*)p);
            __atomic_fetch_or(&tc->valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
          } else tc->offsets[i + 1] = 0;
        }
      }
    } else {
      oracle_col* c = j->cols;
      memcpy(c[0].data + c[0].offsets[i], r + 36, lrn ? lrn - 1 : 0);
      if (refid >= 0) memcpy(c[1].data + c[1].offsets[i], j->ref_names[refid], j->ref_name_len[refid]);
      if (nref >= 0) memcpy(c[7].data + c[7].offsets[i], j->ref_names[nref], j->ref_name_len[nref]);
      uint8_t* d = c[5].data + c[5].offsets[i];
      for (uint32_t k = 0; k < ncig; k++) {
        uint32_t v = rd32(cg + 4 * k), n = v >> 4, nd = ndigits(n);
        for (int q = (int)nd - 1; q >= 0; q--) { d[q] = (uint8_t)('0' + n % 10); n /= 10; }
        d += nd;
        *d++ = (uint8_t)OPS[v & 15];
      }
      d = c[9].data + c[9].offsets[i];
      for (int32_t k = 0; k < lseq; k++) { uint8_t b = sq[k >> 1]; d[k] = (uint8_t)SEQ[(k & 1) ? (b & 15) : (b >> 4)]; }
      d = c[10].data + c[10].offsets[i];
      for (int32_t k = 0; k < lseq; k++) {
        uint32_t ch = ((uint32_t)ql[k] + 33u) & 0xFFu;
        if (ch < 128u) *d++ = (uint8_t)ch;
        else { *d++ = (uint8_t)(0xC0u | (ch >> 6)); *d++ = (uint8_t)(0x80u | (ch & 0x3Fu)); }
      }
      for (int t = 0; t < j->n_tags; t++) {
        if (j->tag_kinds[t] != 3) continue;
        oracle_col* tc = &c[12 + t];
        uint8_t ty = 0;
        const uint8_t* p = find_tag(r, bs, j->tags + 2 * t, &ty);
        if (p) memcpy(tc->data + tc->offsets[i], p, (size_t)(tc->offsets[i + 1] - tc->offsets[i]));
      }
    }
  }
  return NULL;
}

void oracle_free(oracle_result* r) {
  for (int c = 0; c < r->n_cols; c++) { free(r->cols[c].data); free(r->cols[c].offsets); free(r->cols[c].valid); }
  memset(r, 0, sizeof *r);
}

/* Sequential full scan of the first `max_blocks` BGZF members (0 = all) of a BAM file held in
 * memory; records that straddle the sample end are not emitted.  tags: n_tags 2-char names
 * concatenated; tag_kinds[t] 0 = Int32 column, 3 = Utf8 column. */
int oracle_bam_scan_mem(const uint8_t* file, uint64_t file_len, int zero_based, int threads, uint64_t max_blocks,
                        int n_tags, const char* tags, const int* tag_kinds, int build_columns, oracle_result* out) {
  memset(out, 0, sizeof *out);
  load_libdeflate();
  out->used_libdeflate = ld_allocd != NULL;
  if (threads < 1) threads = 1;
  out->threads = threads;
  double t_all = now_s();
  /* framing */
  uint64_t cap = 1024, nb = 0, o = 0, uo = 0;
  uint64_t* coff = (uint64_t*)malloc((cap + 1) * 8);
  uint64_t* uoff = (uint64_t*)malloc((cap + 1) * 8);
  while (o < file_len && (max_blocks == 0 || nb < max_blocks)) {
    if (max_blocks && file_len - o < 18) break;
    if (file_len - o < 18 || file[o] != 0x1f || file[o + 1] != 0x8b) { snprintf(out->error, sizeof out->error, "bad BGZF header at %llu", (unsigned long long)o); return 1; }
    uint32_t xlen = rd16(file + o + 10);
    int64_t bsize = -1;
    for (uint64_t p = o + 12; p + 4 <= o + 12 + xlen;) {
      uint32_t slen = rd16(file + p + 2);
      if (file[p] == 66 && file[p + 1] == 67 && slen == 2) bsize = (int64_t)rd16(file + p + 4) + 1;
      p += 4 + slen;
    }
    if (max_blocks && (bsize < 0 || o + (uint64_t)bsize > file_len)) break; /* sample buffer ends inside a member */
    if (bsize < 0 || o + (uint64_t)bsize > file_len) { snprintf(out->error, sizeof out->error, "bad BGZF size at %llu", (unsigned long long)o); return 1; }
    if (nb == cap) { cap *= 2; coff = (uint64_t*)realloc(coff, (cap + 1) * 8); uoff = (uint64_t*)realloc(uoff, (cap + 1) * 8); }
    coff[nb] = o; uoff[nb] = uo;
    uo += rd32(file + o + bsize - 4);
    o += (uint64_t)bsize;
    nb++;
  }
  coff[nb] = o; uoff[nb] = uo;
  out->n_blocks = nb; out->compressed_bytes = o; out->inflated_bytes = uo;
  /* phase A: inflate */
  double t0 = now_s();
  uint8_t* u = (uint8_t*)malloc(uo + 64);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  inflate_job* ij = (inflate_job*)calloc((size_t)threads, sizeof(inflate_job));
  for (int t = 0; t < threads; t++) {
    ij[t] = (inflate_job){file, coff, uoff, u, nb * (uint64_t)t / (uint64_t)threads, nb * (uint64_t)(t + 1) / (uint64_t)threads, 0};
    pthread_create(&th[t], NULL, inflate_worker, &ij[t]);
  }
  int ierr = 0;
  for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (ij[t].err) ierr = ij[t].err; }
  out->seconds_inflate = now_s() - t0;
  if (ierr) { snprintf(out->error, sizeof out->error, ierr == 2 ? "CRC mismatch" : "inflate failed"); return 1; }
  /* header */
  if (uo < 12 || memcmp(u, "BAM\1", 4)) { snprintf(out->error, sizeof out->error, "not BAM"); return 1; }
  uint64_t p = 8 + (uint64_t)rdi32(u + 4);
  int n_ref = rdi32(u + p); p += 4;
  const char** ref_names = (const char**)malloc(sizeof(char*) * (size_t)(n_ref + 1));
  uint32_t* ref_len = (uint32_t*)malloc(4 * (size_t)(n_ref + 1));
  for (int r = 0; r < n_ref; r++) {
    int32_t ln = rdi32(u + p);
    ref_names[r] = (const char*)u + p + 4;
    ref_len[r] = (uint32_t)ln - 1;
    p += 8 + (uint64_t)ln;
  }
  /* record chain */
  t0 = now_s();
  uint64_t rcap = uo / 300 + 16, n = 0;
  uint64_t* rec = (uint64_t*)malloc(rcap * 8);
  while (p + 4 <= uo) {
    uint32_t bs = rd32(u + p);
    if (bs < 32 || p + 4 + bs > uo) {
      if (max_blocks) break; /* sample ends inside a record */
      snprintf(out->error, sizeof out->error, "truncated record"); return 1;
    }
    if (n == rcap) { rcap *= 2; rec = (uint64_t*)realloc(rec, rcap * 8); }
    rec[n++] = p;
    p += 4 + bs;
  }
  out->seconds_chain = now_s() - t0;
  out->n_rows = n;
  /* phase B: columns */
  t0 = now_s();
  if (build_columns) {
    out->n_cols = 12 + n_tags;
    for (int c = 0; c < out->n_cols; c++) {
      int var = c < 12 ? CORE_VAR[c] : (tag_kinds[c - 12] == 3);
      oracle_col* oc = &out->cols[c];
      if (var) oc->offsets = (int64_t*)calloc(n + 1, 8);
      else { oc->data = (uint8_t*)malloc((n ? n : 1) * 4); oc->data_len = n * 4; }
      int nullable = (c == 1 || c == 2 || c == 3 || c == 7 || c == 8 || c >= 12);
      if (nullable) oc->valid = (uint8_t*)calloc((n + 7) / 8 + 8, 1);
    }
    col_job* cj = (col_job*)calloc((size_t)threads, sizeof(col_job));
    for (int pass = 0; pass < 2; pass++) {
      for (int t = 0; t < threads; t++) {
        cj[t] = (col_job){u, rec, n * (uint64_t)t / (uint64_t)threads, n * (uint64_t)(t + 1) / (uint64_t)threads, zero_based,
                          ref_names, ref_len, n_ref, n_tags, tags, tag_kinds, pass, out->cols, {0}, 0};
        pthread_create(&th[t], NULL, col_worker, &cj[t]);
      }
      int cerr = 0;
      for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (cj[t].err) cerr = cj[t].err; }
      if (cerr) { snprintf(out->error, sizeof out->error, "column build error %d", cerr); return 1; }
      if (pass == 0) {
        for (int c = 0; c < out->n_cols; c++) {
          oracle_col* oc = &out->cols[c];
          if (!oc->offsets) continue;
          for (uint64_t i = 0; i < n; i++) oc->offsets[i + 1] += oc->offsets[i];
          oc->data_len = (uint64_t)oc->offsets[n];
          oc->data = (uint8_t*)malloc(oc->data_len ? oc->data_len : 1);
        }
      }
    }
    free(cj);
  }
  out->seconds_columns = now_s() - t0;
  out->seconds_total = now_s() - t_all;
  free(th); free(ij); free(rec); free(ref_names); free(ref_len); free(u); free(coff); free(uoff);
  return 0;
}

int oracle_bam_scan(const char* path, int zero_based, int threads, uint64_t max_blocks, int n_tags, const char* tags,
                    const int* tag_kinds, int build_columns, oracle_result* out) {
  FILE* f = fopen(path, "rb");
  if (!f) { memset(out, 0, sizeof *out); snprintf(out->error, sizeof out->error, "cannot open %s", path); return 1; }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* d = (uint8_t*)malloc((size_t)sz + 1);
  size_t got = fread(d, 1, (size_t)sz, f);
  fclose(f);
  int rc = got == (size_t)sz ? oracle_bam_scan_mem(d, (uint64_t)sz, zero_based, threads, max_blocks, n_tags, tags, tag_kinds, build_columns, out) : 1;
  free(d);
  return rc;
}

/* =================================================================================================
 * Streaming form of the BAM scan: the shape of the reference's executor, used as the CPU baseline.
 *
 * The reference runs one OS thread per partition (bio-format-core/src/sync_stream.rs:19-29); a partition's thread owns a
 * BGZF reader positioned at a record-aligned virtual offset from the index, inflates ONE member at a time, decodes the
 * records in it, appends to its own column builders and hands a RecordBatch over every `batch_rows` rows
 * (bio-format-bam/src/physical_exec.rs:408-573, 857-862) -- nothing is shared between partitions and nothing is as large
 * as the file.  oracle_bam_scan_mem above materialises whole columns (for parity checks) and pays for that with phases no
 * partition thread of the reference has: one serial walk of the record chain over the whole inflated file, serial prefix
 * sums of six offset columns, first-touch of file-sized arrays.  It stopped scaling at 8 threads, which understated the CPU.
 *
 * oracle_bam_stream_plan plays the index: it returns, for T partitions of the first `max_blocks` members, the virtual
 * offset (compressed offset of a member, offset inside its payload) of the first record that starts in each partition's
 * member range.  It is not timed (an index is built before a scan).  oracle_bam_scan_stream then runs the T partitions.
 * ================================================================================================= */
typedef struct {
  uint64_t n_rows, n_batches, n_blocks, compressed_bytes, inflated_bytes, arrow_bytes;
  uint64_t col_bytes[12];      /* value bytes per core column (4 x rows for the fixed ones) */
  uint64_t col_sum[12];        /* sum of all value bytes per core column (order-independent checksum) */
  uint64_t n_valid[12];        /* non-NULL rows per column */
  double seconds_total;        /* wall time of the parallel region */
  double seconds_inflate_avg, seconds_inflate_max;   /* per-thread time in inflate + CRC32 */
  double seconds_build_avg, seconds_build_max;       /* per-thread time decoding records into builders */
  int threads, used_libdeflate;
  char error[256];
} oracle_stream_result;

int oracle_bam_stream_plan(const uint8_t* file, uint64_t file_len, uint64_t max_blocks, int threads, uint64_t* start_coff,
                           uint64_t* start_within, uint64_t* n_blocks_out, char* err, int err_cap) {
  oracle_result r;
  /* members + record chain through the materialising scan (columns off): untimed planning */
  if (oracle_bam_scan_mem(file, file_len, 1, threads > 16 ? 16 : threads, max_blocks, 0, "", NULL, 0, &r)) { snprintf(err, (size_t)err_cap, "%s", r.error); return 1; }
  /* redo the framing + chain here to get at the offsets (cheap next to the inflate above, and untimed) */
  uint64_t cap = r.n_blocks + 1, nb = 0, o = 0, uo = 0;
  uint64_t* coff = (uint64_t*)malloc((cap + 1) * 8);
  uint64_t* uoff = (uint64_t*)malloc((cap + 1) * 8);
  while (nb < r.n_blocks) {
    uint32_t xlen = rd16(file + o + 10);
    int64_t bsize = -1;
    for (uint64_t p = o + 12; p + 4 <= o + 12 + xlen;) {
      uint32_t slen = rd16(file + p + 2);
      if (file[p] == 66 && file[p + 1] == 67 && slen == 2) bsize = (int64_t)rd16(file + p + 4) + 1;
      p += 4 + slen;
    }
    coff[nb] = o; uoff[nb] = uo;
    uo += rd32(file + o + bsize - 4);
    o += (uint64_t)bsize;
    nb++;
  }
  coff[nb] = o; uoff[nb] = uo;
  uint8_t* u = (uint8_t*)malloc(uo + 64);
  inflate_job ij = {file, coff, uoff, u, 0, nb, 0};
  inflate_worker(&ij);
  if (ij.err) { snprintf(err, (size_t)err_cap, "inflate failed"); return 1; }
  uint64_t p = 8 + (uint64_t)rdi32(u + 4);
  int n_ref = rdi32(u + p); p += 4;
  for (int k = 0; k < n_ref; k++) p += 8 + (uint64_t)rdi32(u + p);
  /* partition t owns members [nb t / T, nb (t + 1) / T): its first record is the first one starting at / after uoff of its first member */
  int t = 0;
  uint64_t b = 0;
  for (t = 0; t < threads; t++) { start_coff[t] = coff[nb]; start_within[t] = 0; }
  t = 0;
  while (p + 4 <= uo && t < threads) {
    const uint64_t lo = uoff[nb * (uint64_t)t / (uint64_t)threads];
    if (p >= lo) {
      while (uoff[b + 1] <= p) b++;
      start_coff[t] = coff[b]; start_within[t] = p - uoff[b];
      t++;
      continue;
    }
    uint32_t bs = rd32(u + p);
    if (bs < 32 || p + 4 + bs > uo) break;
    p += 4 + bs;
  }
  *n_blocks_out = nb;
  free(u); free(coff); free(uoff);
  oracle_free(&r);
  return 0;
}

typedef struct { uint8_t* p; size_t n, cap; } sbuf;
static inline uint8_t* sb_room(sbuf* g, size_t k) {
  if (g->n + k > g->cap) { g->cap = (g->n + k) * 2 + 4096; g->p = (uint8_t*)realloc(g->p, g->cap); }
  uint8_t* d = g->p + g->n;
  g->n += k;
  return d;
}
typedef struct {
  const uint8_t* file;
  uint64_t file_len, c_lo, w_lo, c_hi, w_hi;   /* [start, end) as virtual offsets (member coff, offset in payload) */
  int zero_based;
  uint32_t batch_rows;
  oracle_stream_result acc;                      /* this thread's share */
  double t_inflate, t_build;
  int err;
} stream_job;

/* a batch is handed over: account for its buffers, fold the order-independent checksums, empty the builders for reuse */
static void stream_hand_over(stream_job* j, sbuf* vdata, sbuf* voff, sbuf* fdata, sbuf* valid, uint32_t rows, uint64_t* arrow_bytes) {
  static const int VAR[6] = {0, 1, 5, 7, 9, 10}, FIX[6] = {2, 3, 4, 6, 8, 11};
  for (int c = 0; c < 6; c++) {
    uint64_t sm = 0;
    for (size_t k = 0; k < vdata[c].n; k++) sm += vdata[c].p[k];
    j->acc.col_bytes[VAR[c]] += vdata[c].n; j->acc.col_sum[VAR[c]] += sm;
    *arrow_bytes += vdata[c].n + ((uint64_t)rows + 1) * 4;
    vdata[c].n = 0; voff[c].n = 0;
    sm = 0;
    for (size_t k = 0; k < fdata[c].n; k++) sm += fdata[c].p[k];
    j->acc.col_bytes[FIX[c]] += fdata[c].n; j->acc.col_sum[FIX[c]] += sm;
    *arrow_bytes += fdata[c].n;
    fdata[c].n = 0;
  }
  for (int c = 0; c < 12; c++) {
    if (valid[c].n) {
      uint64_t nv = 0;
      for (size_t k = 0; k < valid[c].n; k++) nv += valid[c].p[k];
      j->acc.n_valid[c] += nv;
      *arrow_bytes += (valid[c].n + 7) / 8;
      valid[c].n = 0;
    } else {
      j->acc.n_valid[c] += rows;   /* columns without NULLs */
    }
  }
  j->acc.n_rows += rows;
  j->acc.n_batches++;
}

static void* stream_worker(void* arg) {
  stream_job* j = (stream_job*)arg;
  static const char SEQ[] = "=ACMGRSVTWYHKDBN";
  static const char OPS[] = "MIDNSHP=X???????";
  void* d = ld_allocd ? ld_allocd() : NULL;
  /* builders of one batch: 6 var-len columns (offsets + bytes), 6 fixed, validity as one byte per row (packed at hand-over) */
  sbuf vdata[6] = {{0}}, voff[6] = {{0}}, fdata[6] = {{0}}, valid[12] = {{0}};
  uint32_t rows = 0;
  size_t ucap = 1 << 18, have = 0;
  uint8_t* u = (uint8_t*)malloc(ucap);
  uint64_t pos_c = j->c_lo;
  size_t off = 0;          /* parse position in u */
  int first = 1;
  /* reference names are needed for chrom / mate_chrom: the header is read by partition 0 of the reference's provider at
     open; here every partition decodes the leading members once (outside its timers would hide real work of nobody: it
     is part of open, not of the scan) */
  const char** ref_names = NULL; uint32_t* ref_len = NULL; int n_ref = 0;
  uint8_t* hdrbuf = NULL;
  {
    size_t hcap = 1 << 20, hh = 0;
    hdrbuf = (uint8_t*)malloc(hcap);
    uint64_t c = 0;
    for (;;) {
      const uint8_t* m = j->file + c;
      uint32_t xlen = rd16(m + 10);
      int64_t bsize = -1;
      for (uint64_t q = 12; q + 4 <= 12 + (uint64_t)xlen;) { uint32_t slen = rd16(m + q + 2); if (m[q] == 66 && m[q + 1] == 67 && slen == 2) bsize = (int64_t)rd16(m + q + 4) + 1; q += 4 + slen; }
      uint32_t isize = rd32(m + bsize - 4);
      if (hh + isize > hcap) { hcap = (hh + isize) * 2; hdrbuf = (uint8_t*)realloc(hdrbuf, hcap); }
      size_t got = 0;
      if (d) { if (ld_decomp(d, m + 12 + xlen, (size_t)(bsize - 12 - xlen - 8), hdrbuf + hh, isize, &got) != 0) { j->err = 1; return NULL; } }
      else { uLongf dl = isize; z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15); zs.next_in = (Bytef*)(m + 12 + xlen); zs.avail_in = (uInt)(bsize - 12 - xlen - 8); zs.next_out = hdrbuf + hh; zs.avail_out = isize; inflate(&zs, Z_FINISH); inflateEnd(&zs); (void)dl; }
      hh += isize; c += (uint64_t)bsize;
      if (hh >= 12) {
        uint64_t q = 8 + (uint64_t)rdi32(hdrbuf + 4);
        if (q + 4 <= hh) {
          int nr = rdi32(hdrbuf + q); uint64_t e = q + 4; int ok = 1;
          for (int k = 0; k < nr; k++) { if (e + 4 > hh) { ok = 0; break; } e += 8 + (uint64_t)rdi32(hdrbuf + e); if (e > hh) { ok = 0; break; } }
          if (ok) {
            n_ref = nr; ref_names = (const char**)malloc(sizeof(char*) * (size_t)(nr + 1)); ref_len = (uint32_t*)malloc(4 * (size_t)(nr + 1));
            e = q + 4;
            for (int k = 0; k < nr; k++) { int32_t ln = rdi32(hdrbuf + e); ref_names[k] = (const char*)hdrbuf + e + 4; ref_len[k] = (uint32_t)ln - 1; e += 8 + (uint64_t)ln; }
            break;
          }
        }
      }
      if (c >= j->file_len) { j->err = 3; return NULL; }
    }
  }
  uint64_t arrow_bytes = 0;
  uint64_t head_c = j->c_lo, head_w = j->w_lo;   /* virtual offset of the carried (cut) record at u[0] */
  size_t data_start = 0;                          /* where the current member's payload begins in u */
  uint64_t cur_c = 0;                             /* compressed offset of the current member */
  for (;;) {
    /* ---- next member: needed while a record of this partition is incomplete, or the next member starts before the end ---- */
    const int pending = off < have;
    if (!pending && !first && (pos_c > j->c_hi || (pos_c == j->c_hi && j->w_hi == 0))) break;
    if (first && (pos_c > j->c_hi || (pos_c == j->c_hi && j->w_lo >= j->w_hi))) break;   /* an empty partition */
    if (pos_c + 28 > j->file_len) break;
    double t0 = now_s();
    const uint8_t* m = j->file + pos_c;
    uint32_t xlen = rd16(m + 10);
    int64_t bsize = -1;
    for (uint64_t q = 12; q + 4 <= 12 + (uint64_t)xlen;) { uint32_t slen = rd16(m + q + 2); if (m[q] == 66 && m[q + 1] == 67 && slen == 2) bsize = (int64_t)rd16(m + q + 4) + 1; q += 4 + slen; }
    if (bsize < 0 || pos_c + (uint64_t)bsize > j->file_len) break;   /* the sample ends inside a member */
    uint32_t isize = rd32(m + bsize - 4), crc = rd32(m + bsize - 8);
    /* keep the unparsed tail (a record cut by the member's end) at the front, drop what has been decoded */
    if (off) {
      if (off >= data_start) { head_c = cur_c; head_w = off - data_start; }
      memmove(u, u + off, have - off);
      have -= off; off = 0;
    }
    data_start = have;
    if (have + isize > ucap) { ucap = (have + isize) * 2; u = (uint8_t*)realloc(u, ucap); }
    size_t got = 0;
    if (d) { if (ld_decomp(d, m + 12 + xlen, (size_t)(bsize - 12 - xlen - 8), u + have, isize, &got) != 0 || got != isize) { j->err = 1; break; } }
    else {
      z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
      zs.next_in = (Bytef*)(m + 12 + xlen); zs.avail_in = (uInt)(bsize - 12 - xlen - 8); zs.next_out = u + have; zs.avail_out = isize;
      int rc = inflate(&zs, Z_FINISH); got = zs.total_out; inflateEnd(&zs);
      if ((rc != Z_STREAM_END && !(rc == Z_BUF_ERROR && isize == 0)) || got != isize) { j->err = 1; break; }
    }
    if (crc_of(u + have, isize) != crc) { j->err = 2; break; }   /* noodles-bgzf checks every block */
    have += isize;
    j->acc.n_blocks++; j->acc.compressed_bytes += (uint64_t)bsize; j->acc.inflated_bytes += isize;
    cur_c = pos_c;
    pos_c += (uint64_t)bsize;
    if (first) { off = (size_t)j->w_lo; first = 0; }
    double t1 = now_s();
    j->t_inflate += t1 - t0;
    /* ---- records that are complete in the buffer ---- */
    int done = 0;
    while (off + 4 <= have) {
      /* a record belongs to this partition when it STARTS before the partition's end (virtual offsets) */
      const uint64_t rc_ = off >= data_start ? cur_c : head_c, rw_ = off >= data_start ? (uint64_t)(off - data_start) : head_w;
      if (rc_ > j->c_hi || (rc_ == j->c_hi && rw_ >= j->w_hi)) { done = 1; break; }
      const uint8_t* r = u + off;
      const uint32_t bs = rd32(r);
      if (bs < 32) { j->err = 4; done = 1; break; }
      if (off + 4 + bs > have) break;   /* cut by the member's end: carried */
      int32_t refid = rdi32(r + 4), pos = rdi32(r + 8);
      uint32_t lrn = r[12], mapq = r[13], ncig = rd16(r + 16), flag = rd16(r + 18);
      int32_t lseq = rdi32(r + 20), nref = rdi32(r + 24), npos = rdi32(r + 28), tlen = rdi32(r + 32);
      const uint8_t* cg = r + 36 + lrn;
      const uint8_t* sq = cg + 4 * ncig;
      const uint8_t* ql = sq + (lseq + 1) / 2;
      if (refid >= n_ref || nref >= n_ref || lseq < 0 || 32u + lrn + 4u * ncig + (uint32_t)((lseq + 1) / 2) + (uint32_t)lseq > bs) { j->err = 5; done = 1; break; }
      /* name */
      { const uint32_t n = lrn ? lrn - 1 : 0; memcpy(sb_room(&vdata[0], n), r + 36, n); }
      /* chrom / mate_chrom */
      if (refid >= 0) memcpy(sb_room(&vdata[1], ref_len[refid]), ref_names[refid], ref_len[refid]);
      if (nref >= 0) memcpy(sb_room(&vdata[3], ref_len[nref]), ref_names[nref], ref_len[nref]);
      /* cigar + reference span */
      uint32_t span = 0;
      for (uint32_t k = 0; k < ncig; k++) {
        uint32_t v = rd32(cg + 4 * k), n = v >> 4, nd = ndigits(n);
        if ((0x18Du >> (v & 15)) & 1u) span += n;
        uint8_t* dd = sb_room(&vdata[2], nd + 1);
        for (int q = (int)nd - 1; q >= 0; q--) { dd[q] = (uint8_t)('0' + n % 10); n /= 10; }
        dd[nd] = (uint8_t)OPS[v & 15];
      }
      /* sequence, qualities */
      { uint8_t* dd = sb_room(&vdata[4], (size_t)lseq); for (int32_t k = 0; k < lseq; k++) { uint8_t b = sq[k >> 1]; dd[k] = (uint8_t)SEQ[(k & 1) ? (b & 15) : (b >> 4)]; } }
      {
        uint8_t* dd = sb_room(&vdata[5], 2 * (size_t)lseq);
        size_t w = 0;
        for (int32_t k = 0; k < lseq; k++) {
          uint32_t ch = ((uint32_t)ql[k] + 33u) & 0xFFu;
          if (ch < 128u) dd[w++] = (uint8_t)ch;
          else { dd[w++] = (uint8_t)(0xC0u | (ch >> 6)); dd[w++] = (uint8_t)(0x80u | (ch & 0x3Fu)); }
        }
        vdata[5].n -= 2 * (size_t)lseq - w;
      }
      for (int c = 0; c < 6; c++) { uint32_t e = (uint32_t)vdata[c].n; memcpy(sb_room(&voff[c], 4), &e, 4); }
      const uint32_t end1 = pos >= 0 ? (uint32_t)pos + span : 0;
      const uint32_t fx[6] = {pos >= 0 ? (uint32_t)(j->zero_based ? pos : pos + 1) : 0, end1, flag, mapq,
                              npos >= 0 ? (uint32_t)(j->zero_based ? npos : npos + 1) : 0, (uint32_t)tlen};
      for (int c = 0; c < 6; c++) memcpy(sb_room(&fdata[c], 4), &fx[c], 4);
      *sb_room(&valid[1], 1) = refid >= 0; *sb_room(&valid[2], 1) = pos >= 0; *sb_room(&valid[3], 1) = end1 != 0;
      *sb_room(&valid[7], 1) = nref >= 0; *sb_room(&valid[8], 1) = npos >= 0;
      rows++;
      off += 4 + (size_t)bs;
      if (rows == j->batch_rows) { stream_hand_over(j, vdata, voff, fdata, valid, rows, &arrow_bytes); rows = 0; }
    }
    j->t_build += now_s() - t1;
    if (done) break;
  }
  if (rows) stream_hand_over(j, vdata, voff, fdata, valid, rows, &arrow_bytes);   /* the short last batch */
  j->acc.arrow_bytes = arrow_bytes;
  for (int c = 0; c < 6; c++) { free(vdata[c].p); free(voff[c].p); free(fdata[c].p); }
  for (int c = 0; c < 12; c++) free(valid[c].p);
  free(u); free(hdrbuf); free(ref_names); free(ref_len);
  if (d) ld_freed(d);
  return NULL;
}

int oracle_bam_scan_stream(const uint8_t* file, uint64_t file_len, int zero_based, int threads, const uint64_t* start_coff,
                           const uint64_t* start_within, uint64_t end_coff, uint32_t batch_rows, oracle_stream_result* out) {
  memset(out, 0, sizeof *out);
  load_libdeflate();
  out->used_libdeflate = ld_allocd != NULL;
  out->threads = threads;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  stream_job* sj = (stream_job*)calloc((size_t)threads, sizeof(stream_job));
  const double t0 = now_s();
  for (int t = 0; t < threads; t++) {
    sj[t].file = file; sj[t].file_len = file_len; sj[t].zero_based = zero_based; sj[t].batch_rows = batch_rows ? batch_rows : 8192;
    sj[t].c_lo = start_coff[t]; sj[t].w_lo = start_within[t];
    sj[t].c_hi = t + 1 < threads ? start_coff[t + 1] : end_coff; sj[t].w_hi = t + 1 < threads ? start_within[t + 1] : 0;
    pthread_create(&th[t], NULL, stream_worker, &sj[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  out->seconds_total = now_s() - t0;
  int rc = 0;
  for (int t = 0; t < threads; t++) {
    if (sj[t].err) { snprintf(out->error, sizeof out->error, "partition %d: error %d", t, sj[t].err); rc = 1; }
    out->n_rows += sj[t].acc.n_rows; out->n_batches += sj[t].acc.n_batches; out->n_blocks += sj[t].acc.n_blocks;
    out->compressed_bytes += sj[t].acc.compressed_bytes; out->inflated_bytes += sj[t].acc.inflated_bytes; out->arrow_bytes += sj[t].acc.arrow_bytes;
    for (int c = 0; c < 12; c++) { out->col_bytes[c] += sj[t].acc.col_bytes[c]; out->col_sum[c] += sj[t].acc.col_sum[c]; out->n_valid[c] += sj[t].acc.n_valid[c]; }
    out->seconds_inflate_avg += sj[t].t_inflate / threads; out->seconds_build_avg += sj[t].t_build / threads;
    if (sj[t].t_inflate > out->seconds_inflate_max) out->seconds_inflate_max = sj[t].t_inflate;
    if (sj[t].t_build > out->seconds_build_max) out->seconds_build_max = sj[t].t_build;
  }
  free(th); free(sj);
  return rc;
}

/* =================================================================================================
 * VCF (oracle/vcf_oracle.py restated in C for medium-scale checks and as the CPU baseline of the VCF
 * bench lines).  Decodes BGZF members [b0, b1) with libdeflate/zlib, splits the text [x0, x1) of that
 * range into lines, and builds every column of the scan into per-thread growable buffers: the 8 core
 * columns (bio-format-vcf/src/physical_exec.rs:1043-1093), the INFO columns given by `info_names` /
 * `info_kinds` (load_infos_single_pass :544-640) and, when n_samples > 0, the FORMAT keys GT/GQ/DP of every
 * sample (MultiSampleFormatBuilder::append_record :1634-1828) followed by the list UDFs of the config-4
 * query (udfs.rs:67-110, 606-650).  The result carries row counts and checksums of every column so the
 * Python oracle (and the GPU path) can be compared against it without shipping the columns.
 * ================================================================================================= */
typedef struct {
  uint64_t n_rows, n_blocks, compressed_bytes, inflated_bytes;
  uint64_t sum_start, sum_end, core_str_bytes, n_qual_valid;
  double sum_qual;
  uint64_t info_int_sum, info_valid, info_float_valid, info_str_bytes, info_flag_true, info_list_elems;
  double info_float_sum;
  uint64_t cells, gt_bytes, gt_valid, gq_sum, gq_valid, dp_sum, dp_valid;
  double avg_gq_sum, avg_dp_sum;
  uint64_t avg_gq_valid, avg_dp_valid, gq_gte_true, dp_gte_true, dp_lte_true;
  double seconds_inflate, seconds_parse, seconds_total;
  int threads, used_libdeflate;
  char error[256];
} oracle_vcf_result;

typedef struct { uint8_t* p; size_t n, cap; } gbuf;
static inline void gb_put(gbuf* g, const void* src, size_t k) {
  if (g->cap - g->n < k) { g->cap = (g->cap + k) * 2 + 4096; g->p = (uint8_t*)realloc(g->p, g->cap); }
  memcpy(g->p + g->n, src, k);
  g->n += k;
}
static inline void gb_u32(gbuf* g, uint32_t v) { gb_put(g, &v, 4); }
static inline void gb_f64(gbuf* g, double v) { gb_put(g, &v, 8); }

typedef struct {
  const uint8_t* u;
  uint64_t a, b;             /* this thread's text range (line aligned) */
  int zero_based, n_info, n_samples;
  const char* const* info_names;
  const int* info_kinds;     /* 0 int, 1 float, 2 flag, 3 string, 4 list<int>, 5 list<float>, 6 list<string> */
  oracle_vcf_result r;       /* per-thread partial sums */
  gbuf cols[40];             /* column bytes: the work a builder does; freed at the end */
  int err;
} vcf_job;

static inline const uint8_t* find_byte(const uint8_t* p, const uint8_t* e, int c) {
  const uint8_t* q = (const uint8_t*)memchr(p, c, (size_t)(e - p));
  return q ? q : e;
}
static void* vcf_worker(void* arg) {
  vcf_job* j = (vcf_job*)arg;
  const uint8_t* u = j->u;
  oracle_vcf_result* r = &j->r;
  const uint8_t* p = u + j->a;
  const uint8_t* end = u + j->b;
  int32_t* gq = j->n_samples ? (int32_t*)malloc(sizeof(int32_t) * (size_t)j->n_samples * 2) : NULL;
  int32_t* dp = gq ? gq + j->n_samples : NULL;
  while (p < end) {
    const uint8_t* le = find_byte(p, end, '\n');
    const uint8_t* lend = le;
    if (lend > p && lend[-1] == '\r') lend--;
    const uint8_t* f[10];
    const uint8_t* fe[10];
    const uint8_t* q = p;
    int nf = 0;
    while (nf < 9) {
      const uint8_t* t = find_byte(q, lend, '\t');
      f[nf] = q; fe[nf] = t; nf++;
      if (t == lend) break;
      q = t + 1;
    }
    if (nf < 8) { j->err = 1; break; }
    const uint8_t* samples = (nf == 9 && fe[8] < lend) ? fe[8] + 1 : lend;
    /* core */
    uint64_t pos = 0;
    for (const uint8_t* c = f[1]; c < fe[1]; c++) pos = pos * 10 + (uint64_t)(*c - '0');
    const size_t rl = (size_t)(fe[3] - f[3]);
    int n_alt = (fe[4] - f[4] == 1 && f[4][0] == '.') ? 0 : 1;
    for (const uint8_t* c = f[4]; c < fe[4]; c++) n_alt += *c == ',';
    int snv = rl == 1 && n_alt == 1 && fe[4] - f[4] == 1 && strchr("ACGT", f[3][0]) && strchr("ACGT", f[4][0]);
    uint64_t vend = pos + rl - 1;
    gb_put(&j->cols[0], f[0], (size_t)(fe[0] - f[0]));
    r->core_str_bytes += (uint64_t)(fe[0] - f[0]);
    for (int k = 2; k <= 6; k++) {
      if (k == 5) continue;
      size_t l = (size_t)(fe[k] - f[k]);
      if (l == 1 && f[k][0] == '.' && k != 3) l = 0;
      if (k == 4) {  /* alleles re-joined with '|' */
        size_t o = j->cols[k].n;
        gb_put(&j->cols[k], f[k], l);
        for (size_t x = 0; x < l; x++) if (j->cols[k].p[o + x] == ',') j->cols[k].p[o + x] = '|';
      } else gb_put(&j->cols[k], f[k], l);
      gb_u32(&j->cols[k + 10], (uint32_t)j->cols[k].n);
      r->core_str_bytes += l;
    }
    if (!(fe[5] - f[5] == 1 && f[5][0] == '.')) {
      char tmp[64];
      size_t l = (size_t)(fe[5] - f[5]);
      if (l > 63) l = 63;
      memcpy(tmp, f[5], l); tmp[l] = 0;
      const double qv = (double)strtof(tmp, NULL);
      gb_f64(&j->cols[5], qv);
      r->sum_qual += qv; r->n_qual_valid++;
    } else gb_f64(&j->cols[5], 0.0);
    /* INFO: one pass, keys matched against the selected names */
    if (!(fe[7] - f[7] == 1 && f[7][0] == '.')) {
      const uint8_t* e7 = fe[7];
      for (const uint8_t* k0 = f[7]; k0 < e7;) {
        const uint8_t* ke = find_byte(k0, e7, ';');
        const uint8_t* eq = find_byte(k0, ke, '=');
        const size_t kl = (size_t)(eq - k0);
        if (kl == 3 && memcmp(k0, "END", 3) == 0 && eq < ke) vend = strtoull((const char*)eq + 1, NULL, 10);
        for (int x = 0; x < j->n_info; x++) {
          if (strlen(j->info_names[x]) != kl || memcmp(j->info_names[x], k0, kl) != 0) continue;
          const int kind = j->info_kinds[x];
          gbuf* g = &j->cols[20 + (x % 18)];
          if (eq == ke) { if (kind == 2) { r->info_flag_true++; gb_put(g, "\1", 1); } break; }
          const uint8_t* v0 = eq + 1;
          if (ke - v0 == 1 && v0[0] == '.') break;
          if (kind == 0) { const long v = strtol((const char*)v0, NULL, 10); r->info_int_sum += (uint64_t)v; r->info_valid++; gb_u32(g, (uint32_t)v); }
          else if (kind == 1) { char t[48]; size_t l = (size_t)(ke - v0); if (l > 47) l = 47; memcpy(t, v0, l); t[l] = 0;
                                const float fv = strtof(t, NULL); r->info_float_sum += (double)fv; r->info_float_valid++; gb_put(g, &fv, 4); }
          else if (kind == 3) { r->info_str_bytes += (uint64_t)(ke - v0); r->info_valid++; gb_put(g, v0, (size_t)(ke - v0)); }
          else {
            r->info_valid++;
            for (const uint8_t* a0 = v0; a0 <= ke;) {
              const uint8_t* ae = find_byte(a0, ke, ',');
              r->info_list_elems++;
              if (!(ae - a0 == 1 && a0[0] == '.')) {
                if (kind == 4) { const long v = strtol((const char*)a0, NULL, 10); r->info_int_sum += (uint64_t)v; gb_u32(g, (uint32_t)v); }
                else if (kind == 5) { char t[48]; size_t l = (size_t)(ae - a0); if (l > 47) l = 47; memcpy(t, a0, l); t[l] = 0;
                                      const float fv = strtof(t, NULL); r->info_float_sum += (double)fv; r->info_float_valid++; gb_put(g, &fv, 4); }
                else { r->info_str_bytes += (uint64_t)(ae - a0); gb_put(g, a0, (size_t)(ae - a0)); }
              }
              if (ae == ke) break;
              a0 = ae + 1;
            }
          }
          break;
        }
        k0 = ke + 1;
      }
    }
    const uint64_t endcol = snv ? pos : vend;
    gb_u32(&j->cols[1], (uint32_t)(j->zero_based ? pos - 1 : pos));
    gb_u32(&j->cols[7], (uint32_t)endcol);
    r->sum_start += j->zero_based ? pos - 1 : pos;
    r->sum_end += endcol;
    /* FORMAT GT:GQ:DP (positions taken from the FORMAT column) */
    if (j->n_samples > 0 && nf == 9) {
      int pgt = -1, pgq = -1, pdp = -1, kidx = 0;
      for (const uint8_t* k0 = f[8]; k0 <= fe[8];) {
        const uint8_t* ke = find_byte(k0, fe[8], ':');
        if (ke - k0 == 2 && k0[0] == 'G' && k0[1] == 'T') pgt = kidx;
        else if (ke - k0 == 2 && k0[0] == 'G' && k0[1] == 'Q') pgq = kidx;
        else if (ke - k0 == 2 && k0[0] == 'D' && k0[1] == 'P') pdp = kidx;
        kidx++;
        if (ke == fe[8]) break;
        k0 = ke + 1;
      }
      const uint8_t* s0 = samples;
      int64_t sgq = 0, sdp = 0;
      uint32_t ngq = 0, ndp = 0;
      for (int s = 0; s < j->n_samples; s++) {
        const uint8_t* se = s0 <= lend ? find_byte(s0, lend, '\t') : lend;
        gq[s] = INT32_MIN; dp[s] = INT32_MIN;
        r->cells++;
        if (s0 < lend && !(se - s0 == 1 && s0[0] == '.')) {
          int ki = 0;
          for (const uint8_t* v0 = s0; v0 <= se;) {
            const uint8_t* ve = find_byte(v0, se, ':');
            const int missing = ve - v0 == 1 && v0[0] == '.';
            if (!missing) {
              if (ki == pgt) { gb_put(&j->cols[38], v0, (size_t)(ve - v0)); r->gt_bytes += (uint64_t)(ve - v0); r->gt_valid++; }
              else if (ki == pgq) { gq[s] = (int32_t)strtol((const char*)v0, NULL, 10); r->gq_sum += (uint64_t)gq[s]; r->gq_valid++; }
              else if (ki == pdp) { dp[s] = (int32_t)strtol((const char*)v0, NULL, 10); r->dp_sum += (uint64_t)dp[s]; r->dp_valid++; }
            }
            ki++;
            if (ve == se) break;
            v0 = ve + 1;
          }
        }
        if (gq[s] != INT32_MIN) { sgq += gq[s]; ngq++; r->gq_gte_true += gq[s] >= 10; }
        if (dp[s] != INT32_MIN) { sdp += dp[s]; ndp++; r->dp_gte_true += dp[s] >= 10; r->dp_lte_true += dp[s] <= 200; }
        s0 = se + 1;
      }
      gb_put(&j->cols[36], gq, sizeof(int32_t) * (size_t)j->n_samples);
      gb_put(&j->cols[37], dp, sizeof(int32_t) * (size_t)j->n_samples);
      if (ngq) { r->avg_gq_sum += (double)sgq / (double)ngq; r->avg_gq_valid++; }
      if (ndp) { r->avg_dp_sum += (double)sdp / (double)ndp; r->avg_dp_valid++; }
    }
    r->n_rows++;
    p = le + 1;
  }
  free(gq);
  for (int k = 0; k < 40; k++) { free(j->cols[k].p); j->cols[k].p = NULL; }
  return NULL;
}

/* x0 / x1: text range relative to the first decoded byte of member b0 (x1 = 0: to the end); lines are
 * assigned to threads by where they start. */
int oracle_vcf_scan_mem(const uint8_t* file, uint64_t file_len, uint64_t b0, uint64_t b1, uint64_t x0, uint64_t x1, int zero_based,
                        int threads, int n_info, const char* const* info_names, const int* info_kinds, int n_samples,
                        oracle_vcf_result* out) {
  memset(out, 0, sizeof(*out));
  load_libdeflate();
  const double t0 = now_s();
  uint64_t nb = 0, cap = 1024;
  uint64_t* coff = (uint64_t*)malloc(8 * (cap + 1));
  uint64_t* uoff = (uint64_t*)malloc(8 * (cap + 1));
  uint64_t o = 0, uo = 0;
  while (o + 28 <= file_len) {
    if (nb == cap) { cap *= 2; coff = (uint64_t*)realloc(coff, 8 * (cap + 1)); uoff = (uint64_t*)realloc(uoff, 8 * (cap + 1)); }
    const uint32_t bsize = rd16(file + o + 16) + 1;
    coff[nb] = o; uoff[nb] = uo;
    uo += rd32(file + o + bsize - 4);
    o += bsize;
    nb++;
  }
  coff[nb] = o; uoff[nb] = uo;
  if (b1 == 0 || b1 > nb) b1 = nb;
  if (b0 > b1) b0 = b1;
  const uint64_t base = uoff[b0], ulen = uoff[b1] - base;
  uint8_t* u = (uint8_t*)malloc(ulen + 64);
  for (uint64_t b = b0; b <= b1; b++) uoff[b] -= base;
  if (threads < 1) threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  inflate_job* ij = (inflate_job*)calloc((size_t)threads, sizeof(inflate_job));
  for (int t = 0; t < threads; t++) {
    ij[t].file = file; ij[t].coff = coff; ij[t].uoff = uoff; ij[t].u = u;
    ij[t].b0 = b0 + (b1 - b0) * (uint64_t)t / (uint64_t)threads;
    ij[t].b1 = b0 + (b1 - b0) * (uint64_t)(t + 1) / (uint64_t)threads;
    pthread_create(&th[t], NULL, inflate_worker, &ij[t]);
  }
  int bad = 0;
  for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); bad |= ij[t].err; }
  const double t1 = now_s();
  if (bad) { snprintf(out->error, sizeof out->error, "BGZF inflate / CRC failure"); free(u); free(coff); free(uoff); free(th); free(ij); return 1; }
  if (x1 == 0 || x1 > ulen) x1 = ulen;
  vcf_job* vj = (vcf_job*)calloc((size_t)threads, sizeof(vcf_job));
  uint64_t prev = x0;
  for (int t = 0; t < threads; t++) {
    uint64_t cut = t + 1 == threads ? x1 : x0 + (x1 - x0) * (uint64_t)(t + 1) / (uint64_t)threads;
    if (t + 1 < threads) {  /* move the cut to the next line start */
      const uint8_t* nlp = (const uint8_t*)memchr(u + cut, '\n', (size_t)(x1 - cut));
      cut = nlp ? (uint64_t)(nlp - u) + 1 : x1;
    }
    if (cut < prev) cut = prev;
    vj[t].u = u; vj[t].a = prev; vj[t].b = cut;
    vj[t].zero_based = zero_based; vj[t].n_info = n_info; vj[t].info_names = info_names; vj[t].info_kinds = info_kinds;
    vj[t].n_samples = n_samples;
    prev = cut;
    pthread_create(&th[t], NULL, vcf_worker, &vj[t]);
  }
  for (int t = 0; t < threads; t++) {
    pthread_join(th[t], NULL);
    bad |= vj[t].err;
    const oracle_vcf_result* r = &vj[t].r;
    out->n_rows += r->n_rows; out->sum_start += r->sum_start; out->sum_end += r->sum_end; out->core_str_bytes += r->core_str_bytes;
    out->n_qual_valid += r->n_qual_valid; out->sum_qual += r->sum_qual;
    out->info_int_sum += r->info_int_sum; out->info_valid += r->info_valid; out->info_float_valid += r->info_float_valid;
    out->info_str_bytes += r->info_str_bytes; out->info_flag_true += r->info_flag_true; out->info_list_elems += r->info_list_elems;
    out->info_float_sum += r->info_float_sum;
    out->cells += r->cells; out->gt_bytes += r->gt_bytes; out->gt_valid += r->gt_valid; out->gq_sum += r->gq_sum; out->gq_valid += r->gq_valid;
    out->dp_sum += r->dp_sum; out->dp_valid += r->dp_valid; out->avg_gq_sum += r->avg_gq_sum; out->avg_dp_sum += r->avg_dp_sum;
    out->avg_gq_valid += r->avg_gq_valid; out->avg_dp_valid += r->avg_dp_valid; out->gq_gte_true += r->gq_gte_true;
    out->dp_gte_true += r->dp_gte_true; out->dp_lte_true += r->dp_lte_true;
  }
  const double t2 = now_s();
  out->n_blocks = b1 - b0;
  out->compressed_bytes = coff[b1] - coff[b0];
  out->inflated_bytes = ulen;
  out->seconds_inflate = t1 - t0; out->seconds_parse = t2 - t1; out->seconds_total = t2 - t0;
  out->threads = threads; out->used_libdeflate = ld_allocd != NULL;
  free(u); free(coff); free(uoff); free(th); free(ij); free(vj);
  if (bad) { snprintf(out->error, sizeof out->error, "VCF read error: invalid record"); return 1; }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * FASTQ: BGZF -> four Utf8 columns (bio-format-fastq/src/physical_exec.rs:393-465 batch_producer, :554-589
 * build_batch_from_builders): name = first line after '@' up to the first space, description = the rest (NULL when
 * empty, :430-434), sequence, quality as raw text.  Threads take text ranges cut at record starts found with the
 * reference's resync rule (:184-248: a line starting with '@' whose line + 2 starts with '+').  Used as the
 * cpu_baseline of bench.py --format fastq and checked against oracle/fastq_oracle.py in tests/test_cpu_fastq_oracle.py.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t n_rows, n_blocks, compressed_bytes, inflated_bytes;
  uint64_t name_bytes, desc_bytes, seq_bytes, qual_bytes, desc_null;
  uint64_t byte_sum;  /* sum over rows of (sum of name bytes + 3 * desc + 5 * seq + 7 * qual bytes) */
  double seconds_inflate, seconds_parse, seconds_total;
  int threads, used_libdeflate;
  char error[256];
} oracle_fastq_result;

typedef struct {
  const uint8_t* u;
  uint64_t a, b, end;      /* records starting in [a, b); text ends at `end` */
  oracle_fastq_result r;
  gbuf cols[4], offs[4], valid;
  int err;
} fastq_job;

/* NULL when there is no such byte (find_byte above returns the end instead) */
static inline const uint8_t* find_or_null(const uint8_t* p, const uint8_t* e, int c) {
  return p < e ? (const uint8_t*)memchr(p, c, (size_t)(e - p)) : NULL;
}
static inline uint64_t bsum(const uint8_t* p, size_t n) { uint64_t s = 0; for (size_t i = 0; i < n; i++) s += p[i]; return s; }

static void* fastq_worker(void* arg) {
  fastq_job* j = (fastq_job*)arg;
  const uint8_t* u = j->u;
  uint64_t x = j->a;
  while (x < j->b) {
    const uint8_t* e = u + j->end;
    const uint8_t* l1 = find_or_null(u + x, e, '\n');
    if (!l1) break;
    const uint8_t* l2 = find_or_null(l1 + 1, e, '\n');
    if (!l2) break;
    const uint8_t* l3 = find_or_null(l2 + 1, e, '\n');
    if (!l3) break;
    const uint8_t* l4 = find_or_null(l3 + 1, e, '\n');
    const uint8_t* q_end = l4 ? l4 : e;                 /* the last record may lack the final newline */
    if (u[x] != '@' || l2[1] != '+') { j->err = 1; break; }
    const uint8_t* h = u + x + 1;
    const uint8_t* sp = (const uint8_t*)memchr(h, ' ', (size_t)(l1 - h));
    const uint8_t* name_e = sp ? sp : l1;
    const size_t n_name = (size_t)(name_e - h), n_desc = sp ? (size_t)(l1 - sp - 1) : 0;
    const size_t n_seq = (size_t)(l2 - l1 - 1), n_qual = (size_t)(q_end - l3 - 1);
    gb_put(&j->cols[0], h, n_name);
    if (n_desc) gb_put(&j->cols[1], sp + 1, n_desc); else j->r.desc_null++;
    gb_put(&j->cols[2], l1 + 1, n_seq);
    gb_put(&j->cols[3], l3 + 1, n_qual);
    for (int c = 0; c < 4; c++) gb_u32(&j->offs[c], (uint32_t)j->cols[c].n);
    { uint8_t v = n_desc ? 1 : 0; gb_put(&j->valid, &v, 1); }
    j->r.name_bytes += n_name; j->r.desc_bytes += n_desc; j->r.seq_bytes += n_seq; j->r.qual_bytes += n_qual;
    j->r.byte_sum += bsum(h, n_name) + 3 * (n_desc ? bsum(sp + 1, n_desc) : 0) + 5 * bsum(l1 + 1, n_seq) + 7 * bsum(l3 + 1, n_qual);
    j->r.n_rows++;
    if (!l4) break;
    x = (uint64_t)(l4 - u) + 1;
  }
  for (int c = 0; c < 4; c++) { free(j->cols[c].p); free(j->offs[c].p); }
  free(j->valid.p);
  return NULL;
}

/* first record start at or after x (reference resync rule); `end` when there is none */
static uint64_t fastq_sync(const uint8_t* u, uint64_t x, uint64_t end) {
  if (x == 0) return 0;
  const uint8_t* e = u + end;
  const uint8_t* p = find_or_null(u + x - 1, e, '\n');
  while (p) {
    const uint8_t* s = p + 1;
    if (s >= e) break;
    if (*s == '@') {
      const uint8_t* l1 = find_or_null(s, e, '\n');
      const uint8_t* l2 = l1 ? find_or_null(l1 + 1, e, '\n') : NULL;
      if (l2 && l2 + 1 < e && l2[1] == '+') return (uint64_t)(s - u);
    }
    p = find_or_null(s, e, '\n');
  }
  return end;
}

int oracle_fastq_scan_mem(const uint8_t* file, uint64_t file_len, uint64_t max_blocks, int threads, oracle_fastq_result* out) {
  memset(out, 0, sizeof(*out));
  load_libdeflate();
  const double t0 = now_s();
  uint64_t nb = 0, cap = 1024;
  uint64_t* coff = (uint64_t*)malloc(8 * (cap + 1));
  uint64_t* uoff = (uint64_t*)malloc(8 * (cap + 1));
  uint64_t o = 0, uo = 0;
  while (o + 28 <= file_len && (max_blocks == 0 || nb < max_blocks)) {
    if (nb == cap) { cap *= 2; coff = (uint64_t*)realloc(coff, 8 * (cap + 1)); uoff = (uint64_t*)realloc(uoff, 8 * (cap + 1)); }
    const uint32_t bsize = rd16(file + o + 16) + 1;
    if (o + bsize > file_len) break;  /* a member cut off by the caller's prefix read */
    coff[nb] = o; uoff[nb] = uo;
    uo += rd32(file + o + bsize - 4);
    o += bsize;
    nb++;
  }
  coff[nb] = o; uoff[nb] = uo;
  const uint64_t ulen = uo;
  uint8_t* u = (uint8_t*)malloc(ulen + 64);
  if (threads < 1) threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  inflate_job* ij = (inflate_job*)calloc((size_t)threads, sizeof(inflate_job));
  for (int t = 0; t < threads; t++) {
    ij[t].file = file; ij[t].coff = coff; ij[t].uoff = uoff; ij[t].u = u;
    ij[t].b0 = nb * (uint64_t)t / (uint64_t)threads;
    ij[t].b1 = nb * (uint64_t)(t + 1) / (uint64_t)threads;
    pthread_create(&th[t], NULL, inflate_worker, &ij[t]);
  }
  int bad = 0;
  for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); bad |= ij[t].err; }
  const double t1 = now_s();
  if (bad) { snprintf(out->error, sizeof out->error, "BGZF inflate / CRC failure"); free(u); free(coff); free(uoff); free(th); free(ij); return 1; }
  fastq_job* fj = (fastq_job*)calloc((size_t)threads, sizeof(fastq_job));
  uint64_t prev = 0;
  for (int t = 0; t < threads; t++) {
    uint64_t cut = t + 1 == threads ? ulen : fastq_sync(u, ulen * (uint64_t)(t + 1) / (uint64_t)threads, ulen);
    if (cut < prev) cut = prev;
    fj[t].u = u; fj[t].a = prev; fj[t].b = cut; fj[t].end = ulen;
    prev = cut;
    pthread_create(&th[t], NULL, fastq_worker, &fj[t]);
  }
  for (int t = 0; t < threads; t++) {
    pthread_join(th[t], NULL);
    bad |= fj[t].err;
    const oracle_fastq_result* r = &fj[t].r;
    out->n_rows += r->n_rows; out->name_bytes += r->name_bytes; out->desc_bytes += r->desc_bytes; out->seq_bytes += r->seq_bytes;
    out->qual_bytes += r->qual_bytes; out->desc_null += r->desc_null; out->byte_sum += r->byte_sum;
  }
  const double t2 = now_s();
  out->n_blocks = nb; out->compressed_bytes = coff[nb]; out->inflated_bytes = ulen;
  out->seconds_inflate = t1 - t0; out->seconds_parse = t2 - t1; out->seconds_total = t2 - t0;
  out->threads = threads; out->used_libdeflate = ld_allocd != NULL;
  free(u); free(coff); free(uoff); free(th); free(ij); free(fj);
  if (bad) { snprintf(out->error, sizeof out->error, "FASTQ read error: invalid record"); return 1; }
  return 0;
}
