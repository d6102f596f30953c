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
    if (ld_allocd && ld_decomp && ld_freed) return;
    ld_allocd = NULL;
  }
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
    if ((uint32_t)crc32(crc32(0, NULL, 0), dst, isize) != crc) { j->err = 2; break; }
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
