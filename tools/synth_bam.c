/* synth_bam.c -- deterministic synthetic coordinate-sorted BGZF-BAM (+BAI) generator.
 *
 * Bench / test infrastructure (not product code).  Produces the "config 2" workload of
 * 25 references (GRCh38 lengths), paired 150 bp reads, names
 * SIM:{lane}:{tile}:{x}:{y}, mapq in {0,20,40,60}, flags from {99,147,83,163} (+0x400 dup),
 * CIGAR 85% 150M / 10% soft-clipped / 5% one indel, ACGT with 0.1% N, Markov-run qualities
 * in [2,40], tlen ~ N(350,50), tags NM:C MD:Z AS:C RG:Z, 0.5% unplaced-unmapped pairs at the
 * end; BGZF members of <= 65280 payload bytes holding whole records (htslib bgzf_flush_try
 * behaviour), DEFLATE level 6 (libdeflate when present, else zlib), 28-byte EOF member.
 *
 * The file is a pure function of (n_blocks, seed): work is cut into tiles of TILE_BLOCKS
 * members, each generated from its own seeded PRNG, so the thread count never changes a byte.
 *
 * usage: synth_bam OUT.bam N_BLOCKS [SEED=42] [THREADS=nproc] [LEVEL=6] [part K W | place K W]
 *
 * The last form builds ONE file from W processes (bench.py --gpus N: every rank compresses its share): `part K W` generates
 * the K-th of W contiguous runs of tiles into OUT.bam.partK (+ .idx: sizes and the run's share of the BAI, offsets relative
 * to the run), `place K W` -- once every part exists -- copies part K to its offset in OUT.bam (tail first, truncating the part
 * as it goes: no second copy of the file in /dev/shm); K = 0 also writes the header member, the EOF member and the merged BAI
 * and prints the JSON line.  The result is byte for byte the file of the one-process form.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <zlib.h>

#define TILE_BLOCKS 64
#define BLOCK_PAYLOAD 65280
#define N_REF 25

static const char* REF_NAMES[N_REF] = {"chr1", "chr2", "chr3", "chr4", "chr5", "chr6", "chr7", "chr8", "chr9",
                                       "chr10", "chr11", "chr12", "chr13", "chr14", "chr15", "chr16", "chr17",
                                       "chr18", "chr19", "chr20", "chr21", "chr22", "chrX", "chrY", "chrM"};
static const int64_t REF_LENS[N_REF] = {248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
                                        145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
                                        101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
                                        50818468, 156040895, 57227415, 16569};

/* ---- libdeflate via dlopen (optional) -------------------------------------------------------- */
typedef void* (*ld_alloc_t)(int);
typedef size_t (*ld_comp_t)(void*, const void*, size_t, void*, size_t);
typedef void (*ld_free_t)(void*);
static ld_alloc_t ld_alloc;
static ld_comp_t ld_comp;
static ld_free_t ld_free;
static void load_libdeflate(void) {
  const char* names[] = {"libdeflate.so.0", "/opt/conda/lib/libdeflate.so.0", "libdeflate.so", NULL};
  for (int i = 0; names[i]; i++) {
    void* h = dlopen(names[i], RTLD_NOW);
    if (!h) continue;
    ld_alloc = (ld_alloc_t)dlsym(h, "libdeflate_alloc_compressor");
    ld_comp = (ld_comp_t)dlsym(h, "libdeflate_deflate_compress");
    ld_free = (ld_free_t)dlsym(h, "libdeflate_free_compressor");
    if (ld_alloc && ld_comp && ld_free) return;
    ld_alloc = NULL;
  }
}

/* ---- PRNG (splitmix64 / xoshiro256**) ------------------------------------------------------- */
typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix(uint64_t* x) {
  uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static void rng_seed(rng_t* r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r) {
  uint64_t* s = r->s;
  uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
  return res;
}
static inline uint32_t rng_u32(rng_t* r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }
static inline double rng_f(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* ---- tiles ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t ref;      /* -1 = unplaced unmapped tile */
  int64_t g0, g1;   /* 0-based position interval on the reference */
  uint32_t n_blocks;
  /* outputs */
  uint8_t* comp; size_t comp_len, comp_cap;
  uint32_t* blk_clen;   /* per block compressed size */
  uint32_t* blk_ulen;
  /* per record index info */
  int32_t* r_pos; int32_t* r_end; uint32_t* r_blk; uint16_t* r_off; uint8_t* r_unmapped;
  size_t n_rec, rec_cap;
} tile_t;

static int g_level = 6;
static uint64_t g_seed = 42;
static tile_t* g_tiles;
static size_t g_ntiles;
static size_t g_next_tile, g_end_tile;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static void put32(uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
static void put16(uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; }

static int reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

/* compress one BGZF member; returns total member size */
static size_t bgzf_member(void* ldc, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
  memcpy(dst, hdr, 16);
  size_t clen;
  if (ldc) {
    clen = ld_comp(ldc, src, n, dst + 18, cap - 26);
    if (!clen) { fprintf(stderr, "libdeflate: block did not fit\n"); exit(2); }
  } else {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n;
    zs.next_out = dst + 18; zs.avail_out = (uInt)(cap - 26);
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "zlib: block did not fit\n"); exit(2); }
    clen = zs.total_out;
    deflateEnd(&zs);
  }
  size_t total = 18 + clen + 8;
  put16(dst + 16, (uint32_t)(total - 1));
  put32(dst + 18 + clen, (uint32_t)crc32(crc32(0, NULL, 0), src, (uInt)n));
  put32(dst + 18 + clen + 4, (uint32_t)n);
  return total;
}

static void tile_push_block(tile_t* t, void* ldc, const uint8_t* buf, size_t n, uint32_t bi) {
  if (t->comp_cap - t->comp_len < 70000) {
    t->comp_cap = t->comp_cap * 2 + 140000;
    t->comp = (uint8_t*)realloc(t->comp, t->comp_cap);
  }
  size_t m = bgzf_member(ldc, buf, n, t->comp + t->comp_len, t->comp_cap - t->comp_len);
  t->blk_clen[bi] = (uint32_t)m;
  t->blk_ulen[bi] = (uint32_t)n;
  t->comp_len += m;
}

static const char BASES[4] = {1, 2, 4, 8}; /* A C G T nibbles */

static void gen_tile(tile_t* t, size_t tile_idx) {
  rng_t rng;
  rng_seed(&rng, g_seed * 0x100000001B3ull + tile_idx * 0x9E3779B97F4A7C15ull + 1);
  void* ldc = ld_alloc ? ld_alloc(g_level) : NULL;
  t->blk_clen = (uint32_t*)calloc(t->n_blocks, 4);
  t->blk_ulen = (uint32_t*)calloc(t->n_blocks, 4);
  t->rec_cap = (size_t)t->n_blocks * 200;
  t->r_pos = (int32_t*)malloc(t->rec_cap * 4);
  t->r_end = (int32_t*)malloc(t->rec_cap * 4);
  t->r_blk = (uint32_t*)malloc(t->rec_cap * 4);
  t->r_off = (uint16_t*)malloc(t->rec_cap * 2);
  t->r_unmapped = (uint8_t*)malloc(t->rec_cap);
  uint8_t* blk = (uint8_t*)malloc(BLOCK_PAYLOAD + 1024);
  size_t bl = 0;
  uint32_t bi = 0;
  const double expect_rec = (double)t->n_blocks * (BLOCK_PAYLOAD / 372.0);
  const double mean_gap = t->ref >= 0 ? (double)(t->g1 - t->g0) / expect_rec : 0;
  double gpos = (double)t->g0;
  uint8_t rec[1024];
  const int lane = 1 + (int)(tile_idx % 8);
  while (bi < t->n_blocks) {
    /* ---- one record ---- */
    int l_seq = 150;
    int32_t pos = -1, mpos = -1, tlen = 0;
    int32_t refid = t->ref;
    uint32_t cig[3];
    int ncig = 0;
    uint32_t flag;
    int mapq;
    int nm = 0;
    if (refid >= 0) {
      gpos += -log(1.0 - rng_f(&rng)) * mean_gap;
      if (gpos > (double)(t->g1 - 1)) gpos = (double)(t->g1 - 1);
      pos = (int32_t)gpos;
      uint32_t c = rng_u32(&rng, 100);
      if (c < 85) { cig[0] = (150u << 4) | 0; ncig = 1; }
      else if (c < 95) {
        uint32_t a = 1 + rng_u32(&rng, 40);
        cig[0] = (a << 4) | 4; cig[1] = ((150 - a) << 4) | 0; ncig = 2;
      } else {
        uint32_t a = 20 + rng_u32(&rng, 100), k = 1 + rng_u32(&rng, 3);
        if (rng_u32(&rng, 2)) { cig[0] = (a << 4); cig[1] = (k << 4) | 1; cig[2] = ((150 - a - k) << 4); }
        else { cig[0] = (a << 4); cig[1] = (k << 4) | 2; cig[2] = ((150 - a) << 4); }
        ncig = 3;
        nm = (int)k;
      }
      static const uint32_t FL[4] = {99, 147, 83, 163};
      flag = FL[rng_u32(&rng, 4)];
      if (rng_u32(&rng, 100) < 5) flag |= 0x400;
      uint32_t mq = rng_u32(&rng, 100);
      mapq = mq < 5 ? 0 : mq < 10 ? 20 : mq < 20 ? 40 : 60;
      /* tlen ~ N(350, 50) via sum of uniforms */
      double g = 0;
      for (int k = 0; k < 6; k++) g += rng_f(&rng);
      int isz = (int)(350.0 + (g - 3.0) * 50.0 * 1.41421356);
      if (isz < 151) isz = 151;
      if (flag & 0x10) { mpos = pos - (isz - 150); if (mpos < 0) mpos = 0; tlen = -isz; }
      else { mpos = pos + (isz - 150); tlen = isz; }
      nm += (int)rng_u32(&rng, 3);
    } else {
      flag = rng_u32(&rng, 2) ? 77 : 141;
      mapq = 0;
    }
    char name[64];
    int ln = snprintf(name, sizeof name, "SIM:%d:%u:%u:%u", lane, (unsigned)(1000 + tile_idx % 9000), rng_u32(&rng, 30000), rng_u32(&rng, 30000)) + 1;
    uint8_t* p = rec + 4;
    int64_t end = pos;
    for (int k = 0; k < ncig; k++) { uint32_t op = cig[k] & 15; if (op == 0 || op == 2) end += cig[k] >> 4; }
    if (refid < 0) end = 0;
    put32(p, (uint32_t)refid); put32(p + 4, (uint32_t)pos);
    p[8] = (uint8_t)ln; p[9] = (uint8_t)mapq;
    put16(p + 10, refid >= 0 ? (uint32_t)reg2bin(pos, end > pos ? end : pos + 1) : 4680);
    put16(p + 12, (uint32_t)ncig); put16(p + 14, flag);
    put32(p + 16, (uint32_t)l_seq); put32(p + 20, (uint32_t)refid); put32(p + 24, (uint32_t)mpos); put32(p + 28, (uint32_t)tlen);
    p += 32;
    memcpy(p, name, (size_t)ln); p += ln;
    for (int k = 0; k < ncig; k++) { put32(p, cig[k]); p += 4; }
    for (int k = 0; k < l_seq; k += 2) {
      uint64_t x = rng_next(&rng);
      uint8_t a = (x & 1023) == 0 ? 15 : BASES[(x >> 10) & 3];
      uint8_t b = ((x >> 20) & 1023) == 0 ? 15 : BASES[(x >> 30) & 3];
      *p++ = (uint8_t)((a << 4) | b);
    }
    { /* Markov-run qualities in [2,40] */
      int q = 30 + (int)rng_u32(&rng, 11);
      for (int k = 0; k < l_seq;) {
        uint64_t x = rng_next(&rng);
        int run = 1 + (int)(x & 7);
        int step = (int)((x >> 3) % 7) - 3;
        if (((x >> 8) & 31) == 0) q = 2 + (int)((x >> 16) % 20);
        else q += step;
        if (q < 2) q = 2;
        if (q > 40) q = 40;
        for (int j = 0; j < run && k < l_seq; j++, k++) *p++ = (uint8_t)q;
      }
    }
    /* tags */
    *p++ = 'N'; *p++ = 'M'; *p++ = 'C'; *p++ = (uint8_t)nm;
    *p++ = 'M'; *p++ = 'D'; *p++ = 'Z';
    if (nm == 0 || refid < 0) { p += sprintf((char*)p, "150") + 1; }
    else {
      int a = 1 + (int)rng_u32(&rng, 140);
      p += sprintf((char*)p, "%d%c%d", a, "ACGT"[rng_u32(&rng, 4)], 149 - a) + 1;
    }
    *p++ = 'A'; *p++ = 'S'; *p++ = 'C'; *p++ = (uint8_t)(refid >= 0 ? 150 - 5 * nm - (int)rng_u32(&rng, 5) : 0);
    *p++ = 'R'; *p++ = 'G'; *p++ = 'Z';
    p += sprintf((char*)p, "grp%d", lane) + 1;
    size_t rl = (size_t)(p - rec);
    put32(rec, (uint32_t)(rl - 4));
    /* ---- block packing: whole records per member ---- */
    if (bl + rl > BLOCK_PAYLOAD) {
      tile_push_block(t, ldc, blk, bl, bi);
      bi++;
      bl = 0;
      if (bi >= t->n_blocks) break; /* record discarded: the tile holds exactly n_blocks members */
    }
    if (t->n_rec == t->rec_cap) {
      t->rec_cap *= 2;
      t->r_pos = (int32_t*)realloc(t->r_pos, t->rec_cap * 4);
      t->r_end = (int32_t*)realloc(t->r_end, t->rec_cap * 4);
      t->r_blk = (uint32_t*)realloc(t->r_blk, t->rec_cap * 4);
      t->r_off = (uint16_t*)realloc(t->r_off, t->rec_cap * 2);
      t->r_unmapped = (uint8_t*)realloc(t->r_unmapped, t->rec_cap);
    }
    t->r_pos[t->n_rec] = pos;
    t->r_end[t->n_rec] = (int32_t)end;
    t->r_blk[t->n_rec] = bi;
    t->r_off[t->n_rec] = (uint16_t)bl;
    t->r_unmapped[t->n_rec] = (flag & 4) ? 1 : 0;
    t->n_rec++;
    memcpy(blk + bl, rec, rl);
    bl += rl;
  }
  free(blk);
  if (ldc) ld_free(ldc);
}

static void* worker(void* arg) {
  (void)arg;
  for (;;) {
    pthread_mutex_lock(&g_mu);
    size_t i = g_next_tile++;
    pthread_mutex_unlock(&g_mu);
    if (i >= g_end_tile) break;
    gen_tile(&g_tiles[i], i);
  }
  return NULL;
}

/* ---- BAI -------------------------------------------------------------------------------------- */
typedef struct { uint64_t beg, end; } chunk_t;
typedef struct { uint32_t bin; chunk_t* c; size_t n, cap; } binrec_t;
typedef struct {
  binrec_t* bins; size_t nbins, capbins;
  int64_t* bin_lookup;  /* bin id -> index+1 (37450 entries) */
  uint64_t* lin; size_t nlin;
  uint64_t ref_beg, ref_end, n_mapped, n_unmapped;
} refidx_t;

static void bin_add(refidx_t* r, uint32_t bin, uint64_t beg, uint64_t end) {
  if (!r->bin_lookup) r->bin_lookup = (int64_t*)calloc(37451, sizeof(int64_t));
  int64_t k = r->bin_lookup[bin];
  if (!k) {
    if (r->nbins == r->capbins) { r->capbins = r->capbins * 2 + 64; r->bins = (binrec_t*)realloc(r->bins, r->capbins * sizeof(binrec_t)); }
    r->bins[r->nbins].bin = bin; r->bins[r->nbins].c = NULL; r->bins[r->nbins].n = r->bins[r->nbins].cap = 0;
    r->nbins++;
    k = r->bin_lookup[bin] = (int64_t)r->nbins;
  }
  binrec_t* b = &r->bins[k - 1];
  if (b->n && b->c[b->n - 1].end == beg) { b->c[b->n - 1].end = end; return; }
  if (b->n == b->cap) { b->cap = b->cap * 2 + 4; b->c = (chunk_t*)realloc(b->c, b->cap * sizeof(chunk_t)); }
  b->c[b->n].beg = beg; b->c[b->n].end = end; b->n++;
}

/* adds the records of tiles [t0, t1) to the index; tile_coff[i - t0] = compressed offset of tile i (absolute, or relative to a part) */
static void index_tiles(refidx_t* ri, size_t t0, size_t t1, const uint64_t* tile_coff, uint64_t* n_no_coor) {
  for (size_t i = t0; i < t1; i++) {
    tile_t* t = &g_tiles[i];
    uint64_t* bc = (uint64_t*)malloc((t->n_blocks + 1) * 8);
    bc[0] = tile_coff[i - t0];
    for (uint32_t b = 0; b < t->n_blocks; b++) bc[b + 1] = bc[b] + t->blk_clen[b];
    for (size_t k = 0; k < t->n_rec; k++) {
      uint64_t v0 = (bc[t->r_blk[k]] << 16) | t->r_off[k];
      uint64_t v1;
      if (k + 1 < t->n_rec && t->r_blk[k + 1] == t->r_blk[k]) v1 = (bc[t->r_blk[k]] << 16) | t->r_off[k + 1];
      else v1 = bc[t->r_blk[k] + 1] << 16; /* end of member = start of the next */
      if (t->ref < 0) { (*n_no_coor)++; continue; }
      refidx_t* r = &ri[t->ref];
      int64_t beg = t->r_pos[k], end = t->r_end[k] > t->r_pos[k] ? t->r_end[k] : t->r_pos[k] + 1;
      bin_add(r, (uint32_t)reg2bin(beg, end), v0, v1);
      size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
      if (w1 + 1 > r->nlin) {
        r->lin = (uint64_t*)realloc(r->lin, (w1 + 1) * 8);
        for (size_t w = r->nlin; w <= w1; w++) r->lin[w] = 0;
        r->nlin = w1 + 1;
      }
      for (size_t w = w0; w <= w1; w++) if (!r->lin[w]) r->lin[w] = v0;
      if (!r->ref_beg) r->ref_beg = v0;
      r->ref_end = v1;
      if (t->r_unmapped[k]) r->n_unmapped++; else r->n_mapped++;
    }
    free(bc);
  }
}

static int write_bai(const char* out_path, refidx_t* ri, uint64_t n_no_coor) {
  char bai_path[4096];
  snprintf(bai_path, sizeof bai_path, "%s.bai", out_path);
  FILE* f = fopen(bai_path, "wb");
  if (!f) { perror(bai_path); return 1; }
  uint8_t w8[16];
  fwrite("BAI\1", 1, 4, f);
  put32(w8, N_REF); fwrite(w8, 1, 4, f);
  for (int r = 0; r < N_REF; r++) {
    refidx_t* x = &ri[r];
    int has_meta = x->n_mapped + x->n_unmapped > 0;
    put32(w8, (uint32_t)(x->nbins + (has_meta ? 1 : 0))); fwrite(w8, 1, 4, f);
    for (size_t b = 0; b < x->nbins; b++) {
      put32(w8, x->bins[b].bin); put32(w8 + 4, (uint32_t)x->bins[b].n); fwrite(w8, 1, 8, f);
      fwrite(x->bins[b].c, sizeof(chunk_t), x->bins[b].n, f);
    }
    if (has_meta) {
      put32(w8, 37450); put32(w8 + 4, 2); fwrite(w8, 1, 8, f);
      uint64_t m[4] = {x->ref_beg, x->ref_end, x->n_mapped, x->n_unmapped};
      fwrite(m, 8, 4, f);
    }
    /* backfill empty linear windows with the previous entry (htslib) */
    for (size_t w = 1; w < x->nlin; w++) if (!x->lin[w]) x->lin[w] = x->lin[w - 1];
    put32(w8, (uint32_t)x->nlin); fwrite(w8, 1, 4, f);
    fwrite(x->lin, 8, x->nlin, f);
  }
  fwrite(&n_no_coor, 8, 1, f);
  fclose(f);
  return 0;
}

/* a part's sidecar: sizes, then its share of the index with offsets relative to the part's first byte */
typedef struct { uint64_t comp_len, total_u, n_rec, n_no_coor, n_blocks; } part_hdr_t;
static int idx_save(const char* path, const part_hdr_t* h, refidx_t* ri) {
  FILE* f = fopen(path, "wb");
  if (!f) { perror(path); return 1; }
  fwrite(h, sizeof *h, 1, f);
  for (int r = 0; r < N_REF; r++) {
    refidx_t* x = &ri[r];
    uint64_t m[6] = {x->nbins, x->nlin, x->ref_beg, x->ref_end, x->n_mapped, x->n_unmapped};
    fwrite(m, 8, 6, f);
    for (size_t b = 0; b < x->nbins; b++) {
      uint64_t bh[2] = {x->bins[b].bin, x->bins[b].n};
      fwrite(bh, 8, 2, f);
      fwrite(x->bins[b].c, sizeof(chunk_t), x->bins[b].n, f);
    }
    fwrite(x->lin, 8, x->nlin, f);
  }
  return fclose(f) ? 1 : 0;
}
/* merges a part's index (read from `path`) behind what `ri` holds, its offsets shifted by the part's place in the file */
static int idx_merge(const char* path, uint64_t base, refidx_t* ri) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); return 1; }
  part_hdr_t h;
  if (fread(&h, sizeof h, 1, f) != 1) return 1;
  const uint64_t sh = (base - 1) << 16;   /* (a part counts its offsets from 1) */
  for (int r = 0; r < N_REF; r++) {
    refidx_t* x = &ri[r];
    uint64_t m[6];
    if (fread(m, 8, 6, f) != 6) return 1;
    for (uint64_t b = 0; b < m[0]; b++) {
      uint64_t bh[2];
      if (fread(bh, 8, 2, f) != 2) return 1;
      chunk_t* c = (chunk_t*)malloc((size_t)(bh[1] ? bh[1] : 1) * sizeof(chunk_t));
      if (fread(c, sizeof(chunk_t), (size_t)bh[1], f) != (size_t)bh[1]) return 1;
      for (uint64_t k = 0; k < bh[1]; k++) bin_add(x, (uint32_t)bh[0], c[k].beg + sh, c[k].end + sh);
      free(c);
    }
    uint64_t* lin = (uint64_t*)malloc((size_t)(m[1] ? m[1] : 1) * 8);
    if (fread(lin, 8, (size_t)m[1], f) != (size_t)m[1]) return 1;
    if (m[1] > x->nlin) {
      x->lin = (uint64_t*)realloc(x->lin, (size_t)m[1] * 8);
      for (size_t w = x->nlin; w < m[1]; w++) x->lin[w] = 0;
      x->nlin = (size_t)m[1];
    }
    for (size_t w = 0; w < m[1]; w++) if (!x->lin[w] && lin[w]) x->lin[w] = lin[w] + sh;
    free(lin);
    if (!x->ref_beg && m[2]) x->ref_beg = m[2] + sh;
    if (m[3]) x->ref_end = m[3] + sh;
    x->n_mapped += m[4]; x->n_unmapped += m[5];
  }
  fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s OUT.bam N_BLOCKS [SEED] [THREADS] [LEVEL] [part K W | place K W]\n", argv[0]); return 1; }
  const char* out_path = argv[1];
  uint64_t n_blocks = strtoull(argv[2], NULL, 10);
  if (argc > 3) g_seed = strtoull(argv[3], NULL, 10);
  int threads = argc > 4 ? atoi(argv[4]) : (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (argc > 5) g_level = atoi(argv[5]);
  int mode = 0, part_k = 0, part_w = 1;  /* 0: the whole file; 1: part K of W; 2: place part K of W */
  if (argc > 8) {
    mode = !strcmp(argv[6], "part") ? 1 : !strcmp(argv[6], "place") ? 2 : -1;
    part_k = atoi(argv[7]); part_w = atoi(argv[8]);
    if (mode < 0 || part_w < 1 || part_k < 0 || part_k >= part_w) { fprintf(stderr, "bad part arguments\n"); return 1; }
  }
  if (threads < 1) threads = 1;
  if (!getenv("SYNTH_BAM_ZLIB")) load_libdeflate();

  /* header member (its own BGZF block, like samtools) counts as block 0 */
  if (n_blocks < 3) n_blocks = 3;
  uint64_t data_blocks = n_blocks - 2; /* minus header member and EOF member */
  size_t ntiles = (size_t)((data_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS);
  g_tiles = (tile_t*)calloc(ntiles, sizeof(tile_t));
  g_ntiles = ntiles;
  size_t unm_tiles = ntiles >= 2 ? (ntiles / 200 ? ntiles / 200 : 1) : 0;
  size_t map_tiles = ntiles - unm_tiles;
  /* tiles per reference proportional to length (>= 1 while tiles remain) */
  size_t per_ref[N_REF];
  {
    double total = 0;
    for (int r = 0; r < N_REF; r++) total += (double)REF_LENS[r];
    size_t used = 0;
    for (int r = 0; r < N_REF; r++) {
      per_ref[r] = (size_t)floor((double)map_tiles * (double)REF_LENS[r] / total);
      used += per_ref[r];
    }
    for (int r = 0; used < map_tiles; r = (r + 1) % N_REF) { per_ref[r]++; used++; }
  }
  size_t ti = 0;
  uint64_t left = data_blocks;
  for (int r = 0; r < N_REF; r++)
    for (size_t k = 0; k < per_ref[r]; k++, ti++) {
      tile_t* t = &g_tiles[ti];
      t->ref = r;
      t->g0 = (int64_t)((double)REF_LENS[r] * (double)k / (double)per_ref[r]);
      t->g1 = (int64_t)((double)REF_LENS[r] * (double)(k + 1) / (double)per_ref[r]);
      if (t->g1 <= t->g0) t->g1 = t->g0 + 1;
    }
  for (; ti < ntiles; ti++) g_tiles[ti].ref = -1;
  for (size_t i = 0; i < ntiles; i++) {
    uint64_t nb = left < TILE_BLOCKS ? left : TILE_BLOCKS;
    g_tiles[i].n_blocks = (uint32_t)nb;
    left -= nb;
  }
  /* the tiles this process generates: all of them, or the K-th of W contiguous runs */
  const size_t t0 = mode ? ntiles * (size_t)part_k / (size_t)part_w : 0;
  const size_t t1 = mode ? ntiles * (size_t)(part_k + 1) / (size_t)part_w : ntiles;
  char part_path[4096], idx_path[4096];
  snprintf(part_path, sizeof part_path, "%s.part%d", out_path, part_k);
  snprintf(idx_path, sizeof idx_path, "%s.part%d.idx", out_path, part_k);

  if (mode != 2) {
    g_next_tile = t0; g_end_tile = t1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, NULL);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
  }

  if (mode == 1) {
    /* the run's members back to back, and its sidecar */
    FILE* f = fopen(part_path, "wb");
    if (!f) { perror(part_path); return 1; }
    part_hdr_t h = {0, 0, 0, 0, 0};
    uint64_t* tile_coff = (uint64_t*)malloc((t1 - t0 + 1) * 8);
    for (size_t i = t0; i < t1; i++) {
      tile_coff[i - t0] = h.comp_len + 1;   /* (+ 1: a virtual offset of 0 means "none" in the index structures) */
      if (fwrite(g_tiles[i].comp, 1, g_tiles[i].comp_len, f) != g_tiles[i].comp_len) { perror(part_path); return 1; }
      h.comp_len += g_tiles[i].comp_len;
      h.n_rec += g_tiles[i].n_rec;
      for (uint32_t b = 0; b < g_tiles[i].n_blocks; b++) h.total_u += g_tiles[i].blk_ulen[b];
      h.n_blocks += g_tiles[i].n_blocks;
    }
    if (fclose(f)) { perror(part_path); return 1; }
    refidx_t* ri = (refidx_t*)calloc(N_REF, sizeof(refidx_t));
    index_tiles(ri, t0, t1, tile_coff, &h.n_no_coor);
    if (idx_save(idx_path, &h, ri)) return 1;
    printf("{\"part\": %d, \"of\": %d, \"compressed_bytes\": %llu, \"n_blocks\": %llu}\n", part_k, part_w,
           (unsigned long long)h.comp_len, (unsigned long long)h.n_blocks);
    return 0;
  }

  /* ---- header member ---- */
  char text[8192];
  int tl = snprintf(text, sizeof text, "@HD\tVN:1.6\tSO:coordinate\n");
  for (int r = 0; r < N_REF; r++) tl += snprintf(text + tl, sizeof text - (size_t)tl, "@SQ\tSN:%s\tLN:%lld\n", REF_NAMES[r], (long long)REF_LENS[r]);
  for (int l = 1; l <= 8; l++) tl += snprintf(text + tl, sizeof text - (size_t)tl, "@RG\tID:grp%d\tSM:synth\tPL:ILLUMINA\n", l);
  tl += snprintf(text + tl, sizeof text - (size_t)tl, "@PG\tID:synth_bam\tPN:synth_bam\tVN:1\n");
  uint8_t* hb = (uint8_t*)malloc(65536);
  size_t hl = 0;
  memcpy(hb, "BAM\1", 4); put32(hb + 4, (uint32_t)tl); memcpy(hb + 8, text, (size_t)tl); hl = 8 + (size_t)tl;
  put32(hb + hl, N_REF); hl += 4;
  for (int r = 0; r < N_REF; r++) {
    size_t ln = strlen(REF_NAMES[r]) + 1;
    put32(hb + hl, (uint32_t)ln); memcpy(hb + hl + 4, REF_NAMES[r], ln); put32(hb + hl + 4 + ln, (uint32_t)REF_LENS[r]);
    hl += 8 + ln;
  }
  uint8_t* hmem = (uint8_t*)malloc(70000);
  void* ldc = ld_alloc ? ld_alloc(g_level) : NULL;
  size_t hmem_len = bgzf_member(ldc, hb, hl, hmem, 70000);
  if (ldc) ld_free(ldc);
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  if (mode == 2) {
    /* every part's size -> this part's place; copy it there tail first, giving the part's pages back as they are copied */
    part_hdr_t* ph = (part_hdr_t*)calloc((size_t)part_w, sizeof(part_hdr_t));
    uint64_t* base = (uint64_t*)calloc((size_t)part_w + 1, 8);
    base[0] = hmem_len;
    for (int k = 0; k < part_w; k++) {
      char ip[4096];
      snprintf(ip, sizeof ip, "%s.part%d.idx", out_path, k);
      FILE* f = fopen(ip, "rb");
      if (!f || fread(&ph[k], sizeof(part_hdr_t), 1, f) != 1) { perror(ip); return 1; }
      fclose(f);
      base[k + 1] = base[k] + ph[k].comp_len;
    }
    int out_fd = open(out_path, O_WRONLY | O_CREAT, 0644);
    int in_fd = open(part_path, O_RDWR);
    if (out_fd < 0 || in_fd < 0) { perror(out_fd < 0 ? out_path : part_path); return 1; }
    const size_t CH = 64u << 20;
    uint8_t* buf = (uint8_t*)malloc(CH);
    uint64_t left_b = ph[part_k].comp_len;
    while (left_b) {
      const size_t n = left_b < CH ? (size_t)left_b : CH;
      const uint64_t off = left_b - n;
      if (pread(in_fd, buf, n, (off_t)off) != (ssize_t)n || pwrite(out_fd, buf, n, (off_t)(base[part_k] + off)) != (ssize_t)n) { perror("copy"); return 1; }
      if (ftruncate(in_fd, (off_t)off)) { perror("ftruncate"); return 1; }
      left_b = off;
    }
    close(in_fd);
    unlink(part_path);
    if (part_k != 0) { close(out_fd); printf("{\"placed\": %d}\n", part_k); return 0; }
    if (pwrite(out_fd, hmem, hmem_len, 0) != (ssize_t)hmem_len || pwrite(out_fd, eof, 28, (off_t)base[part_w]) != 28) { perror(out_path); return 1; }
    close(out_fd);
    refidx_t* ri = (refidx_t*)calloc(N_REF, sizeof(refidx_t));
    uint64_t total_u = hl, total_rec = 0, total_blocks = 2, n_no_coor = 0;
    for (int k = 0; k < part_w; k++) {
      char ip[4096];
      snprintf(ip, sizeof ip, "%s.part%d.idx", out_path, k);
      if (idx_merge(ip, base[k], ri)) { fprintf(stderr, "cannot merge %s\n", ip); return 1; }   /* (the caller removes the sidecars: the other parts read them too) */
      total_u += ph[k].total_u; total_rec += ph[k].n_rec; total_blocks += ph[k].n_blocks; n_no_coor += ph[k].n_no_coor;
    }
    if (write_bai(out_path, ri, n_no_coor)) return 1;
    printf("{\"path\": \"%s\", \"n_blocks\": %llu, \"n_records\": %llu, \"compressed_bytes\": %llu, \"inflated_bytes\": %llu, "
           "\"n_no_coor\": %llu, \"seed\": %llu, \"level\": %d, \"deflate\": \"%s\", \"threads\": %d}\n",
           out_path, (unsigned long long)total_blocks, (unsigned long long)total_rec, (unsigned long long)(base[part_w] + 28),
           (unsigned long long)total_u, (unsigned long long)n_no_coor, (unsigned long long)g_seed, g_level,
           ld_alloc ? "libdeflate" : "zlib", threads);
    return 0;
  }

  /* ---- write BAM ---- */
  FILE* f = fopen(out_path, "wb");
  if (!f) { perror(out_path); return 1; }
  fwrite(hmem, 1, hmem_len, f);
  uint64_t coff = hmem_len, total_u = hl, total_rec = 0, total_blocks = 1;
  uint64_t* tile_coff = (uint64_t*)malloc(ntiles * 8);
  for (size_t i = 0; i < ntiles; i++) {
    tile_coff[i] = coff;
    fwrite(g_tiles[i].comp, 1, g_tiles[i].comp_len, f);
    coff += g_tiles[i].comp_len;
    total_rec += g_tiles[i].n_rec;
    for (uint32_t b = 0; b < g_tiles[i].n_blocks; b++) total_u += g_tiles[i].blk_ulen[b];
    total_blocks += g_tiles[i].n_blocks;
  }
  fwrite(eof, 1, 28, f);
  total_blocks++;
  uint64_t file_len = coff + 28;
  fclose(f);

  /* ---- BAI ---- */
  refidx_t* ri = (refidx_t*)calloc(N_REF, sizeof(refidx_t));
  uint64_t n_no_coor = 0;
  index_tiles(ri, 0, ntiles, tile_coff, &n_no_coor);
  if (write_bai(out_path, ri, n_no_coor)) return 1;
  printf("{\"path\": \"%s\", \"n_blocks\": %llu, \"n_records\": %llu, \"compressed_bytes\": %llu, \"inflated_bytes\": %llu, "
         "\"n_no_coor\": %llu, \"seed\": %llu, \"level\": %d, \"deflate\": \"%s\", \"threads\": %d}\n",
         out_path, (unsigned long long)total_blocks, (unsigned long long)total_rec, (unsigned long long)file_len,
         (unsigned long long)total_u, (unsigned long long)n_no_coor, (unsigned long long)g_seed, g_level,
         ld_alloc ? "libdeflate" : "zlib", threads);
  return 0;
}
