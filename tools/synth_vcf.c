/* synth_vcf.c -- synthetic BGZF-compressed VCF + tabix index (bench / test tooling, not part of the product).
 *
 *   synth_vcf sites   OUT.vcf.gz N_LINES [SEED] [THREADS] [LEVEL]        config 3: sites-only, gnomAD-like INFO
 *   synth_vcf samples OUT.vcf.gz N_LINES N_SAMPLES [SEED] [THREADS] [LEVEL]   config 4: FORMAT GT:GQ:DP
 *
 * Lines are coordinate-sorted over chr1..chr22 (GRCh38 lengths, line counts proportional to length), generated
 * in parallel tiles, each tile cut into 65280-byte BGZF members like bgzip does (a line may span members).
 * OUT.vcf.gz.tbi is written with the tabix layout (bins of the UCSC scheme, 16 KiB linear index, pseudo-bin
 * 37450), BGZF-compressed.  Deterministic for a given (mode, sizes, seed): tiles are seeded by index. */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#define N_REF 22
#define BLOCK_PAYLOAD 65280
#define TILE_LINES_SITES 20000
#define TILE_LINES_SAMPLES 64

static const int64_t REF_LENS[N_REF] = {248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
                                        145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
                                        101991189, 90338345,  83257441,  80373285,  58617616,  64444167,  46709983, 50818468};

typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix(uint64_t* x) {
  uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static void rng_seed(rng_t* r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r) {
  uint64_t* s = r->s;
  const uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
  return result;
}
static inline uint32_t rng_u32(rng_t* r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }
static inline double rng_f(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

static int g_mode = 0; /* 0 sites, 1 samples */
static int g_level = 6;
static uint64_t g_seed = 42;
static uint32_t g_nsamples = 0;

typedef struct {
  int32_t ref;
  int64_t g0, g1;      /* 1-based position interval [g0, g1) */
  uint64_t n_lines;
  uint8_t* text; size_t text_len, text_cap;
  uint8_t* comp; size_t comp_len;
  uint32_t n_blocks; uint32_t* blk_clen;
  uint64_t* l_off; int32_t* l_pos; int32_t* l_rlen;  /* per line: text offset, POS, len(REF) */
} tile_t;
static tile_t* g_tiles;
static size_t g_ntiles, g_next_tile;
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
static size_t bgzf_member(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
  memcpy(dst, hdr, 16);
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  deflateInit2(&zs, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n;
  zs.next_out = dst + 18; zs.avail_out = (uInt)(cap - 26);
  if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "zlib: block did not fit\n"); exit(2); }
  size_t clen = zs.total_out;
  deflateEnd(&zs);
  size_t total = 18 + clen + 8;
  put16(dst + 16, (uint32_t)(total - 1));
  put32(dst + 18 + clen, (uint32_t)crc32(crc32(0, NULL, 0), src, (uInt)n));
  put32(dst + 18 + clen + 4, (uint32_t)n);
  return total;
}
/* cut `text` into members of BLOCK_PAYLOAD bytes */
static void compress_text(const uint8_t* text, size_t len, uint8_t** comp, size_t* comp_len, uint32_t* n_blocks, uint32_t** blk_clen) {
  uint32_t nb = (uint32_t)((len + BLOCK_PAYLOAD - 1) / BLOCK_PAYLOAD);
  uint8_t* out = (uint8_t*)malloc((size_t)nb * 66000 + 64);
  uint32_t* cl = (uint32_t*)malloc(sizeof(uint32_t) * (nb + 1));
  size_t o = 0;
  for (uint32_t b = 0; b < nb; b++) {
    size_t a = (size_t)b * BLOCK_PAYLOAD, n = len - a < BLOCK_PAYLOAD ? len - a : BLOCK_PAYLOAD;
    size_t m = bgzf_member(text + a, n, out + o, 66000);
    cl[b] = (uint32_t)m;
    o += m;
  }
  *comp = out; *comp_len = o; *n_blocks = nb; *blk_clen = cl;
}

static void tile_reserve(tile_t* t, size_t more) {
  if (t->text_cap - t->text_len < more) {
    t->text_cap = (t->text_cap + more) * 2;
    t->text = (uint8_t*)realloc(t->text, t->text_cap);
  }
}
static const char ACGT[4] = {'A', 'C', 'G', 'T'};

static void gen_tile(tile_t* t, size_t tile_idx) {
  rng_t rng;
  rng_seed(&rng, g_seed * 1000003ull + tile_idx);
  const uint64_t n = t->n_lines;
  t->l_off = (uint64_t*)malloc(8 * (n + 1));
  t->l_pos = (int32_t*)malloc(4 * (n + 1));
  t->l_rlen = (int32_t*)malloc(4 * (n + 1));
  /* sorted positions: even spacing with jitter */
  const double span = (double)(t->g1 - t->g0) / (double)(n ? n : 1);
  char* p;
  for (uint64_t k = 0; k < n; k++) {
    int64_t pos = t->g0 + (int64_t)(span * (double)k) + (int64_t)(rng_f(&rng) * (span > 1.0 ? span - 1.0 : 0.0));
    if (pos < 1) pos = 1;
    tile_reserve(t, 4096 + (size_t)g_nsamples * 16);
    t->l_off[k] = t->text_len;
    t->l_pos[k] = (int32_t)pos;
    p = (char*)t->text + t->text_len;
    /* REF / ALT */
    char ref[16], alt[40];
    int rlen = 1;
    const uint32_t kind = rng_u32(&rng, 100);
    ref[0] = ACGT[rng_u32(&rng, 4)];
    if (kind < 82) {  /* SNV */
      ref[1] = 0;
      char a = ACGT[rng_u32(&rng, 4)];
      while (a == ref[0]) a = ACGT[rng_u32(&rng, 4)];
      alt[0] = a; alt[1] = 0;
      if (kind < 8) {  /* multi-allelic */
        char b = ACGT[rng_u32(&rng, 4)];
        while (b == ref[0] || b == a) b = ACGT[rng_u32(&rng, 4)];
        alt[1] = ','; alt[2] = b; alt[3] = 0;
      }
    } else if (kind < 91) {  /* deletion */
      rlen = 2 + (int)rng_u32(&rng, 6);
      for (int j = 1; j < rlen; j++) ref[j] = ACGT[rng_u32(&rng, 4)];
      ref[rlen] = 0;
      alt[0] = ref[0]; alt[1] = 0;
    } else if (kind < 99) {  /* insertion */
      ref[1] = 0;
      int il = 2 + (int)rng_u32(&rng, 6);
      alt[0] = ref[0];
      for (int j = 1; j < il; j++) alt[j] = ACGT[rng_u32(&rng, 4)];
      alt[il] = 0;
    } else {  /* missing ALT */
      ref[1] = 0;
      alt[0] = '.'; alt[1] = 0;
    }
    t->l_rlen[k] = rlen;
    int n_alt = 1;
    for (const char* q = alt; *q; q++) n_alt += *q == ',';
    p += sprintf(p, "chr%d\t%lld\t", t->ref + 1, (long long)pos);
    if (rng_u32(&rng, 100) < 70) p += sprintf(p, "rs%u", 1000000u + rng_u32(&rng, 900000000u));
    else *p++ = '.';
    p += sprintf(p, "\t%s\t%s\t", ref, alt);
    {
      const uint32_t q = rng_u32(&rng, 100);
      if (q < 5) *p++ = '.';
      else if (q < 40) p += sprintf(p, "%u", 10 + rng_u32(&rng, 5000));
      else p += sprintf(p, "%u.%02u", 10 + rng_u32(&rng, 5000), rng_u32(&rng, 100));
    }
    {
      const uint32_t q = rng_u32(&rng, 100);
      if (q < 80) p += sprintf(p, "\tPASS\t");
      else if (q < 90) p += sprintf(p, "\tLowQual\t");
      else if (q < 97) p += sprintf(p, "\tLowQual;AC0\t");
      else p += sprintf(p, "\t.\t");
    }
    /* INFO */
    const uint32_t an = 2 * (g_mode ? (g_nsamples ? g_nsamples : 1) : 70000 + rng_u32(&rng, 6000));
    p += sprintf(p, "AC=");
    uint32_t ac[3];
    for (int a = 0; a < n_alt; a++) {
      ac[a] = rng_u32(&rng, 100) < 60 ? 1 + rng_u32(&rng, 20) : 1 + rng_u32(&rng, an / 2);
      p += sprintf(p, "%s%u", a ? "," : "", ac[a]);
    }
    p += sprintf(p, ";AN=%u;AF=", an);
    for (int a = 0; a < n_alt; a++) {
      const double af = (double)ac[a] / (double)an;
      if (rng_u32(&rng, 200) == 0) p += sprintf(p, "%s.", a ? "," : "");
      else if (af < 1e-3) p += sprintf(p, "%s%.5e", a ? "," : "", af);
      else p += sprintf(p, "%s%.6g", a ? "," : "", af);
    }
    p += sprintf(p, ";DP=%u", 1000 + rng_u32(&rng, 4000000));
    if (!g_mode) {
      if (rng_u32(&rng, 100) < 30) p += sprintf(p, ";DB");
      if (rng_u32(&rng, 100) < 6) p += sprintf(p, ";SEGDUP");
      if (rng_u32(&rng, 100) < 9) p += sprintf(p, ";LCR");
      p += sprintf(p, ";VT=%s", kind < 82 ? "SNP" : (kind < 99 ? "INDEL" : "NOALT"));
      if (rng_u32(&rng, 100) < 50) {
        static const char* SRC[5] = {"dbSNP", "ExAC", "TOPMed", "1000G", "UK10K"};
        const uint32_t ns = 1 + rng_u32(&rng, 3);
        p += sprintf(p, ";RSRC=");
        for (uint32_t j = 0; j < ns; j++) p += sprintf(p, "%s%s", j ? "," : "", rng_u32(&rng, 40) == 0 ? "." : SRC[rng_u32(&rng, 5)]);
      }
      if (rng_u32(&rng, 100) < 20) p += sprintf(p, ";CULPRIT=%s", rng_u32(&rng, 2) ? "MQ" : "FS");
      if (rng_u32(&rng, 100) < 3) p += sprintf(p, ";VQSLOD=%s%u.%03u", rng_u32(&rng, 2) ? "-" : "", rng_u32(&rng, 30), rng_u32(&rng, 1000));
    }
    if (g_mode) {
      p += sprintf(p, "\tGT:GQ:DP");
      for (uint32_t s = 0; s < g_nsamples; s++) {
        const uint32_t m = rng_u32(&rng, 1000);
        if (m < 4) { *p++ = '\t'; *p++ = '.'; continue; }  /* whole sample missing */
        const uint32_t g = rng_u32(&rng, 100);
        const char sep = rng_u32(&rng, 10) == 0 ? '|' : '/';
        *p++ = '\t';
        if (g < 2) { *p++ = '.'; *p++ = sep; *p++ = '.'; }
        else {
          const int a1 = g < 80 ? 0 : (int)(1 + rng_u32(&rng, (uint32_t)n_alt)), a2 = g < 70 ? 0 : (int)(1 + rng_u32(&rng, (uint32_t)n_alt));
          *p++ = (char)('0' + (a1 < a2 ? a1 : a2)); *p++ = sep; *p++ = (char)('0' + (a1 < a2 ? a2 : a1));
        }
        if (m >= 4 && m < 8) continue;  /* trailing fields dropped */
        *p++ = ':';
        if (rng_u32(&rng, 100) == 0) *p++ = '.'; else p += sprintf(p, "%u", rng_u32(&rng, 100));
        *p++ = ':';
        if (rng_u32(&rng, 100) == 0) *p++ = '.'; else p += sprintf(p, "%u", rng_u32(&rng, 251));
      }
    }
    *p++ = '\n';
    t->text_len = (size_t)((uint8_t*)p - t->text);
  }
  t->l_off[n] = t->text_len;
  compress_text(t->text, t->text_len, &t->comp, &t->comp_len, &t->n_blocks, &t->blk_clen);
  free(t->text);
  t->text = NULL;
}
static void* worker(void* arg) {
  (void)arg;
  for (;;) {
    pthread_mutex_lock(&g_mu);
    size_t i = g_next_tile++;
    pthread_mutex_unlock(&g_mu);
    if (i >= g_ntiles) break;
    gen_tile(&g_tiles[i], i);
  }
  return NULL;
}

typedef struct { uint64_t beg, end; } chunk_t;
typedef struct { uint32_t bin; chunk_t* c; size_t n, cap; } binrec_t;
typedef struct {
  binrec_t* bins; size_t nbins, capbins;
  int64_t* bin_lookup;
  uint64_t* lin; size_t nlin;
  uint64_t ref_beg, ref_end, n_mapped;
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

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s sites OUT.vcf.gz N_LINES [SEED] [THREADS] [LEVEL]\n       %s samples OUT.vcf.gz N_LINES N_SAMPLES [SEED] [THREADS] [LEVEL]\n", argv[0], argv[0]);
    return 1;
  }
  g_mode = strcmp(argv[1], "samples") == 0;
  const char* out_path = argv[2];
  uint64_t n_lines = strtoull(argv[3], NULL, 10);
  int ai = 4;
  if (g_mode) { if (argc < 5) return 1; g_nsamples = (uint32_t)strtoul(argv[4], NULL, 10); ai = 5; }
  if (argc > ai) g_seed = strtoull(argv[ai], NULL, 10);
  int threads = argc > ai + 1 ? atoi(argv[ai + 1]) : (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (argc > ai + 2) g_level = atoi(argv[ai + 2]);
  if (threads < 1) threads = 1;
  const uint64_t tile_lines = g_mode ? TILE_LINES_SAMPLES : TILE_LINES_SITES;

  /* lines per contig proportional to length */
  uint64_t per_ref[N_REF];
  {
    double total = 0;
    for (int r = 0; r < N_REF; r++) total += (double)REF_LENS[r];
    uint64_t used = 0;
    for (int r = 0; r < N_REF; r++) { per_ref[r] = (uint64_t)floor((double)n_lines * (double)REF_LENS[r] / total); used += per_ref[r]; }
    for (int r = 0; used < n_lines; r = (r + 1) % N_REF) { per_ref[r]++; used++; }
  }
  size_t ntiles = 0;
  for (int r = 0; r < N_REF; r++) ntiles += (size_t)((per_ref[r] + tile_lines - 1) / tile_lines);
  g_tiles = (tile_t*)calloc(ntiles ? ntiles : 1, sizeof(tile_t));
  g_ntiles = ntiles;
  size_t ti = 0;
  for (int r = 0; r < N_REF; r++) {
    const uint64_t nt = (per_ref[r] + tile_lines - 1) / tile_lines;
    uint64_t left = per_ref[r];
    for (uint64_t k = 0; k < nt; k++, ti++) {
      tile_t* t = &g_tiles[ti];
      t->ref = r;
      t->n_lines = left < tile_lines ? left : tile_lines;
      left -= t->n_lines;
      t->g0 = 10000 + (int64_t)((double)(REF_LENS[r] - 20000) * (double)k / (double)nt);
      t->g1 = 10000 + (int64_t)((double)(REF_LENS[r] - 20000) * (double)(k + 1) / (double)nt);
    }
  }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, NULL);
  for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);

  /* ---- header ---- */
  size_t hcap = 8192 + (size_t)g_nsamples * 16, hl = 0;
  char* hdr = (char*)malloc(hcap);
  hl += (size_t)sprintf(hdr + hl, "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n"
                                  "##FILTER=<ID=LowQual,Description=\"Low quality\">\n##FILTER=<ID=AC0,Description=\"Allele count is zero\">\n");
  for (int r = 0; r < N_REF; r++) hl += (size_t)sprintf(hdr + hl, "##contig=<ID=chr%d,length=%lld>\n", r + 1, (long long)REF_LENS[r]);
  hl += (size_t)sprintf(hdr + hl,
                        "##INFO=<ID=AC,Number=A,Type=Integer,Description=\"Alternate allele count\">\n"
                        "##INFO=<ID=AN,Number=1,Type=Integer,Description=\"Total number of alleles\">\n"
                        "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Alternate allele frequency\">\n"
                        "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Depth of informative coverage\">\n");
  if (!g_mode)
    hl += (size_t)sprintf(hdr + hl,
                          "##INFO=<ID=DB,Number=0,Type=Flag,Description=\"dbSNP membership\">\n"
                          "##INFO=<ID=SEGDUP,Number=0,Type=Flag,Description=\"Segmental duplication\">\n"
                          "##INFO=<ID=LCR,Number=0,Type=Flag,Description=\"Low complexity region\">\n"
                          "##INFO=<ID=VT,Number=1,Type=String,Description=\"Variant type\">\n"
                          "##INFO=<ID=RSRC,Number=.,Type=String,Description=\"Resources listing the site\">\n"
                          "##INFO=<ID=CULPRIT,Number=1,Type=String,Description=\"Worst-performing annotation\">\n"
                          "##INFO=<ID=VQSLOD,Number=1,Type=Float,Description=\"Log odds of being a true variant\">\n");
  else
    hl += (size_t)sprintf(hdr + hl,
                          "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
                          "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">\n"
                          "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read depth\">\n");
  hl += (size_t)sprintf(hdr + hl, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO");
  if (g_mode) {
    hl += (size_t)sprintf(hdr + hl, "\tFORMAT");
    for (uint32_t s = 0; s < g_nsamples; s++) hl += (size_t)sprintf(hdr + hl, "\tS%05u", s + 1);
  }
  hdr[hl++] = '\n';
  uint8_t* hcomp; size_t hcomp_len; uint32_t hnb; uint32_t* hcl;
  compress_text((const uint8_t*)hdr, hl, &hcomp, &hcomp_len, &hnb, &hcl);

  FILE* f = fopen(out_path, "wb");
  if (!f) { perror(out_path); return 1; }
  fwrite(hcomp, 1, hcomp_len, f);
  uint64_t coff = hcomp_len, total_u = hl, total_blocks = hnb, total_lines = 0;
  uint64_t* tile_coff = (uint64_t*)malloc((ntiles + 1) * 8);
  for (size_t i = 0; i < ntiles; i++) {
    tile_coff[i] = coff;
    fwrite(g_tiles[i].comp, 1, g_tiles[i].comp_len, f);
    coff += g_tiles[i].comp_len;
    total_blocks += g_tiles[i].n_blocks;
    total_lines += g_tiles[i].n_lines;
    total_u += g_tiles[i].l_off[g_tiles[i].n_lines];
  }
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  fwrite(eof, 1, 28, f);
  total_blocks++;
  const uint64_t file_len = coff + 28;
  fclose(f);

  /* ---- tabix index ---- */
  refidx_t* ri = (refidx_t*)calloc(N_REF, sizeof(refidx_t));
  uint64_t per_chr1 = 0;
  for (size_t i = 0; i < ntiles; i++) {
    tile_t* t = &g_tiles[i];
    uint64_t* bc = (uint64_t*)malloc((t->n_blocks + 2) * 8);
    bc[0] = tile_coff[i];
    for (uint32_t b = 0; b < t->n_blocks; b++) bc[b + 1] = bc[b] + t->blk_clen[b];
    if (t->ref == 0) per_chr1 += t->n_lines;
    for (uint64_t k = 0; k < t->n_lines; k++) {
      const uint64_t o0 = t->l_off[k], o1 = t->l_off[k + 1];
      const uint64_t v0 = (bc[o0 / BLOCK_PAYLOAD] << 16) | (o0 % BLOCK_PAYLOAD);
      uint64_t v1;
      if (o1 == t->l_off[t->n_lines] && (o1 % BLOCK_PAYLOAD) == 0) v1 = bc[t->n_blocks] << 16;
      else if (o1 % BLOCK_PAYLOAD == 0) v1 = bc[o1 / BLOCK_PAYLOAD] << 16;
      else if (o1 == t->l_off[t->n_lines]) v1 = bc[t->n_blocks] << 16;  /* end of the tile's last member = start of the next */
      else v1 = (bc[o1 / BLOCK_PAYLOAD] << 16) | (o1 % BLOCK_PAYLOAD);
      refidx_t* r = &ri[t->ref];
      const int64_t beg = t->l_pos[k] - 1, end = beg + t->l_rlen[k];
      bin_add(r, (uint32_t)reg2bin(beg, end), v0, v1);
      const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
      if (w1 + 1 > r->nlin) {
        r->lin = (uint64_t*)realloc(r->lin, (w1 + 1) * 8);
        for (size_t w = r->nlin; w <= w1; w++) r->lin[w] = 0;
        r->nlin = w1 + 1;
      }
      for (size_t w = w0; w <= w1; w++) if (!r->lin[w]) r->lin[w] = v0;
      if (!r->ref_beg) r->ref_beg = v0;
      r->ref_end = v1;
      r->n_mapped++;
    }
    free(bc);
  }
  size_t icap = 1 << 20, il = 0;
  uint8_t* ib = (uint8_t*)malloc(icap);
#define IRES(n) do { if (icap - il < (size_t)(n) + 64) { icap = (icap + (size_t)(n)) * 2; ib = (uint8_t*)realloc(ib, icap); } } while (0)
  memcpy(ib, "TBI\1", 4); il = 4;
  int n_used = 0;
  for (int r = 0; r < N_REF; r++) if (ri[r].n_mapped) n_used = r + 1;
  char names[512];
  int nl = 0;
  for (int r = 0; r < n_used; r++) { nl += sprintf(names + nl, "chr%d", r + 1); names[nl++] = 0; }
  put32(ib + il, (uint32_t)n_used); put32(ib + il + 4, 2); put32(ib + il + 8, 1); put32(ib + il + 12, 2); put32(ib + il + 16, 0);
  put32(ib + il + 20, '#'); put32(ib + il + 24, 0); put32(ib + il + 28, (uint32_t)nl);
  il += 32;
  memcpy(ib + il, names, (size_t)nl); il += (size_t)nl;
  for (int r = 0; r < n_used; r++) {
    refidx_t* x = &ri[r];
    const int has_meta = x->n_mapped > 0;
    IRES(8);
    put32(ib + il, (uint32_t)(x->nbins + (has_meta ? 1 : 0))); il += 4;
    for (size_t b = 0; b < x->nbins; b++) {
      IRES(8 + x->bins[b].n * 16);
      put32(ib + il, x->bins[b].bin); put32(ib + il + 4, (uint32_t)x->bins[b].n); il += 8;
      memcpy(ib + il, x->bins[b].c, x->bins[b].n * 16); il += x->bins[b].n * 16;
    }
    if (has_meta) {
      IRES(48);
      put32(ib + il, 37450); put32(ib + il + 4, 2); il += 8;
      uint64_t m[4] = {x->ref_beg, x->ref_end, x->n_mapped, 0};
      memcpy(ib + il, m, 32); il += 32;
    }
    for (size_t w = 1; w < x->nlin; w++) if (!x->lin[w]) x->lin[w] = x->lin[w - 1];
    IRES(8 + x->nlin * 8);
    put32(ib + il, (uint32_t)x->nlin); il += 4;
    memcpy(ib + il, x->lin, x->nlin * 8); il += x->nlin * 8;
  }
  IRES(8);
  memset(ib + il, 0, 8); il += 8;
  uint8_t* icomp; size_t icomp_len; uint32_t inb; uint32_t* icl;
  compress_text(ib, il, &icomp, &icomp_len, &inb, &icl);
  char tbi_path[4096];
  snprintf(tbi_path, sizeof tbi_path, "%s.tbi", out_path);
  f = fopen(tbi_path, "wb");
  if (!f) { perror(tbi_path); return 1; }
  fwrite(icomp, 1, icomp_len, f);
  fwrite(eof, 1, 28, f);
  fclose(f);
  printf("{\"path\": \"%s\", \"mode\": \"%s\", \"n_blocks\": %llu, \"n_lines\": %llu, \"n_lines_chr1\": %llu, \"n_samples\": %u, "
         "\"compressed_bytes\": %llu, \"inflated_bytes\": %llu, \"seed\": %llu, \"level\": %d, \"threads\": %d}\n",
         out_path, g_mode ? "samples" : "sites", (unsigned long long)total_blocks, (unsigned long long)total_lines,
         (unsigned long long)per_chr1, g_nsamples, (unsigned long long)file_len, (unsigned long long)total_u,
         (unsigned long long)g_seed, g_level, threads);
  return 0;
}
