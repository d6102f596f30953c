/* synth_fastq.c -- deterministic synthetic BGZF-FASTQ (+GZI) generator (bench / test tooling).
 * Illumina-like 101 bp reads: "@SIM.{n} HSQ:{lane}:{tile}:{x}:{y}/1", ACGT with 0.1% N, Markov-run
 * qualities (Phred+33 in '#'..'J'), '+' separator.  The text is cut into BGZF members of 65280 bytes
 * like bgzip (records straddle members); the file is a pure function of (n_blocks, seed).
 * usage: synth_fastq OUT.fastq.bgz N_BLOCKS [SEED=42] [THREADS=nproc] [LEVEL=6]
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#define TILE_BLOCKS 64
#define BLOCK_PAYLOAD 65280

typedef void* (*ld_alloc_t)(int);
typedef size_t (*ld_comp_t)(void*, const void*, size_t, void*, size_t);
typedef void (*ld_free_t)(void*);
static ld_alloc_t ld_alloc; static ld_comp_t ld_comp; static ld_free_t ld_free;
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
typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix(uint64_t* x) { uint64_t z = (*x += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static void rng_seed(rng_t* r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r) { uint64_t* s = r->s; uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return res; }

typedef struct { uint8_t* comp; size_t comp_len; uint32_t n_blocks; uint32_t* clen; uint32_t* ulen; uint64_t n_rec; } tile_t;
static tile_t* g_tiles; static size_t g_ntiles, g_next; static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static uint64_t g_seed = 42; static int g_level = 6;

static void put32(uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
static size_t bgzf_member(void* ldc, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
  memcpy(dst, hdr, 16);
  size_t clen;
  if (ldc) { clen = ld_comp(ldc, src, n, dst + 18, cap - 26); if (!clen) { fprintf(stderr, "block did not fit\n"); exit(2); } }
  else {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n; zs.next_out = dst + 18; zs.avail_out = (uInt)(cap - 26);
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "block did not fit\n"); exit(2); }
    clen = zs.total_out; deflateEnd(&zs);
  }
  size_t total = 18 + clen + 8;
  dst[16] = (uint8_t)(total - 1); dst[17] = (uint8_t)((total - 1) >> 8);
  put32(dst + 18 + clen, (uint32_t)crc32(crc32(0, NULL, 0), src, (uInt)n));
  put32(dst + 18 + clen + 4, (uint32_t)n);
  return total;
}

static void gen_tile(tile_t* t, size_t idx) {
  rng_t rng; rng_seed(&rng, g_seed * 0x100000001B3ull + idx * 0x9E3779B97F4A7C15ull + 7);
  const size_t want = (size_t)t->n_blocks * BLOCK_PAYLOAD;
  uint8_t* text = (uint8_t*)malloc(want + 1024);
  size_t tl = 0;
  uint64_t n = 0;
  while (tl < want) {
    char* p = (char*)text + tl;
    int k = sprintf(p, "@SIM.%zu.%llu HSQ:%d:%u:%u:%u/1\n", idx, (unsigned long long)n, 1 + (int)(idx % 8), (unsigned)(1000 + idx % 2000),
                    (unsigned)(rng_next(&rng) % 20000), (unsigned)(rng_next(&rng) % 200000));
    p += k;
    for (int i = 0; i < 101; i += 4) {
      uint64_t x = rng_next(&rng);
      for (int j = 0; j < 4 && i + j < 101; j++) { p[j] = ((x >> (16 * j + 2)) & 1023) == 0 ? 'N' : "ACGT"[(x >> (16 * j)) & 3]; }
      p += (i + 4 <= 101) ? 4 : 101 - i;
    }
    *p++ = '\n'; *p++ = '+'; *p++ = '\n';
    int q = 30 + (int)(rng_next(&rng) % 11);
    for (int i = 0; i < 101;) {
      uint64_t x = rng_next(&rng);
      int run = 1 + (int)(x & 7), step = (int)((x >> 3) % 7) - 3;
      if (((x >> 8) & 31) == 0) q = 2 + (int)((x >> 16) % 20); else q += step;
      if (q < 2) q = 2;
      if (q > 41) q = 41;
      for (int j = 0; j < run && i < 101; j++, i++) *p++ = (char)(33 + q);
    }
    *p++ = '\n';
    tl = (size_t)((uint8_t*)p - text);
    n++;
  }
  t->n_rec = n;
  /* the tile ends on a record boundary: its last member is longer/shorter than 65280 as needed */
  uint32_t nb = (uint32_t)((tl + BLOCK_PAYLOAD - 1) / BLOCK_PAYLOAD);
  t->n_blocks = nb;
  t->clen = (uint32_t*)calloc(nb, 4); t->ulen = (uint32_t*)calloc(nb, 4);
  t->comp = (uint8_t*)malloc((size_t)nb * 66000);
  void* ldc = ld_alloc ? ld_alloc(g_level) : NULL;
  size_t o = 0;
  for (uint32_t b = 0; b < nb; b++) {
    size_t a = (size_t)b * BLOCK_PAYLOAD, len = tl - a < BLOCK_PAYLOAD ? tl - a : BLOCK_PAYLOAD;
    size_t m = bgzf_member(ldc, text + a, len, t->comp + o, 66000);
    t->clen[b] = (uint32_t)m; t->ulen[b] = (uint32_t)len; o += m;
  }
  t->comp_len = o;
  if (ldc) ld_free(ldc);
  free(text);
}
static void* worker(void* a) { (void)a; for (;;) { pthread_mutex_lock(&g_mu); size_t i = g_next++; pthread_mutex_unlock(&g_mu); if (i >= g_ntiles) break; gen_tile(&g_tiles[i], i); } return NULL; }

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s OUT.fastq.bgz N_BLOCKS [SEED] [THREADS] [LEVEL]\n", argv[0]); return 1; }
  uint64_t n_blocks = strtoull(argv[2], NULL, 10);
  if (argc > 3) g_seed = strtoull(argv[3], NULL, 10);
  int threads = argc > 4 ? atoi(argv[4]) : (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (argc > 5) g_level = atoi(argv[5]);
  if (threads < 1) threads = 1;
  if (!getenv("SYNTH_ZLIB")) load_libdeflate();
  if (n_blocks < 2) n_blocks = 2;
  uint64_t data_blocks = n_blocks - 1;
  g_ntiles = (size_t)((data_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS);
  g_tiles = (tile_t*)calloc(g_ntiles, sizeof(tile_t));
  uint64_t left = data_blocks;
  for (size_t i = 0; i < g_ntiles; i++) { uint64_t nb = left < TILE_BLOCKS ? left : TILE_BLOCKS; g_tiles[i].n_blocks = (uint32_t)nb; left -= nb; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, NULL);
  for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
  FILE* f = fopen(argv[1], "wb");
  if (!f) { perror(argv[1]); return 1; }
  char gp[4096]; snprintf(gp, sizeof gp, "%s.gzi", argv[1]);
  FILE* g = fopen(gp, "wb");
  uint64_t coff = 0, uoff = 0, nrec = 0, nblk = 0, nent = 0;
  fwrite(&nent, 8, 1, g);
  for (size_t i = 0; i < g_ntiles; i++) {
    fwrite(g_tiles[i].comp, 1, g_tiles[i].comp_len, f);
    for (uint32_t b = 0; b < g_tiles[i].n_blocks; b++) {
      coff += g_tiles[i].clen[b]; uoff += g_tiles[i].ulen[b]; nblk++;
      int last = (i + 1 == g_ntiles) && (b + 1 == g_tiles[i].n_blocks);
      if (!last) { fwrite(&coff, 8, 1, g); fwrite(&uoff, 8, 1, g); nent++; }
    }
    nrec += g_tiles[i].n_rec;
  }
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  fwrite(eof, 1, 28, f); fclose(f);
  fseek(g, 0, SEEK_SET); fwrite(&nent, 8, 1, g); fclose(g);
  printf("{\"path\": \"%s\", \"n_blocks\": %llu, \"n_records\": %llu, \"compressed_bytes\": %llu, \"inflated_bytes\": %llu, \"seed\": %llu, \"level\": %d, \"deflate\": \"%s\"}\n",
         argv[1], (unsigned long long)(nblk + 1), (unsigned long long)nrec, (unsigned long long)(coff + 28), (unsigned long long)uoff,
         (unsigned long long)g_seed, g_level, ld_alloc ? "libdeflate" : "zlib");
  return 0;
}
