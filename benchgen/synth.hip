// synth.hip — deterministic synthetic VCF text (bench / test tooling, not product code).
//
// Row i is a pure function of (seed, i): the same bytes come out of the host functions and of
// the HIP kernels, so parity tests at small sizes and the device-resident bench at full size read
// one data model:
//   fixed columns  "1 <POS> rs<id> <REF> <ALT> 100 PASS AC=..;AF=..;AN=..;NS=..;DP=..;VT=.. [GT]"
//   samples        "x|y" per sample, alleles placed per haplotype from the 1000-Genomes allele-count
//                  spectrum (AC=1 40.5 %, 2 11.9 %, 3-10 19.9 %, 11-100 16.4 %, 101-1000 6.6 %,
//                  >1000 4.7 %); one haplotype always carries an ALT so no row is dropped by ac==0
// Profile knobs (per 10 000 rows): multiallelic, indel and malformed rows (BASELINE config 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define HD __host__ __device__ inline

struct SynthCfg {
  uint64_t seed;
  uint32_t n_samples;
  uint32_t p_multi;  // per 10 000
  uint32_t p_indel;  // per 10 000
  uint32_t p_bad;    // per 10 000
  uint32_t pos0;
  uint32_t reserved;
};

HD uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

HD uint32_t put_str(uint8_t *b, uint32_t n, const char *s) {
  while (*s) b[n++] = (uint8_t)*s++;
  return n;
}

HD uint32_t put_u(uint8_t *b, uint32_t n, uint64_t v) {
  char tmp[24];
  int k = 0;
  do {
    tmp[k++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (k) b[n++] = (uint8_t)tmp[--k];
  return n;
}

HD uint32_t put_upad(uint8_t *b, uint32_t n, uint32_t v, int width) {
  for (int i = width - 1; i >= 0; i--) {
    b[n + i] = (uint8_t)('0' + v % 10);
    v /= 10;
  }
  return n + width;
}

struct RowParams {
  uint32_t n_alts;   // ALT alleles a genotype may carry (1..3)
  uint32_t thr;      // carrier probability per haplotype, as a 32-bit threshold
  uint32_t h0;       // the haplotype that always carries an ALT
};

HD char base_of(uint32_t i) { return "ACGT"[i & 3]; }

// fixed columns of row r (up to and including FORMAT when there are samples) into buf (>= 320 B);
// returns the length.  Single source of truth for both the length pass and the fill pass.
HD uint32_t fixed_part(const SynthCfg &c, uint64_t r, uint8_t *buf, RowParams *rp) {
  const uint64_t h = splitmix64(c.seed ^ (r * 0x9E3779B97F4A7C15ull));
  const uint64_t h2 = splitmix64(h);
  const uint64_t h3 = splitmix64(h2);
  const uint32_t ns = c.n_samples;
  const uint32_t nhap = ns ? 2 * ns : 2;
  // allele count target from the 1KG spectrum
  uint32_t u = (uint32_t)(h % 1000), k;
  if (u < 405) k = 1;
  else if (u < 524) k = 2;
  else if (u < 723) k = 3 + (uint32_t)(h2 % 8);
  else if (u < 887) k = 11 + (uint32_t)(h2 % 90);
  else if (u < 953) k = 101 + (uint32_t)(h2 % 900);
  else k = 1001 + (uint32_t)(h2 % 4000);
  // "dense" rows (reserved bit 2): a common variant, AF ~ 0.3 -- ~1 050 heterozygotes and ~225 homozygotes of 2 504
  // samples, i.e. the longest output rows of the reference's own golden file (20 KB) on every line
  if (c.reserved & 4u) k = nhap * 3u / 10u + (uint32_t)(h2 % 64);
  if (k > nhap) k = nhap;
  rp->thr = (uint32_t)((((uint64_t)k << 32) / nhap) > 0xFFFFFFFFull ? 0xFFFFFFFFull : (((uint64_t)k << 32) / nhap));
  rp->h0 = (uint32_t)((h3 >> 20) % nhap);

  const uint32_t kind_u = (uint32_t)((h >> 32) % 10000);
  // 0 SNP, 1 multiallelic, 2 INS, 3 DEL, 4 padded DEL, 5 malformed
  uint32_t kind = 0;
  if (kind_u < c.p_multi) kind = 1;
  else if (kind_u < c.p_multi + c.p_indel / 2) kind = 2;
  else if (kind_u < c.p_multi + c.p_indel) kind = ((h3 & 3) == 0) ? 4 : 3;
  else if (kind_u < c.p_multi + c.p_indel + c.p_bad) kind = 5;

  const uint32_t rb = (uint32_t)(h2 >> 8) & 3;
  uint32_t n = 0;
  buf[n++] = '1';
  buf[n++] = '\t';
  n = put_u(buf, n, (uint64_t)c.pos0 + r * 151ull + (h3 % 150));
  buf[n++] = '\t';
  n = put_str(buf, n, "rs");
  n = put_u(buf, n, 1 + (h2 >> 16) % 999999999ull);
  buf[n++] = '\t';
  rp->n_alts = 1;
  const char *vt = "SNP";
  if (kind == 0) {
    buf[n++] = base_of(rb);
    buf[n++] = '\t';
    buf[n++] = base_of(rb + 1 + (uint32_t)((h2 >> 12) % 3));
  } else if (kind == 1) {
    const uint32_t na = 2 + (uint32_t)((h3 >> 8) & 1);
    rp->n_alts = na;
    buf[n++] = base_of(rb);
    buf[n++] = '\t';
    for (uint32_t i = 0; i < na; i++) {
      if (i) buf[n++] = ',';
      buf[n++] = base_of(rb + 1 + i);
    }
    vt = "MULTI";
  } else if (kind == 2) {
    const uint32_t li = 1 + (uint32_t)((h3 >> 10) % 8);
    buf[n++] = base_of(rb);
    buf[n++] = '\t';
    buf[n++] = base_of(rb);
    for (uint32_t i = 0; i < li; i++) buf[n++] = base_of((uint32_t)(h3 >> (16 + 2 * i)));
    vt = "INDEL";
  } else if (kind == 3) {
    const uint32_t lr = 2 + (uint32_t)((h3 >> 10) % 8);
    buf[n++] = base_of(rb);
    for (uint32_t i = 1; i < lr; i++) buf[n++] = base_of((uint32_t)(h3 >> (16 + 2 * i)));
    buf[n++] = '\t';
    buf[n++] = base_of(rb);
    vt = "INDEL";
  } else if (kind == 4) {
    // X + del + pad  ->  X + pad : exercises left-normalisation (main.go:971-998)
    const uint32_t ld = 1 + (uint32_t)((h3 >> 10) % 5), lp = 1 + (uint32_t)((h3 >> 14) % 3);
    const uint32_t x = rb;
    buf[n++] = base_of(x);
    for (uint32_t i = 0; i < ld; i++) buf[n++] = base_of(x + 1 + (uint32_t)(h3 >> (20 + 2 * i)) % 3);
    for (uint32_t i = 0; i < lp; i++) buf[n++] = base_of(x + 2 + i);
    buf[n++] = '\t';
    buf[n++] = base_of(x);
    for (uint32_t i = 0; i < lp; i++) buf[n++] = base_of(x + 2 + i);
    vt = "INDEL";
  } else {
    buf[n++] = base_of(rb);
    buf[n++] = '\t';
    if (h3 & 1) n = put_str(buf, n, "<CN0>");
    else {  // deletion-shaped with a mismatched padding base
      buf[n - 2] = base_of(rb);
      buf[n - 1] = base_of(rb + 1);
      buf[n++] = base_of(rb + 2);
      buf[n++] = '\t';
      buf[n++] = base_of(rb + 3);
    }
    vt = "SV";
  }
  n = put_str(buf, n, "\t100\tPASS\tAC=");
  n = put_u(buf, n, k);
  n = put_str(buf, n, ";AF=0.");
  n = put_upad(buf, n, (uint32_t)((uint64_t)k * 9999 / nhap), 4);
  n = put_str(buf, n, ";AN=");
  n = put_u(buf, n, nhap);
  n = put_str(buf, n, ";NS=");
  n = put_u(buf, n, ns);
  n = put_str(buf, n, ";DP=");
  n = put_u(buf, n, 1000 + (h3 >> 40) % 30000);
  // population tags of the real 1KG INFO column bring the fixed columns to ~152 B
  const uint32_t af4 = (uint32_t)((uint64_t)k * 9999 / nhap);
  n = put_str(buf, n, ";EAS_AF=0.");
  n = put_upad(buf, n, af4 / 2, 4);
  n = put_str(buf, n, ";AMR_AF=0.");
  n = put_upad(buf, n, af4, 4);
  n = put_str(buf, n, ";AFR_AF=0.");
  n = put_upad(buf, n, af4 / 3, 4);
  n = put_str(buf, n, ";EUR_AF=0.");
  n = put_upad(buf, n, af4 / 4, 4);
  n = put_str(buf, n, ";AA=.|||");
  n = put_str(buf, n, ";VT=");
  n = put_str(buf, n, vt);
  if (c.reserved & 1u) {  // experiment knob: pad INFO so every sample region starts 16-byte aligned
    n = put_str(buf, n, ";P=");
    while ((n + (ns ? 4u : 1u)) % 16u) buf[n++] = 'X';
  }
  if (ns) n = put_str(buf, n, (c.reserved & 2u) ? "\tGT:DP:GQ" : "\tGT");
  return n;
}

// Extended FORMAT (reserved bit 1): "\tx|y:DP:GQ" with a 1- or 2-digit DP, so fields are 9 or 10
// bytes and not on a fixed stride.  Sample s of row r is short iff (s + r) % 3 == 0.
HD uint32_t ext_short_before(uint64_t r, uint32_t s) {
  const uint32_t c0 = (uint32_t)((3u - (uint32_t)(r % 3u)) % 3u);  // first short sample
  return s > c0 ? (s - c0 + 2u) / 3u : 0u;
}
HD uint32_t ext_offset(uint64_t r, uint32_t s) { return 10u * s - ext_short_before(r, s); }

HD uint32_t samples_bytes(const SynthCfg &c, uint64_t r) {
  return (c.reserved & 2u) ? ext_offset(r, c.n_samples) : 4u * c.n_samples;
}

HD void put_sample(const SynthCfg &c, uint64_t r, const RowParams &rp, uint32_t s, uint8_t *row_samples);

HD uint32_t row_length(const SynthCfg &c, uint64_t r) {
  uint8_t buf[320];
  RowParams rp;
  return fixed_part(c, r, buf, &rp) + samples_bytes(c, r) + 1u;
}

// allele digit carried by haplotype hap of row r
HD uint8_t hap_allele(const SynthCfg &c, uint64_t r, const RowParams &rp, uint32_t hap) {
  const uint64_t x = splitmix64((c.seed + 0x51ED27ull) ^ (r * 0xD1B54A32D192ED03ull) ^ ((uint64_t)hap * 0x9E3779B97F4A7C15ull));
  const bool carrier = (uint32_t)x < rp.thr || hap == rp.h0;
  if (!carrier) return '0';
  return (uint8_t)('1' + (uint32_t)(x >> 40) % rp.n_alts);
}

HD void put_sample(const SynthCfg &c, uint64_t r, const RowParams &rp, uint32_t s, uint8_t *row_samples) {
  if (!(c.reserved & 2u)) {
    uint8_t *p = row_samples + 4u * s;
    p[0] = '\t';
    p[1] = hap_allele(c, r, rp, 2 * s);
    p[2] = '|';
    p[3] = hap_allele(c, r, rp, 2 * s + 1);
    return;
  }
  uint8_t *p = row_samples + ext_offset(r, s);
  const bool is_short = (s + r) % 3u == 0;
  const uint64_t x = splitmix64(c.seed ^ (r * 0x2545F4914F6CDD1Dull) ^ s);
  uint32_t n = 0;
  // (reserved bit 3: one sample in twenty -- the same ones on every row, as the males of a chrX file -- has haploid
  // calls, "x:" with two more digits of DP in place of "|y": the field keeps its length)
  const bool haploid = (c.reserved & 8u) && splitmix64((c.seed + 0x4D414C45ull) ^ ((uint64_t)s * 0x9E3779B97F4A7C15ull)) % 20u == 0;
  p[n++] = '\t';
  p[n++] = hap_allele(c, r, rp, 2 * s);
  if (haploid) {
    p[n++] = ':';
    p[n++] = (uint8_t)('0' + (x >> 32) % 10);
    p[n++] = (uint8_t)('0' + (x >> 36) % 10);
  } else {
    p[n++] = '|';
    p[n++] = hap_allele(c, r, rp, 2 * s + 1);
    p[n++] = ':';
  }
  if (is_short) {
    p[n++] = (uint8_t)('0' + x % 10);
  } else {
    p[n++] = (uint8_t)('1' + x % 9);
    p[n++] = (uint8_t)('0' + (x >> 8) % 10);
  }
  p[n++] = ':';
  p[n++] = (uint8_t)('0' + (x >> 16) % 10);
  p[n++] = (uint8_t)('0' + (x >> 24) % 10);
}

__global__ void k_lengths(SynthCfg c, uint64_t first, uint32_t n, int64_t *len) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) len[i] = row_length(c, first + i);
}

// one workgroup per row
__global__ __launch_bounds__(256) void k_fill(SynthCfg c, uint64_t first, uint32_t n, const int64_t *off, uint8_t *out) {
  __shared__ uint8_t s_fixed[320];
  __shared__ uint32_t s_len;
  __shared__ RowParams s_rp;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint64_t r = first + i;
    if (threadIdx.x == 0) {
      RowParams rp;
      s_len = fixed_part(c, r, s_fixed, &rp);
      s_rp = rp;
    }
    __syncthreads();
    uint8_t *row = out + off[i];
    const uint32_t fl = s_len;
    const RowParams rp = s_rp;
    for (uint32_t k = threadIdx.x; k < fl; k += blockDim.x) row[k] = s_fixed[k];
    for (uint32_t s = threadIdx.x; s < c.n_samples; s += blockDim.x) put_sample(c, r, rp, s, row + fl);
    if (threadIdx.x == 0) row[fl + samples_bytes(c, r)] = '\n';
    __syncthreads();
  }
}

extern "C" {

// "##fileformat..." + "#CHROM ..." header for this config; returns its length
size_t synth_header(const SynthCfg *c, char *out, size_t cap) {
  size_t n = 0;
  auto put = [&](const char *s) {
    size_t l = strlen(s);
    if (out && n + l < cap) memcpy(out + n, s, l);
    n += l;
  };
  put("##fileformat=VCFv4.1\n##source=bvcf_synth\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO");
  if (c->n_samples) put("\tFORMAT");
  char nm[16];
  for (uint32_t s = 0; s < c->n_samples; s++) {
    snprintf(nm, sizeof nm, "\tHG%05u", s);
    put(nm);
  }
  put("\n");
  if (out && n < cap) out[n] = 0;
  return n;
}

// total bytes of rows [first, first + n)
uint64_t synth_rows_bytes_host(const SynthCfg *c, uint64_t first, uint64_t n) {
  uint64_t t = 0;
  for (uint64_t i = 0; i < n; i++) t += row_length(*c, first + i);
  return t;
}

// rows [first, first + n) written back to back; returns bytes written (<= cap or 0 on overflow)
uint64_t synth_fill_host(const SynthCfg *c, uint64_t first, uint64_t n, uint8_t *out, uint64_t cap) {
  uint64_t o = 0;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t r = first + i;
    RowParams rp;
    uint8_t fx[320];
    const uint32_t fl = fixed_part(*c, r, fx, &rp);
    const uint32_t sb = samples_bytes(*c, r);
    const uint64_t need = (uint64_t)fl + sb + 1;
    if (o + need > cap) return 0;
    memcpy(out + o, fx, fl);
    for (uint32_t s = 0; s < c->n_samples; s++) put_sample(*c, r, rp, s, out + o + fl);
    out[o + fl + sb] = '\n';
    o += need;
  }
  return o;
}

// device side: len[i] (int64, so torch.cumsum applies directly) then fill at off[i]
int synth_lengths_dev(const SynthCfg *c, uint64_t first, uint32_t n, int64_t *d_len, void *stream) {
  hipLaunchKernelGGL(k_lengths, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *c, first, n, d_len);
  return (int)hipGetLastError();
}

int synth_fill_dev(const SynthCfg *c, uint64_t first, uint32_t n, const int64_t *d_off, uint8_t *d_out, void *stream) {
  const uint32_t grid = n < 65536u ? (n ? n : 1) : 65536u;
  hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, (hipStream_t)stream, *c, first, n, d_off, d_out);
  return (int)hipGetLastError();
}

}  // extern "C"
