// bvcf_streamgen.hip.h — the streaming path over lines whose sample fields are not the bare 4-byte "x|y<TAB>":
// FORMAT with sub-fields beyond GT (every GATK-style file, examples/test.query.vcf), main.go:1042-1194
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
//
// k_stream's regular scan predicts a line's end from 4 bytes per sample; a line that does not fit is only
// delimited there and scanned again by k_gt (two reads of the text, the second one per-field).  k_stream_gen is the
// kernel for files made of such lines (the host launches it in k_stream's place once a batch has shown the file's
// shape, bvcf_core.hip): the wave walks its run as ONE stream of 1 KiB chunks on a fixed grid, a ring of them in flight, and a small state
// machine (in the head: count TABs to the 9th; in the samples: count TABs, look at the field after each) takes the
// bytes as they come -- every byte is loaded once, there is no per-line load latency, and the line's end is
// simply the terminator the stream runs into.
//
// The chunk of a cohort file is nearly always "samples, no terminator, every field starts with the reference
// genotype": that case is decided from byte flags packed in the loaded dwords (0x80 per TAB byte), without turning
// them into bit masks and without a prefix sum:
//   * a dword holds at most one field start worth looking at -- the byte after its first TAB flag (a second start in
//     the same dword means the first field is at most two bytes long: it cannot equal the reference word and sends
//     the chunk to the exact handler anyway);
//   * the four bytes at that start (v_alignbyte over the dword and its successor) are compared with ONE word R, the
//     reference genotype as this file writes it ("0/0:", "0|0:", "0/0<TAB>", ...; adopted from the fields seen);
//   * TABs are counted per lane (v_bcnt) and summed when the line ends.
// Anything else -- a terminator, a head, a field that is not R, a byte >= 0x80, the end of the block -- goes through
// the exact handler below (bit masks, sample index by prefix sum, class by table).  It accepts the fields the
// reference's own fast gate accepts (main.go:1063-1124: "x<sep>y" followed by ':' or the field's end, x and y
// digits or '.'); a line with any other field, or whose field count is not the header's, is listed as deferred
// and k_gt scans it the byte-serial way, as before.  Counts then follow from the classes: ac = het + 2 hom,
// an = 2 (samples - missing).
// (Round 4 tried to keep polyploid / two-digit / empty fields in the kernel as well: a mark in the line's list where such
// a field is met, the field found again and classified by the general branch's restatement when the line ends.  Parity
// was green, and the kernel took 19 % longer on every GATK-shaped file (0.645 -> 0.766 ms per 2.4 GB block): it runs at
// the limit of its scalar registers, and the cold code -- inlined, marked unlikely, or called out of line -- doubled the
// scalar spills around the hot loop (28 -> 48-60) or cost a wave per SIMD (84 vector registers).  Not kept: such
// fields are rare, the file shape is not)
//
// The class map of ALT #1 is a short list while few samples carry the allele (BVCF_ALLELE_CMAP_SPARSE, sorted and
// merged per map byte when the line ends), a dense map staged in LDS otherwise.  A sample that carries a further
// ALT index makes the map dense and leaves those indices to k_gt (the list form promises that no further index is
// carried).
#pragma once

// (included by bvcf_stream.hip.h, after its helpers)

namespace bvcf_dev {

// (measured on the GT:DP:GQ profile, 2 504 samples: a ring of 8 with the 4 workgroups per CU its LDS allows reads
// 3.4 TB/s, a ring of 4 with 6 workgroups 3.8 TB/s -- the kernel is bound by instruction issue, more waves hide more)
#ifndef BVCF_GEN_RING
#define BVCF_GEN_RING 4
#endif
constexpr int kGenRing = BVCF_GEN_RING;  // chunks of the LDS ring: one being read, the others in flight
// s_waitcnt immediate of gfx9: vmcnt(n), nothing asked of expcnt / lgkmcnt
constexpr int vmcnt_imm(int n) { return (n & 0xF) | 0x70 | 0xF00 | ((n >> 4) << 14); }

constexpr uint32_t kNoneGen = kNone;

// walks the lines that start in [p0, r1); p0 is a line start
template <class Commit, class MapSlot>
__device__ __forceinline__ void stream_general_run(const KernelArgs &a, uint32_t p0, uint32_t r1, uint32_t n_map_chunks,
                                                   uint8_t *stage, uint32_t *glist, const uint8_t *ring, uint32_t &seen,
                                                   uint32_t &n_regular, Commit &&commit, MapSlot &&map_slot) {
  const int lane = lane_id();
  const uint32_t ns = a.n_samples;
  const uint32_t nb = a.nbytes;
  const bool maps = a.want_cmap != 0;
  const bool list_ok = maps && a.cmap_stride >= 4u * kSparseWords;
  // glist: list mode, (sample << 8 | first allele digit << 4 | second allele digit) of every field that is not the
  // reference genotype -- digits as the byte ^ '0' & 15, i.e. 14 for '.' (the classes of every ALT index follow from them)
  uint32_t *stage32 = reinterpret_cast<uint32_t *>(stage);  // dense mode: the 2-bit map
  const uint32_t base = p0 & ~3u;
  const uint32_t cap_off = (a.cap - 16u) & ~3u;
  // (32-bit offsets do not wrap: a block ends below 4 GiB - 1 MiB, the stream stops at the first chunk past it and
  // loads run kGenRing chunks ahead)
  auto chunk_start = [&](uint32_t c) -> uint32_t { return base + c * kChunk; };

  // ---- state (wave-uniform unless said otherwise)
  enum : uint32_t { kHead = 0, kSamples = 1 };
  uint32_t mode = kHead;
  uint32_t ls = p0;        // start of the line in progress
  uint32_t cur_pos = p0;   // first byte the state machine has not consumed
  uint32_t found = 0;      // head: TABs seen
  uint32_t s_begin = 0;    // samples: first byte of the sample region
  uint32_t tabs_base = 0;  // samples: TABs of the region counted so far (exact handler) ...
  uint32_t tabs_lane = 0;  // ... plus what each lane counted in fast chunks (per lane)
  uint32_t het = 0, hom = 0, miss = 0;  // per lane
  // haploid calls (one allele character: chrX / chrY / chrM samples), per lane: how many are not missing (each counts one
  // allele towards an, not two) | how many of them carry ALT #1 (hom by class: alt == gt, main.go:1186; one allele
  // towards ac) << 16.  Taken here unless a dosage matrix is asked for (a haploid carrier's dosage is 1, its class 2).
  uint32_t hap = 0;
  // (with a dosage matrix too, round 4: a line with haploid calls is marked "not regular" in its entry, and k_dosage scans
  // it itself instead of expanding its class map -- a haploid carrier's dosage is 1, its class 2)
  const bool hap_ok = true;
  // wave-uniform: a haploid reference call with sub-fields ("0:...") has been seen -- a chrX-like file.  From then on the
  // packed-flag tiers take such a field for what it is (one allele towards an, nothing else) instead of sending its chunk
  // to the field-at-a-time code; files without them do not pay for the test
  bool hapref = false;
  constexpr uint32_t kHapRef = 0x3A30u;  // "0:"
  uint32_t n_sp = 0;       // list entries; kDenseMode once the map is dense
  uint32_t bad = 0;        // the line has a field this scan does not take: deferred (per lane until the line ends)
  uint32_t pl = 0;         // 0x80000000 if the last byte of the previous chunk was a TAB
  uint32_t R = 0x3A302F30u;  // "0/0:" -- the reference genotype word of this file, adopted as seen
  bool done = false;

#ifdef BVCF_EXP_TIMES
  // (tools/gen_tiers.py) cycles per tier of the chunk loop: 0 loop top (wait, ring slot, flags), 1 packed-flag tier, 2 medium,
  // 3 exact handler without 4 = its line ends (finish_line); 5..7: chunks through 1 / 2 / 3
  unsigned long long gph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, glast_ = __builtin_readcyclecounter();
#define GSTAMP(k)                                                   \
  {                                                                 \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    gph_[k] += now_ - glast_;                                       \
    glast_ = now_;                                                  \
  }
#define GCOUNT(k) gph_[k]++;
#else
#define GSTAMP(k)
#define GCOUNT(k)
#endif
  auto begin_line_samples = [&](uint32_t sb) {
    mode = kSamples;
    s_begin = sb;
    tabs_base = 0;
    tabs_lane = 0;
    het = hom = miss = 0;
    hap = 0;
    bad = 0;
    n_sp = list_ok ? 0u : kDenseMode;
    if (maps && !list_ok) zero_stage(stage, n_map_chunks);
  };
  // class of a list entry for ALT index k: '.' in either place makes the sample missing, else one point per allele == k
  // (second digit 15: a haploid call -- its one allele is all its alleles)
  auto entry_class = [](uint32_t e, uint32_t k) -> uint32_t {
    const uint32_t a4 = (e >> 4) & 15u, b4 = e & 15u;
    if (b4 == 15u) return a4 == 14u ? 3u : (a4 == k ? 2u : 0u);
    return (a4 == 14u || b4 == 14u) ? 3u : (a4 == k ? 1u : 0u) + (b4 == k ? 1u : 0u);
  };
  auto to_dense = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = n_sp;
    uint32_t e = 0;
    if ((uint32_t)lane < n) e = glist[lane];
    zero_stage(stage, n_map_chunks);
    if ((uint32_t)lane < n) {
      const uint32_t s = e >> 8;  // (a line with more fields than samples lists them too; it ends up deferred)
      if (s < ns) atomicOr(stage32 + (s >> 4), entry_class(e, 1u) << (2u * (s & 15u)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    n_sp = kDenseMode;
  };
  // one field per lane that is not the reference genotype (wave-uniform control flow): into the list while it lasts,
  // else -- ALT #1's class only -- into the stage
  auto record = [&](bool nonref, uint32_t sidx, uint32_t digits, uint32_t cls) {
    const unsigned long long br = __ballot(nonref);
    if (!br) return;
    if (n_sp < kDenseMode) {
      const uint32_t n_new = (uint32_t)__popcll(br);
      if (n_sp + n_new <= BVCF_CMAP_SPARSE_MAX) {
        const uint32_t at = n_sp + __builtin_amdgcn_mbcnt_hi((uint32_t)(br >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)br, 0u));
        if (nonref) glist[at] = (sidx << 8) | digits;
        n_sp = bcast0(n_sp + n_new);
        return;
      }
      to_dense();
    }
    if (nonref && cls != 0 && sidx < ns) atomicOr(stage32 + (sidx >> 4), cls << (2u * (sidx & 15u)));
  };
  // the line ended at terminator position e
  auto finish_line = [&](uint32_t e, bool last_is_tab) {
    seen++;
    const uint32_t ls0 = ls;
    ls = e + 1u;
    cur_pos = e + 1u;
    mode = kHead;
    found = 0;
    if (e + 1u < s_begin + a.eol_chars) return;  // chomping numChars bytes (main.go:535) eats the 9th TAB: <= 9 fields
    const uint32_t cend = e + 1u - a.eol_chars;
    if (cend - ls0 + 1u < a.n_header) return;  // cannot have n_header fields: never listed (see k_stream)
    const uint32_t tabs = tabs_base + wave_sum(tabs_lane);
    bool ok = !__any(bad != 0) && tabs + 1u == ns && !last_is_tab && cend > s_begin;
    // a slot of the wave's class-map range is owed to lines of at least 4 ns + 8 bytes (see k_stream)
    if (maps && (unsigned long long)(cend - ls0) + a.eol_chars < 4ull * ns + 8ull) ok = false;
    GtStats st = {0, 0, 0, 0, 0};
    if (!ok) {
      commit(ls0, cend, st, true, BVCF_NO_CMAP);
      return;
    }
    wave_sum3(het, hom, miss, ns, &st.n_het, &st.n_hom, &st.n_miss);
    const uint32_t hap_all = __any(hap != 0) ? wave_sum(hap) : 0u;  // (ns <= 16 384: both halves stay below 2^16)
    st.ac = st.n_het + 2u * st.n_hom - (hap_all >> 16);
    st.an = 2u * (ns - st.n_miss) - (hap_all & 0xFFFFu);
    const bool irregular = hap_all != 0u;  // the class map does not tell the dosage of a haploid carrier
    uint32_t cm_off = BVCF_NO_CMAP;
    if (maps) {
      const uint32_t slot = bcast0(map_slot());
      if (slot != BVCF_NO_CMAP) {
        uint8_t *cm = a.cmap + slot;
        // lists fit the slot for ALT #1 .. #max_k
        const uint32_t max_k = min(kListAlleles, a.cmap_stride / (4u * kSparseWords));
        uint32_t kmax = 0;  // 0: a dense map after all
        uint32_t ent = 0xFFFFFFFFu;
        if (n_sp < kDenseMode) {
          // ---- the entries in sample order (they were appended chunk by chunk, but field by field inside a chunk)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const uint32_t n = n_sp;
          const uint32_t my = (uint32_t)lane < n ? glist[lane] : 0xFFFFFFFFu;
          uint32_t rank = 0;
#pragma nounroll
          for (uint32_t j = 0; j < n; j++) rank += lane_value(my, (int)j) < my ? 1u : 0u;
          __builtin_amdgcn_wave_barrier();
          if ((uint32_t)lane < n) glist[rank] = my;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if ((uint32_t)lane < n) ent = glist[lane];
          // the highest ALT index a sample carries
          const uint32_t a4 = (ent >> 4) & 15u, b4 = ent & 15u;
          const uint32_t top = (uint32_t)lane < n ? max(a4 <= 9u ? a4 : 0u, b4 <= 9u ? b4 : 0u) : 0u;
          kmax = 1;
          if (__any(top >= 2u)) {  // (a biallelic line never gets here)
#pragma nounroll
            for (uint32_t k = 9; k >= 2; k--) {
              if (__any(top == k)) {
                kmax = k;
                break;
              }
            }
          }
          if (kmax > max_k) kmax = 0;  // no room for that many lists: ALT #1 as a map, the others through k_gt
          // (k_head takes the counts of a further index from its list as het + 2 hom: not so for a haploid carrier)
          if (kmax > 1u && hap_all) kmax = 0;
        }
        if (kmax) {
          // ---- one class list per ALT index (BVCF_ALLELE_CMAP_SPARSE): ascending, one entry per map byte; k_head takes
          // the counts of the further indices from them and no wave reads the line again (main.go:549-556 rescans)
          const uint32_t n = n_sp;
          const uint32_t smp = ent >> 8, idx = smp >> 2, sh = 2u * (smp & 3u);
          const bool have = (uint32_t)lane < n;
          // (n <= 15: the entries sit in lanes 0..14, one DPP row)
          const uint32_t prev_idx = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(have ? idx : 0xFFFFFFFEu), 0x111, 0xF, 0xF, false);
          const bool leader = have && idx != prev_idx;
#pragma nounroll
          for (uint32_t k = 1; k <= kmax; k++) {
            const uint32_t key = have ? (idx << 8) | (entry_class(ent, k) << sh) : 0xFFFFFFFFu;
            uint32_t merged = key & 0xFFu;
            const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x101, 0xF, 0xF, false);  // row_shl:1
            const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x102, 0xF, 0xF, false);
            const uint32_t n3 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x103, 0xF, 0xF, false);
            if ((n1 >> 8) == idx) merged |= n1 & 0xFFu;
            if ((n2 >> 8) == idx) merged |= n2 & 0xFFu;
            if ((n3 >> 8) == idx) merged |= n3 & 0xFFu;
            const bool w = leader && merged != 0;
            const unsigned long long bw = __ballot(w);
            uint32_t *list = reinterpret_cast<uint32_t *>(cm + 64u * (k - 1u));
            const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(bw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bw, 0u));
            if (w) __builtin_nontemporal_store((idx << 8) | merged, list + 1u + at);
            if (lane == 0) __builtin_nontemporal_store((uint32_t)__popcll(bw), list);
          }
          cm_off = slot | 1u | ((kmax - 1u) << 1);  // bit 0: lists; bits 1-3: ALT #2..#kmax have theirs, none is higher
        } else {
          if (n_sp < kDenseMode) to_dense();
          flush_stage(stage, cm, 0u, n_map_chunks * 64u, a.cmap_stride);
          cm_off = slot;
        }
      }
    }
    commit(ls0, cend, st, false, cm_off, irregular);
  };

  // ---- the exact handler: one chunk, any state
  auto slow = [&](const u32x4 v, uint32_t nx0, uint32_t cs) {
    const uint32_t off = cs + 16u * (uint32_t)lane;
    const uint32_t valid = bits_until(nb, off);
    const uint32_t mT = eq_mask16(v, '\t') & valid;
    const uint32_t mE = eq_mask16(v, a.eol_byte) & valid;
    const uint32_t carry = (uint32_t)__builtin_amdgcn_update_dpp((int)(pl >> 31), (int)(mT >> 15), 0x138, 0xF, 0xF, false) & 1u;
    const uint32_t starts_all = ((mT << 1) | carry) & valid & 0xFFFFu;
    // bytes 16..19 of the lane's window: the next lane's first dword (lane 63: the next chunk's)
    const uint32_t d4 = (uint32_t)__builtin_amdgcn_update_dpp((int)nx0, (int)v.x, 0x130, 0xF, 0xF, false);
    const uint32_t ce = min(cs + kChunk, nb);
    uint32_t cur = max(cs, cur_pos);
#pragma nounroll
    while (cur < ce) {
      const uint32_t from = ~bits_until(cur, off);  // the lane's bytes at positions >= cur
      const uint32_t mEc = mE & from;
      const unsigned long long be = __ballot(mEc != 0);
      uint32_t e = kNoneGen;
      if (be) e = lane_value(off + (uint32_t)__ffs(mEc) - 1u, __ffsll((long long)be) - 1);
      const uint32_t hi = be ? e : ce;
      const uint32_t rng = bits_until(hi, off) & from;
      if (mode == kSamples) {
        // ---- the fields that start in [cur, hi)
        tabs_base += wave_sum(tabs_lane);
        tabs_lane = 0;
        const uint32_t mTr = mT & rng;
        uint32_t tot;
        const uint32_t pre = wave_excl_scan(__popc(mTr), &tot);
        uint32_t st = starts_all & rng;
        // fields that are the reference word need nothing here (their TABs are counted above): the first start of each
        // of the lane's dwords is compared with R up front, so that the field-at-a-time loop below -- two rounds for
        // every line end otherwise, a lane's 16 bytes hold two ten-byte fields -- only runs for what is left
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          const uint32_t sq = (st >> (4u * q)) & 15u;
          const uint32_t b = (uint32_t)__ffs(sq) - 1u;  // (no start: 0xFFFFFFFF, byte 3 of nothing that is looked at)
          const uint32_t lo = q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w));
          const uint32_t hw = q == 0 ? v.y : (q == 1 ? v.z : (q == 2 ? v.w : d4));
          const uint32_t w = __builtin_amdgcn_alignbyte(hw, lo, b & 3u);
          if (sq != 0u && w == R) st &= ~(1u << (4u * q + b));
        }
        uint32_t cand = 0;
        bool seen_hapref = false;
#pragma nounroll
        while (__any(st != 0)) {
          const bool act = st != 0;
          const uint32_t k = act ? (uint32_t)__ffs(st) - 1u : 0u;
          st &= st - 1u;
          const uint32_t s = tabs_base + pre + __popc(mTr & ((1u << k) - 1u));
          const uint32_t i = k >> 2;
          const uint32_t lo = i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
          const uint32_t hw = i == 0 ? v.y : (i == 1 ? v.z : (i == 2 ? v.w : d4));
          const uint32_t w = __builtin_amdgcn_alignbyte(hw, lo, k & 3u);
          const uint32_t c1 = (w >> 8) & 0xFFu, c3 = w >> 24;
          const uint32_t v0 = (w & 0xFFu) ^ '0', v2 = ((w >> 16) & 0xFFu) ^ '0';
          // the reference's fast gate, main.go:1063-1064; a field of three bytes that ends the line counts as well
          const bool frame = (bool)((uint32_t)(c1 == '|') | (uint32_t)(c1 == '/')) &
                             (bool)((uint32_t)(c3 == ':') | (uint32_t)(c3 == '\t') | (uint32_t)(c3 == a.eol_byte));
          const bool plain = v0 < 32u && v2 < 32u && ((0x400003FFu >> v0) & (0x400003FFu >> v2) & 1u);
          const bool take = act && frame && plain;
          const bool isref = take && (v0 | v2) == 0;
          uint32_t cls = 0;
          // one allele character, then ':' or the field's end: the general branch's single token (main.go:1130-1190)
          const bool take_h = act && !take && hap_ok && v0 < 32u && ((0x400003FFu >> v0) & 1u) &&
                              (bool)((uint32_t)(c1 == ':') | (uint32_t)(c1 == '\t') | (uint32_t)(c1 == a.eol_byte));
          if (take_h) {
            cls = v0 == 30u ? 3u : (v0 == 1u ? 2u : 0u);  // ('.' ^ '0' = 30)
            hom += cls == BVCF_CLS_HOM;
            miss += cls == BVCF_CLS_MISSING;
            hap += (v0 != 30u ? 1u : 0u) + (v0 == 1u ? 0x10000u : 0u);
            if (v0 == 0u && c1 == ':') seen_hapref = true;
          }
          if (act && !take && !take_h) bad = 1;
          if (isref && w != R) cand = w;
          if (take && !isref) {
            // digit 1 scores one, '.' makes the sample missing (table of ALT #1, as in gt_scan_general)
            const uint32_t table = (1u << 2) | (3u << 28);
            const uint32_t code = ((table >> ((v0 & 15u) * 2u)) & 3u) + ((table >> ((v2 & 15u) * 2u)) & 3u);
            cls = code < 3u ? code : 3u;
            het += cls == BVCF_CLS_HET;
            hom += cls == BVCF_CLS_HOM;
            miss += cls == BVCF_CLS_MISSING;
          }
          // (one call for the whole wave: record() ballots)
          record(maps && ((take && !isref) || (take_h && v0 != 0u)), s, ((v0 & 15u) << 4) | (take_h ? 15u : (v2 & 15u)), cls);
        }
        tabs_base += tot;
        const unsigned long long bc = __ballot(cand != 0);
        if (bc) R = lane_value(cand, __ffsll((long long)bc) - 1);
        if (__any(seen_hapref)) hapref = true;
        if (!be) break;
        // the byte before the line's content end: a TAB there is an empty last field
        const uint32_t cend = e + 1u - a.eol_chars;
        bool last_is_tab = false;
        if (cend > cs) {
          const uint32_t q = cend - 1u - cs;  // byte of this chunk
          last_is_tab = (lane_value(mT, (int)(q >> 4)) >> (q & 15u)) & 1u;
        } else if (cend == cs) {
          last_is_tab = (pl >> 31) != 0;
        }  // (cend < cs: the terminator's "\r" was in the previous chunk; a TAB before it is caught by the field test)
        GSTAMP(3)
        finish_line(e, last_is_tab);
        GSTAMP(4)
        cur = e + 1u;
        if (ls >= r1) {
          done = true;
          break;
        }
        if ((unsigned long long)s_begin + 4ull * ns == (unsigned long long)e + 2u - a.eol_chars) n_regular++;  // (the shape k_stream is for)
      } else {
        // ---- head: the 9th TAB, or the terminator if it comes first (main.go:535)
        const uint32_t mTh = mT & rng;
        const uint32_t cnt = __popc(mTh);
        uint32_t tot;
        const uint32_t prefix = wave_excl_scan(cnt, &tot);
        if (found + tot >= 9u) {
          const uint32_t target = 8u - found;
          const bool mine = prefix <= target && target < prefix + cnt;
          const unsigned long long bm = __ballot(mine);
          const uint32_t pos = mine ? off + nth_bit(mTh, target - prefix) : 0u;
          const uint32_t tab9 = lane_value(pos, __ffsll((long long)bm) - 1);
          begin_line_samples(tab9 + 1u);
          cur = tab9 + 1u;
          cur_pos = cur;
          continue;
        }
        if (be) {  // fewer than 10 fields: cannot pass linePasses
          seen++;
          ls = e + 1u;
          cur = e + 1u;
          cur_pos = cur;
          found = 0;
          if (ls >= r1) {
            done = true;
            break;
          }
          continue;
        }
        found += tot;
        break;
      }
    }
    pl = (lane_value(mT, kWave - 1) >> 15) << 31;
  };

  // ---- a chunk of samples without a terminator in which some field is not R (t: TAB flags, S: field starts, u: the
  // word at the first start of each dword).  The lane's four words ^ R are the "t" words of the regular scan --
  // allele bytes ^ '0' in bytes 0 and 2, zero elsewhere when separator and end are R's -- so its byte-parallel
  // classification serves all four fields at once (gather4 / classes4, bvcf_gtscan.hip.h); a field framed otherwise
  // ("0/0:" in a file of "0|0:", a haploid call ...) goes through the field-at-a-time code.
  auto medium = [&](uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t S0, uint32_t S1, uint32_t S2, uint32_t S3,
                    uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3) {
    const uint32_t c0 = __popc(t0), c1 = c0 + __popc(t1), c2 = c1 + __popc(t2), cnt = c2 + __popc(t3);
    // one scan for both: the TABs of this chunk before the lane (low half) and what the lanes counted so far (high half)
    const uint32_t inc = wave_incl_scan(cnt | (tabs_lane << 16));
    const uint32_t total = lane_value(inc, kWave - 1);
    const uint32_t before = tabs_base + (total >> 16) + (inc & 0xFFFFu) - cnt;  // TABs of the region before the lane's bytes
    tabs_base += (total >> 16) + (total & 0xFFFFu);
    tabs_lane = 0;
    // sample index of the field at the first start of dword q: the TAB before it is this dword's unless it is byte 0
    auto sample_of = [&](uint32_t S, uint32_t cb) -> uint32_t {
      return cb + ((S & 0xFFu) ? 0u : 1u);
    };
    const uint32_t s0 = sample_of(S0, before), s1 = sample_of(S1, before + c0), s2 = sample_of(S2, before + c1),
                   s3 = sample_of(S3, before + c2);
    uint32_t x0 = S0 ? u0 ^ R : 0u, x1 = S1 ? u1 ^ R : 0u, x2 = S2 ? u2 ^ R : 0u, x3 = S3 ? u3 ^ R : 0u;
    // ---- fields whose separator / end byte differ from R's, or whose allele bytes are not '0' ^ [0, 31]
    // (a chrX-like file: its haploid reference calls are settled here, four at a time)
    if (hapref) {
      const uint32_t hx = (kHapRef ^ R) & 0xFFFFu;  // (x = 0, no start in the dword, cannot match: see the packed-flag tier)
      const bool h0 = (x0 & 0xFFFFu) == hx, h1 = (x1 & 0xFFFFu) == hx, h2 = (x2 & 0xFFFFu) == hx, h3 = (x3 & 0xFFFFu) == hx;
      hap += (uint32_t)h0 + (uint32_t)h1 + (uint32_t)h2 + (uint32_t)h3;
      x0 = h0 ? 0u : x0;
      x1 = h1 ? 0u : x1;
      x2 = h2 ? 0u : x2;
      x3 = h3 ? 0u : x3;
    }
    if (__any(((x0 | x1 | x2 | x3) & 0xFFE0FFE0u) != 0)) {
      uint32_t cand = 0;
      bool seen_hapref = false;
      auto odd = [&](uint32_t &x, uint32_t u, uint32_t sidx) {
        const bool m = (x & 0xFFE0FFE0u) != 0;
        if (!__any(m)) return;
        const uint32_t c1b = (u >> 8) & 0xFFu, c3b = u >> 24;
        const uint32_t v0 = (u & 0xFFu) ^ '0', v2 = ((u >> 16) & 0xFFu) ^ '0';
        const bool frame = (bool)((uint32_t)(c1b == '|') | (uint32_t)(c1b == '/')) &
                           (bool)((uint32_t)(c3b == ':') | (uint32_t)(c3b == '\t') | (uint32_t)(c3b == a.eol_byte));
        const bool plain = v0 < 32u && v2 < 32u && ((0x400003FFu >> v0) & (0x400003FFu >> v2) & 1u);
        const bool take = m && frame && plain;
        const bool isref = take && (v0 | v2) == 0;
        const bool take_h = m && !take && hap_ok && v0 < 32u && ((0x400003FFu >> v0) & 1u) &&
                            (bool)((uint32_t)(c1b == ':') | (uint32_t)(c1b == '\t') | (uint32_t)(c1b == a.eol_byte));
        if (m && !take && !take_h) bad = 1;
        if (isref) cand = u;
        uint32_t cls = 0;
        if (take_h) {  // a haploid call (see the exact handler)
          cls = v0 == 30u ? 3u : (v0 == 1u ? 2u : 0u);  // ('.' ^ '0' = 30)
          hom += cls == BVCF_CLS_HOM;
          miss += cls == BVCF_CLS_MISSING;
          hap += (v0 != 30u ? 1u : 0u) + (v0 == 1u ? 0x10000u : 0u);
          if (v0 == 0u && c1b == ':') seen_hapref = true;
        }
        if (take && !isref) {
          const uint32_t table = (1u << 2) | (3u << 28);
          const uint32_t code = ((table >> ((v0 & 15u) * 2u)) & 3u) + ((table >> ((v2 & 15u) * 2u)) & 3u);
          cls = code < 3u ? code : 3u;
          het += cls == BVCF_CLS_HET;
          hom += cls == BVCF_CLS_HOM;
          miss += cls == BVCF_CLS_MISSING;
        }
        record(maps && ((take && !isref) || (take_h && v0 != 0u)), sidx, ((v0 & 15u) << 4) | (take_h ? 15u : (v2 & 15u)), cls);
        if (m) x = 0;  // settled here
      };
      odd(x0, u0, s0);
      odd(x1, u1, s1);
      odd(x2, u2, s2);
      odd(x3, u3, s3);
      const unsigned long long bc = __ballot(cand != 0);
      if (bc) R = lane_value(cand, __ffsll((long long)bc) - 1);
      if (__any(seen_hapref)) hapref = true;
    }
    // ---- the rest, four fields at a time
    if (!__any((x0 | x1 | x2 | x3) != 0)) return;
    const Alleles4 g = gather4(x0, x1, x2, x3);
    if (alphabet_bad(g)) bad = 1;
    uint32_t LO, HI;
    classes4(g, 0x01010101u, &LO, &HI);
    het += __popc(LO & ~HI);
    hom += __popc(HI & ~LO);
    miss += __popc(LO & HI);
    if (!maps) return;
    const uint32_t cls_any = LO | HI;                                       // bit 8q + 7: field q has a class for ALT #1
    const uint32_t nz = g.A | g.B;
    const uint32_t nonref = ((nz + 0x7F7F7F7Fu) | nz) & 0x80808080u;        // ... is not the reference genotype
    auto cls_of = [&](uint32_t q) -> uint32_t { return ((LO >> (8u * q + 7u)) & 1u) | (((HI >> (8u * q + 7u)) & 1u) << 1); };
    auto digits_of = [&](uint32_t q) -> uint32_t { return (((g.A >> (8u * q)) & 15u) << 4) | ((g.B >> (8u * q)) & 15u); };
    if (n_sp < kDenseMode) {
      const unsigned long long bl = __ballot(nonref != 0);
      if (!bl) return;
      // entries of the lanes in lane order, a lane's fields in field order
      const uint32_t mine = __popc(nonref);
      uint32_t n_new, at;
      if ((bl & (bl - 1ull)) == 0) {  // one lane (the usual case): no prefix sum
        n_new = lane_value(mine, __ffsll((long long)bl) - 1);
        at = n_sp;
      } else {
        at = n_sp + wave_excl_scan(mine, &n_new);
      }
      if (n_sp + n_new <= BVCF_CMAP_SPARSE_MAX) {
        if (nonref & 0x00000080u) glist[at++] = (s0 << 8) | digits_of(0);
        if (nonref & 0x00008000u) glist[at++] = (s1 << 8) | digits_of(1);
        if (nonref & 0x00800000u) glist[at++] = (s2 << 8) | digits_of(2);
        if (nonref & 0x80000000u) glist[at++] = (s3 << 8) | digits_of(3);
        n_sp = bcast0(n_sp + n_new);
        return;
      }
      to_dense();
    }
    if (!__any(cls_any != 0)) return;
    if ((cls_any & 0x00000080u) && s0 < ns) atomicOr(stage32 + (s0 >> 4), cls_of(0) << (2u * (s0 & 15u)));
    if ((cls_any & 0x00008000u) && s1 < ns) atomicOr(stage32 + (s1 >> 4), cls_of(1) << (2u * (s1 & 15u)));
    if ((cls_any & 0x00800000u) && s2 < ns) atomicOr(stage32 + (s2 >> 4), cls_of(2) << (2u * (s2 & 15u)));
    if ((cls_any & 0x80000000u) && s3 < ns) atomicOr(stage32 + (s3 >> 4), cls_of(3) << (2u * (s3 & 15u)));
  };

  // ---- the stream: a ring of kGenRing chunks in LDS, filled by LDS-DMA loads (global_load_lds_dwordx4: no register
  // destination, so a plain loop -- one body, one handler each -- keeps kGenRing - 1 KiB in flight; with registers
  // the ring would have to be unrolled, and hipcc's code for that spilled).  hipcc does not count loads issued from an
  // asm statement: the waits below are ours.  vmcnt also counts the entry / class-map stores of finished lines,
  // which only makes a wait longer than needed, never shorter.
  const uint32_t e4 = a.eol_byte * 0x01010101u;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)as_lds(ring);
  auto issue = [&](uint32_t c) {
    const uint8_t *gsrc = a.buf + min(chunk_start(c) + 16u * (uint32_t)lane, cap_off);
    const uint32_t dst = ring_lds + (c % (uint32_t)kGenRing) * kChunk;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  };
  auto slot = [&](uint32_t c) -> u32x4 {
    return *reinterpret_cast<const u32x4 *>(ring + (c % (uint32_t)kGenRing) * kChunk + 16u * (uint32_t)lane);
  };
  static_assert((kGenRing & (kGenRing - 1)) == 0 && kGenRing >= 2 && kGenRing <= 16, "ring: a power of two");
#pragma unroll
  for (int j = 0; j < kGenRing; j++) issue((uint32_t)j);
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(kGenRing - 1));  // chunk 0 has landed
  u32x4 v = slot(0);
  uint32_t since_fold = 0;
#pragma nounroll
  for (uint32_t c = 0; !done; c++) {
    const uint32_t cs = chunk_start(c);
    if (cs >= nb) break;  // the unterminated tail of the block: dropped (main.go:354-358)
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(kGenRing - 2));  // chunk c + 1 has landed
    const u32x4 nv = slot(c + 1u);
    issue(c + (uint32_t)kGenRing);  // into the slot of chunk c, which is in `v` since the last round
    bool handled = false;
    if (mode == kSamples && cs + kChunk <= nb && cur_pos <= cs) {
      // flags: 0x80 in every byte that is a TAB (exact while all bytes are < 0x80, which is checked)
      const uint32_t x0 = v.x ^ 0x09090909u, x1 = v.y ^ 0x09090909u, x2 = v.z ^ 0x09090909u, x3 = v.w ^ 0x09090909u;
      const uint32_t t0 = ~(x0 + 0x7F7F7F7Fu) & 0x80808080u, t1 = ~(x1 + 0x7F7F7F7Fu) & 0x80808080u;
      const uint32_t t2 = ~(x2 + 0x7F7F7F7Fu) & 0x80808080u, t3 = ~(x3 + 0x7F7F7F7Fu) & 0x80808080u;
      // a terminator clears bit 7 of its byte in one of these
      const uint32_t eand = ((v.x ^ e4) + 0x7F7F7F7Fu) & ((v.y ^ e4) + 0x7F7F7F7Fu) & ((v.z ^ e4) + 0x7F7F7F7Fu) &
                            ((v.w ^ e4) + 0x7F7F7F7Fu);
      const uint32_t hard = (~eand | v.x | v.y | v.z | v.w) & 0x80808080u;  // a terminator, or a byte >= 0x80
      // field starts: the byte after a TAB
      const uint32_t pt = (uint32_t)__builtin_amdgcn_update_dpp((int)pl, (int)t3, 0x138, 0xF, 0xF, false);
      const uint32_t S0 = __builtin_amdgcn_alignbyte(t0, pt, 3u), S1 = __builtin_amdgcn_alignbyte(t1, t0, 3u);
      const uint32_t S2 = __builtin_amdgcn_alignbyte(t2, t1, 3u), S3 = __builtin_amdgcn_alignbyte(t3, t2, 3u);
      // the four bytes at the first start of a dword: its flag sits at bit 8 r + 7 (no flag: any r will do, the word is
      // not looked at)
      auto at_start = [](uint32_t hi, uint32_t lo, uint32_t S) -> uint32_t {
        uint32_t bit;  // (v_ffbl_b32 of 0 is -1, which C's ctz cannot say: the byte picked then is 3, and not looked at)
        asm("v_ffbl_b32 %0, %1" : "=v"(bit) : "v"(S));
        return __builtin_amdgcn_alignbyte(hi, lo, __builtin_amdgcn_ubfe(bit, 3u, 2u));
      };
      const uint32_t u0 = at_start(v.y, v.x, S0), u1 = at_start(v.z, v.y, S1), u2 = at_start(v.w, v.z, S2);
      // (the next chunk's first dword is asked for last: its LDS read has had the lines above to arrive)
      const uint32_t nx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nv.x);
      const uint32_t d4 = (uint32_t)__builtin_amdgcn_update_dpp((int)nx0, (int)v.x, 0x130, 0xF, 0xF, false);
      const uint32_t u3 = at_start(d4, v.w, S3);
      // (the kernel is bound by instructions issued, of any kind: selecting u ^ R costs one instruction less per dword
      // than two compares and the scalar logic on their masks)
      uint32_t mis = (S0 ? u0 ^ R : 0u) | (S1 ? u1 ^ R : 0u) | (S2 ? u2 ^ R : 0u) | (S3 ? u3 ^ R : 0u);
      const bool is_hard = hard != 0;
      const uint32_t two_starts = (S0 & (S0 - 1u)) | (S1 & (S1 - 1u)) | (S2 & (S2 - 1u)) | (S3 & (S3 - 1u));
      uint32_t hq = 0;
      if (hapref && hap_ok) {
        // haploid reference calls count as reference here (a dword with two field starts is not looked at twice: exact
        // handler).  "0:" against R = "0<sep>0:" differs in one known pattern of the low half (byte 0 equal, byte 1 =
        // ':' ^ sep, never zero): one xor serves both tests, and a dword without a start (x = 0) cannot match
        const uint32_t hx = (kHapRef ^ R) & 0xFFFFu;
        const uint32_t y0 = S0 ? u0 ^ R : 0u, y1 = S1 ? u1 ^ R : 0u, y2 = S2 ? u2 ^ R : 0u, y3 = S3 ? u3 ^ R : 0u;
        const bool h0 = (y0 & 0xFFFFu) == hx, h1 = (y1 & 0xFFFFu) == hx, h2 = (y2 & 0xFFFFu) == hx, h3 = (y3 & 0xFFFFu) == hx;
        mis = (h0 ? 0u : y0) | (h1 ? 0u : y1) | (h2 ? 0u : y2) | (h3 ? 0u : y3) | two_starts;
        hq = (uint32_t)h0 + (uint32_t)h1 + (uint32_t)h2 + (uint32_t)h3;
      }
      GSTAMP(0)
      if (!__any((mis | hard) != 0)) {
        hap += hq;
        tabs_lane = (uint32_t)__builtin_popcount(t0) + tabs_lane;
        tabs_lane = (uint32_t)__builtin_popcount(t1) + tabs_lane;
        tabs_lane = (uint32_t)__builtin_popcount(t2) + tabs_lane;
        tabs_lane = (uint32_t)__builtin_popcount(t3) + tabs_lane;
        pl = lane_value(t3, kWave - 1) & 0x80000000u;
        handled = true;
        GSTAMP(1)
        GCOUNT(5)
      } else if (!__any(is_hard || two_starts != 0u)) {
        // (medium() looks at the first field start of a dword only: a dword with two -- a one-character field, i.e. a haploid
        // call without sub-fields -- sends the chunk to the exact handler)
        medium(t0, t1, t2, t3, S0, S1, S2, S3, u0, u1, u2, u3);
        pl = lane_value(t3, kWave - 1) & 0x80000000u;
        handled = true;
        GSTAMP(2)
        GCOUNT(6)
      }
    }
    if (!handled) {
      GSTAMP(0)
      slow(v, (uint32_t)__builtin_amdgcn_readfirstlane((int)nv.x), cs);
      GSTAMP(3)
      GCOUNT(7)
    }
    v = nv;
    // (a lane's count shares a register with another in medium(): folded long before it could reach 16 bits)
    if (++since_fold >= 2048u) {
      tabs_base += wave_sum(tabs_lane);
      tabs_lane = 0;
      since_fold = 0;
    }
  }
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));  // no LDS-DMA may land after the wave has gone
#ifdef BVCF_EXP_TIMES
  if (lane == 0)
    for (int k = 0; k < 8; k++) g_phase_t[k][wave_in_grid() & 32767u] = gph_[k];
#endif
}

// ------------------------------------------------------------------ k_stream_gen: k_stream's frame around the general stream
// (runs of tiles per wave, tile-local entries, per-wave class-map slot ranges: all as in k_stream, whose k_order /
// k_head_lean / k_gt / k_finish follow unchanged)
// LDS of a workgroup, sized at launch (gen_lds_bytes): per wave the chunk ring, the stage of a dense class map -- 1 KiB
// per 4 096 samples, so that cohorts of a few thousand leave room for a fourth workgroup per CU -- and the carrier list
__host__ __device__ inline uint32_t gen_stage_bytes(uint32_t n_samples) {
  const uint32_t n_chunks = (n_samples * 4u + kChunk - 1u) / kChunk;
  return ((n_chunks + 15u) / 16u) * 1024u;
}
__host__ __device__ inline uint32_t gen_lds_bytes(uint32_t n_samples) {
  return (uint32_t)kWavesPerWg * ((uint32_t)kGenRing * kChunk + gen_stage_bytes(n_samples) + 64u * 4u);
}
__global__ __launch_bounds__(kWgThreads) void k_stream_gen(KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_gen[];
  const uint32_t stage_bytes = gen_stage_bytes(a.n_samples);
  const uint8_t *ring = s_gen + wave_in_wg() * ((uint32_t)kGenRing * kChunk);
  uint8_t *stage = s_gen + (uint32_t)kWavesPerWg * ((uint32_t)kGenRing * kChunk) + wave_in_wg() * stage_bytes;
  uint32_t *list = reinterpret_cast<uint32_t *>(s_gen + (uint32_t)kWavesPerWg * ((uint32_t)kGenRing * kChunk + stage_bytes)) + wave_in_wg() * 64u;
  const int lane = lane_id();
  const uint32_t wave = wave_in_grid();
#ifdef BVCF_EXP_TIMES
  if (lane == 0 && wave < 32768u) g_wave_t[0][wave] = wall_clock64();
#endif
  const uint32_t n_waves = gridDim.x * kWavesPerWg;
  const uint32_t ns = a.n_samples;
  const uint32_t nb = a.nbytes;
  const uint32_t T = a.tile_bytes;
  const bool maps = a.want_cmap != 0;
  const uint32_t n_chunks = (ns * 4u + kChunk - 1u) / kChunk;  // map bytes / 64 of a line
  const uint32_t q_tiles = a.n_tiles / n_waves, r_tiles = a.n_tiles % n_waves;
  const uint32_t per_wave = q_tiles + (r_tiles ? 1u : 0u);
  const uint32_t tile_lo = wave * q_tiles + min(wave, r_tiles);
  const uint32_t tile_hi = tile_lo + q_tiles + (wave < r_tiles ? 1u : 0u);
  const uint32_t r0 = tile_lo * T;
  const uint32_t r1 = (uint32_t)min((unsigned long long)tile_hi * T, (unsigned long long)nb);
  uint32_t tile = tile_lo, n_local = 0, n_listed = 0;
  uint32_t p = kNone;
  if (tile_lo < tile_hi) {
    p = 0;
    if (r0 > 0) {
      const uint32_t q = find_eol(a, r0 - 1, r1);
      p = q == kNone ? kNone : q + 1;
    }
  }
  // class-map slots: see k_stream (a line takes one only if it is at least 4 ns + 8 bytes long)
  const uint32_t slots_per_wave = (uint32_t)(((unsigned long long)per_wave * T) / (4ull * ns + 8ull)) + 2u;
  uint32_t cm_next = wave * slots_per_wave;
  const uint32_t cm_end = cm_next + slots_per_wave;
  if (maps && wave == 0 && lane == 0) a.counters->cmap_maps = min(n_waves, a.n_tiles) * slots_per_wave;
  auto map_slot = [&]() -> uint32_t {
    if (!maps) return BVCF_NO_CMAP;
    if (cm_next >= cm_end) {
      if (lane == 0) a.counters->pad[0] = 1;
      return BVCF_NO_CMAP;
    }
    return cmap_of(a, cm_next, true);
  };
  auto commit = [&](uint32_t ls, uint32_t cend, const GtStats &st, bool deferred, uint32_t cm_off, bool irregular = false) {
    while (ls >= (tile + 1) * T) {
      if (lane == 0) a.census[tile] = n_local;
      n_listed += n_local;
      tile++;
      n_local = 0;
    }
    if (n_local >= a.tile_quota) {
      if (lane == 0) a.counters->pad[0] = 1;  // cannot happen: see tile_quota
      return;
    }
    if (lane == 0) {
      StreamEntry en;
      en.ls = ls;
      en.len = (cend - ls) | (irregular ? kNotRegular : 0u);
      en.ac = st.ac;
      en.an = st.an;
      en.n_het = st.n_het;
      en.n_hom = st.n_hom;
      en.n_miss = deferred ? kDeferred : st.n_miss;
      en.cmap_off = cm_off;
      a.entries[(size_t)tile * a.tile_quota + n_local] = en;
    }
    n_local++;
    if (maps && !deferred) cm_next++;
  };
  uint32_t seen = 0, n_regular = 0;
  if (p != kNone && p < r1) stream_general_run(a, p, r1, n_chunks, stage, list, ring, seen, n_regular, commit, map_slot);
  for (; tile < tile_hi; tile++) {
    if (lane == 0) a.census[tile] = n_local;
    n_listed += n_local;
    n_local = 0;
  }
  if (lane == 0) a.run_lines[wave] = n_listed;  // (every wave of the grid: k_order adds them up)
  if (lane == 0 && seen) atomicAdd(&a.counters->lines_seen, seen);
  if (lane == 0 && n_regular) atomicAdd(&a.counters->n_other_shape, n_regular);
#ifdef BVCF_EXP_TIMES
  if (lane == 0 && wave < 32768u) g_wave_t[1][wave] = wall_clock64();
#endif
}

}  // namespace bvcf_dev
