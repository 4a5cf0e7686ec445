#!/bin/bash
# Builds bystro-vcf_amd/libbvcf_exp_<tag>.so for tools/kstream_stores.py: the chain stops after k_stream, and the kernels own
# stores are changed (a: all there; b: no class-list / class-map stores; c: none -- careful: the compiler then also drops the
# computations that only fed them, so b and c measure "no result computation", not "no stores"; j: every store to the
# waves first slot -- same work, no new cache lines; d plain instead of non-temporal list stores; e, f, g, h: entry stored by
# eight lanes / four lines per store instruction / lists in a compact area).  Results of such builds are NOT valid.
# usage: tools/kstream_stores_build.sh <tag>
R=$(cd "$(dirname "$0")/.." && pwd)
T=$1
D=/tmp/exp_$T
rm -rf $D && mkdir -p $D/bystro-vcf_amd && cp -r $R/bystro-vcf_amd/csrc $D/bystro-vcf_amd/ && cp -r $R/include $D/ && rm -rf $D/bystro-vcf_amd/csrc/build
python3 - "$T" "$D" <<'P'
import sys
T,D=sys.argv[1],sys.argv[2]
p=D+'/bystro-vcf_amd/csrc/bvcf_core.hip'
s=open(p).read()
old='''    if (ev_gt1) hipEventRecord(ev_gt1, st);
    hipLaunchKernelGGL(k_scan_groups, dim3(n_groups ? n_groups : 1), dim3(kWgThreads), 0, st, a, a.n_tiles);'''
assert old in s
s=s.replace(old,'''    if (ev_gt1) hipEventRecord(ev_gt1, st);
    return;  // EXPERIMENT: k_stream only
    hipLaunchKernelGGL(k_scan_groups, dim3(n_groups ? n_groups : 1), dim3(kWgThreads), 0, st, a, a.n_tiles);''',1)
open(p,'w').write(s)
if T in ('b','c'):
    p=D+'/bystro-vcf_amd/csrc/bvcf_gtscan.hip.h'
    s=open(p).read()
    s=s.replace('  if ((uint32_t)lane <= n) __builtin_nontemporal_store(lane == 0 ? n : prev, reinterpret_cast<uint32_t *>(cmap) + lane);','  if (stride == 1u && (uint32_t)lane <= n) __builtin_nontemporal_store(lane == 0 ? n : prev, reinterpret_cast<uint32_t *>(cmap) + lane);  // EXPERIMENT',1)
    s=s.replace('''  n = min(n, stride - g0);  // the slot is `stride` bytes (a multiple of 16)
''','''  n = min(n, stride - g0);  // the slot is `stride` bytes (a multiple of 16)
  if (stride != 1u) return;  // EXPERIMENT
''',1)
    open(p,'w').write(s)
if T=='d':
    p=D+'/bystro-vcf_amd/csrc/bvcf_gtscan.hip.h'
    s=open(p).read()
    s=s.replace('  if ((uint32_t)lane <= n) __builtin_nontemporal_store(lane == 0 ? n : prev, reinterpret_cast<uint32_t *>(cmap) + lane);','  if ((uint32_t)lane <= n) reinterpret_cast<uint32_t *>(cmap)[lane] = lane == 0 ? n : prev;  // EXPERIMENT: plain store',1)
    s=s.replace('''    __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(stage + i),
                                reinterpret_cast<u32x4 *>(cmap + g0 + i));  // written once, read by the host''','''    *reinterpret_cast<u32x4 *>(cmap + g0 + i) = *reinterpret_cast<const u32x4 *>(stage + i);  // EXPERIMENT: plain store''',1)
    open(p,'w').write(s)
if T=='e':
    p=D+'/bystro-vcf_amd/csrc/bvcf_stream.hip.h'
    s=open(p).read()
    old=s[s.index('    if (lane == 0) {\n      StreamEntry en;'):s.index('    if (bits_ok && lane < 16) a.head_bits[')]
    new='''    {
      // EXPERIMENT: the entry's eight dwords by eight lanes, one store instruction
      const uint32_t f[8] = {ls, (cend - ls) | (bits_ok ? kHasHeadBits : 0u), st.ac, st.an, st.n_het, st.n_hom, deferred ? kDeferred : st.n_miss, cm_off};
      uint32_t v = f[0];
#pragma unroll
      for (int q = 1; q < 8; q++) v = lane == q ? f[q] : v;
      if (lane < 8) reinterpret_cast<uint32_t *>(&a.entries[(size_t)tile * a.tile_quota + n_local])[lane] = v;
    }
'''
    s=s.replace(old,new,1)
    open(p,'w').write(s)
if T in ('f','g'):
    p=D+'/bystro-vcf_amd/csrc/bvcf_gtscan.hip.h'
    s=open(p).read()
    s=s.replace('  if ((uint32_t)lane <= n) __builtin_nontemporal_store(lane == 0 ? n : prev, reinterpret_cast<uint32_t *>(cmap) + lane);','''  // EXPERIMENT: the lists of four lines in one store instruction (the line in every fourth SLOT writes its own and the three
  // slots before it; the caller says so in bit 31 of max_k)
  if (max_k >> 31)
    __builtin_nontemporal_store(lane == 0 ? n : prev, reinterpret_cast<uint32_t *>(cmap - (size_t)((uint32_t)lane >> 4) * stride) + (lane & 15));''',1)
    s=s.replace('const bool sparse1 = n <= BVCF_CMAP_SPARSE_MAX && kmax <= max_k;','const bool sparse1 = n <= BVCF_CMAP_SPARSE_MAX && kmax <= (max_k & 0x7FFFFFFFu);',1)
    open(p,'w').write(s)
    p=D+'/bystro-vcf_amd/csrc/bvcf_stream.hip.h'
    s=open(p).read()
    old2='enc = finish_list(sparse, acc, cm, min(kListAlleles, a.cmap_stride / (4u * kSparseWords)), stage, nc, a.cmap_stride,'
    assert old2 in s
    s=s.replace(old2,'enc = finish_list(sparse, acc, cm, min(kListAlleles, a.cmap_stride / (4u * kSparseWords)) | ((((cmA / a.cmap_stride) & 3u) == 3u) ? 0x80000000u : 0u), stage, nc, a.cmap_stride,',1)
    open(p,'w').write(s)
if T=='g':
    p=D+'/bystro-vcf_amd/csrc/bvcf_stream.hip.h'
    s=open(p).read()
    old=s[s.index('    if (lane == 0) {\n      StreamEntry en;'):s.index('    n_local++;\n    if (maps && !deferred) cm_next += n_slots;')]
    new='''    {
      // EXPERIMENT: entries and head bits of four lines in one store instruction each
      const uint32_t f[8] = {ls, (cend - ls) | (bits_ok ? kHasHeadBits : 0u), st.ac, st.an, st.n_het, st.n_hom, deferred ? kDeferred : st.n_miss, cm_off};
      uint32_t v = f[0];
#pragma unroll
      for (int q = 1; q < 8; q++) v = (lane & 7) == q ? f[q] : v;
      if ((n_local & 3u) == 3u) {
        if (lane < 32) reinterpret_cast<uint32_t *>(&a.entries[(size_t)tile * a.tile_quota + n_local - 3u])[lane] = v;
        a.head_bits[((size_t)tile * a.tile_quota + n_local - 3u) * 16u + (uint32_t)lane] = (uint16_t)bits;
      }
    }
'''
    s=s.replace(old,new,1)
    open(p,'w').write(s)
if T=='h':
    # class lists into a compact, per-wave sequential area (64 B per line) instead of the 640-byte-strided slots
    p=D+'/bystro-vcf_amd/csrc/bvcf_stream.hip.h'
    s=open(p).read()
    old2='          uint8_t *cm = cmA != BVCF_NO_CMAP ? a.cmap + cmA : nullptr;\n          // the line starts in list mode'
    assert old2 in s
    s=s.replace(old2,'''          uint8_t *cm = cmA != BVCF_NO_CMAP ? a.cmap + cmA : nullptr;
          if (cm) cm = a.cmap + ((size_t)wave * lines_bound + (cm_next - wave * slots_per_wave)) * 64u;  // EXPERIMENT: compact list area
          // the line starts in list mode''',1)
    open(p,'w').write(s)
if T=='j':
    # every store of a line goes to the wave's FIRST entry / head-bits / class-map slot: the same instructions and the same
    # computations, but no new cache lines are dirtied
    p=D+'/bystro-vcf_amd/csrc/bvcf_stream.hip.h'
    s=open(p).read()
    s=s.replace('      a.entries[(size_t)tile * a.tile_quota + n_local] = en;','      a.entries[(size_t)tile_lo * a.tile_quota] = en;  // EXPERIMENT',1)
    s=s.replace('    if (bits_ok && lane < 16) a.head_bits[((size_t)tile * a.tile_quota + n_local) * 16u + (uint32_t)lane] = (uint16_t)bits;','    if (bits_ok && lane < 16) a.head_bits[((size_t)tile_lo * a.tile_quota) * 16u + (uint32_t)lane] = (uint16_t)bits;  // EXPERIMENT',1)
    old2='          uint8_t *cm = cmA != BVCF_NO_CMAP ? a.cmap + cmA : nullptr;\n          // the line starts in list mode'
    assert old2 in s
    s=s.replace(old2,'''          uint8_t *cm = cmA != BVCF_NO_CMAP ? a.cmap + cmA : nullptr;
          if (cm) cm = a.cmap + (size_t)cmap_of(a, wave * slots_per_wave, true);  // EXPERIMENT: always the wave's first slot
          // the line starts in list mode''',1)
    open(p,'w').write(s)
if T=='c':
    p=D+'/bystro-vcf_amd/csrc/bvcf_stream.hip.h'
    s=open(p).read()
    s=s.replace('      a.entries[(size_t)tile * a.tile_quota + n_local] = en;','      if (a.nbytes == 1u) a.entries[(size_t)tile * a.tile_quota + n_local] = en;  // EXPERIMENT',1)
    s=s.replace('    if (bits_ok && lane < 16) a.head_bits[','    if (a.nbytes == 1u && bits_ok && lane < 16) a.head_bits[',1)
    open(p,'w').write(s)
P
(cd $D/bystro-vcf_amd/csrc && make -s -j4 OUT=$D OBJ=$D/obj 2>&1 | grep -v "warning: unused\|^$" | head -5)
cp $D/libbvcf.so $R/bystro-vcf_amd/libbvcf_exp_$T.so
