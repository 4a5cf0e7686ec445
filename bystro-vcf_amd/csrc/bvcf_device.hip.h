// bvcf_device.hip.h — gfx950 device code of the per-line variant pipeline.
//
// Kernels, in launch order per batch:
//   k_count_eol    newline census per 1 KiB chunk              (readVcf's ReadBytes, main.go:354)
//   k_scan_groups  exclusive scan of the census, level 1
//   k_scan_top     level 2 + batch totals
//   k_scatter_eol  line start offsets from the census           ("workQueue <- buff", main.go:366)
//   k_head         16 lanes per line: tokenise the fixed columns, FILTER gate, getAlleles;
//                  emits allele records and one genotype-scan task per (line, ALT index)
//                                                                (main.go:535-545, 723-1038)
//   k_gt           one wavefront per task: the per-sample GT byte scan -> ac/an/het/hom/missing
//                  and the 2-bit class map.  THE HBM-bound kernel.   (main.go:1042-1194)
//   (from 32 768 samples up, in front of k_gt: the scans of one line split over waves --
//    k_gt_wide          one wave per (task, window of 16 384 samples) of a regular region
//    k_tabs_wide        TABs per 64 KiB share of an irregular region (a field's sample index = TABs before it)
//    k_gt_wide_general  one wave per (task, share): the general scan of the fields that start in the share
//    k_gt then adds up, or rescans a task whose regular windows met an irregular field)
//   k_finish       field-count verdict per line, scan results into the allele records
//   k_dosage       (bvcf_params.want_dosage) one wave per output allele: the int8 dosage row
//                  (k_dosage_wide: per share, for the lines k_gt_wide_general scanned)
//                                                                (main.go:1069-1178)
//
// Streaming variant for files with samples (KernelArgs.fused): the census, its scans, the scatter
// and the ALT #1 genotype scan are replaced by ONE pass over the text,
//   k_stream       one wave walks a 64 KiB tile: finds the lines that start in it, tokenises their
//                  fixed columns and scans ALT #1 straight away (the scan's loads ARE the newline
//                  search: a regular line ends where 4*ns bytes of "x|y<TAB>" end)
//   (k_stream_gen  the same for sample fields of any shape -- GT:DP:GQ ... --, bvcf_streamgen.hip.h; chosen per batch)
//   k_order        tile-local line entries -> input-ordered line_off / line_len / results and the batch's line count
//                  (a workgroup adds up the one-pass kernel's per-wave line totals and scans its own tiles' counts itself)
// after which k_head (k_head_lean when blocks are in flight), k_gt -- only the task slots k_head listed as holding a scan:
// deferred lines and further ALT indices of dense lines; it fills their allele records itself -- and k_finish (lines whose
// ALT #1 was deferred) run.
//
//   (bvcf_params.want_name_lists, after k_finish: k_name_len / k_name_scan / k_name_write render the het / hom / missing
//    sample-name lists of every output allele as text -- main.go:612-656 -- see bvcf_names.hip.h)
//
// Sites-only input (no sample columns): k_count_eol + k_scan_* as above, then ONE pass,
//   k_sites        a wave walks a run of 8 KiB windows: text ring + TAB bit ring in LDS, line ends into a FIFO, then
//                  one lane per line for strings.Split / linePasses / getAlleles / trTv and the records
//                  (replaces k_scatter_eol + k_head + k_finish there)
// ... the default since round 3 (bvcf_sites1.hip.h), census + scans as above, then
//   k_sites2       a wave takes 7 KiB tiles behind 1 KiB of lead-in: text + TAB and terminator bitmaps in LDS, the tile's
//                  first line number from the census; one lane per line, the common lines (SNPs, lines the gate
//                  rejects) settled on fast lanes without the general getAlleles code
// ... and, kept as BVCF_SITES=3, the same body without the census (the text is read once, and it is slower):
//   k_sites1       the tile's line count published at once, the line numbers found by a decoupled look-back
//
// ... and, on request (bvcf_params.render_sites, bvcf_render.hip.h), behind k_sites2p:
//   k_render_len / k_render_scan / k_render_rows   the TSV rows of the lines the packed form settles, in input order
//
// Everything is byte/integer work over the line bytes; no MFMA.  The genotype scans are bound by
// VALU issue at 57-70 % of the HBM peak.
// Loads are 16 B per lane, 1 KiB per wave-instruction, from the dword at or before the byte the
// record window starts at; the 0-3 byte shift is undone in registers (realign), after which every
// dword of a lane in a regular sample region ("x|y\t" per sample) is exactly one sample field.
#pragma once



#include "bvcf_common.hip.h"
#include "bvcf_index.hip.h"
#include "bvcf_alleles.hip.h"
#include "bvcf_gtscan.hip.h"
#include "bvcf_stream.hip.h"
#include "bvcf_head.hip.h"
#include "bvcf_sites.hip.h"
#include "bvcf_sites1.hip.h"
#include "bvcf_render.hip.h"
#include "bvcf_names.hip.h"
#include "bvcf_inflate.hip.h"
