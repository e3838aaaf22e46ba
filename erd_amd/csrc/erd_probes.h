// Every compile-time probe of liberd_hip.so in ONE place ("fence them in one header").
//
// The shipped library is built with NONE of these defined (erd_amd/csrc/Makefile passes no -D); tools/build_probe.sh and
// tools/build_abl.sh build variants under erd_amd/lib/abl/ that are selected with ERD_HIP_LIB for same-box A/B runs.  A translation
// unit compiled with any RESULT-CHANGING or TRACING probe emits the weak symbol `erd_probe_build_marker`; erd_probe_build() reports
// it, and erd_amd/_lib.py refuses to load such a library from the product path (tests/test_host_boundary.py asserts the shipped
// library is clean).
//
//   timing probes (results are WRONG: parts of a kernel removed to see what they cost)
//     conv_mfma.hip   ERD_X3_NOMFMA  ERD_X3_NOLOAD  ERD_X3_NOBREAD  ERD_X3_NOVALU  ERD_X3_NOSYNC      three-limb implicit GEMM K loop
//                     ERD_IG_NOMFMA  ERD_IG_NOLOAD  ERD_IG_NOFRAG   ERD_IG_NOSYNC                      bf16-mode implicit GEMM K loop
//     winograd.hip    ERD_WX3_NOMFMA ERD_WX3_NOSPLIT ERD_WX3_NOLOAD ERD_WX3_VREAD1 ERD_WX3_VWRITE1 ERD_WX3_NORAW ERD_WINO_GNPROBE                   three-limb / fp32 Winograd
//     conv_mfma.hip   ERD_WG3_NOSLAB                                                                    three-limb weight gradient without its partial-slab stores
//     conv_thin.hip   ERD_THIN_NOSTORE ERD_THIN_NOMFMA                                                    thin-K kernel: no output stores / no MFMAs
//     elementwise.hip ERD_GN_NOSTATS                                                                    GroupNorm without its statistics pass
//   accuracy probe (results differ in the last bits)
//     conv_mfma.hip, conv_thin.hip   ERD_X3_NINE      all nine limb products instead of six
//   tracing (results unchanged; s_memtime stamps + a device-side trace buffer)
//     conv_mfma.hip   ERD_IGEMM_TRACE  (erd_igemm_trace)          winograd.hip   ERD_WINO_TRACE  (erd_wino_trace)
//     conv_thin.hip   ERD_THIN_TRACE   (erd_thin_trace)
//   tuning parameters (results unchanged; defaults are the shipped values)
//     ERD_SGB  ERD_W3X3_MINW  ERD_WINO_NCH  ERD_WINO_DATA_PRIO  ERD_WINO_MMA_PRIO  ERD_WX3_RD
#pragma once

#if defined(ERD_X3_NOMFMA) || defined(ERD_X3_NOLOAD) || defined(ERD_X3_NOBREAD) || defined(ERD_X3_NOVALU) || defined(ERD_X3_NOSYNC) || \
    defined(ERD_IG_NOMFMA) || defined(ERD_IG_NOLOAD) || defined(ERD_IG_NOFRAG) || defined(ERD_IG_NOSYNC) || defined(ERD_WX3_NOMFMA) ||    \
    defined(ERD_WX3_NOSPLIT) || defined(ERD_WX3_NOLOAD) || defined(ERD_WX3_VREAD1) || defined(ERD_WX3_VWRITE1) || defined(ERD_THIN_NOSTORE) || defined(ERD_WG3_NOSLAB) || defined(ERD_THIN_NOMFMA) || defined(ERD_WX3_NORAW) || defined(ERD_WINO_GNPROBE) || defined(ERD_GN_NOSTATS) || defined(ERD_X3_NINE) || \
    defined(ERD_IGEMM_TRACE) || defined(ERD_WINO_TRACE) || defined(ERD_THIN_TRACE)
#define ERD_PROBE_BUILD 1
extern "C" __attribute__((weak, visibility("default"))) int erd_probe_build_marker = 1;
#endif
